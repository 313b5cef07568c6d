// Brute-force Chamfer nearest-neighbour search for gfx950 (MI355X).
//
// Replaces the chamferdist extension called at nnutils/loss_utils.py:85-86.
// Forward: for every point of one cloud, the minimum squared distance to the other
// cloud and the (lowest) index where it is attained.  FP32 VALU bound: 8 FLOP per
// (query, candidate) pair on 12 bytes per point.
//
// Layout of one workgroup (W waves; 4, or 16 for calls so small that they leave most CUs idle and
// a wave's scan length is the whole run time):
//   * 64*Q query points live in registers, Q per lane, the SAME queries in all W waves;
//   * candidate points are staged through LDS in tiles of TILE points padded to
//     float4, so one broadcast ds_read_b128 feeds Q distance evaluations;
//   * wave w scans share w of every tile, so each wave sees its candidates in
//     increasing index order and a strict `<` keeps the lowest index; the W
//     per-wave results are merged through LDS by (distance, index) lexicographic
//     order, which is the global "lowest index among the minima".
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "bucket_sort.h"
#include "sv3d_hip.h"

namespace {

#ifndef SV3D_CHAMFER_SMALL
#define SV3D_CHAMFER_SMALL 1
#endif
#ifndef SV3D_CHAMFER_SMALL_WGS
#define SV3D_CHAMFER_SMALL_WGS 512             // ... used up to this many workgroups of the tiled kernel (measured crossover)
#endif
#ifndef SV3D_CHAMFER_SMALL_QW
#define SV3D_CHAMFER_SMALL_QW 2                 // queries per wave of the small-call kernel
#endif
#ifndef SV3D_CHAMFER_Q4_MIN
#define SV3D_CHAMFER_Q4_MIN (4L * 256 * 64 * 4)  // points in the call from which a lane takes four queries
#endif
#ifndef SV3D_CHAMFER_Q2_MIN
#define SV3D_CHAMFER_Q2_MIN (4L * 256 * 64)      // ... two queries (measured: 64 x 642^2 19.4 -> 15.8 us; half of it loses at 32 x 642^2, 8 x 2048^2)
#endif
#ifndef SV3D_CHAMFER_TINY_WGS
#define SV3D_CHAMFER_TINY_WGS 512               // sixteen waves per workgroup up to this many workgroups (8 x 2048^2 25.8 -> 21.1 us, 2 x 8192^2 92 -> 73 us; 1024 loses at 32 x 642^2)
#endif
constexpr int kThreads = 256;
constexpr int kTile = 1024;          // candidates staged per LDS tile (16 KiB)
constexpr int kGroup = 8;            // candidates per group minimum

template <int Q, int kWaves>
__global__ __launch_bounds__(64 * kWaves) void chamfer_nn_kernel(
    const float *__restrict__ xyz1, const float *__restrict__ xyz2, int N, int M,
    float *__restrict__ dist1, float *__restrict__ dist2,
    int32_t *__restrict__ idx1, int32_t *__restrict__ idx2, int blocks_dir0)
{
    __shared__ float4 s_pts[kTile];
    __shared__ float s_best[kWaves][64 * Q];
    __shared__ int s_bidx[kWaves][64 * Q];

    const int b = blockIdx.y;
    // direction 0: queries = cloud 1, candidates = cloud 2; direction 1: swapped
    const bool dir1 = (int)blockIdx.x >= blocks_dir0;
    const int qblock = dir1 ? (int)blockIdx.x - blocks_dir0 : (int)blockIdx.x;
    const int nq = dir1 ? M : N;
    const int nc = dir1 ? N : M;
    const float *qp = (dir1 ? xyz2 + (size_t)b * M * 3 : xyz1 + (size_t)b * N * 3);
    const float *cp = (dir1 ? xyz1 + (size_t)b * N * 3 : xyz2 + (size_t)b * M * 3);
    float *dout = (dir1 ? dist2 + (size_t)b * M : dist1 + (size_t)b * N);
    int32_t *iout = (dir1 ? idx2 + (size_t)b * M : idx1 + (size_t)b * N);

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q0 = qblock * 64 * Q;

    float qx[Q], qy[Q], qz[Q], best[Q];
    int bi[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const int qi = min(q0 + k * 64 + lane, nq - 1);      // clamp: tail lanes redo the last point
        qx[k] = qp[3 * qi];
        qy[k] = qp[3 * qi + 1];
        qz[k] = qp[3 * qi + 2];
        best[k] = INFINITY;
        bi[k] = 0;
    }

    for (int t0 = 0; t0 < nc; t0 += kTile) {
        const int tn = min(kTile, nc - t0);
        __syncthreads();
        for (int j = threadIdx.x; j < tn; j += 64 * kWaves) {
            const float *c = cp + 3 * (size_t)(t0 + j);
            s_pts[j] = make_float4(c[0], c[1], c[2], 0.f);
        }
        __syncthreads();
        const int quarter = (tn + kWaves - 1) / kWaves;
        const int jb = wave * quarter;
        const int je = min(jb + quarter, tn);
        // Candidates are taken in groups of kGroup: the group minimum costs half a v_min3 per
        // distance, and the compare / select pair that a running argmin needs per distance is
        // only paid when some lane's minimum improves (ever rarer as the scan proceeds).  The
        // result is the one the plain scan gives: a group only wins with a strictly smaller
        // distance, and inside the group the first index attaining the minimum is taken.
        int j = jb;
        for (; j + kGroup <= je; j += kGroup) {
            float d[Q][kGroup];
#pragma unroll
            for (int u = 0; u < kGroup; ++u) {
                const float4 c = s_pts[j + u];
#pragma unroll
                for (int k = 0; k < Q; ++k) {
                    const float dx = c.x - qx[k];
                    const float dy = c.y - qy[k];
                    const float dz = c.z - qz[k];
                    d[k][u] = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                }
            }
            bool better = false;
            float gm[Q];
#pragma unroll
            for (int k = 0; k < Q; ++k) {
                const float m1 = fminf(fminf(d[k][0], d[k][1]), d[k][2]);
                const float m2 = fminf(fminf(d[k][3], d[k][4]), d[k][5]);
                gm[k] = fminf(fminf(d[k][6], d[k][7]), fminf(m1, m2));
                better = better || (gm[k] < best[k]);
            }
            if (__any(better)) {
#pragma unroll
                for (int k = 0; k < Q; ++k) {
                    if (gm[k] < best[k]) {
                        best[k] = gm[k];
                        int u0 = kGroup - 1;
#pragma unroll
                        for (int u = kGroup - 2; u >= 0; --u) u0 = d[k][u] == gm[k] ? u : u0;
                        bi[k] = t0 + j + u0;
                    }
                }
            }
        }
        for (; j < je; ++j) {                       // tail of the quarter
            const float4 c = s_pts[j];
            const int gj = t0 + j;
#pragma unroll
            for (int k = 0; k < Q; ++k) {
                const float dx = c.x - qx[k];
                const float dy = c.y - qy[k];
                const float dz = c.z - qz[k];
                const float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                const bool lt = d < best[k];
                best[k] = lt ? d : best[k];
                bi[k] = lt ? gj : bi[k];
            }
        }
    }

#pragma unroll
    for (int k = 0; k < Q; ++k) {
        s_best[wave][k * 64 + lane] = best[k];
        s_bidx[wave][k * 64 + lane] = bi[k];
    }
    __syncthreads();
    for (int s = threadIdx.x; s < 64 * Q; s += 64 * kWaves) {
        float bd = s_best[0][s];
        int bj = s_bidx[0][s];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) {
            const float d = s_best[w][s];
            const int j = s_bidx[w][s];
            if (d < bd || (d == bd && j < bj)) {
                bd = d;
                bj = j;
            }
        }
        const int k = s >> 6, l = s & 63;
        const int qi = q0 + k * 64 + l;
        if (qi < nq) {
            dout[qi] = bd;
            iout[qi] = bj;
        }
    }
}

// Lexicographic (distance, index) minimum over the 64 lanes with a fixed DPP tree; valid in lane 63.
template <int CTRL, int ROW_MASK>
__device__ inline void dpp_lexmin(float &d, int &i)
{
    // lanes the row mask leaves out, and lanes whose source lies outside their row, read their own value
    const float od = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(d), __float_as_int(d), CTRL, ROW_MASK, 0xf, false));
    const int oi = __builtin_amdgcn_update_dpp(i, i, CTRL, ROW_MASK, 0xf, false);
    const bool take = od < d || (od == d && oi < i);
    d = take ? od : d;
    i = take ? oi : i;
}

// Calls so small that the tiled kernel leaves most of the chip idle (the reference's own: two
// halves of a batch of 4-8 meshes with 642 vertices, nnutils/loss_utils.py:85-86) are nothing but
// latency there: stage, barrier, scan, barrier, merge.  Here a wave works alone -- no LDS, no
// barrier: its lanes hold the candidate cloud between them (CL points each, point c * 64 + lane in
// slot c), the wave takes QW queries one after the other, every lane scans its own candidates in
// increasing index with a strict `<`, and a (distance, index) lexicographic minimum over the lanes
// gives the lowest index among the minima -- the same result, bit for bit, as the tiled kernel.
template <int CL, int QW>
__global__ __launch_bounds__(kThreads) void chamfer_nn_small_kernel(
    const float *__restrict__ xyz1, const float *__restrict__ xyz2, int N, int M,
    float *__restrict__ dist1, float *__restrict__ dist2,
    int32_t *__restrict__ idx1, int32_t *__restrict__ idx2, int waves_dir0)
{
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave_b = __builtin_amdgcn_readfirstlane((int)blockIdx.x * (kThreads / 64) + (int)(threadIdx.x >> 6));
    const bool dir1 = wave_b >= waves_dir0;
    const int w = dir1 ? wave_b - waves_dir0 : wave_b;
    const int nq = dir1 ? M : N;
    const int nc = dir1 ? N : M;
    const int q0 = w * QW;
    if (q0 >= nq) return;                           // whole wave
    const float *qp = (dir1 ? xyz2 + (size_t)b * M * 3 : xyz1 + (size_t)b * N * 3);
    const float *cp = (dir1 ? xyz1 + (size_t)b * N * 3 : xyz2 + (size_t)b * M * 3);
    float *dout = (dir1 ? dist2 + (size_t)b * M : dist1 + (size_t)b * N);
    int32_t *iout = (dir1 ? idx2 + (size_t)b * M : idx1 + (size_t)b * N);

    float cx[CL], cy[CL], cz[CL];
#pragma unroll
    for (int c = 0; c < CL; ++c) {
        const int j = min(c * 64 + lane, nc - 1);   // slots past the cloud repeat its last point and are skipped below
        cx[c] = cp[3 * j];
        cy[c] = cp[3 * j + 1];
        cz[c] = cp[3 * j + 2];
    }
#pragma unroll
    for (int k = 0; k < QW; ++k) {
        const int qi = min(q0 + k, nq - 1);         // wave-uniform
        const float qx = qp[3 * qi], qy = qp[3 * qi + 1], qz = qp[3 * qi + 2];
        float best = INFINITY;
        int bi = 0x7fffffff;                        // a lane without candidates never wins a tie
#pragma unroll
        for (int c = 0; c < CL; ++c) {
            const float dx = cx[c] - qx;
            const float dy = cy[c] - qy;
            const float dz = cz[c] - qz;
            const float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
            const bool lt = c * 64 + lane < nc && d < best;
            best = lt ? d : best;
            bi = lt ? c * 64 + lane : bi;
        }
        dpp_lexmin<0xB1, 0xf>(best, bi);            // quad_perm [1,0,3,2]
        dpp_lexmin<0x4E, 0xf>(best, bi);            // quad_perm [2,3,0,1]
        dpp_lexmin<0x141, 0xf>(best, bi);           // row_half_mirror
        dpp_lexmin<0x140, 0xf>(best, bi);           // row_mirror: every lane of a row holds the row's minimum
        dpp_lexmin<0x142, 0xa>(best, bi);           // row_bcast:15: rows 1 and 3 take in the row before them
        dpp_lexmin<0x143, 0xc>(best, bi);           // row_bcast:31: rows 2 and 3 take in rows 0 and 1
        if (lane == 63 && q0 + k < nq) {
            dout[q0 + k] = best;
            // no candidate compared below +inf (a NaN coordinate, or distances that overflow): index 0
            // with distance +inf, as the tiled kernel reports it -- never an index outside the cloud
            iout[q0 + k] = bi == 0x7fffffff ? 0 : bi;
        }
    }
}

// Backward with the indices frozen.  One thread per (direction, batch, own point):
//   own term   g_own[i]  = 2 gd_own[i] (own[i] - other[idx_own[i]])
//   scattered  g_own[i] -= sum over j with idx_other[j] == i of 2 gd_other[j] (other[j] - own[i])
// The scatter is evaluated as a gather: every thread scans the other cloud's index array
// (staged through LDS) in ascending j, so the sum order is fixed and no atomics are needed.
__global__ __launch_bounds__(kThreads) void chamfer_grad_kernel(
    const float *__restrict__ xyz1, const float *__restrict__ xyz2, int N, int M,
    const float *__restrict__ gd1, const float *__restrict__ gd2,
    const int32_t *__restrict__ idx1, const int32_t *__restrict__ idx2,
    float *__restrict__ g1, float *__restrict__ g2, int blocks_dir0)
{
    __shared__ int s_idx[kTile];
    __shared__ float s_g[kTile];

    const int b = blockIdx.y;
    const bool dir1 = (int)blockIdx.x >= blocks_dir0;
    const int qblock = dir1 ? (int)blockIdx.x - blocks_dir0 : (int)blockIdx.x;
    const int n_own = dir1 ? M : N;
    const int n_oth = dir1 ? N : M;
    const float *own = dir1 ? xyz2 + (size_t)b * M * 3 : xyz1 + (size_t)b * N * 3;
    const float *oth = dir1 ? xyz1 + (size_t)b * N * 3 : xyz2 + (size_t)b * M * 3;
    const float *gd_own = dir1 ? gd2 : gd1;
    const float *gd_oth = dir1 ? gd1 : gd2;
    const int32_t *idx_own = dir1 ? idx2 + (size_t)b * M : idx1 + (size_t)b * N;
    const int32_t *idx_oth = dir1 ? idx1 + (size_t)b * N : idx2 + (size_t)b * M;
    float *gout = dir1 ? g2 : g1;
    if (gout == nullptr) return;                         // uniform over the block
    gout += (size_t)b * n_own * 3;

    const int i = qblock * kThreads + threadIdx.x;
    const bool live = i < n_own;
    const int ic = live ? i : n_own - 1;
    const float ox = own[3 * ic], oy = own[3 * ic + 1], oz = own[3 * ic + 2];
    float ax = 0.f, ay = 0.f, az = 0.f;
    if (gd_own != nullptr) {
        const int j = min(max(idx_own[ic], 0), n_oth - 1);     // a foreign index array cannot fault
        const float g = 2.f * gd_own[(size_t)b * n_own + ic];
        ax = g * (ox - oth[3 * j]);
        ay = g * (oy - oth[3 * j + 1]);
        az = g * (oz - oth[3 * j + 2]);
    }
    if (gd_oth != nullptr) {
        for (int t0 = 0; t0 < n_oth; t0 += kTile) {
            const int tn = min(kTile, n_oth - t0);
            __syncthreads();
            for (int j = threadIdx.x; j < tn; j += kThreads) {
                s_idx[j] = idx_oth[t0 + j];
                s_g[j] = 2.f * gd_oth[(size_t)b * n_oth + t0 + j];
            }
            __syncthreads();
            for (int j = 0; j < tn; ++j) {
                if (s_idx[j] == i) {
                    const float g = s_g[j];
                    const float *o = oth + 3 * (size_t)(t0 + j);
                    ax -= g * (o[0] - ox);
                    ay -= g * (o[1] - oy);
                    az -= g * (o[2] - oz);
                }
            }
        }
    }
    if (live) {
        gout[3 * i] = ax;
        gout[3 * i + 1] = ay;
        gout[3 * i + 2] = az;
    }
}

// The same backward with the scattered term read from buckets instead of a scan: one
// 1024-thread workgroup per (direction, batch element) sorts the other cloud's points j by their
// nearest-neighbour index idx_other[j] (LDS counting sort) and every own point i adds its bucket
// in ascending j -- O(N + M) instead of O(N M), the sum order of the scan form, no atomics on
// floats.  Dynamic LDS: (2 n_own + n_other) ints.
__global__ __launch_bounds__(sv3d_dev::kSortThreads) void chamfer_grad_sorted_kernel(
    const float *__restrict__ xyz1, const float *__restrict__ xyz2, int N, int M,
    const float *__restrict__ gd1, const float *__restrict__ gd2,
    const int32_t *__restrict__ idx1, const int32_t *__restrict__ idx2,
    float *__restrict__ g1, float *__restrict__ g2)
{
    extern __shared__ int smem[];
    __shared__ int s_wave[sv3d_dev::kSortThreads / 64];
    const int b = blockIdx.y;
    const bool dir1 = blockIdx.x == 1;
    const int n_own = dir1 ? M : N;
    const int n_oth = dir1 ? N : M;
    const float *own = dir1 ? xyz2 + (size_t)b * M * 3 : xyz1 + (size_t)b * N * 3;
    const float *oth = dir1 ? xyz1 + (size_t)b * N * 3 : xyz2 + (size_t)b * M * 3;
    const float *gd_own = dir1 ? gd2 : gd1;
    const float *gd_oth = dir1 ? gd1 : gd2;
    const int32_t *idx_own = dir1 ? idx2 + (size_t)b * M : idx1 + (size_t)b * N;
    const int32_t *idx_oth = dir1 ? idx1 + (size_t)b * N : idx2 + (size_t)b * M;
    float *gout = dir1 ? g2 : g1;
    if (gout == nullptr) return;                         // uniform over the block
    gout += (size_t)b * n_own * 3;
    int *fill = smem, *start = smem + n_own, *bucket = smem + 2 * n_own;
    if (gd_oth != nullptr) sv3d_dev::build_buckets(idx_oth, n_own, n_oth, fill, start, bucket, s_wave);
    for (int i = threadIdx.x; i < n_own; i += sv3d_dev::kSortThreads) {
        const float ox = own[3 * i], oy = own[3 * i + 1], oz = own[3 * i + 2];
        float ax = 0.f, ay = 0.f, az = 0.f;
        if (gd_own != nullptr) {
            const int j = min(max(idx_own[i], 0), n_oth - 1);
            const float g = 2.f * gd_own[(size_t)b * n_own + i];
            ax = g * (ox - oth[3 * j]);
            ay = g * (oy - oth[3 * j + 1]);
            az = g * (oz - oth[3 * j + 2]);
        }
        if (gd_oth != nullptr) {
            sv3d_dev::for_each_slot_ascending(bucket, start[i], fill[i], idx_oth, i, n_own, n_oth, [&](int j) {
                // an index outside the own cloud was clamped into a bucket by the sort; the scan
                // form never matches it, so it is skipped here as well
                if (idx_oth[j] != i) return;
                const float g = 2.f * gd_oth[(size_t)b * n_oth + j];
                const float *o = oth + 3 * (size_t)j;
                ax -= g * (o[0] - ox);
                ay -= g * (o[1] - oy);
                az -= g * (o[2] - oz);
            });
        }
        gout[3 * i] = ax;
        gout[3 * i + 1] = ay;
        gout[3 * i + 2] = az;
    }
}

constexpr size_t kSortLdsDefault = 48 * 1024;    // above this the launch raises the kernel's dynamic-LDS limit (static
                                                 // arrays count against the same 64 KiB default)
constexpr size_t kSortLdsMax = 150 * 1024;

inline int launch_status()
{
    return hipGetLastError() == hipSuccess ? SV3D_OK : SV3D_ERR_LAUNCH;
}

}  // namespace

extern "C" int sv3d_chamfer_forward(const float *xyz1, const float *xyz2, int32_t B, int32_t N,
                                    int32_t M, float *dist1, float *dist2, int32_t *idx1,
                                    int32_t *idx2, void *stream)
{
    if (!xyz1 || !xyz2 || !dist1 || !dist2 || !idx1 || !idx2) return SV3D_ERR_INVALID_ARG;
    if (B <= 0 || N <= 0 || M <= 0) return SV3D_ERR_INVALID_ARG;
    if (B > 65535) return SV3D_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // Q queries per lane: more reuse of each staged candidate once there is enough work
    // to keep 256 CUs busy with the smaller grid.
    const long total = (long)B * ((long)N + M);
    const int Q = total >= SV3D_CHAMFER_Q4_MIN ? 4 : (total >= SV3D_CHAMFER_Q2_MIN ? 2 : 1);
    const int per = 64 * Q;
    const int b0 = (N + per - 1) / per, b1 = (M + per - 1) / per;
    dim3 grid(b0 + b1, B), block(kThreads);
    // at most two workgroups per CU: sixteen waves split every tile, so that the scan of a wave --
    // the whole run time of such a call -- is a quarter as long (4 x 642 points: 8.4 -> about 6 us)
    const bool tiny = (long)(b0 + b1) * B <= SV3D_CHAMFER_TINY_WGS;
    // ... and when both clouds fit the lanes of one wave (16 points each), waves that work alone
    const int cmax = N > M ? N : M;
    if (SV3D_CHAMFER_SMALL && (long)(b0 + b1) * B <= SV3D_CHAMFER_SMALL_WGS && cmax <= 1024) {
        constexpr int QW = SV3D_CHAMFER_SMALL_QW;
        const int w0 = (N + QW - 1) / QW, w1 = (M + QW - 1) / QW;
        const dim3 sgrid((w0 + w1 + kThreads / 64 - 1) / (kThreads / 64), B);
#define SV3D_LAUNCH_SMALL(CL_)                                                                         \
    hipLaunchKernelGGL((chamfer_nn_small_kernel<CL_, QW>), sgrid, block, 0, s, xyz1, xyz2, N, M, dist1, dist2, \
                       idx1, idx2, w0)
        if (cmax <= 256) SV3D_LAUNCH_SMALL(4);
        else if (cmax <= 512) SV3D_LAUNCH_SMALL(8);
        else if (cmax <= 768) SV3D_LAUNCH_SMALL(12);
        else SV3D_LAUNCH_SMALL(16);
#undef SV3D_LAUNCH_SMALL
        return launch_status();
    }
    if (Q == 4)
        hipLaunchKernelGGL((chamfer_nn_kernel<4, 4>), grid, block, 0, s, xyz1, xyz2, N, M, dist1, dist2, idx1, idx2, b0);
    else if (Q == 2)
        hipLaunchKernelGGL((chamfer_nn_kernel<2, 4>), grid, block, 0, s, xyz1, xyz2, N, M, dist1, dist2, idx1, idx2, b0);
    else if (tiny)
        hipLaunchKernelGGL((chamfer_nn_kernel<1, 16>), grid, dim3(1024), 0, s, xyz1, xyz2, N, M, dist1, dist2, idx1, idx2, b0);
    else
        hipLaunchKernelGGL((chamfer_nn_kernel<1, 4>), grid, block, 0, s, xyz1, xyz2, N, M, dist1, dist2, idx1, idx2, b0);
    return launch_status();
}

extern "C" int sv3d_chamfer_backward(const float *xyz1, const float *xyz2, int32_t B, int32_t N,
                                     int32_t M, const float *grad_dist1, const float *grad_dist2,
                                     const int32_t *idx1, const int32_t *idx2, float *grad_xyz1,
                                     float *grad_xyz2, void *stream)
{
    if (!xyz1 || !xyz2 || !idx1 || !idx2) return SV3D_ERR_INVALID_ARG;
    if (B <= 0 || N <= 0 || M <= 0) return SV3D_ERR_INVALID_ARG;
    if (B > 65535) return SV3D_ERR_UNSUPPORTED;
    if (!grad_xyz1 && !grad_xyz2) return SV3D_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // The scattered term (the gradient a cloud receives through the OTHER cloud's distances) is
    // what costs: with it present and tables that fit the LDS, the sorted form; else the scan form
    // (whose scan loop is skipped when that term is absent, as in the reference's use).
    const size_t lds = ((size_t)2 * (N > M ? N : M) + (size_t)(N > M ? N : M)) * sizeof(int);
    const bool scattered = (grad_xyz1 && grad_dist2) || (grad_xyz2 && grad_dist1);
    bool sorted = scattered && lds <= kSortLdsMax;
    if (sorted && lds > kSortLdsDefault &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(&chamfer_grad_sorted_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        sorted = false;
    }
    if (sorted) {
        hipLaunchKernelGGL(chamfer_grad_sorted_kernel, dim3(2, B), dim3(sv3d_dev::kSortThreads), lds, s, xyz1,
                           xyz2, N, M, grad_dist1, grad_dist2, idx1, idx2, grad_xyz1, grad_xyz2);
        return launch_status();
    }
    const int b0 = (N + kThreads - 1) / kThreads, b1 = (M + kThreads - 1) / kThreads;
    dim3 grid(b0 + b1, B), block(kThreads);
    hipLaunchKernelGGL(chamfer_grad_kernel, grid, block, 0, s, xyz1, xyz2, N, M, grad_dist1,
                       grad_dist2, idx1, idx2, grad_xyz1, grad_xyz2, b0);
    return launch_status();
}
