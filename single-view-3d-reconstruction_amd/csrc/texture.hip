// The texture path either side of the rasterizer:
//  * bilinear sampling of images by a per-texel flow field, written straight in the rasterizer's
//    texel layout [B, F*T*T, C] (geom_utils.py:11-26 `sample_textures`: grid_sample + permute,
//    and loss_utils.py:242-283, the same sampling of a distance map), with its backward to the flow;
//  * the barrier distance transform of the ground-truth masks (utils/image.py:94-113), which the
//    reference computes with scipy on the host for every sample of every iteration (main.py:259-260).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "sv3d_hip.h"

namespace {

constexpr int kSampleThreads = 256;
constexpr int kMaxChannels = 4;

struct Bilinear {
    int x0, y0;                 // north-west corner
    float wx0, wy0;             // weights of the west / north side: (x0 + 1) - ix, (y0 + 1) - iy
    float wx1, wy1;             // weights of the east / south side: ix - x0, iy - y0
};

// grid_sample's coordinate rule for align_corners=True: -1 -> 0, +1 -> size - 1.
__device__ inline Bilinear bilinear_setup(float gx, float gy, int W, int H)
{
    const float ix = (gx + 1.f) * 0.5f * (float)(W - 1);
    const float iy = (gy + 1.f) * 0.5f * (float)(H - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    Bilinear r;
    // keeps the int conversion defined for wild or non-finite coordinates: such corners lie far
    // outside the image on either side and read as zero like any other outside corner
    r.x0 = (int)fminf(fmaxf(fx, -2.f), (float)W);
    r.y0 = (int)fminf(fmaxf(fy, -2.f), (float)H);
    r.wx0 = (fx + 1.f) - ix;
    r.wy0 = (fy + 1.f) - iy;
    r.wx1 = ix - fx;
    r.wy1 = iy - fy;
    return r;
}

__device__ inline float texel(const float *plane, int x, int y, int W, int H)
{
    return (x >= 0 && x < W && y >= 0 && y < H) ? plane[(size_t)y * W + x] : 0.f;     // zero padding
}

// One thread per sample n of image b.  out[b,n,c] = sum over the four corners, in the order
// nw, ne, sw, se, of image[b,c,corner] * weight.
template <int C>
__global__ __launch_bounds__(kSampleThreads) void texture_sample_kernel(
    const float *__restrict__ images, const float *__restrict__ flow, int H, int W, int N,
    float *__restrict__ out)
{
    const int b = blockIdx.y;
    const int n = blockIdx.x * kSampleThreads + threadIdx.x;
    if (n >= N) return;
    const float2 g = reinterpret_cast<const float2 *>(flow)[(size_t)b * N + n];
    const Bilinear s = bilinear_setup(g.x, g.y, W, H);
    const float w_nw = s.wx0 * s.wy0, w_ne = s.wx1 * s.wy0, w_sw = s.wx0 * s.wy1, w_se = s.wx1 * s.wy1;
    float *o = out + ((size_t)b * N + n) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float *plane = images + ((size_t)b * C + c) * H * W;
        float v = texel(plane, s.x0, s.y0, W, H) * w_nw;
        v += texel(plane, s.x0 + 1, s.y0, W, H) * w_ne;
        v += texel(plane, s.x0, s.y0 + 1, W, H) * w_sw;
        v += texel(plane, s.x0 + 1, s.y0 + 1, W, H) * w_se;
        o[c] = v;
    }
}

// grad_flow[b,n,:] from grad_out[b,n,:]: the derivative of the four weights with respect to the
// sample position, times the corner values, chained through the coordinate rule.
template <int C>
__global__ __launch_bounds__(kSampleThreads) void texture_sample_backward_kernel(
    const float *__restrict__ images, const float *__restrict__ flow, const float *__restrict__ grad_out,
    int H, int W, int N, float *__restrict__ grad_flow)
{
    const int b = blockIdx.y;
    const int n = blockIdx.x * kSampleThreads + threadIdx.x;
    if (n >= N) return;
    const float2 g = reinterpret_cast<const float2 *>(flow)[(size_t)b * N + n];
    const Bilinear s = bilinear_setup(g.x, g.y, W, H);
    const float *go = grad_out + ((size_t)b * N + n) * C;
    float gix = 0.f, giy = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float *plane = images + ((size_t)b * C + c) * H * W;
        const float nw = texel(plane, s.x0, s.y0, W, H), ne = texel(plane, s.x0 + 1, s.y0, W, H);
        const float sw = texel(plane, s.x0, s.y0 + 1, W, H), se = texel(plane, s.x0 + 1, s.y0 + 1, W, H);
        const float gc = go[c];
        gix += gc * ((ne - nw) * s.wy0 + (se - sw) * s.wy1);
        giy += gc * ((sw - nw) * s.wx0 + (se - ne) * s.wx1);
    }
    float2 r;
    r.x = gix * (0.5f * (float)(W - 1));
    r.y = giy * (0.5f * (float)(H - 1));
    reinterpret_cast<float2 *>(grad_flow)[(size_t)b * N + n] = r;
}

// ---------------------------------------------------------------------------------------------
// Exact Euclidean distance transform of two fields at once: "outside" = 1 - mask (zero where
// mask == 1) and "inside" = mask (zero where mask == 0).  A field without any zero gets scipy's
// value for that case, the distance to the point (row -1, column 0).
//
// Separable and exact in integers, two launches:
// (1) dt_rows_kernel: a wave per image row finds for every pixel the horizontal distance to the
//     nearest zero of each field with wave scans (nearest zero to the left: running maximum of
//     zero positions; to the right: running minimum) and stores the pair as two uint16 (16384 =
//     no zero in this row) in the workspace.
// (2) dt_columns_kernel: a workgroup per (image, 64-column strip, 16 output rows) stages the strip
//     of pairs in the LDS (read straight from the workspace when the image is too tall) and every
//     pixel takes min over rows y' of (y - y')^2 + g(y', x)^2, walking outwards from its own row
//     until (y - y')^2 alone reaches the best found for both fields.
constexpr int kDtThreads = 256;
constexpr int kDtMaxSize = 4096;
constexpr int kDtStrip = 64;
constexpr int kDtRowsPerGroup = 16;
constexpr int kDtLdsRows = 512;                  // 512 x 64 x 4 B = 128 KB
constexpr unsigned kNoZero = 16384u;             // its square is kFar; real distances stay below 4096
constexpr int kFar = 1 << 28;

__global__ __launch_bounds__(kDtThreads) void dt_rows_kernel(const float *__restrict__ masks, int H, int W,
                                                            uint32_t *__restrict__ pairs)
{
    const int lane = threadIdx.x & 63;
    const int y = blockIdx.x * (kDtThreads / 64) + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (y >= H) return;
    const float *m = masks + ((size_t)b * H + y) * W;
    uint32_t *p = pairs + ((size_t)b * H + y) * W;
    const int chunks = (W + 63) / 64;
    // left to right: position of the nearest zero at or before x, per field
    int carry_out = -kFar, carry_in = -kFar;
    for (int c = 0; c < chunks; ++c) {
        const int x = c * 64 + lane;
        const float v = x < W ? m[x] : 0.5f;     // padding is a zero of neither field
        int z_out = (1.f - v) == 0.f ? x : -kFar, z_in = v == 0.f ? x : -kFar;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(z_out, d), i = __shfl_up(z_in, d);
            if (lane >= d) {
                z_out = max(z_out, o);
                z_in = max(z_in, i);
            }
        }
        z_out = max(z_out, carry_out);
        z_in = max(z_in, carry_in);
        carry_out = __shfl(z_out, 63);
        carry_in = __shfl(z_in, 63);
        if (x < W) {
            const unsigned d_out = (unsigned)min(x - z_out, (int)kNoZero), d_in = (unsigned)min(x - z_in, (int)kNoZero);
            p[x] = d_out | (d_in << 16);
        }
    }
    // right to left: nearest zero at or after x; keep the smaller of the two distances
    carry_out = kFar;
    carry_in = kFar;
    for (int c = chunks - 1; c >= 0; --c) {
        const int x = c * 64 + lane;
        const float v = x < W ? m[x] : 0.5f;
        int z_out = (1.f - v) == 0.f ? x : kFar, z_in = v == 0.f ? x : kFar;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_down(z_out, d), i = __shfl_down(z_in, d);
            if (lane + d < 64) {
                z_out = min(z_out, o);
                z_in = min(z_in, i);
            }
        }
        z_out = min(z_out, carry_out);
        z_in = min(z_in, carry_in);
        carry_out = __shfl(z_out, 0);
        carry_in = __shfl(z_in, 0);
        if (x < W) {
            const uint32_t left = p[x];
            const unsigned d_out = min((unsigned)min(z_out - x, (int)kNoZero), left & 0xffffu);
            const unsigned d_in = min((unsigned)min(z_in - x, (int)kNoZero), left >> 16);
            p[x] = d_out | (d_in << 16);
        }
    }
}

__device__ inline float dt_phantom(int x, int y) { return sqrtf((float)((y + 1) * (y + 1) + x * x)); }

__device__ inline int dt_square(unsigned g) { return (int)(g * g); }     // kNoZero^2 == kFar

// The four rows of a wave are walked together: their reads are independent, so the LDS latency
// of one overlaps the arithmetic of the others.
constexpr int kDtRowsPerWave = kDtRowsPerGroup / (kDtThreads / 64);

template <bool STAGED>
__global__ __launch_bounds__(kDtThreads) void dt_columns_kernel(const uint32_t *__restrict__ pairs, int H, int W,
                                                               float k, int barrier, float *__restrict__ out)
{
    extern __shared__ uint32_t s_pairs[];        // STAGED: [H][kDtStrip]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * kDtStrip + lane;
    const int b = blockIdx.z;
    const uint32_t *g = pairs + (size_t)b * H * W;
    const int xc = min(x, W - 1);
    if (STAGED) {
        for (int r = wave; r < H; r += kDtThreads / 64) s_pairs[r * kDtStrip + lane] = g[(size_t)r * W + xc];
        __syncthreads();
    }
    // rows outside the image are clamped: a clamped row was already seen at its true, smaller
    // distance, so the repeated candidate cannot lower a minimum
    auto pair_at = [&](int r) -> uint32_t {
        r = min(max(r, 0), H - 1);
        return STAGED ? s_pairs[r * kDtStrip + lane] : g[(size_t)r * W + xc];
    };
    const int y0 = blockIdx.y * kDtRowsPerGroup + wave * kDtRowsPerWave;
    int best_out[kDtRowsPerWave], best_in[kDtRowsPerWave];
#pragma unroll
    for (int j = 0; j < kDtRowsPerWave; ++j) {
        const uint32_t own = pair_at(y0 + j);
        best_out[j] = dt_square(own & 0xffffu);
        best_in[j] = barrier ? dt_square(own >> 16) : 0;
    }
    for (int dy = 1; dy < H; ++dy) {
        const int dy2 = dy * dy;
        int worst = 0;
#pragma unroll
        for (int j = 0; j < kDtRowsPerWave; ++j) worst = max(worst, max(best_out[j], best_in[j]));
        if (dy2 >= worst) break;
#pragma unroll
        for (int j = 0; j < kDtRowsPerWave; ++j) {
            const uint32_t up = pair_at(y0 + j - dy), down = pair_at(y0 + j + dy);
            best_out[j] = min(best_out[j], dy2 + min(dt_square(up & 0xffffu), dt_square(down & 0xffffu)));
            best_in[j] = min(best_in[j], dy2 + min(dt_square(up >> 16), dt_square(down >> 16)));
        }
    }
    const float inv_size = 1.f / (float)max(H, W);
#pragma unroll
    for (int j = 0; j < kDtRowsPerWave; ++j) {
        const int y = y0 + j;
        const float dist_out = best_out[j] >= kFar ? dt_phantom(x, y) : sqrtf((float)best_out[j]);
        float r = dist_out * inv_size;
        if (barrier) {
            const float dist_in = best_in[j] >= kFar ? dt_phantom(x, y) : sqrtf((float)best_in[j]);
            r = 1.f / (1.f + expf(-k * ((dist_out - dist_in) * inv_size)));
        }
        if (x < W && y < H) out[((size_t)b * H + y) * W + x] = r;
    }
}

inline int launch_status()
{
    return hipGetLastError() == hipSuccess ? SV3D_OK : SV3D_ERR_LAUNCH;
}

inline bool sample_args_ok(const void *images, const void *flow, int B, int C, int H, int W, int N)
{
    return images && flow && B > 0 && C >= 1 && C <= kMaxChannels && H > 0 && W > 0 && N > 0;
}

}  // namespace

#define SV3D_BY_CHANNELS(KERNEL, ...)                                                                   \
    switch (C) {                                                                                        \
    case 1: hipLaunchKernelGGL(KERNEL<1>, grid, dim3(kSampleThreads), 0, s, __VA_ARGS__); break;        \
    case 2: hipLaunchKernelGGL(KERNEL<2>, grid, dim3(kSampleThreads), 0, s, __VA_ARGS__); break;        \
    case 3: hipLaunchKernelGGL(KERNEL<3>, grid, dim3(kSampleThreads), 0, s, __VA_ARGS__); break;        \
    default: hipLaunchKernelGGL(KERNEL<4>, grid, dim3(kSampleThreads), 0, s, __VA_ARGS__); break;       \
    }

extern "C" int sv3d_texture_sample(const float *images, const float *flow, int32_t B, int32_t C, int32_t H,
                                   int32_t W, int32_t N, float *out, void *stream)
{
    if (!sample_args_ok(images, flow, B, C, H, W, N) || !out) return SV3D_ERR_INVALID_ARG;
    if (B > 65535) return SV3D_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((N + kSampleThreads - 1) / kSampleThreads, B);
    SV3D_BY_CHANNELS(texture_sample_kernel, images, flow, H, W, N, out)
    return launch_status();
}

extern "C" int sv3d_texture_sample_backward(const float *images, const float *flow, const float *grad_out,
                                            int32_t B, int32_t C, int32_t H, int32_t W, int32_t N,
                                            float *grad_flow, void *stream)
{
    if (!sample_args_ok(images, flow, B, C, H, W, N) || !grad_out || !grad_flow) return SV3D_ERR_INVALID_ARG;
    if (B > 65535) return SV3D_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((N + kSampleThreads - 1) / kSampleThreads, B);
    SV3D_BY_CHANNELS(texture_sample_backward_kernel, images, flow, grad_out, H, W, N, grad_flow)
    return launch_status();
}

extern "C" int sv3d_distance_transform_workspace(int32_t B, int32_t H, int32_t W, size_t *bytes)
{
    if (!bytes || B <= 0 || H <= 0 || W <= 0) return SV3D_ERR_INVALID_ARG;
    *bytes = (size_t)B * H * W * sizeof(uint32_t);
    return SV3D_OK;
}

namespace {

// Both launches of the transform; `pairs` holds B*H*W words of device memory.
int distance_transform_launch(const float *masks, int B, int H, int W, int barrier, float k, float *out,
                              uint32_t *pairs, hipStream_t s)
{
    const int rows_per_wg = kDtThreads / 64;
    hipLaunchKernelGGL(dt_rows_kernel, dim3((H + rows_per_wg - 1) / rows_per_wg, B), dim3(kDtThreads), 0, s, masks,
                       H, W, pairs);
    const dim3 grid((W + kDtStrip - 1) / kDtStrip, (H + kDtRowsPerGroup - 1) / kDtRowsPerGroup, B);
    const size_t lds = (size_t)H * kDtStrip * sizeof(uint32_t);
    bool staged = H <= kDtLdsRows;
    if (staged && lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(&dt_columns_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        staged = false;
    }
    if (staged)
        hipLaunchKernelGGL(dt_columns_kernel<true>, grid, dim3(kDtThreads), lds, s, pairs, H, W, k, barrier ? 1 : 0, out);
    else
        hipLaunchKernelGGL(dt_columns_kernel<false>, grid, dim3(kDtThreads), 0, s, pairs, H, W, k, barrier ? 1 : 0, out);
    return launch_status();
}

// Row-distance words of the entry point that allocates nothing: one array per GPU the library
// runs on, held until the process ends.  In host code this name is only a handle for
// hipGetSymbolAddress; kernels get the per-device address looked up below.
static __device__ __align__(256) uint32_t scratch[SV3D_DT_STATIC_MAX_PIXELS];

constexpr int kMaxDevices = 64;
uint32_t *g_scratch_ptr[kMaxDevices];            // device address of `scratch` on each GPU, once looked up
std::mutex g_scratch_mutex;

// Address of the current device's copy of `scratch`, or nullptr with `status` saying why.  The
// first call on a device looks the address up, which may load the library's code object on that
// device: not something to do while `stream` is capturing, so such a call is refused instead.
uint32_t *scratch_on_current_device(hipStream_t stream, int &status)
{
    int dev = -1;
    status = SV3D_ERR_LAUNCH;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cap) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        if (cap != hipStreamCaptureStatusNone) {
            status = SV3D_ERR_NOT_READY;
            return nullptr;
        }
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) {
            (void)hipGetLastError();
            return nullptr;
        }
        g_scratch_ptr[dev] = static_cast<uint32_t *>(p);
    }
    return g_scratch_ptr[dev];
}

}  // namespace

extern "C" int sv3d_distance_transform(const float *masks, int32_t B, int32_t H, int32_t W, int32_t barrier,
                                       float k, float *out, void *workspace, size_t workspace_bytes,
                                       void *stream)
{
    if (!masks || !out || !workspace || B <= 0 || H <= 0 || W <= 0) return SV3D_ERR_INVALID_ARG;
    if (W > kDtMaxSize || H > kDtMaxSize || B > 65535) return SV3D_ERR_UNSUPPORTED;
    if (workspace_bytes < (size_t)B * H * W * sizeof(uint32_t)) return SV3D_ERR_WORKSPACE;
    return distance_transform_launch(masks, B, H, W, barrier, k, out, static_cast<uint32_t *>(workspace),
                                     (hipStream_t)stream);
}

extern "C" int sv3d_distance_transform_static(const float *masks, int32_t B, int32_t H, int32_t W,
                                              int32_t barrier, float k, float *out, void *stream)
{
    if (!masks || !out || B <= 0 || H <= 0 || W <= 0) return SV3D_ERR_INVALID_ARG;
    if (W > kDtMaxSize || H > kDtMaxSize || B > 65535) return SV3D_ERR_UNSUPPORTED;
    if ((size_t)B * H * W > (size_t)SV3D_DT_STATIC_MAX_PIXELS) return SV3D_ERR_WORKSPACE;
    int why = SV3D_OK;
    uint32_t *pairs = scratch_on_current_device((hipStream_t)stream, why);
    if (!pairs) return why;
    // every word the column pass reads was written by the row pass of this same call: nothing to clear
    return distance_transform_launch(masks, B, H, W, barrier, k, out, pairs, (hipStream_t)stream);
}
