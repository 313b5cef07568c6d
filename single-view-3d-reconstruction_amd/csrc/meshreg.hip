// Mesh regularisers of the training step, each as ONE launch that returns the per-mesh loss and
// its gradient with respect to the vertices:
//   uniform Laplacian loss   sum_v |x_v - mean(neighbours of v)|^2        (sr.LaplacianLoss, main.py:202,515)
//   flatten loss             sum over interior edges of (cos(dihedral) + 1)^2  (sr.FlattenLoss, main.py:203,516)
// and a batched CSR product y = L x with three columns, the forward and the backward of the
// cotangent Laplacian of nnutils/laplacian.py:63-103 (a scipy product on the host there).
// All sums run in a fixed order (no float atomics): results are bitwise repeatable.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sv3d_hip.h"

namespace {

constexpr int kRegThreads = 1024;
constexpr int kSpmvThreads = 256;
constexpr size_t kLdsDefault = 48 * 1024;     // static arrays count against the same 64 KiB default
constexpr size_t kLdsMax = 150 * 1024;

__device__ inline int clampi(int v, int n) { return min(max(v, 0), n - 1); }

// Sum of `v` over the workgroup in a fixed order, returned to every thread.
__device__ inline float block_sum(float v, float *s_part)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    __syncthreads();
    if (lane == 0) s_part[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < kRegThreads / 64; ++w) t += s_part[w];
    return t;
}

// ---------------------------------------------------------------------------------------------
// Uniform Laplacian.  y_v = x_v - (1/deg_v) sum_{n in N(v)} x_n; loss = sum_v |y_v|^2;
// grad_v = 2 (y_v - sum_{n in N(v)} y_n / deg_n)   (the neighbour relation is symmetric).
// One workgroup per mesh; y and 1/deg live in the LDS (4 V floats).  A vertex without
// neighbours has y = 0.
__global__ __launch_bounds__(kRegThreads) void laplacian_loss_kernel(
    const float *__restrict__ x, const int32_t *__restrict__ row_start, const int32_t *__restrict__ nbr,
    int V, float *__restrict__ loss, float *__restrict__ grad)
{
    extern __shared__ float s_dyn[];
    __shared__ float s_part[kRegThreads / 64];
    float *s_y = s_dyn, *s_inv = s_dyn + 3 * (size_t)V;
    const int b = blockIdx.x;
    const float *xb = x + (size_t)b * V * 3;
    float acc = 0.f;
    for (int v = threadIdx.x; v < V; v += kRegThreads) {
        const int s0 = row_start[v], s1 = row_start[v + 1];
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int k = s0; k < s1; ++k) {
            const float *p = xb + 3 * (size_t)clampi(nbr[k], V);
            sx += p[0];
            sy += p[1];
            sz += p[2];
        }
        const float inv = s1 > s0 ? 1.f / (float)(s1 - s0) : 0.f;
        const float live = s1 > s0 ? 1.f : 0.f;
        const float yx = live * xb[3 * v] - sx * inv;
        const float yy = live * xb[3 * v + 1] - sy * inv;
        const float yz = live * xb[3 * v + 2] - sz * inv;
        s_y[3 * v] = yx;
        s_y[3 * v + 1] = yy;
        s_y[3 * v + 2] = yz;
        s_inv[v] = inv;
        acc += yx * yx + yy * yy + yz * yz;
    }
    const float total = block_sum(acc, s_part);      // its barriers also publish s_y / s_inv
    if (threadIdx.x == 0) loss[b] = total;
    if (grad == nullptr) return;
    float *gb = grad + (size_t)b * V * 3;
    for (int v = threadIdx.x; v < V; v += kRegThreads) {
        const int s0 = row_start[v], s1 = row_start[v + 1];
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int k = s0; k < s1; ++k) {
            const int n = clampi(nbr[k], V);
            const float w = s_inv[n];
            sx += w * s_y[3 * n];
            sy += w * s_y[3 * n + 1];
            sz += w * s_y[3 * n + 2];
        }
        gb[3 * v] = 2.f * (s_y[3 * v] - sx);
        gb[3 * v + 1] = 2.f * (s_y[3 * v + 1] - sy);
        gb[3 * v + 2] = 2.f * (s_y[3 * v + 2] - sz);
    }
}

// ---------------------------------------------------------------------------------------------
// Flatten loss of one interior edge (v0,v1) with opposite vertices v2, v3:
//   a = v1 - v0;  for k in {2,3}: b = vk - v0, c_k = b - a (a.b)/(|a|^2 + eps),
//   l_k = sqrt(|b|^2 + eps) sqrt(1 - cos_k^2 + eps), cos_k = a.b / (sqrt(|a|^2+eps) sqrt(|b|^2+eps) + eps)
//   cos = c_2 . c_3 / (l_2 l_3 + eps);  value = (cos + 1)^2.
struct V3 {
    float x, y, z;
};
__device__ inline V3 ld3(const float *p) { return {p[0], p[1], p[2]}; }
__device__ inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

struct Altitude {
    V3 b, c;
    float ab, bl1, cosk, sink, dk, t, l;
};

__device__ inline Altitude altitude(V3 a, float al2, float al1e, V3 b, float eps)
{
    Altitude r;
    r.b = b;
    r.bl1 = sqrtf(dot(b, b) + eps);
    r.ab = dot(a, b);
    r.dk = al1e * r.bl1 + eps;
    r.cosk = r.ab / r.dk;
    r.sink = sqrtf(1.f - r.cosk * r.cosk + eps);
    r.t = r.ab / (al2 + eps);
    r.c = b - r.t * a;
    r.l = r.bl1 * r.sink;
    return r;
}

// Reverse pass through one altitude: given dL/dc and dL/dl, adds to dL/da and returns dL/db.
__device__ inline V3 altitude_backward(const Altitude &r, V3 a, float al2, float al1e, float eps, V3 dc,
                                       float dl, V3 &da)
{
    V3 db = dc;
    da = da + (-r.t) * dc;
    const float dt = -dot(a, dc);
    float dbl1 = dl * r.sink;
    const float dsin = dl * r.bl1;
    const float dcosk = -dsin * r.cosk / r.sink;
    float dab = dcosk / r.dk;
    const float ddk = -dcosk * r.cosk / r.dk;
    const float dal1e = ddk * r.bl1;
    dbl1 += ddk * al1e;
    dab += dt / (al2 + eps);
    float dal2 = -dt * r.t / (al2 + eps);
    dal2 += dal1e / (2.f * al1e);
    db = db + (dbl1 / r.bl1) * r.b;
    da = da + dab * r.b;
    db = db + dab * a;
    da = da + (2.f * dal2) * a;
    return db;
}

// Value of one edge term; with GRAD also its gradient with respect to the four vertices.
template <bool GRAD>
__device__ inline float flatten_edge(V3 p0, V3 p1, V3 p2, V3 p3, float eps, V3 *g)
{
    const V3 a = p1 - p0;
    const float al2 = dot(a, a), al1e = sqrtf(al2 + eps);
    const Altitude r2 = altitude(a, al2, al1e, p2 - p0, eps);
    const Altitude r3 = altitude(a, al2, al1e, p3 - p0, eps);
    const float den = r2.l * r3.l + eps;
    const float cosd = dot(r2.c, r3.c) / den;
    const float value = (cosd + 1.f) * (cosd + 1.f);
    if (GRAD) {
        const float dcos = 2.f * (cosd + 1.f);
        const float dden = -dcos * cosd / den;
        V3 da = {0.f, 0.f, 0.f};
        const V3 db2 = altitude_backward(r2, a, al2, al1e, eps, (dcos / den) * r3.c, dden * r3.l, da);
        const V3 db3 = altitude_backward(r3, a, al2, al1e, eps, (dcos / den) * r2.c, dden * r2.l, da);
        g[1] = da;
        g[2] = db2;
        g[3] = db3;
        g[0] = (-1.f) * (da + db2 + db3);
    }
    return value;
}

// One workgroup per mesh.  Pass 1: a thread per edge, fixed-order sum -> loss[b].  Pass 2: a thread
// per vertex walks its incidence list (entries edge*4 + role, ascending) and re-evaluates those
// edges, so every vertex gradient is a fixed-order sum without a scatter.
__global__ __launch_bounds__(kRegThreads) void flatten_loss_kernel(
    const float *__restrict__ x, const int32_t *__restrict__ quads, const int32_t *__restrict__ inc_start,
    const int32_t *__restrict__ inc, int V, int E, float eps, float *__restrict__ loss,
    float *__restrict__ grad)
{
    __shared__ float s_part[kRegThreads / 64];
    const int b = blockIdx.x;
    const float *xb = x + (size_t)b * V * 3;
    float acc = 0.f;
    for (int e = threadIdx.x; e < E; e += kRegThreads) {
        const int32_t *q = quads + 4 * (size_t)e;
        acc += flatten_edge<false>(ld3(xb + 3 * clampi(q[0], V)), ld3(xb + 3 * clampi(q[1], V)),
                                   ld3(xb + 3 * clampi(q[2], V)), ld3(xb + 3 * clampi(q[3], V)), eps, nullptr);
    }
    const float total = block_sum(acc, s_part);
    if (threadIdx.x == 0) loss[b] = total;
    if (grad == nullptr) return;
    float *gb = grad + (size_t)b * V * 3;
    for (int v = threadIdx.x; v < V; v += kRegThreads) {
        V3 sum = {0.f, 0.f, 0.f};
        for (int k = inc_start[v]; k < inc_start[v + 1]; ++k) {
            const int entry = inc[k];
            const int e = clampi(entry >> 2, E), role = entry & 3;
            const int32_t *q = quads + 4 * (size_t)e;
            V3 g[4];
            flatten_edge<true>(ld3(xb + 3 * clampi(q[0], V)), ld3(xb + 3 * clampi(q[1], V)),
                               ld3(xb + 3 * clampi(q[2], V)), ld3(xb + 3 * clampi(q[3], V)), eps, g);
            const V3 lo = role & 1 ? g[1] : g[0], hi = role & 1 ? g[3] : g[2];
            sum = sum + (role & 2 ? hi : lo);
        }
        gb[3 * v] = sum.x;
        gb[3 * v + 1] = sum.y;
        gb[3 * v + 2] = sum.z;
    }
}

// ---------------------------------------------------------------------------------------------
// y[b,v,:] = sum_k values[vb,k] x[b, col[k], :] over the CSR row v, k ascending; a thread per (b,v).
__global__ __launch_bounds__(kSpmvThreads) void mesh_spmv_kernel(
    const float *__restrict__ values, const int32_t *__restrict__ row_start, const int32_t *__restrict__ col,
    const float *__restrict__ x, int V, int nnz, int values_batch, float *__restrict__ y)
{
    const int b = blockIdx.y;
    const int v = blockIdx.x * kSpmvThreads + threadIdx.x;
    if (v >= V) return;
    const float *xb = x + (size_t)b * V * 3;
    const float *vb = values + (values_batch > 1 ? (size_t)b * nnz : 0);
    float sx = 0.f, sy = 0.f, sz = 0.f;
    const int s0 = max(row_start[v], 0), s1 = min(row_start[v + 1], nnz);
    for (int k = s0; k < s1; ++k) {
        const float w = vb[k];
        const float *p = xb + 3 * (size_t)clampi(col[k], V);
        sx += w * p[0];
        sy += w * p[1];
        sz += w * p[2];
    }
    float *yb = y + ((size_t)b * V + v) * 3;
    yb[0] = sx;
    yb[1] = sy;
    yb[2] = sz;
}

inline int launch_status()
{
    return hipGetLastError() == hipSuccess ? SV3D_OK : SV3D_ERR_LAUNCH;
}

}  // namespace

extern "C" int sv3d_laplacian_loss(const float *vertices, const int32_t *row_start, const int32_t *nbr,
                                   int32_t B, int32_t V, float *loss, float *grad_vertices, void *stream)
{
    if (!vertices || !row_start || !nbr || !loss || B <= 0 || V <= 0) return SV3D_ERR_INVALID_ARG;
    const size_t lds = (size_t)4 * V * sizeof(float);
    if (lds > kLdsMax) return SV3D_ERR_UNSUPPORTED;
    if (lds > kLdsDefault &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(&laplacian_loss_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return SV3D_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(laplacian_loss_kernel, dim3(B), dim3(kRegThreads), lds, (hipStream_t)stream, vertices,
                       row_start, nbr, V, loss, grad_vertices);
    return launch_status();
}

extern "C" int sv3d_flatten_loss(const float *vertices, const int32_t *quads, const int32_t *inc_start,
                                 const int32_t *inc, int32_t B, int32_t V, int32_t E, float eps, float *loss,
                                 float *grad_vertices, void *stream)
{
    if (!vertices || !quads || !loss || B <= 0 || V <= 0 || E <= 0) return SV3D_ERR_INVALID_ARG;
    if (grad_vertices && (!inc_start || !inc)) return SV3D_ERR_INVALID_ARG;
    if ((int64_t)E * 4 + 3 > INT32_MAX) return SV3D_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(flatten_loss_kernel, dim3(B), dim3(kRegThreads), 0, (hipStream_t)stream, vertices, quads,
                       inc_start, inc, V, E, eps, loss, grad_vertices);
    return launch_status();
}

extern "C" int sv3d_mesh_spmv(const float *values, const int32_t *row_start, const int32_t *col, const float *x,
                              int32_t B, int32_t V, int32_t nnz, int32_t values_batch, float *y, void *stream)
{
    if (!values || !row_start || !col || !x || !y || B <= 0 || V <= 0 || nnz < 0) return SV3D_ERR_INVALID_ARG;
    if (values_batch != 1 && values_batch != B) return SV3D_ERR_INVALID_ARG;
    if (B > 65535) return SV3D_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(mesh_spmv_kernel, dim3((V + kSpmvThreads - 1) / kSpmvThreads, B), dim3(kSpmvThreads), 0,
                       (hipStream_t)stream, values, row_start, col, x, V, nnz, values_batch, y);
    return launch_status();
}
