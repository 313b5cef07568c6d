// Mesh glue kernels for gfx950: the `vertices[faces]` gather that soft_renderer's Mesh
// performs for every render (reached from nnutils/softras.py:64,72,77) fused with the
// renderer's affine pre-stage (y flip of softras.py:60, look_at z shift), and its
// backward as a deterministic gather-form sum.  Both are HBM/latency bound.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <limits.h>

#include "bucket_sort.h"
#include "camera.h"
#include "sv3d_hip.h"

namespace {

using sv3d_dev::Affine;
using sv3d_dev::Camera;

constexpr int kThreads = 256;

// one thread per (b, f, corner): 12 B gathered, 12 B stored contiguously
__global__ __launch_bounds__(kThreads) void face_gather_kernel(
    const float *__restrict__ values, const int32_t *__restrict__ faces, int B, int V, int F,
    int faces_batch, Affine a, float *__restrict__ out)
{
    const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
    const long per = (long)F * 3;
    if (gid >= (long)B * per) return;
    const int b = (int)(gid / per);
    const long slot = gid - (long)b * per;
    const int32_t *fb = faces + (faces_batch > 1 ? (long)b * per : 0);
    int v = fb[slot];
    v = min(max(v, 0), V - 1);
    const float *src = values + ((long)b * V + v) * 3;
    float *dst = out + gid * 3;
    dst[0] = src[0] * a.s0 + a.t0;
    dst[1] = src[1] * a.s1 + a.t1;
    dst[2] = src[2] * a.s2 + a.t2;
}

// one thread per (b, v); the image's corner->vertex table is streamed through LDS and every
// thread scans it in ascending slot order, adding the slots that reference its vertex.  The
// sum order is fixed, there are no atomics and no index has to be built beforehand; the scan
// costs 3F integer compares per vertex, which stays far below the rasterizer's own work.
constexpr int kScanTile = 3072;      // corner slots per LDS tile (12 KiB)

__global__ __launch_bounds__(kThreads) void face_scatter_kernel(
    const float *__restrict__ grad_out, const int32_t *__restrict__ faces, int B, int V, int F,
    int faces_batch, Affine a, float *__restrict__ grad_values)
{
    __shared__ int4 s_slots[kScanTile / 4];
    const int b = blockIdx.y;
    const int v = blockIdx.x * kThreads + threadIdx.x;
    const int nslots = F * 3;
    const int32_t *fb = faces + (faces_batch > 1 ? (long)b * nslots : 0);
    const float *g = grad_out + (long)b * nslots * 3;
    float x = 0.f, y = 0.f, z = 0.f;
    for (int t0 = 0; t0 < nslots; t0 += kScanTile) {
        const int tn = min(kScanTile, nslots - t0);
        __syncthreads();
        int *flat = reinterpret_cast<int *>(s_slots);
        for (int i = threadIdx.x; i < kScanTile; i += kThreads)
            flat[i] = i < tn ? min(max(fb[t0 + i], 0), V - 1) : -1;      // clamped like the gather's reads
        __syncthreads();
        const int n4 = (tn + 3) / 4;
        for (int i = 0; i < n4; ++i) {
            const int4 q = s_slots[i];
            const bool h0 = q.x == v, h1 = q.y == v, h2 = q.z == v, h3 = q.w == v;
            if (h0 | h1 | h2 | h3) {
                const float *gs = g + 3 * (long)(t0 + 4 * i);
                if (h0) { x += gs[0]; y += gs[1]; z += gs[2]; }
                if (h1) { x += gs[3]; y += gs[4]; z += gs[5]; }
                if (h2) { x += gs[6]; y += gs[7]; z += gs[8]; }
                if (h3) { x += gs[9]; y += gs[10]; z += gs[11]; }
            }
        }
    }
    if (v < V) {
        float *dst = grad_values + ((long)b * V + v) * 3;
        dst[0] = x * a.s0;
        dst[1] = y * a.s1;
        dst[2] = z * a.s2;
    }
}

// Camera + gather in one pass (nnutils/softras.py:58-61 followed by `vertices[faces]`): one
// thread per (b, f, corner) projects the object-space vertex it gathers, so the projected
// [B,V,3] tensor is never materialised.
__global__ __launch_bounds__(kThreads) void project_gather_kernel(
    const float *__restrict__ vertices, const float *__restrict__ cams,
    const int32_t *__restrict__ faces, int B, int V, int F, int faces_batch, float offset_z, Affine a,
    float *__restrict__ out)
{
    const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
    const long per = (long)F * 3;
    if (gid >= (long)B * per) return;
    const int b = (int)(gid / per);
    const long slot = gid - (long)b * per;
    const int32_t *fb = faces + (faces_batch > 1 ? (long)b * per : 0);
    int v = fb[slot];
    v = min(max(v, 0), V - 1);
    const Camera c = sv3d_dev::load_camera(cams + (long)b * 7, offset_z);
    const float *src = vertices + ((long)b * V + v) * 3;
    float *dst = out + gid * 3;
    float ox, oy, oz;
    sv3d_dev::camera_project(c, a, src[0], src[1], src[2], ox, oy, oz);
    dst[0] = ox; dst[1] = oy; dst[2] = oz;
}

// Backward of the camera for one vertex: g = dL/d(projected point) after the output scale.
// Returns dL/dX and adds this vertex's terms of dL/d(scale, tx, ty, qw, qx, qy, qz) to acc.
__device__ inline void camera_backward(const Camera &c, float X, float Y, float Z, float gx, float gy,
                                       float gz, float &dX, float &dY, float &dZ, float acc[7])
{
    dX = c.s * (c.m00 * gx + c.m10 * gy + c.m20 * gz);
    dY = c.s * (c.m01 * gx + c.m11 * gy + c.m21 * gz);
    dZ = c.s * (c.m02 * gx + c.m12 * gy + c.m22 * gz);
    float rx, ry, rz;
    sv3d_dev::camera_rotate(c, X, Y, Z, rx, ry, rz);
    acc[0] += gx * rx + gy * ry + gz * rz;
    acc[1] += gx;
    acc[2] += gy;
    // d(R X)/dw = 2 (w X + u x X);  d(R X)/du_i = 2 (-u_i X + X_i u + (u.X) e_i + w e_i x X)
    const float gX = gx * X + gy * Y + gz * Z;
    const float gu = gx * c.ux + gy * c.uy + gz * c.uz;
    const float uX = c.ux * X + c.uy * Y + c.uz * Z;
    const float cx = c.uy * Z - c.uz * Y, cy = c.uz * X - c.ux * Z, cz = c.ux * Y - c.uy * X;   // u x X
    const float kx = Y * gz - Z * gy, ky = Z * gx - X * gz, kz = X * gy - Y * gx;               // X x g
    const float s2 = 2.f * c.s;
    acc[3] += s2 * (c.w * gX + gx * cx + gy * cy + gz * cz);
    acc[4] += s2 * (-gX * c.ux + gu * X + uX * gx + c.w * kx);
    acc[5] += s2 * (-gX * c.uy + gu * Y + uX * gy + c.w * ky);
    acc[6] += s2 * (-gX * c.uz + gu * Z + uX * gz + c.w * kz);
}

using sv3d_dev::build_buckets;
using sv3d_dev::kSortThreads;
using sv3d_dev::for_each_slot_ascending;

// Sum of 7 per-thread values over a 1024-thread block in a fixed order; thread i < 7 stores
// total i to dst[i].

__device__ inline void block_sum7(float acc[7], float (*s_red)[8], float *dst)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        float v = acc[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        if (lane == 0) s_red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        float t = 0.f;
        for (int w = 0; w < kSortThreads / 64; ++w) t += s_red[w][threadIdx.x];
        dst[threadIdx.x] = t;
    }
}

// Backward of the gather (and, with PROJECT, of the camera in front of it): one 1024-thread
// block per image groups the image's 3F corner slots by vertex with an LDS counting sort, then
// one thread per vertex adds its slots' gradients in ascending slot order.  The sort places
// slots with LDS integer atomics, so the order inside a bucket varies; picking the next larger
// slot id on every step makes the float sum order -- and the result bits -- fixed.  No float
// atomics, no index kept between calls.  Dynamic LDS: (2V + 3F) ints.
template <bool PROJECT>
__global__ __launch_bounds__(kSortThreads) void mesh_scatter_kernel(
    const float *__restrict__ grad_out, const int32_t *__restrict__ faces, int B, int V, int F,
    int faces_batch, Affine a, const float *__restrict__ vertices, const float *__restrict__ cams,
    float offset_z, float *__restrict__ grad_values, float *__restrict__ grad_cams)
{
    extern __shared__ int smem[];
    __shared__ int s_wave[kSortThreads / 64];
    __shared__ float s_red[kSortThreads / 64][8];
    int *fill = smem;                 // per vertex: slot count, then running fill pointer
    int *start = smem + V;            // per vertex: first entry of its bucket
    int *bucket = smem + 2 * V;       // slot ids grouped by vertex
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nslots = F * 3;
    const int32_t *fb = faces + (faces_batch > 1 ? (long)b * nslots : 0);

    build_buckets(fb, V, nslots, fill, start, bucket, s_wave);

    Camera c;
    if (PROJECT) c = sv3d_dev::load_camera(cams + (long)b * 7, offset_z);
    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float *g = grad_out + (long)b * nslots * 3;
    for (int v = tid; v < V; v += kSortThreads) {
        const int s0 = start[v], s1 = fill[v];
        float x = 0.f, y = 0.f, z = 0.f;
        for_each_slot_ascending(bucket, s0, s1, fb, v, V, nslots, [&](int slot) {
            const float *gs = g + 3 * (long)slot;
            x += gs[0]; y += gs[1]; z += gs[2];
        });
        x *= a.s0; y *= a.s1; z *= a.s2;
        float *dst = grad_values + ((long)b * V + v) * 3;
        if (PROJECT) {
            const float *X = vertices + ((long)b * V + v) * 3;
            float dX, dY, dZ;
            camera_backward(c, X[0], X[1], X[2], x, y, z, dX, dY, dZ, acc);
            dst[0] = dX; dst[1] = dY; dst[2] = dZ;
        } else {
            dst[0] = x; dst[1] = y; dst[2] = z;
        }
    }
    if (PROJECT) block_sum7(acc, s_red, grad_cams + (long)b * 7);
}

// Camera backward on its own, for meshes whose sort tables exceed the LDS: grad_values holds
// dL/d(projected vertex) on entry (already scaled) and dL/d(vertex) on exit.
__global__ __launch_bounds__(kSortThreads) void camera_backward_kernel(
    int V, const float *__restrict__ vertices, const float *__restrict__ cams, float offset_z,
    float *__restrict__ grad_values, float *__restrict__ grad_cams)
{
    __shared__ float s_red[kSortThreads / 64][8];
    const int b = blockIdx.x;
    const Camera c = sv3d_dev::load_camera(cams + (long)b * 7, offset_z);
    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int v = threadIdx.x; v < V; v += kSortThreads) {
        const float *X = vertices + ((long)b * V + v) * 3;
        float *dst = grad_values + ((long)b * V + v) * 3;
        float dX, dY, dZ;
        camera_backward(c, X[0], X[1], X[2], dst[0], dst[1], dst[2], dX, dY, dZ, acc);
        dst[0] = dX; dst[1] = dY; dst[2] = dZ;
    }
    block_sum7(acc, s_red, grad_cams + (long)b * 7);
}

constexpr size_t kSortLdsDefault = 48 * 1024;    // dynamic LDS a launch asks for unprompted; the kernels' static
                                                 // arrays (under 1 KiB) count against the same 64 KiB default
constexpr size_t kSortLdsMax = 150 * 1024;       // of the CU's 160 KiB, less the static arrays

inline size_t sort_lds_bytes(int V, int F) { return ((size_t)2 * V + (size_t)3 * F) * sizeof(int); }

// Launches the sorted scatter when its tables fit the LDS; false = caller takes the scan kernel.
template <bool PROJECT>
bool launch_sorted_scatter(const float *grad_out, const int32_t *faces, int B, int V, int F,
                           int faces_batch, Affine a, const float *vertices, const float *cams,
                           float offset_z, float *grad_values, float *grad_cams, hipStream_t s)
{
    const size_t lds = sort_lds_bytes(V, F);
    if (lds > kSortLdsMax) return false;
    if (lds > kSortLdsDefault &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(&mesh_scatter_kernel<PROJECT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    hipLaunchKernelGGL(mesh_scatter_kernel<PROJECT>, dim3(B), dim3(kSortThreads), lds, s, grad_out, faces,
                       B, V, F, faces_batch, a, vertices, cams, offset_z, grad_values, grad_cams);
    return true;
}

// Gather / scatter of C <= 8 scalar channels per vertex (part maps): out[b,f,k,:] = values[b, faces[fb,f,k], :].
__global__ __launch_bounds__(kThreads) void face_gather_n_kernel(
    const float *__restrict__ values, const int32_t *__restrict__ faces, int B, int V, int F,
    int faces_batch, int C, float *__restrict__ out)
{
    const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
    const long per = (long)F * 3;
    if (gid >= (long)B * per) return;
    const int b = (int)(gid / per);
    const int32_t *fb = faces + (faces_batch > 1 ? (long)b * per : 0);
    const int v = min(max(fb[gid - (long)b * per], 0), V - 1);
    const float *src = values + ((long)b * V + v) * C;
    float *dst = out + gid * C;
    for (int c = 0; c < C; ++c) dst[c] = src[c];
}

__global__ __launch_bounds__(kSortThreads) void mesh_scatter_n_kernel(
    const float *__restrict__ grad_out, const int32_t *__restrict__ faces, int B, int V, int F,
    int faces_batch, int C, float *__restrict__ grad_values)
{
    extern __shared__ int smem[];
    __shared__ int s_wave[kSortThreads / 64];
    int *fill = smem, *start = smem + V, *bucket = smem + 2 * V;
    const int b = blockIdx.x, nslots = F * 3;
    const int32_t *fb = faces + (faces_batch > 1 ? (long)b * nslots : 0);
    build_buckets(fb, V, nslots, fill, start, bucket, s_wave);
    const float *g = grad_out + (long)b * nslots * C;
    for (int v = threadIdx.x; v < V; v += kSortThreads) {
        const int s0 = start[v], s1 = fill[v];
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for_each_slot_ascending(bucket, s0, s1, fb, v, V, nslots, [&](int slot) {
            const float *gs = g + (long)slot * C;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < C) a[c] += gs[c];
        });
        float *dst = grad_values + ((long)b * V + v) * C;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < C) dst[c] = a[c];
    }
}

Affine make_affine(const float *scale, const float *shift)
{
    Affine a = {1.f, 1.f, 1.f, 0.f, 0.f, 0.f};
    if (scale) { a.s0 = scale[0]; a.s1 = scale[1]; a.s2 = scale[2]; }
    if (shift) { a.t0 = shift[0]; a.t1 = shift[1]; a.t2 = shift[2]; }
    return a;
}

}  // namespace

extern "C" int sv3d_face_gather(const float *values, const int32_t *faces, int32_t B, int32_t V,
                                int32_t F, int32_t faces_batch, const float *scale,
                                const float *shift, float *out, void *stream)
{
    if (!values || !faces || !out) return SV3D_ERR_INVALID_ARG;
    if (B <= 0 || V <= 0 || F <= 0) return SV3D_ERR_INVALID_ARG;
    if (faces_batch != 1 && faces_batch != B) return SV3D_ERR_INVALID_ARG;
    const long n = (long)B * F * 3;
    if (n > 0x7fffffffL * (long)kThreads / 2) return SV3D_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(face_gather_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)),
                       dim3(kThreads), 0, static_cast<hipStream_t>(stream), values, faces, B, V, F,
                       faces_batch, make_affine(scale, shift), out);
    return hipGetLastError() == hipSuccess ? SV3D_OK : SV3D_ERR_LAUNCH;
}

extern "C" int sv3d_face_scatter(const float *grad_out, const int32_t *faces, int32_t B, int32_t V,
                                 int32_t F, int32_t faces_batch, const float *scale,
                                 float *grad_values, void *stream)
{
    if (!grad_out || !faces || !grad_values) return SV3D_ERR_INVALID_ARG;
    if (B <= 0 || V <= 0 || F <= 0) return SV3D_ERR_INVALID_ARG;
    if (faces_batch != 1 && faces_batch != B) return SV3D_ERR_INVALID_ARG;
    if (B > 65535) return SV3D_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Affine a = make_affine(scale, nullptr);
    if (!launch_sorted_scatter<false>(grad_out, faces, B, V, F, faces_batch, a, nullptr, nullptr, 0.f,
                                      grad_values, nullptr, s))
        hipLaunchKernelGGL(face_scatter_kernel, dim3((V + kThreads - 1) / kThreads, B), dim3(kThreads), 0,
                           s, grad_out, faces, B, V, F, faces_batch, a, grad_values);
    return hipGetLastError() == hipSuccess ? SV3D_OK : SV3D_ERR_LAUNCH;
}

static int check_mesh(const sv3d_mesh_input *m, int32_t B, int32_t F)
{
    if (!m || !m->vertices || !m->cams || !m->faces) return SV3D_ERR_INVALID_ARG;
    if (B <= 0 || F <= 0 || m->num_vertices <= 0) return SV3D_ERR_INVALID_ARG;
    if (m->faces_batch != 1 && m->faces_batch != B) return SV3D_ERR_INVALID_ARG;
    if (B > 65535) return SV3D_ERR_UNSUPPORTED;
    if ((long)B * F * 3 > 0x7fffffffL * (long)kThreads / 2) return SV3D_ERR_UNSUPPORTED;
    return SV3D_OK;
}

extern "C" int sv3d_project_gather(const sv3d_mesh_input *m, int32_t B, int32_t F,
                                   float *face_vertices, void *stream)
{
    const int st = check_mesh(m, B, F);
    if (st != SV3D_OK) return st;
    if (!face_vertices) return SV3D_ERR_INVALID_ARG;
    const long n = (long)B * F * 3;
    hipLaunchKernelGGL(project_gather_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)),
                       dim3(kThreads), 0, static_cast<hipStream_t>(stream), m->vertices, m->cams,
                       m->faces, B, m->num_vertices, F, m->faces_batch, m->offset_z,
                       make_affine(m->scale, m->shift), face_vertices);
    return hipGetLastError() == hipSuccess ? SV3D_OK : SV3D_ERR_LAUNCH;
}

extern "C" int sv3d_project_scatter(const sv3d_mesh_input *m, int32_t B, int32_t F,
                                    const float *grad_face_vertices, float *grad_vertices,
                                    float *grad_cams, void *stream)
{
    const int st = check_mesh(m, B, F);
    if (st != SV3D_OK) return st;
    if (!grad_face_vertices || !grad_vertices || !grad_cams) return SV3D_ERR_INVALID_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int V = m->num_vertices;
    const Affine a = make_affine(m->scale, nullptr);
    if (!launch_sorted_scatter<true>(grad_face_vertices, m->faces, B, V, F, m->faces_batch, a,
                                     m->vertices, m->cams, m->offset_z, grad_vertices, grad_cams, s)) {
        hipLaunchKernelGGL(face_scatter_kernel, dim3((V + kThreads - 1) / kThreads, B), dim3(kThreads), 0,
                           s, grad_face_vertices, m->faces, B, V, F, m->faces_batch, a, grad_vertices);
        hipLaunchKernelGGL(camera_backward_kernel, dim3(B), dim3(kSortThreads), 0, s, V, m->vertices,
                           m->cams, m->offset_z, grad_vertices, grad_cams);
    }
    return hipGetLastError() == hipSuccess ? SV3D_OK : SV3D_ERR_LAUNCH;
}

extern "C" int sv3d_face_gather_n(const float *values, const int32_t *faces, int32_t B, int32_t V,
                                  int32_t F, int32_t faces_batch, int32_t C, float *out, void *stream)
{
    if (!values || !faces || !out) return SV3D_ERR_INVALID_ARG;
    if (B <= 0 || V <= 0 || F <= 0 || C <= 0) return SV3D_ERR_INVALID_ARG;
    if (faces_batch != 1 && faces_batch != B) return SV3D_ERR_INVALID_ARG;
    if (C > 8) return SV3D_ERR_UNSUPPORTED;
    const long n = (long)B * F * 3;
    if (n > 0x7fffffffL * (long)kThreads / 2) return SV3D_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(face_gather_n_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads),
                       0, static_cast<hipStream_t>(stream), values, faces, B, V, F, faces_batch, C, out);
    return hipGetLastError() == hipSuccess ? SV3D_OK : SV3D_ERR_LAUNCH;
}

extern "C" int sv3d_face_scatter_n(const float *grad_out, const int32_t *faces, int32_t B, int32_t V,
                                   int32_t F, int32_t faces_batch, int32_t C, float *grad_values,
                                   void *stream)
{
    if (!grad_out || !faces || !grad_values) return SV3D_ERR_INVALID_ARG;
    if (B <= 0 || V <= 0 || F <= 0 || C <= 0) return SV3D_ERR_INVALID_ARG;
    if (faces_batch != 1 && faces_batch != B) return SV3D_ERR_INVALID_ARG;
    if (C > 8 || B > 65535) return SV3D_ERR_UNSUPPORTED;
    const size_t lds = sort_lds_bytes(V, F);
    if (lds > kSortLdsMax) return SV3D_ERR_UNSUPPORTED;
    if (lds > kSortLdsDefault &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(&mesh_scatter_n_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return SV3D_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(mesh_scatter_n_kernel, dim3(B), dim3(kSortThreads), lds,
                       static_cast<hipStream_t>(stream), grad_out, faces, B, V, F, faces_batch, C, grad_values);
    return hipGetLastError() == hipSuccess ? SV3D_OK : SV3D_ERR_LAUNCH;
}
