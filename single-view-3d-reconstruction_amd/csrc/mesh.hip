// Mesh glue kernels for gfx950: the `vertices[faces]` gather that soft_renderer's Mesh
// performs for every render (reached from nnutils/softras.py:64,72,77) fused with the
// renderer's affine pre-stage (y flip of softras.py:60, look_at z shift), and its
// backward as a deterministic gather-form sum.  Both are HBM/latency bound.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sv3d_hip.h"

namespace {

constexpr int kThreads = 256;

struct Affine {
    float s0, s1, s2, t0, t1, t2;
};

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
            flat[i] = i < tn ? fb[t0 + i] : -1;
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
    hipLaunchKernelGGL(face_scatter_kernel, dim3((V + kThreads - 1) / kThreads, B), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), grad_out, faces, B, V, F, faces_batch,
                       make_affine(scale, nullptr), grad_values);
    return hipGetLastError() == hipSuccess ? SV3D_OK : SV3D_ERR_LAUNCH;
}
