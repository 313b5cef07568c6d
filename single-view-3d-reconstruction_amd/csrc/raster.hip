// Soft rasterizer for gfx950 (MI355X): forward and backward.
//
// Replaces the soft_renderer CUDA extension reached from nnutils/softras.py:64,75,77.
// The arithmetic is FP32 VALU work (no contraction, so no MFMA).  Each pass is organised
// around its OUTPUT so that nothing is scattered:
//
//   setup     one thread per face: pixel-independent coefficients ("face record", 32 floats)
//             and the face's pixel bounding box (the formula's own cull box).
//   forward   one workgroup per 16x16 pixel tile, one pixel per lane, one wave per 8x8
//             block.  Faces whose box touches the tile are compacted IN FACE ORDER into an
//             LDS list, their records staged in LDS, and each wave walks the sub-list that
//             touches its block (ballot over 64 records at a time), reading every record by
//             broadcast ds_read_b128.  A pixel reduces over faces in registers.
//   backward  one WAVE per face, lanes enumerate the pixels of the face's box 64 at a time.
//             A face reduces over pixels in registers; one fixed-tree DPP reduction per face
//             at the end; each gradient element is written exactly once.  No atomics to
//             global memory, bitwise repeatable.
//
// Numerics: distances are evaluated in pixel-relative coordinates (vertex minus pixel
// centre) as point-to-segment distances, which is the same quantity the barycentric
// `t - w` algebra of SoftRas produces but without its cancellation; depth enters the
// z-softmax as zp - z_ref (z_ref = first vertex depth of the image) so that the 1/gamma
// amplification acts on a small, exactly representable offset.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "camera.h"
#include "sv3d_hip.h"
#include "tile_order.h"

#ifdef SV3D_TRACE
// Diagnostic build only: per-workgroup start / end stamps and phase stamps of the forward kernel.
__device__ unsigned long long g_trace[4 * 16384];
__device__ unsigned long long g_phase[8 * 16384];
#define SV3D_WG_ID ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x)
#define SV3D_STAMP(i) do { if (threadIdx.x == 0 && SV3D_WG_ID < 16384) g_phase[SV3D_WG_ID * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int sv3d_debug_read_trace(void *host_out, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_trace), bytes, 0, hipMemcpyDeviceToHost);
}
extern "C" int sv3d_debug_read_phase(void *host_out, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_phase), bytes, 0, hipMemcpyDeviceToHost);
}
#else
#define SV3D_STAMP(i) do { } while (0)
#endif

namespace {

constexpr int kThreads = 256;
constexpr int kRec = 32;          // floats per face record (128 B, eight float4)
constexpr int kRec4 = kRec / 4;
constexpr int kTile = 16;         // tile side in raster pixels
#ifndef SV3D_SPLIT_MAX_TILES
#define SV3D_SPLIT_MAX_TILES 2048               // two waves per block up to this many tiles per launch
#endif
#ifndef SV3D_SPLIT_MAX_TILES_COLOUR
#define SV3D_SPLIT_MAX_TILES_COLOUR 3072        // ... in the colour modes
#endif
#ifndef SV3D_DIRECT_BOXES
#define SV3D_DIRECT_BOXES 1
#endif
#ifndef SV3D_SPLIT_ADJACENT
#define SV3D_SPLIT_ADJACENT 1
#endif
#ifndef SV3D_SPIRAL
#define SV3D_SPIRAL 1
#endif
#ifndef SV3D_LIST_CAP
#define SV3D_LIST_CAP 128
#endif
#ifndef SV3D_EARLY_EXIT
#define SV3D_EARLY_EXIT 1
#endif
// per-tile face lists (see raster_tile_lists_kernel and raster_setup_lists_kernel)
#ifndef SV3D_TILED
#define SV3D_TILED 1
#endif
#ifndef SV3D_TILED_TWO_BLOCKS_MIN_TILES
#define SV3D_TILED_TWO_BLOCKS_MIN_TILES 12288   // two blocks per wave from this many tiles per launch
#endif
#ifndef SV3D_TILED_SHARED_MIN_TILES
#define SV3D_TILED_SHARED_MIN_TILES 1536            // from this many tiles on a tile's four waves share all its hits
#endif
#ifndef SV3D_TILED_SHARED_MAX_TILES
#define SV3D_TILED_SHARED_MAX_TILES 12287           // ... up to this many, for meshes of four faces per tile or more
#endif
#ifndef SV3D_TILED_FUSED_MAX_TILES
#define SV3D_TILED_FUSED_MAX_TILES 4096             // at most this many tiles: setup and lists in one launch
#endif
#ifndef SV3D_TILED_MAX_FACES_PER_TILE
#define SV3D_TILED_MAX_FACES_PER_TILE 8             // faces of the mesh per tile of an image, at most
#endif
#ifndef SV3D_BWD_UNROLL
#define SV3D_BWD_UNROLL 1                           // silhouette backward: two steps per loop trip
#endif
#ifndef SV3D_BINNING_LISTS_MIN_TILES
#define SV3D_BINNING_LISTS_MIN_TILES 4096           // other launches above this many tiles: lists instead of binning
#endif
#ifndef SV3D_TILED_MIN_TILES
#define SV3D_TILED_MIN_TILES 4096                   // silhouette launches above this many tiles: lists from a kernel of their own
#endif
constexpr int kListCap = SV3D_LIST_CAP;   // faces staged in LDS at a time (128 B each)
constexpr int kRecPad4 = kRec4 + 1;       // LDS record stride (144 B): lane-strided reads are conflict free

// Colour channels of a texture mode: RGB, or eight scalar per-corner channels for VERTEX8.
__host__ __device__ constexpr int mode_channels(int mode) { return mode == SV3D_TEX_VERTEX8 ? 8 : 3; }
__host__ __device__ constexpr bool mode_vertex(int mode) { return mode == SV3D_TEX_VERTEX || mode == SV3D_TEX_VERTEX8; }

struct DevParams {
    int B, F, S, aa, mode, R, double_side;
    int TB;                       // texture sets: image b reads set b % TB (TB == B: one per image)
    float bg0, bg1, bg2;
    float near_plane, far_plane, eps;
    float neg_l2e_over_sigma;     // -log2(e) / sigma
    float inv_sigma;
    float threshold;              // ln(1/dist_eps - 1) * sigma
    float margin;                 // sqrt(threshold)
    float visit_threshold;        // squared distance beyond which the forward need not visit a pair
    float l2e_over_gamma;         // log2(e) / gamma
    float inv_gamma;
    float inv_range;              // 1 / (far - near)
    float ssum0;                  // exp(eps / gamma): weight of the background term
    float inv_S;                  // 1 / S
    int alpha_fast;               // dist_eps so small that 1 + exp(-threshold/sigma) == 1 in fp32
    int alpha_only;               // silhouette mode: the image (and its gradient) is the alpha plane alone
    unsigned inv_F;               // floor(2^32 / F): the image of a face index by a multiply, see image_of_face
    float c_scale;                // sqrt(log2(e) / sigma): lengths times this, squared, are exponents to base 2
    float visit_r;                // sqrt(visit_threshold) * c_scale
};

// A parameter block read field by field from constant memory (the argument block of a kernel);
// fields the caller never uses are never loaded.
typedef const DevParams __attribute__((address_space(4))) *ConstantParams;
__device__ inline DevParams copy_params(ConstantParams p)
{
    DevParams d;
    d.B = p->B; d.F = p->F; d.S = p->S; d.aa = p->aa; d.mode = p->mode; d.R = p->R; d.double_side = p->double_side;
    d.TB = p->TB;
    d.bg0 = p->bg0; d.bg1 = p->bg1; d.bg2 = p->bg2;
    d.near_plane = p->near_plane; d.far_plane = p->far_plane; d.eps = p->eps;
    d.neg_l2e_over_sigma = p->neg_l2e_over_sigma; d.inv_sigma = p->inv_sigma; d.threshold = p->threshold;
    d.margin = p->margin; d.visit_threshold = p->visit_threshold; d.l2e_over_gamma = p->l2e_over_gamma;
    d.inv_gamma = p->inv_gamma; d.inv_range = p->inv_range; d.ssum0 = p->ssum0; d.inv_S = p->inv_S;
    d.alpha_fast = p->alpha_fast; d.alpha_only = p->alpha_only; d.inv_F = p->inv_F;
    d.c_scale = p->c_scale; d.visit_r = p->visit_r;
    return d;
}

// b = gf / F for 0 <= gf < B * F <= 2^29, without the ~25 instructions of an integer division (the
// backward's waves start with it, eight to a SIMD): the multiply gives gf / F or one less.
__device__ inline int image_of_face(int gf, int F, unsigned inv_F)
{
    int b = (int)__umulhi((unsigned)gf, inv_F);
    b += (gf - b * F >= F) ? 1 : 0;
    return b;
}

// Face record layout (float index):
//  0..5   p0x p0y p1x p1y p2x p2y
//  6..11  e0x e0y e1x e1y e2x e2y        e_k = p_{k+1} - p_k
// 12..14  1/|e_k|^2 (0 for a zero-length edge)
// 15      1/det, det = cross(p1-p0, p2-p0) clamped away from 0 by 1e-10
// 16..19  f0x f0y f1x f1y        f_k = e_k / det: cross(f_k, pixel - p_k) is the barycentric
//                                coordinate of the vertex opposite to edge k
// 20..22  1/z_k
// 23..25  (z_k - z_ref)/z_k
// 26      flags (bit 0: front facing)
// 27      z_ref
// 28..29  pixel box, four int16: col0 col1 | row0 row1   (also kept in a compact array)
// 30..31  f2x f2y
struct FaceRec {
    float p0x, p0y, p1x, p1y, p2x, p2y;
    float e0x, e0y, e1x, e1y, e2x, e2y;
    float il0, il1, il2, inv_det;
    float f0x, f0y, f1x, f1y, f2x, f2y;
    // Colour modes: 1/z_k and (z_k - z_ref)/z_k.  Silhouette launches have no use for depth and
    // keep -e_k / |e_k|^2 in these six floats instead (g0x, g0y, g1x, g1y, g2x, g2y in this
    // order): the edge parameter of a pixel is then one multiply-add less per edge.
    float iz0, iz1, iz2;
    float q0, q1, q2;
    int flags;
    float zref;
    unsigned box_x, box_y;        // the pixel box as in the compact array
};

__device__ inline FaceRec load_rec(const float4 *r)
{
    const float4 a = r[0], b = r[1], c = r[2], d = r[3], e = r[4], f = r[5], g = r[6], h = r[7];
    FaceRec o;
    o.p0x = a.x; o.p0y = a.y; o.p1x = a.z; o.p1y = a.w;
    o.p2x = b.x; o.p2y = b.y; o.e0x = b.z; o.e0y = b.w;
    o.e1x = c.x; o.e1y = c.y; o.e2x = c.z; o.e2y = c.w;
    o.il0 = d.x; o.il1 = d.y; o.il2 = d.z; o.inv_det = d.w;
    o.f0x = e.x; o.f0y = e.y; o.f1x = e.z; o.f1y = e.w; o.f2x = h.z; o.f2y = h.w;
    o.iz0 = f.x; o.iz1 = f.y; o.iz2 = f.z; o.q0 = f.w;
    o.q1 = g.x; o.q2 = g.y; o.flags = __float_as_int(g.z); o.zref = g.w;
    o.box_x = __float_as_uint(h.x); o.box_y = __float_as_uint(h.y);
    return o;
}

// The record of a wave-uniform face through the scalar cache, all 128 bytes requested at once and
// waited for once, in one statement.  (Written as loads of the record's float4s the compiler splits
// them into three batches near their uses with a wait after each; scalar loads return in any order,
// so every wait is for everything outstanding, and the batches become trips to memory in a row.
// Request and wait are not split into two statements with work between them: nothing could stop
// the register allocator from copying a register block that the load has not written yet.)
__device__ inline FaceRec load_rec_uniform(const float4 *r)
{
    typedef float f8 __attribute__((ext_vector_type(8)));
    f8 a, b, c, d;
    asm volatile("s_load_dwordx8 %0, %4, 0x0\n\t"
                 "s_load_dwordx8 %1, %4, 0x20\n\t"
                 "s_load_dwordx8 %2, %4, 0x40\n\t"
                 "s_load_dwordx8 %3, %4, 0x60\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d)
                 : "s"(r)
                 : "memory");
    FaceRec o;
    o.p0x = a[0]; o.p0y = a[1]; o.p1x = a[2]; o.p1y = a[3];
    o.p2x = a[4]; o.p2y = a[5]; o.e0x = a[6]; o.e0y = a[7];
    o.e1x = b[0]; o.e1y = b[1]; o.e2x = b[2]; o.e2y = b[3];
    o.il0 = b[4]; o.il1 = b[5]; o.il2 = b[6]; o.inv_det = b[7];
    o.f0x = c[0]; o.f0y = c[1]; o.f1x = c[2]; o.f1y = c[3];
    o.iz0 = c[4]; o.iz1 = c[5]; o.iz2 = c[6]; o.q0 = c[7];
    o.q1 = d[0]; o.q2 = d[1]; o.flags = __float_as_int(d[2]); o.zref = d[3];
    o.box_x = __float_as_uint(d[4]); o.box_y = __float_as_uint(d[5]); o.f2x = d[6]; o.f2y = d[7];
    return o;
}

__device__ inline float sat(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, 1.f); }
__device__ inline float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ inline float rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Pixel centre in normalised device coordinates; row 0 is the top of the image.
__device__ inline void pixel_centre(int col, int row, int S, float inv_S, float &xp, float &yp)
{
    xp = (float)(2 * col + 1 - S) * inv_S;
    yp = (float)(2 * (S - 1 - row) + 1 - S) * inv_S;
}

// Everything about one (pixel, face) pair that does not depend on colours.
struct PairGeom {
    float d2, t, dx, dy;          // squared distance, edge parameter, foot - pixel
    float w0, w1, w2;             // barycentric coordinates of the pixel
    bool k0, k1, k2;              // which edge is nearest (first wins ties)
    bool inside, act;
    float wmin;                   // min(w0, w1, w2): > 0 inside
    float D, omD;                 // sigmoid fragment and 1 - D
    float D_any;                  // the fragment before the activity test (D where act, else arbitrary)
};

// A pixel centre as the kernels carry it.  (2c + 1 - S) / S is exact in fp32 when S is a power of
// two; for other sizes x, y hold the rounded value and xl, yl the remainder (about 3e-8), which
// the LO variants subtract as well: without it the rounding moves every pixel by that much against
// the faces, and the 1/sigma slope of the fragments turns it into 1e-5 .. 1e-4 of relative error.
struct Pixel {
    float x, y, xl, yl;
};

template <bool LO>
__device__ inline Pixel make_pixel(float kx, float ky, float xp, float yp, float fS, float inv_S)
{
    // kx, ky: the integers 2c + 1 - S; xp, yp: their fp32 quotients by S (any rounding)
    Pixel q;
    q.x = xp; q.y = yp;
    q.xl = LO ? fmaf(-xp, fS, kx) * inv_S : 0.f;
    q.yl = LO ? fmaf(-yp, fS, ky) * inv_S : 0.f;
    return q;
}

// SIL: the record of a silhouette launch (see FaceRec).
template <bool LO, bool SIL>
__device__ inline PairGeom pair_geom(const FaceRec &r, const Pixel &px, const DevParams &P)
{
    PairGeom g;
    const float xp = px.x, yp = px.y;
    const float a0x = LO ? (r.p0x - xp) - px.xl : r.p0x - xp, a0y = LO ? (r.p0y - yp) - px.yl : r.p0y - yp;
    const float a1x = LO ? (r.p1x - xp) - px.xl : r.p1x - xp, a1y = LO ? (r.p1y - yp) - px.yl : r.p1y - yp;
    const float a2x = LO ? (r.p2x - xp) - px.xl : r.p2x - xp, a2y = LO ? (r.p2y - yp) - px.yl : r.p2y - yp;
    const float t0 = SIL ? sat(a0x * r.iz0 + a0y * r.iz1) : sat(-(a0x * r.e0x + a0y * r.e0y) * r.il0);
    const float t1 = SIL ? sat(a1x * r.iz2 + a1y * r.q0) : sat(-(a1x * r.e1x + a1y * r.e1y) * r.il1);
    const float t2 = SIL ? sat(a2x * r.q1 + a2y * r.q2) : sat(-(a2x * r.e2x + a2y * r.e2y) * r.il2);
    const float d0x = fmaf(t0, r.e0x, a0x), d0y = fmaf(t0, r.e0y, a0y);
    const float d1x = fmaf(t1, r.e1x, a1x), d1y = fmaf(t1, r.e1y, a1y);
    const float d2x = fmaf(t2, r.e2x, a2x), d2y = fmaf(t2, r.e2y, a2y);
    const float dd0 = d0x * d0x + d0y * d0y;
    const float dd1 = d1x * d1x + d1y * d1y;
    const float dd2 = d2x * d2x + d2y * d2y;
    // scaled edge functions cross(e_k, pixel - p_k) / det: w of the vertex opposite to edge k
    g.w2 = r.f0y * a0x - r.f0x * a0y;
    g.w0 = r.f1y * a1x - r.f1x * a1y;
    g.w1 = r.f2y * a2x - r.f2x * a2y;
    g.k0 = (dd0 <= dd1) && (dd0 <= dd2);
    g.k1 = !g.k0 && (dd1 <= dd2);
    g.k2 = !g.k0 && !g.k1;
    g.d2 = fminf(dd0, fminf(dd1, dd2));            // what the selection below picks; one v_min3
    g.t = g.k0 ? t0 : (g.k1 ? t1 : t2);
    g.dx = g.k0 ? d0x : (g.k1 ? d1x : d2x);
    g.dy = g.k0 ? d0y : (g.k1 ? d1y : d2y);
    g.wmin = fminf(g.w0, fminf(g.w1, g.w2));                      // one v_min3
    g.inside = g.wmin > 0.f;
    // The squared distance with the sign of wmin, + inside: one bit-field insert.  (Where wmin is
    // zero -- on the boundary, counted as outside -- the distance is zero as well.)
    const float xs = copysignf(g.d2, g.wmin);
    // A pixel outside the face's cull box (bbox widened by sqrt(threshold)) is farther than
    // sqrt(threshold) from the triangle, so the distance test alone implements both culls.
    g.act = xs > -P.threshold;
    // exp(-x) with x = +-d2/sigma; D = 1/(1+exp(-x)); 1-D = exp(-x) D (no cancellation)
    const float e = ex2(xs * P.neg_l2e_over_sigma);
    const float D = rcp(1.f + e);
    g.D_any = D;
    g.D = g.act ? D : 0.f;
    g.omD = g.act ? e * D : 1.f;
    return g;
}

// 1 - D of a pair, all the silhouette forward needs.  With x = +-d2/sigma (+ inside),
// 1 - D = 1 / (1 + exp(x)) directly: no cancellation, exp overflow gives exactly 0, and for an
// outside pixel at or beyond the cull distance exp(x) <= dist_eps/(1-dist_eps) = 1e-10 is absorbed
// by the 1 + ..., so the value is exactly 1 there -- the same as skipping the pair -- without
// testing for it.  (Only taken when dist_eps is below half an fp32 ulp of 1: DevParams::alpha_fast.)
template <bool LO>
__device__ inline float pair_one_minus_D(const FaceRec &r, const Pixel &px, const DevParams &P)
{
    const float xp = px.x, yp = px.y;
    const float a0x = LO ? (r.p0x - xp) - px.xl : r.p0x - xp, a0y = LO ? (r.p0y - yp) - px.yl : r.p0y - yp;
    const float a1x = LO ? (r.p1x - xp) - px.xl : r.p1x - xp, a1y = LO ? (r.p1y - yp) - px.yl : r.p1y - yp;
    const float a2x = LO ? (r.p2x - xp) - px.xl : r.p2x - xp, a2y = LO ? (r.p2y - yp) - px.yl : r.p2y - yp;
    const float t0 = sat(a0x * r.iz0 + a0y * r.iz1);       // a silhouette record: see FaceRec
    const float t1 = sat(a1x * r.iz2 + a1y * r.q0);
    const float t2 = sat(a2x * r.q1 + a2y * r.q2);
    const float d0x = fmaf(t0, r.e0x, a0x), d0y = fmaf(t0, r.e0y, a0y);
    const float d1x = fmaf(t1, r.e1x, a1x), d1y = fmaf(t1, r.e1y, a1y);
    const float d2x = fmaf(t2, r.e2x, a2x), d2y = fmaf(t2, r.e2y, a2y);
    const float d2 = fminf(d0x * d0x + d0y * d0y, fminf(d1x * d1x + d1y * d1y, d2x * d2x + d2y * d2y));
    const float w2 = r.f0y * a0x - r.f0x * a0y;
    const float w0 = r.f1y * a1x - r.f1x * a1y;
    const float w1 = r.f2y * a2x - r.f2x * a2y;
    const bool inside = fminf(w0, fminf(w1, w2)) > 0.f;
    const float e = ex2((inside ? d2 : -d2) * -P.neg_l2e_over_sigma);
    return rcp(1.f + e);
}

// Depth and normalised clamped barycentrics of a pair.
struct PairDepth {
    float n0, n1, n2, rs;         // clamp(w) / sum, 1 / sum
    float u, zp, zn;              // zp - z_ref, zp, -(zp - z_ref)/(far-near)
    bool contrib;
};

__device__ inline PairDepth pair_depth(const FaceRec &r, const PairGeom &g, const DevParams &P)
{
    PairDepth d;
    const float wc0 = sat(g.w0), wc1 = sat(g.w1), wc2 = sat(g.w2);
    const float num = wc0 * r.q0 + wc1 * r.q1 + wc2 * r.q2;
    const float den = wc0 * r.iz0 + wc1 * r.iz1 + wc2 * r.iz2;
    // one Newton step on the reciprocal: the z-softmax divides this offset by (far-near)*gamma
    float rden = rcp(den);
    rden = rden * (2.f - den * rden);
    d.u = num * rden;
    d.zp = r.zref + d.u;
    d.rs = rcp(wc0 + wc1 + wc2);
    d.n0 = wc0 * d.rs;
    d.n1 = wc1 * d.rs;
    d.n2 = wc2 * d.rs;
    d.zn = -d.u * P.inv_range;
    d.contrib = g.act && (d.zp >= P.near_plane) && (d.zp <= P.far_plane) &&
                (P.double_side || (r.flags & 1));
    return d;
}

__device__ inline int texel_index(float n0, float n1, float n2, int R)
{
    const float fx = n0 * (float)R, fy = n1 * (float)R;
    const int wx = min((int)fx, R - 1), wy = min((int)fy, R - 1);
    // (n0 + n1) written as (1 - n2): identical in exact arithmetic, and exact on a clamped edge
    const bool lower = ((1.f - n2) * (float)R - (float)wx) - (float)wy <= 1.f;
    // R <= 36: a 24-bit multiply-add (the 32-bit one is a quarter-rate 64-bit instruction here; the
    // masks below tell the compiler the range)
    const int sx = lower ? wx : R - 1 - wx, sy = lower ? wy : R - 1 - wy;
    return (int)(((unsigned)sy & 63u) * ((unsigned)R & 63u) + (unsigned)sx);
}

// ------------------------------------------------------------------------------------
// setup
// ------------------------------------------------------------------------------------
// ax*by - ay*bx with both products rounded before the subtraction: no fma contraction, so equal
// products cancel to exactly 0.
__device__ inline float cross_rounded(float ax, float ay, float bx, float by)
{
#pragma clang fp contract(off)
    const float u = ax * by;
    const float v = ay * bx;
    return u - v;
}

// Record of face `gid` from its three camera-space corners; zref = z of the image's first corner.
// Edge A -> B of the oriented silhouette record: A and c times the unit direction.
__device__ inline float4 oriented_edge(float ax, float ay, float bx, float by, float c)
{
    const float ex = bx - ax, ey = by - ay;
    const float len2 = ex * ex + ey * ey;
    const float inv_len = len2 > 0.f ? 1.f / sqrtf(len2) : 0.f;
    return make_float4(ax, ay, len2 > 0.f ? c * (ex * inv_len) : c, c * (ey * inv_len));
}

// Pixel box of a face from the xy of its corners: the pixels whose centre lies inside the cull box
// (bounding box widened by the margin); an empty box (col0 > col1) for a face that shows nowhere.
// `raw_det`: the face's doubled signed area, see setup_face.
__device__ inline void face_box(const DevParams &P, float x0, float y0, float x1, float y1, float x2, float y2,
                                float raw_det, int &col0, int &col1, int &row0, int &row1)
{
    const float bx0 = fminf(x0, fminf(x1, x2)) - P.margin, bx1 = fmaxf(x0, fmaxf(x1, x2)) + P.margin;
    const float by0 = fminf(y0, fminf(y1, y2)) - P.margin, by1 = fmaxf(y0, fmaxf(y1, y2)) + P.margin;
    // A pixel outside the cull box is farther than the margin from the face, hence inactive, so the
    // box bounds the work exactly; 0.01 pixel of slack covers the rounding of the conversion (below
    // 1e-3 pixel up to the largest S).
    const float fs = (float)P.S, hi = fs;
    const float cx0 = ceilf(fminf(fmaxf((fs * bx0 + fs - 1.f) * 0.5f, -2.f), hi) - 0.01f);
    const float cx1 = floorf(fminf(fmaxf((fs * bx1 + fs - 1.f) * 0.5f, -2.f), hi) + 0.01f);
    const float cy0 = ceilf(fminf(fmaxf((fs * by0 + fs - 1.f) * 0.5f, -2.f), hi) - 0.01f);
    const float cy1 = floorf(fminf(fmaxf((fs * by1 + fs - 1.f) * 0.5f, -2.f), hi) + 0.01f);
    col0 = max((int)cx0, 0);
    col1 = min((int)cx1, P.S - 1);
    int yi0 = max((int)cy0, 0), yi1 = min((int)cy1, P.S - 1);
    row0 = P.S - 1 - yi1;
    row1 = P.S - 1 - yi0;
    const bool finite = (bx0 == bx0) && (bx1 == bx1) && (by0 == by0) && (by1 == by1);
    // A face whose projected area is below the clamp on the determinant (a point, a segment, an
    // exactly edge-on face) has no barycentric coordinates: upstream's algebra then yields
    // distances of the order 1/1e-10 and the face shows nowhere.  It is dropped here.
    const bool no_area = fabsf(raw_det) < 1e-10f;
    if (!finite || no_area || col0 > col1 || row0 > row1) {
        col0 = 32767; col1 = -1; row0 = 32767; row1 = -1;
    }
}

__device__ inline uint2 pack_box(int col0, int col1, int row0, int row1)
{
    return make_uint2((uint32_t)(col0 & 0xffff) | ((uint32_t)(col1 & 0xffff) << 16),
                      (uint32_t)(row0 & 0xffff) | ((uint32_t)(row1 & 0xffff) << 16));
}

// r: where the face's record goes (8 float4); q: its oriented silhouette record (3 float4) or nullptr.
__device__ inline void setup_face(const DevParams &P, int gid, float x0, float y0, float z0, float x1,
                                  float y1, float z1, float x2, float y2, float z2, float zref,
                                  float4 *__restrict__ r, uint2 *__restrict__ pbox, int &col0, int &col1,
                                  int &row0, int &row1, float4 *__restrict__ q)
{
    const float e0x = x1 - x0, e0y = y1 - y0;
    const float e1x = x2 - x1, e1y = y2 - y1;
    const float e2x = x0 - x2, e2y = y0 - y2;
    const float l0 = e0x * e0x + e0y * e0y, l1 = e1x * e1x + e1y * e1y, l2 = e2x * e2x + e2y * e2y;
    // both products rounded before the subtraction (no fma contraction), so that a face with two
    // equal corners gets exactly 0
    const float raw_det = cross_rounded(e0x, e0y, x2 - x0, y2 - y0);
    const float det = raw_det > 0.f ? fmaxf(raw_det, 1e-10f) : fminf(raw_det, -1e-10f);
    face_box(P, x0, y0, x1, y1, x2, y2, raw_det, col0, col1, row0, row1);
    const uint2 box = pack_box(col0, col1, row0, row1);
    const float iz0 = 1.f / z0, iz1 = 1.f / z1, iz2 = 1.f / z2;
    r[0] = make_float4(x0, y0, x1, y1);
    r[1] = make_float4(x2, y2, e0x, e0y);
    r[2] = make_float4(e1x, e1y, e2x, e2y);
    const float inv_det = 1.f / det;
    const float il0 = l0 > 0.f ? 1.f / l0 : 0.f, il1 = l1 > 0.f ? 1.f / l1 : 0.f, il2 = l2 > 0.f ? 1.f / l2 : 0.f;
    r[3] = make_float4(il0, il1, il2, inv_det);
    r[4] = make_float4(e0x * inv_det, e0y * inv_det, e1x * inv_det, e1y * inv_det);
    if (P.mode == SV3D_TEX_NONE) {              // silhouettes: -e_k / |e_k|^2 where the depths would be
        r[5] = make_float4(-e0x * il0, -e0y * il0, -e1x * il1, -e1y * il1);
        r[6] = make_float4(-e2x * il2, -e2y * il2, __int_as_float(raw_det < 0.f ? 1 : 0), zref);
    } else {
        r[5] = make_float4(iz0, iz1, iz2, (z0 - zref) * iz0);
        r[6] = make_float4((z1 - zref) * iz1, (z2 - zref) * iz2,
                           __int_as_float(raw_det < 0.f ? 1 : 0), zref);
    }
    r[7] = make_float4(__uint_as_float(box.x), __uint_as_float(box.y), e2x * inv_det, e2y * inv_det);
    pbox[gid] = box;
    if (q) {
        // Oriented silhouette record (per-tile lists): the three edges taken so that the inside is on
        // their left -- for a face of the other orientation the corners go round the other way --
        // each as its start point and c times its unit direction, c = sqrt(log2(e) / sigma).
        if (raw_det > 0.f) {
            q[0] = oriented_edge(x0, y0, x1, y1, P.c_scale);
            q[1] = oriented_edge(x1, y1, x2, y2, P.c_scale);
            q[2] = oriented_edge(x2, y2, x0, y0, P.c_scale);
        } else {
            q[0] = oriented_edge(x0, y0, x2, y2, P.c_scale);
            q[1] = oriented_edge(x2, y2, x1, y1, P.c_scale);
            q[2] = oriented_edge(x1, y1, x0, y0, P.c_scale);
        }
    }
}

// The setup kernels run on a (faces, image) grid, so that a wave holds 64 consecutive faces of
// one image: the union of their pixel boxes is stored as the box of face group f / 64.  Meshes
// number their faces coherently (an icosphere's 64 consecutive faces cover a sixth of its
// extent), so a tile of the forward pass can drop most groups on that one box.
constexpr int kGroupFaces = 64;
__host__ __device__ inline int face_groups(int F) { return (F + kGroupFaces - 1) / kGroupFaces; }

__device__ inline void store_group_box(const DevParams &P, int b, int f, bool valid, int col0, int col1, int row0,
                                       int row1, uint2 *__restrict__ gbox)
{
    int c0 = valid ? col0 : 32767, c1 = valid ? col1 : -1, r0 = valid ? row0 : 32767, r1 = valid ? row1 : -1;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        c0 = min(c0, __shfl_xor(c0, d));
        c1 = max(c1, __shfl_xor(c1, d));
        r0 = min(r0, __shfl_xor(r0, d));
        r1 = max(r1, __shfl_xor(r1, d));
    }
    if ((threadIdx.x & 63) == 0 && f < P.F)
        gbox[(size_t)b * face_groups(P.F) + f / kGroupFaces] =
            make_uint2((uint32_t)(c0 & 0xffff) | ((uint32_t)(c1 & 0xffff) << 16),
                       (uint32_t)(r0 & 0xffff) | ((uint32_t)(r1 & 0xffff) << 16));
}

// A thread's record is 128 bytes, its oriented record 48: stored by the thread that made them, every
// store instruction of a wave would touch 64 cache lines with 16 bytes each.  The records go through
// LDS instead (per wave: 64 records at a stride of nine float4, conflict free, and 64 x 3 float4 of
// oriented records as they lie) and leave as the 8 KB and 3 KB blocks they form in memory, one
// kilobyte per instruction.  kSetupStage4: float4 per wave.
constexpr int kSetupStageRec4 = 64 * (kRec4 + 1);
constexpr int kSetupStage4 = kSetupStageRec4 + 64 * 3;

// The wave's records out of its staging area: faces f0 .. f0 + 63 of image b, `f0` the face of lane 0.
__device__ inline void store_staged_records(const DevParams &P, int b, int f0, const float4 *s_wave,
                                            float4 *__restrict__ rec, float4 *__restrict__ srec)
{
    const int lane = threadIdx.x & 63;
    const int nvalid = min(64, P.F - f0);               // <= 0: a wave past the image's faces
    __builtin_amdgcn_wave_barrier();                    // the wave's own LDS writes are in order before its reads
    float4 *dst = rec + ((size_t)b * P.F + f0) * kRec4;
#pragma unroll
    for (int i = 0; i < kRec4; ++i) {
        const int qi = i * 64 + lane, face = qi >> 3, part = qi & 7;
        const float4 v = s_wave[face * (kRec4 + 1) + part];
        if (face < nvalid) dst[qi] = v;
    }
    if (srec) {
        float4 *dq = srec + ((size_t)b * P.F + f0) * 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int qi = i * 64 + lane;
            const float4 v = s_wave[kSetupStageRec4 + qi];
            if (qi < 3 * nvalid) dq[qi] = v;
        }
    }
}

// s_stage: kSetupStage4 float4 of LDS for each wave of the workgroup that calls this.
__device__ inline void setup_from_face_vertices(
    const DevParams &P, int b, int f, const float *__restrict__ fv, float4 *__restrict__ rec,
    uint2 *__restrict__ pbox, uint2 *__restrict__ gbox, float4 *__restrict__ srec, float4 *s_stage)
{
    const bool valid = f < P.F;
    const int lane = threadIdx.x & 63;
    float4 *s_wave = s_stage + (threadIdx.x >> 6) * kSetupStage4;
    int col0 = 32767, col1 = -1, row0 = 32767, row1 = -1;
    if (valid) {
        const int gid = b * P.F + f;
        const float *v = fv + (size_t)gid * 9;
        setup_face(P, gid, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8],
                   fv[(size_t)b * P.F * 9 + 2], s_wave + lane * (kRec4 + 1), pbox, col0, col1, row0, row1,
                   srec ? s_wave + kSetupStageRec4 + lane * 3 : nullptr);
    }
    store_staged_records(P, b, f - lane, s_wave, rec, srec);
    if (gbox) store_group_box(P, b, f, valid, col0, col1, row0, row1, gbox);
}

__global__ __launch_bounds__(kThreads) void raster_setup_kernel(
    DevParams P, const float *__restrict__ fv, float4 *__restrict__ rec, uint2 *__restrict__ pbox,
    uint2 *__restrict__ gbox,                   // gbox == nullptr: the forward will not read group boxes
    float4 *__restrict__ srec)                  // oriented silhouette records, or nullptr
{
    __shared__ float4 s_stage[(kThreads / 64) * kSetupStage4];
    setup_from_face_vertices(P, blockIdx.y, blockIdx.x * kThreads + threadIdx.x, fv, rec, pbox, gbox, srec, s_stage);
}

// Setup straight from the object-space mesh: each thread projects its face's three vertices
// through the image's camera (camera.h), optionally stores them as face_vertices, and builds
// the record from those same values (per corner the arithmetic is project_gather_kernel's).
struct MeshArgs {
    const float *vertices, *cams;
    const int32_t *faces;
    int V, faces_batch;
    float offset_z;
    sv3d_dev::Affine a;
};

// The three projected corners of face f of image b (x, y, z each).
__device__ inline void project_face(const MeshArgs &m, int F, int b, int f, float *__restrict__ o)
{
    const int32_t *fb = m.faces + (m.faces_batch > 1 ? (size_t)b * F * 3 : 0);
    const sv3d_dev::Camera c = sv3d_dev::load_camera(m.cams + (size_t)b * 7, m.offset_z);
    const float *vb = m.vertices + (size_t)b * m.V * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int v = min(max(fb[f * 3 + k], 0), m.V - 1);
        sv3d_dev::camera_project(c, m.a, vb[v * 3], vb[v * 3 + 1], vb[v * 3 + 2], o[3 * k], o[3 * k + 1],
                                 o[3 * k + 2]);
    }
}

__device__ inline void setup_from_mesh(
    const DevParams &P, const MeshArgs &m, int b, int f, float *__restrict__ fv_out, float4 *__restrict__ rec,
    uint2 *__restrict__ pbox, uint2 *__restrict__ gbox, float4 *__restrict__ srec, float4 *s_stage)
{
    const bool valid = f < P.F;
    const int lane = threadIdx.x & 63;
    float4 *s_wave = s_stage + (threadIdx.x >> 6) * kSetupStage4;
    int col0 = 32767, col1 = -1, row0 = 32767, row1 = -1;
    if (valid) {
    const int gid = b * P.F + f;
    const int32_t *fb = m.faces + (m.faces_batch > 1 ? (size_t)b * P.F * 3 : 0);
    const sv3d_dev::Camera c = sv3d_dev::load_camera(m.cams + (size_t)b * 7, m.offset_z);
    const float *vb = m.vertices + (size_t)b * m.V * 3;
    float o[9];
    project_face(m, P.F, b, f, o);
    const int v00 = min(max(fb[0], 0), m.V - 1);
    float rx, ry, zref;
    sv3d_dev::camera_project(c, m.a, vb[v00 * 3], vb[v00 * 3 + 1], vb[v00 * 3 + 2], rx, ry, zref);
    if (fv_out) {
#pragma unroll
        for (int i = 0; i < 9; ++i) fv_out[(size_t)gid * 9 + i] = o[i];
    }
    setup_face(P, gid, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8], zref, s_wave + lane * (kRec4 + 1),
               pbox, col0, col1, row0, row1, srec ? s_wave + kSetupStageRec4 + lane * 3 : nullptr);
    }
    store_staged_records(P, b, f - lane, s_wave, rec, srec);
    if (gbox) store_group_box(P, b, f, valid, col0, col1, row0, row1, gbox);
}

__global__ __launch_bounds__(kThreads) void raster_setup_mesh_kernel(
    DevParams P, MeshArgs m, float *__restrict__ fv_out, float4 *__restrict__ rec,
    uint2 *__restrict__ pbox, uint2 *__restrict__ gbox, float4 *__restrict__ srec)
{
    __shared__ float4 s_stage[(kThreads / 64) * kSetupStage4];
    setup_from_mesh(P, m, blockIdx.y, blockIdx.x * kThreads + threadIdx.x, fv_out, rec, pbox, gbox, srec, s_stage);
}

__device__ inline void unpack_box(uint2 b, int &col0, int &col1, int &row0, int &row1)
{
    col0 = (int)(short)(b.x & 0xffff);
    col1 = (int)(short)(b.x >> 16);
    row0 = (int)(short)(b.y & 0xffff);
    row1 = (int)(short)(b.y >> 16);
}

// Inclusive prefix sum over the 64 lanes (row shifts inside each row of 16 lanes, then the row
// totals are broadcast into the following rows).
template <int CTRL, int ROW_MASK>
__device__ inline int dpp_add_i(int v)
{
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, true);
}

__device__ inline int wave_inclusive_scan(int v)
{
    v = dpp_add_i<0x111, 0xf>(v);       // row_shr:1
    v = dpp_add_i<0x112, 0xf>(v);       // row_shr:2
    v = dpp_add_i<0x114, 0xf>(v);       // row_shr:4
    v = dpp_add_i<0x118, 0xf>(v);       // row_shr:8
    v = dpp_add_i<0x142, 0xa>(v);       // row_bcast:15 into rows 1 and 3
    v = dpp_add_i<0x143, 0xc>(v);       // row_bcast:31 into rows 2 and 3
    return v;
}

// ------------------------------------------------------------------------------------
// coarse bins (meshes with many faces)
// ------------------------------------------------------------------------------------
// Every tile workgroup of the forward tests the pixel boxes of its image's faces; with thousands
// of faces that scan outweighs the fragment work (5120 faces at 512^2: ~40 % of the forward's
// instructions).  For such meshes a first kernel lists, per 64x64-pixel bin (4x4 tiles), the
// faces whose pixel box touches the bin, in face order; a tile then scans its bin's list only.
constexpr int kBinTiles = 4;                    // tiles per bin side
constexpr int kBinPix = kBinTiles * kTile;

struct CoarseBins {
    const unsigned short *list;                 // [B][nbins][cap] face ids, ascending; nullptr = not used
    const int *count;                           // [B][nbins]
    int bins_x, nbins, cap;
    // per-tile lists made by raster_tile_lists_kernel (nullptr = not used): a tile whose list holds
    // all its faces takes them from there and bins nothing
    const int *tile_count;                      // [B][tiles][tiles]
    const unsigned short *tile_list;            // [B][tiles][tiles][kListCap] face ids, ascending
};

__global__ __launch_bounds__(kThreads) void raster_coarse_kernel(
    DevParams P, const uint2 *__restrict__ pbox, unsigned short *__restrict__ list,
    int *__restrict__ count, int bins_x, int nbins, int cap)
{
    __shared__ int s_cnt[4][4];
    const int b = blockIdx.x / nbins, bin = blockIdx.x - b * nbins;
    const int by = bin / bins_x, bx = bin - by * bins_x;
    const int c0 = bx * kBinPix, c1 = c0 + kBinPix - 1, r0 = by * kBinPix, r1 = r0 + kBinPix - 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    const uint2 *pb = pbox + (size_t)b * P.F;
    unsigned short *out = list + ((size_t)b * nbins + bin) * cap;
    int base = 0;                               // hits so far (uniform over the workgroup)
    for (int f0 = 0; f0 < P.F; f0 += 4 * kThreads) {
        uint2 box[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {           // four loads in flight per thread
            const int f = f0 + k * kThreads + threadIdx.x;
            box[k] = f < P.F ? pb[f] : make_uint2(0xffff7fffu, 0xffff7fffu);
        }
        unsigned long long mask[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int a0, a1, b0, b1;
            unpack_box(box[k], a0, a1, b0, b1);
            mask[k] = __ballot(!(a1 < c0 || a0 > c1 || b1 < r0 || b0 > r1));
            if (lane == 0) s_cnt[k][wave] = __popcll(mask[k]);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {           // face order: chunk k, then wave, then lane
            int off = base;
            for (int w = 0; w < wave; ++w) off += s_cnt[k][w];
            if ((mask[k] >> lane) & 1ull)
                out[off + __popcll(mask[k] & lane_lt)] = (unsigned short)(f0 + k * kThreads + threadIdx.x);
            base += s_cnt[k][0] + s_cnt[k][1] + s_cnt[k][2] + s_cnt[k][3];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) count[(size_t)b * nbins + bin] = base;
}

// ------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------
// Tile of a workgroup's tile number.  Workgroups start in id order and a CU takes the next one as
// soon as it has room, so the numbering is a schedule.  Small launches (`rings`, even tile counts)
// number their tiles from the image centre outwards: the tiles an object usually covers start first
// and the empty border is left for the tail.  For 4, 8, 16 or 32 tiles a side the order is by the
// distance of the tile's centre from the image centre (a table, tile_order.h): an object's outline
// is closer to a circle than to a square, and the corners of a square ring are the emptiest tiles
// of it.  Other even counts go in square rings (ring k holds the 8k + 4 tiles at Chebyshev
// distance k + 1/2 from the centre).  Large launches lose by either (the heavy tiles then
// share their CUs); they go row by row, every tile row rotated by 5 more columns than the row
// above, so that the tiles one CU receives do not share a screen column.
__device__ inline void tile_position(int gx, int gy, int tiles_x, bool rings, int &tile_x, int &tile_y)
{
    // (gx, gy): the workgroup's place in the grid of tiles; its number there is gy * tiles_x + gx
    if (rings && (tiles_x == 4 || tiles_x == 8 || tiles_x == 16 || tiles_x == 32)) {
        const int n = gy * tiles_x + gx;                    // 0 <= n < tiles_x^2 by the launch grid
        const int t = sv3d_dev::kTileOrder[sv3d_dev::tile_order_offset(tiles_x) + (n & (tiles_x * tiles_x - 1))];
        tile_x = t & (tiles_x - 1);
        tile_y = t / tiles_x;
    } else if (rings && !(tiles_x & 1)) {
        const int tile = gy * tiles_x + gx;
        int k = (int)(__builtin_amdgcn_sqrtf((float)tile) * 0.5f);     // raw v_sqrt_f32, corrected below
        if (4 * k * k > tile) --k;
        else if (4 * (k + 1) * (k + 1) <= tile) ++k;
        const int j = tile - 4 * k * k, w = 2 * k + 1;
        const int side = (j >= w) + (j >= 2 * w) + (j >= 3 * w), off = j - side * w;
        const int lo = (tiles_x >> 1) - 1 - k, hi = (tiles_x >> 1) + k;
        tile_x = side == 0 ? lo + off : (side == 1 ? hi : (side == 2 ? hi - off : lo));
        tile_y = side == 0 ? lo : (side == 1 ? lo + off : (side == 2 ? hi : hi - off));
    } else {
        tile_y = gy;
        tile_x = (gx + 5 * gy) % tiles_x;
    }
}

// SPLIT = 1: four waves, one per 8x8 block of the tile.  SPLIT = 2 (small grids): eight waves, two
// per block, which take alternate hits of the block's face list and merge their partial pixel
// states through LDS at the end -- the kernel's span at small batches is its longest single wave
// (a silhouette-ring block with ~60 faces), and this halves it.
template <int MODE, int SPLIT, bool LO, bool COARSE>
__global__ __launch_bounds__(kThreads * SPLIT)
__attribute__((amdgpu_waves_per_eu(MODE == SV3D_TEX_VERTEX8 ? 7 : 8))) void raster_forward_kernel(
    // The leading arguments are what a wave needs for its first loads (its share of the face boxes,
    // or the image's group boxes); built with kernel argument preloading they arrive in registers
    // with the wave, and the loads go out at once instead of after a first trip to memory for the
    // argument block.  num_faces = P.F.
    const uint2 *__restrict__ pbox, const uint2 *__restrict__ gbox, int num_faces,
    const float4 *__restrict__ rec, const float *__restrict__ tex, float *__restrict__ rgba,
    float *__restrict__ saved, DevParams P, CoarseBins CB)
{
    // Small launches (SPLIT == 2) are bound by latency, not by instructions: they test every face's
    // pixel box at once (one round trip to memory) instead of the group boxes first and then the
    // boxes of the touched groups (two dependent round trips, fewer instructions).
    constexpr bool kDirectBoxes = SV3D_DIRECT_BOXES && SPLIT == 2;
    constexpr int NT = kThreads * SPLIT;        // threads per workgroup
    constexpr int NW = 4 * SPLIT;               // waves per workgroup
    __shared__ float4 s_rec[kListCap * kRecPad4];
    __shared__ int s_ids[kListCap];
    __shared__ unsigned long long s_mask[64];

    // Workgroup id -> (image, tile) with the image index fastest: workgroups that land on one
    // CU (ids a multiple of the CU count apart) then come from tile positions spread over the
    // screen, instead of the same position of every image, which balances silhouette-ring tiles
    // (many faces) against empty corners.
#ifdef SV3D_TRACE
    const unsigned long long trace_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int tiles_x = (P.S + kTile - 1) / kTile;
    // grid (image, tile column, tile row): workgroups start with the image index running fastest
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = P.S, F = num_faces;
    const uint2 *pbox_b = pbox + (size_t)b * F;
    // Small launches send the first box loads of the binning pass out before anything else: the way
    // through the memory system is the longest single wait of a tile, and the tile arithmetic
    // below fits inside it.
    constexpr int kEarly = 3;                           // masks per wave loaded ahead (3 * NW * 64 faces)
    constexpr bool kEarlyBoxes = kDirectBoxes && MODE == SV3D_TEX_NONE;     // colour modes lack the registers
    uint2 bx_early[kEarly];
    if (kEarlyBoxes) {
#pragma unroll
        for (int k = 0; k < kEarly; ++k) {
            const int f = (wave + NW * k) * 64 + lane;
            bx_early[k] = f < F ? pbox_b[f] : make_uint2(0xffff7fffu, 0xffff7fffu);
        }
    }
    // ... and large launches the boxes of the image's first 64 face groups, one per lane
    // (the RGB modes lack the registers to hold either across the tile arithmetic)
    constexpr bool kEarlyGroups = !COARSE && !kDirectBoxes && (MODE == SV3D_TEX_NONE || MODE == SV3D_TEX_VERTEX8);
    uint2 gb_early = make_uint2(0xffff7fffu, 0xffff7fffu);
    if (kEarlyGroups && lane < face_groups(F)) gb_early = gbox[(size_t)b * face_groups(F) + lane];
    const int tb = P.TB == P.B ? b : b % P.TB;          // texture set of this image
    int tile_x, tile_y;
    tile_position(blockIdx.y, blockIdx.z, tiles_x, SV3D_SPIRAL && SPLIT == 2, tile_x, tile_y);
    const int tile_c0 = tile_x * kTile, tile_r0 = tile_y * kTile;
    // lane -> pixel of the wave's 8x8 block; lanes 4q..4q+3 form the 2x2 quad q
    const int quad = lane >> 2, sub = lane & 3;
#if SV3D_SPLIT_ADJACENT
    const int blk = SPLIT == 2 ? wave >> 1 : wave, part = SPLIT == 2 ? wave & 1 : 0;
#else
    const int blk = wave & 3, part = wave >> 2;         // block of the tile, share of its face list
#endif
    const int blk_c0 = tile_c0 + (blk & 1) * 8, blk_r0 = tile_r0 + (blk >> 1) * 8;
    const int col = blk_c0 + (quad & 3) * 2 + (sub & 1);
    const int row = blk_r0 + (quad >> 2) * 2 + (sub >> 1);
    float xp, yp;
    pixel_centre(col, row, S, P.inv_S, xp, yp);
    const Pixel px = make_pixel<LO>((float)(2 * col + 1 - S), (float)(2 * (S - 1 - row) + 1 - S), xp, yp,
                                    (float)S, P.inv_S);
    // NDC extent of the pixel centres of this wave's block
    float bxa, bxb, bya, byb;
    pixel_centre(blk_c0, blk_r0 + 7, S, P.inv_S, bxa, bya);
    pixel_centre(blk_c0 + 7, blk_r0, S, P.inv_S, bxb, byb);

    const float4 *rec_b = rec + (size_t)b * F * kRec4;
    const float zref = ((const float *)rec_b)[27];
    const float c0shift = (P.far_plane - zref) * P.inv_range;

#ifdef SV3D_TRACE
    int trace_visits = 0;
#endif
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    float A = 1.f;                              // prod (1 - D)
    float m = P.eps - c0shift;                  // running softmax maximum, shifted by c0shift
    float ssum = part == 0 ? P.ssum0 : 0.f;     // the background term enters once, through part 0
    constexpr int NC = mode_channels(MODE);     // colour channels
    float Cc[NC];                               // softmax-weighted colour sums
#pragma unroll
    for (int c = 0; c < NC; ++c)                // VERTEX8: every channel is "channel 0" of its own render
        Cc[c] = (MODE == SV3D_TEX_VERTEX8 ? P.bg0 : (c == 0 ? P.bg0 : (c == 1 ? P.bg1 : P.bg2))) * ssum;
    int hits_seen = 0;                          // hits of this block so far (wave-uniform)

    // Conservative per-(block, face) test run lane-parallel over list entries: the block can only
    // hold an active pixel if it overlaps the face's cull box AND no edge line has the whole
    // block on its outer side at more than the cull distance (the triangle lies on the inner
    // side of each of its edge lines).
    const float bcx = 0.5f * (bxa + bxb), bcy = 0.5f * (bya + byb);
    const float bhx = 0.5f * (bxb - bxa), bhy = 0.5f * (byb - bya);
    auto block_may_touch = [&](const float4 *r) -> bool {
        const float4 q7 = r[7];
        int fc0, fc1, fr0, fr1;                                 // the face's pixel box
        unpack_box(make_uint2(__float_as_uint(q7.x), __float_as_uint(q7.y)), fc0, fc1, fr0, fr1);
        if (fc1 < blk_c0 || fc0 > blk_c0 + 7 || fr1 < blk_r0 || fr0 > blk_r0 + 7) return false;
        const float4 q0 = r[0], q1 = r[1], q2 = r[2], q3 = r[3];
        const float sgn = q3.w > 0.f ? 1.f : -1.f;              // orientation (sign of 1/det)
        // edge k: start point, direction, 1/|e|^2
        const float px[3] = {q0.x, q0.z, q1.x}, py[3] = {q0.y, q0.w, q1.y};
        const float ex[3] = {q1.z, q2.x, q2.z}, ey[3] = {q1.w, q2.y, q2.w};
        const float il[3] = {q3.x, q3.y, q3.z};
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float c = ey[k] * (px[k] - bcx) - ex[k] * (py[k] - bcy);     // edge function at the centre
            const float v = sgn * c + fabsf(ey[k]) * bhx + fabsf(ex[k]) * bhy;  // its maximum over the block
            ok = ok && !(v < 0.f && v * v * il[k] > P.visit_threshold);
        }
        return ok;
    };

    // `last`: no window follows this one, so nothing will overwrite the staged records; a wave that
    // has walked its hits then goes straight to its epilogue instead of waiting for the tile's
    // slowest wave (SPLIT == 2 still meets its partner: the merge below reuses the records' LDS).
    auto process_list = [&](int n, bool last) {
        SV3D_STAMP(3);
        for (int i = threadIdx.x; i < n * kRec4; i += NT)
            s_rec[(i / kRec4) * kRecPad4 + (i % kRec4)] = rec_b[(size_t)s_ids[i / kRec4] * kRec4 + (i % kRec4)];
        __syncthreads();
        SV3D_STAMP(4);
        for (int j0 = 0; j0 < n; j0 += 64) {
            const int j = j0 + lane;
            const bool hit = j < n && block_may_touch(&s_rec[min(j, n - 1) * kRecPad4]);
            unsigned long long mask = __ballot(hit);
            if (SPLIT == 2) {                   // the block's hits alternate between its two waves
                const bool mine = hit && (((hits_seen + __popcll(mask & lane_lt)) & 1) == part);
                hits_seen += __popcll(mask);
                mask = __ballot(mine);
            }
            while (mask) {
                const int jj = j0 + __builtin_ctzll(mask);
                mask &= mask - 1;
                const FaceRec r = load_rec(&s_rec[jj * kRecPad4]);
                if (MODE == SV3D_TEX_NONE && P.alpha_fast) {
                    A *= pair_one_minus_D<LO>(r, px, P);
#ifdef SV3D_TRACE
                    ++trace_visits;
#endif
                    continue;
                }
                const PairGeom g = pair_geom<LO, MODE == SV3D_TEX_NONE>(r, px, P);
                A *= g.omD;
#ifdef SV3D_TRACE
                ++trace_visits;
#endif
                if (MODE != SV3D_TEX_NONE) {
                    const PairDepth d = pair_depth(r, g, P);
                    float kc[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c) kc[c] = 0.f;
                    // the face is the same for all lanes: its textures' address is worked out once, in
                    // scalar registers, and a lane adds a 32-bit offset
                    const int fid = __builtin_amdgcn_readfirstlane(s_ids[jj]);
                    if (d.contrib) {
                        if (MODE == SV3D_TEX_SURFACE) {
                            const int Ts = P.R * P.R;
                            const float *tface = tex + ((size_t)tb * F + fid) * Ts * 3;
                            const unsigned off = __umul24((unsigned)texel_index(d.n0, d.n1, d.n2, P.R), 12u);
                            const float *tp = reinterpret_cast<const float *>(reinterpret_cast<const char *>(tface) + off);
                            kc[0] = tp[0]; kc[1] = tp[1]; kc[2] = tp[2];
                        } else {
                            const float *tp = tex + ((size_t)tb * F + fid) * (3 * NC);
#pragma unroll
                            for (int c = 0; c < NC; ++c)
                                kc[c] = d.n0 * tp[c] + d.n1 * tp[NC + c] + d.n2 * tp[2 * NC + c];
                        }
                    }
                    const float mnew = d.contrib ? fmaxf(m, d.zn) : m;
                    const float rs = ex2((m - mnew) * P.l2e_over_gamma);
                    const float ez = d.contrib ? ex2((d.zn - mnew) * P.l2e_over_gamma) : 0.f;
                    const float wgt = ez * g.D;
                    ssum = fmaf(rs, ssum, wgt);
#pragma unroll
                    for (int c = 0; c < NC; ++c) Cc[c] = fmaf(rs, Cc[c], wgt * kc[c]);
                    m = mnew;
                }
            }
        }
        if (!SV3D_EARLY_EXIT || !last || SPLIT == 2) __syncthreads();
    };

    // ---- binning: the faces whose pixel box touches this tile, in face order ----------------
    // One pass over the compact box array, eight independent loads in flight per wave.  The hit
    // bits of 64 consecutive faces are kept as one 64-bit mask in LDS; every wave then scans the
    // mask popcounts for itself (DPP prefix sum, no second barrier), which gives each hit its
    // list position, and the list is consumed in windows of kListCap faces.
    const int tc1 = tile_c0 + kTile - 1, tr1 = tile_r0 + kTile - 1;
    // candidates: all faces, or the faces listed for this tile's coarse bin
    const unsigned short *cl = nullptr;
    int ncand = F;
    if (COARSE) {
        const size_t bi = (size_t)b * CB.nbins + (tile_r0 / kBinPix) * CB.bins_x + tile_c0 / kBinPix;
        cl = CB.list + bi * CB.cap;
        ncand = CB.count[bi];
    }
    SV3D_STAMP(0);
    if (CB.tile_count != nullptr) {                                     // uniform over the launch
        const size_t tile = ((size_t)b * tiles_x + tile_y) * tiles_x + tile_x;
        const int n = CB.tile_count[tile];                              // uniform over the workgroup
        if (n <= kListCap) {
            if ((int)threadIdx.x < n) s_ids[threadIdx.x] = CB.tile_list[tile * kListCap + threadIdx.x];
            __syncthreads();
            if (n > 0) process_list(n, true);
            ncand = 0;                                                  // nothing left to bin
        }
    }
    for (int fbase = 0; fbase < ncand; fbase += 64 * 64) {
        const int nmask = (min(ncand - fbase, 64 * 64) + 63) >> 6;  // at most 64 masks per pass
        if (!COARSE && !kDirectBoxes) {
            // Face groups first: lane l tests the box of group l (the 64 faces of mask l) against
            // the tile; only the groups that touch it have their faces' boxes loaded, dealt out to
            // the waves in turn, up to four per wave and trip so that the loads overlap.  A tile no
            // group touches is done after this one small load.
            const uint2 *gbox_b = gbox + (size_t)b * face_groups(F) + (fbase >> 6);
            bool touch = false;
            if (lane < nmask) {
                int c0, c1, r0, r1;
                unpack_box(kEarlyGroups && fbase == 0 ? gb_early : gbox_b[lane], c0, c1, r0, r1);
                touch = !(c1 < tile_c0 || c0 > tc1 || r1 < tile_r0 || r0 > tr1);
            }
            unsigned long long todo = __ballot(touch);
            if (lane < nmask && !touch && (lane & (NW - 1)) == wave) s_mask[lane] = 0ull;
            while (todo != 0ull) {                                      // wave-uniform
                int sel[4] = {-1, -1, -1, -1};
                int ord = 0, got = 0;
                unsigned long long bits = todo;
                for (; bits != 0ull && ord < 4 * NW; bits &= bits - 1ull, ++ord) {
                    if ((ord & (NW - 1)) == wave) {
                        const int mi = __builtin_ctzll(bits);
                        if (got == 0) sel[0] = mi; else if (got == 1) sel[1] = mi;
                        else if (got == 2) sel[2] = mi; else sel[3] = mi;
                        ++got;
                    }
                }
                todo = bits;                                            // the groups of later trips
                uint2 bx[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int cand = fbase + sel[k] * 64 + lane;
                    bx[k] = (sel[k] >= 0 && cand < ncand) ? pbox_b[cand] : make_uint2(0xffff7fffu, 0xffff7fffu);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (sel[k] >= 0) {                                  // wave-uniform
                        int c0, c1, r0, r1;
                        unpack_box(bx[k], c0, c1, r0, r1);
                        const bool hit = !(c1 < tile_c0 || c0 > tc1 || r1 < tile_r0 || r0 > tr1);
                        const unsigned long long mask = __ballot(hit);
                        if (lane == 0) s_mask[sel[k]] = mask;
                    }
                }
            }
        } else {
        auto test_box = [&](int mi, uint2 box) {
            if (mi < nmask) {                                       // wave-uniform
                int c0, c1, r0, r1;
                unpack_box(box, c0, c1, r0, r1);
                const bool hit = !(c1 < tile_c0 || c0 > tc1 || r1 < tile_r0 || r0 > tr1);
                const unsigned long long mask = __ballot(hit);
                if (lane == 0) s_mask[mi] = mask;
            }
        };
        int m_first = wave;
        if (kEarlyBoxes && fbase == 0) {                            // the boxes loaded at the start
#pragma unroll
            for (int k = 0; k < kEarly; ++k) test_box(wave + NW * k, bx_early[k]);
            m_first += kEarly * NW;
        }
        for (int m0 = m_first; m0 < nmask; m0 += 8 * NW) {          // eight masks per wave per trip
            uint2 bx[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int cand = fbase + (m0 + NW * k) * 64 + lane;
                const int f = cand < ncand ? (COARSE ? (int)cl[cand] : cand) : -1;
                bx[k] = f >= 0 ? pbox_b[f] : make_uint2(0xffff7fffu, 0xffff7fffu);   // empty box
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) test_box(m0 + NW * k, bx[k]);
        }
        }
        __syncthreads();
        SV3D_STAMP(1);
        // lane l holds mask l; exclusive prefix of the popcounts over the wave
        const unsigned long long my_mask = lane < nmask ? s_mask[lane] : 0ull;
        const int my_cnt = __popcll(my_mask);
        const int incl = wave_inclusive_scan(my_cnt);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        const int excl = incl - my_cnt;
        SV3D_STAMP(2);
        for (int win = 0; win < total; win += kListCap) {
            for (int mi = wave; mi < nmask; mi += NW) {
                const unsigned long long mask = s_mask[mi];
                const int off = __builtin_amdgcn_readlane(excl, mi);
                if ((mask >> lane) & 1ull) {
                    const int pos = off + __popcll(mask & lane_lt) - win;
                    if (pos >= 0 && pos < kListCap) {
                        const int cand = fbase + mi * 64 + lane;
                        s_ids[pos] = COARSE ? (int)cl[cand] : cand;
                    }
                }
            }
            __syncthreads();
            process_list(min(kListCap, total - win), win + kListCap >= total && fbase + 64 * 64 >= ncand);
        }
        if (!SV3D_EARLY_EXIT || fbase + 64 * 64 < ncand) __syncthreads();
    }

    if (SPLIT == 2) {
        // merge the two partial pixel states of every block (s_rec is free after the last barrier)
        float *s_part = reinterpret_cast<float *>(s_rec) + (blk * 64 + lane) * (3 + NC);
        if (part == 1) {
            s_part[0] = A;
            if (MODE != SV3D_TEX_NONE) {
                s_part[1] = m; s_part[2] = ssum;
#pragma unroll
                for (int c = 0; c < NC; ++c) s_part[3 + c] = Cc[c];
            }
        }
        __syncthreads();
        if (part == 1) return;
        A *= s_part[0];
        if (MODE != SV3D_TEX_NONE) {
            const float m1 = s_part[1];
            const float mn = fmaxf(m, m1);
            const float sa = ex2((m - mn) * P.l2e_over_gamma), sb = ex2((m1 - mn) * P.l2e_over_gamma);
            ssum = sa * ssum + sb * s_part[2];
#pragma unroll
            for (int c = 0; c < NC; ++c) Cc[c] = sa * Cc[c] + sb * s_part[3 + c];
            m = mn;
        }
    }

    // epilogue: o[0..NC-1] colours, o[NC] alpha
    SV3D_STAMP(5);
    float o[NC + 1];
    if (MODE != SV3D_TEX_NONE) {
        const float is = 1.f / ssum;
#pragma unroll
        for (int c = 0; c < NC; ++c) o[c] = Cc[c] * is;
    } else {
        o[0] = P.bg0; o[1] = P.bg1; o[2] = P.bg2;
    }
    A = fminf(A, 1.f);                          // e*rcp(1+e) may round a hair above 1
    o[NC] = 1.f - A;
    const bool inimg = col < S && row < S;
    // (per image a base in scalar registers; per lane a 24-bit pixel index: S <= 4096)
    constexpr int kSavedFloats = MODE == SV3D_TEX_NONE ? 1 : (NC == 3 ? 6 : 12);
    float *saved_b = saved + (size_t)b * S * S * kSavedFloats;
    if (inimg) {
        const unsigned pix = __umul24((unsigned)row, (unsigned)S) + (unsigned)col;
        if (MODE == SV3D_TEX_NONE) {
            saved_b[pix] = A;
        } else if (NC == 3) {
            float2 *sp = (float2 *)(saved_b + pix * 6u);
            sp[0] = make_float2(A, ssum);
            sp[1] = make_float2(m, o[0]);
            sp[2] = make_float2(o[1], o[2]);
        } else {                                // 12 floats: A, sum, max, eight colours, pad
            float4 *sp = (float4 *)(saved_b + pix * 12u);
            auto oc = [&](int i) { return o[i <= NC ? i : NC]; };     // (indices in range for every NC)
            sp[0] = make_float4(A, ssum, m, oc(0));
            sp[1] = make_float4(oc(1), oc(2), oc(3), oc(4));
            sp[2] = make_float4(oc(5), oc(6), oc(7), 0.f);
        }
    }
    // planes of the image: colours then alpha, or the alpha plane alone
    const bool only_alpha = MODE == SV3D_TEX_NONE && P.alpha_only;
    const int planes = only_alpha ? 1 : NC + 1, first = only_alpha ? NC : 0;
    const int H = __builtin_amdgcn_readfirstlane(P.aa == 2 ? S >> 1 : S);
    const unsigned plane = (unsigned)H * (unsigned)H;
    float *rgba_b = rgba + (size_t)b * planes * plane;
    if (P.aa == 2) {
        const unsigned ob = __umul24((unsigned)(row >> 1), (unsigned)H) + (unsigned)(col >> 1);
#pragma unroll
        for (int c = 0; c <= NC; ++c) {
            if (c < first) continue;
            float v = inimg ? o[c] : 0.f;
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            if (sub == 0 && inimg) (rgba_b + (size_t)(c - first) * plane)[ob] = 0.25f * v;
        }
    } else if (inimg) {
        const unsigned ob = __umul24((unsigned)row, (unsigned)S) + (unsigned)col;
#pragma unroll
        for (int c = 0; c <= NC; ++c) {
            if (c >= first) (rgba_b + (size_t)(c - first) * plane)[ob] = o[c];
        }
    }
    SV3D_STAMP(6);
#ifdef SV3D_TRACE
    if (lane == 0 && SV3D_WG_ID < 16384 && part == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long *t = g_trace + ((size_t)SV3D_WG_ID * 4 + blk) * 4;
        t[0] = trace_t0;
        t[1] = __builtin_amdgcn_s_memrealtime();
        t[2] = ((unsigned long long)xcc << 32) | hwid;
        t[3] = (unsigned long long)trace_visits;
    }
#endif
}


// ------------------------------------------------------------------------------------
// per-tile face lists
// ------------------------------------------------------------------------------------
// Which launches take them: silhouettes of at most SV3D_TILED_FUSED_MAX_TILES tiles that come with
// face_vertices (lists from raster_setup_lists_kernel) and of more than SV3D_TILED_MIN_TILES (lists
// from raster_tile_lists_kernel), both into raster_forward_tiled_kernel; every other launch of more
// than SV3D_BINNING_LISTS_MIN_TILES tiles, into the general forward kernel in place of its binning.
// In a launch the per-tile work of the forward that is not fragment arithmetic -- every tile
// testing the boxes of its image's faces, numbering its hits, copying their 128-byte records and
// testing each against its four blocks -- is nearly half of the kernel's instructions.  Here a
// kernel of its own makes, per tile, the ordered list of the faces whose pixel box touches it
// (16-bit ids: lists of whole records cost more in memory traffic than they saved), and the
// forward builds for each listed face a compact record RELATIVE TO THE TILE'S CENTRE, in LDS,
// that the fragment loop can use as it is.  One workgroup per (image, bin of 4x4 tiles): the
// bin's faces first (ordered compaction over the image's pixel boxes, into LDS), then each of
// the bin's tiles against those.
//
// Tile-relative record (16 floats).  With c = sqrt(log2(e) / sigma), for each edge k taken from A
// to B so that the face's inside is on its left (the two are swapped for faces of the other
// orientation), u = its unit direction, and a pixel centre x = tile centre + o:
//     s = c u.(x - A)                              along the edge from A,   (Sx, Sy) = c u
//     h = c n.(x - A) = H0 - Sy ox + Sx oy         across the edge, > 0 inside,  n = (-uy, ux)
// The squared distance to the segment, times log2(e)/sigma, is h^2 + (how far s lies outside
// [0, c |e|])^2, and the pixel is inside the face iff min(h0, h1, h2) > 0.  The interval is stored
// by its middle and half length, and cut to what the tile can reach (make_tile_record):
//     s - middle = C + Sx ox + Sy oy,    outside by max(|s - middle| - L, 0)
// eight multiply-adds, subtractions and maxima per edge where the point-to-segment form on absolute
// coordinates takes 14.  o is exact in fp32 (a small odd multiple of 1/S), and C, L, H0 come from
// centre - A and centre - B formed with one rounding each, so the 1/sigma slope acts on numbers no
// larger than the tile.
// Layout: [Sx0 Sy0 C0 L0] [H00 Sx1 Sy1 C1] [L1 H01 Sx2 Sy2] [C2 L2 H02 box], box = the
// face's pixel box inside the tile, four 4-bit fields (col0, col1, row0, row1; 0..15).
constexpr int kTileCap = 128;                   // records per tile slot
static_assert(kTileCap == kListCap, "a tile's list fills one window of the general forward kernel");

constexpr int kTRecPad4 = 5;                    // LDS stride (80 B): lane-strided b128 reads are conflict free

constexpr int kListThreads = 256;               // the list kernel: four waves, each takes a row of the bin's tiles
constexpr int kListFaces = 5;                   // ... and a thread five faces per pass (1280 per workgroup)
constexpr int kFusedThreads = 1024;             // setup + lists in one kernel: sixteen waves, one per tile

// The lists of one (image, bin) by one workgroup of NT threads.  Faces are taken NT * U at a time,
// U per thread (thread t: faces f0 + u NT + t), all of a pass's boxes requested at once.
// stage(f0): called by all threads before the faces f0 .. f0 + NT U - 1 are looked at (it may
// synchronise the workgroup); box_of(f): the pixel box of face f of the image (or a box that
// contains it: the forward drops a listed face whose own box misses its tile).  `s_list`: LDS for
// F boxes and F 16-bit ids.  Second phase: a wave takes 16 / (NT / 64) of the bin's sixteen tiles
// and tests each candidate against all of them.
template <int NT, int U, class Stage, class BoxOf>
__device__ inline void tile_lists_of_bin(const DevParams &P, int b, int bin, uint2 *s_list, Stage stage, BoxOf box_of,
                                         int *__restrict__ tcount, unsigned short *__restrict__ tlist, int bins_x,
                                         int tiles_x)
{
    constexpr int NW = NT / 64;
    constexpr int K = kBinTiles * kBinTiles / NW;       // tiles per wave
    static_assert(K >= 1 && K * NW == kBinTiles * kBinTiles && (K == 1 || K == kBinTiles), "a wave: one tile or one row");
    __shared__ int s_cnt[U * NW];
    uint2 *s_box = s_list;
    unsigned short *s_id = reinterpret_cast<unsigned short *>(s_list + P.F);
    const int by = bin / bins_x, bx = bin - by * bins_x;
    const int c0 = bx * kBinPix, c1 = c0 + kBinPix - 1, r0 = by * kBinPix, r1 = r0 + kBinPix - 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    int base = 0;                               // hits so far (uniform over the workgroup)
    for (int f0 = 0; f0 < P.F; f0 += NT * U) {
        stage(f0);
        uint2 box[U];
        unsigned long long mask[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int f = f0 + u * NT + (int)threadIdx.x;
            box[u] = f < P.F ? box_of(f) : make_uint2(0xffff7fffu, 0xffff7fffu);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int a0, a1, b0, b1;
            unpack_box(box[u], a0, a1, b0, b1);
            mask[u] = __ballot(!(a1 < c0 || a0 > c1 || b1 < r0 || b0 > r1));
            if (lane == 0) s_cnt[u * NW + wave] = __popcll(mask[u]);
        }
        __syncthreads();
        int run = base;
#pragma unroll
        for (int u = 0; u < U; ++u) {           // face order: pass slot u, wave, lane
            int off = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const int cw = s_cnt[u * NW + w];
                off = w == wave ? run : off;
                run += cw;
            }
            if ((mask[u] >> lane) & 1ull) {
                const int pos = off + __popcll(mask[u] & lane_lt);
                s_box[pos] = box[u];
                s_id[pos] = (unsigned short)(f0 + u * NT + (int)threadIdx.x);
            }
        }
        base = run;
        __syncthreads();
    }
    const int nbin = base;
    // wave -> its tiles: K == 1: tile `wave` of the bin; K == 4: the four tiles of row `wave`
    const int ty = by * kBinTiles + (K == 1 ? wave / kBinTiles : wave);
    const int tx0 = bx * kBinTiles + (K == 1 ? (wave & (kBinTiles - 1)) : 0);
    if (ty >= tiles_x || tx0 >= tiles_x) return;                    // whole wave; no barrier follows
    const int tr0 = ty * kTile, tr1 = tr0 + kTile - 1;
    int n[K];
#pragma unroll
    for (int k = 0; k < K; ++k) n[k] = 0;
    const size_t tile0 = ((size_t)b * tiles_x + ty) * tiles_x + tx0;
    for (int j0 = 0; j0 < nbin; j0 += 64) {
        const int j = j0 + lane;
        const uint2 bxj = j < nbin ? s_box[j] : make_uint2(0xffff7fffu, 0xffff7fffu);
        const unsigned short id = j < nbin ? s_id[j] : (unsigned short)0;
        int a0, a1, b0, b1;
        unpack_box(bxj, a0, a1, b0, b1);
        const bool rows = !(b1 < tr0 || b0 > tr1);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int tc0 = (tx0 + k) * kTile, tc1 = tc0 + kTile - 1;
            const bool hit = rows && !(a1 < tc0 || a0 > tc1);
            const unsigned long long mask = __ballot(hit);
            const int pos = n[k] + __popcll(mask & lane_lt);
            if (hit && pos < kTileCap && tx0 + k < tiles_x) tlist[(tile0 + k) * kTileCap + pos] = id;
            n[k] += __popcll(mask);
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (lane == 0 && tx0 + k < tiles_x) tcount[tile0 + k] = n[k];
}

extern __shared__ uint2 s_dyn_list[];           // the list kernels' dynamic LDS

__global__ __launch_bounds__(kListThreads) void raster_tile_lists_kernel(
    DevParams P, const uint2 *__restrict__ pbox, int *__restrict__ tcount, unsigned short *__restrict__ tlist,
    int bins_x, int tiles_x)
{
    const uint2 *pb = pbox + (size_t)blockIdx.y * P.F;
    tile_lists_of_bin<kListThreads, kListFaces>(P, blockIdx.y, blockIdx.x, s_dyn_list, [](int) {},
                                                [&](int f) { return pb[f]; }, tcount, tlist, bins_x, tiles_x);
}

// Setup and lists in one launch, for launches small enough to be bound by the chain of their
// launches rather than by their work.  The first workgroups of an image do the setup of 256 faces
// each (a quarter of their threads: spread over many CUs, as the setup kernel's are), the `nbins`
// behind them build the lists of one bin each -- they work the faces' pixel boxes out of the
// corners themselves, every one of them for all the image's faces, instead of waiting for another
// kernel to store them.  The corners come through LDS: a face's nine floats lie 36 bytes apart, and
// read per face they would cost sixteen waves nine times eighteen cache lines each; read as they
// lie they are two lines per instruction, and the next 1024 faces are under way while these are
// looked at.  A box worked out twice could come out a pixel apart if the compiler contracted the
// two copies of the arithmetic differently, so the list builders widen theirs by one pixel and the
// forward tests the stored box again.
constexpr int kFusedFaces = 2;                  // faces per thread per pass: 2048 per workgroup
constexpr int kStageFaces = kFusedThreads * kFusedFaces;
constexpr int kStageFloats = kStageFaces * 9;
__host__ __device__ inline size_t lists_lds_bytes(int F) { return ((size_t)F * 10 + 15) & ~(size_t)15; }

__global__ __launch_bounds__(kFusedThreads) void raster_setup_lists_kernel(
    DevParams P, const float *__restrict__ fv, float4 *__restrict__ rec, uint2 *__restrict__ pbox,
    float4 *__restrict__ srec, int *__restrict__ tcount, unsigned short *__restrict__ tlist, int bins_x,
    int tiles_x, int setup_wgs)
{
    const int b = blockIdx.y;
    if ((int)blockIdx.x < setup_wgs) {          // uniform over the workgroup
        // (their records leave through the dynamic LDS the list builders use for their lists)
        if (threadIdx.x < kThreads)
            setup_from_face_vertices(P, b, (int)blockIdx.x * kThreads + (int)threadIdx.x, fv, rec, pbox, nullptr, srec,
                                     reinterpret_cast<float4 *>(s_dyn_list));
        return;
    }
    float *s_fv = reinterpret_cast<float *>(reinterpret_cast<char *>(s_dyn_list) + lists_lds_bytes(P.F));
    const float *src = fv + (size_t)b * P.F * 9;
    constexpr int kLoads = 9 * kFusedFaces;     // per thread and pass, all in flight together
    float nxt[kLoads];
    auto fetch = [&](int f0) {
        const int n9 = min(kStageFaces, P.F - f0) * 9;
#pragma unroll
        for (int k = 0; k < kLoads; ++k) {
            const int i = k * kFusedThreads + (int)threadIdx.x;
            nxt[k] = i < n9 ? src[(size_t)f0 * 9 + i] : 0.f;
        }
    };
    fetch(0);
    auto stage = [&](int f0) {                  // the loop's closing barrier has freed s_fv
#pragma unroll
        for (int k = 0; k < kLoads; ++k) s_fv[k * kFusedThreads + threadIdx.x] = nxt[k];
        if (f0 + kStageFaces < P.F) fetch(f0 + kStageFaces);
        __syncthreads();
    };
    auto box_of = [&](int f) {
        const float *v = s_fv + (f & (kStageFaces - 1)) * 9;         // stride 9 words: no bank conflicts
        const float x0 = v[0], y0 = v[1], x1 = v[3], y1 = v[4], x2 = v[6], y2 = v[7];
        const float raw_det = cross_rounded(x1 - x0, y1 - y0, x2 - x0, y2 - y0);
        int col0, col1, row0, row1;
        face_box(P, x0, y0, x1, y1, x2, y2, raw_det, col0, col1, row0, row1);
        if (col0 <= col1) { col0 -= 1; col1 += 1; row0 -= 1; row1 += 1; }
        return pack_box(col0, col1, row0, row1);
    };
    tile_lists_of_bin<kFusedThreads, kFusedFaces>(P, b, (int)blockIdx.x - setup_wgs, s_dyn_list, stage, box_of, tcount, tlist, bins_x, tiles_x);
}

// Tile-relative record of a face from its oriented silhouette record (e_k = Ax, Ay, Sx, Sy; the end
// of edge k is the start of edge k+1) and pixel box.  (cx_num, cy_num): numerators over S of the
// tile centre, which lies between pixel columns tc0+7 / tc0+8 and rows tr0+7 / tr0+8.
constexpr unsigned kNoFace = 1u << 16;         // box field of a record slot without a face

__device__ inline void make_tile_record(const float4 e0, const float4 e1, const float4 e2, uint2 box, int tc0, int tr0,
                                        float cx_num, float cy_num, float fS, float inv_S, float4 *__restrict__ o)
{
    // centre - A with one rounding: centre = num / S with num an integer, so (num - A S) / S
    const float d0x = fmaf(-e0.x, fS, cx_num) * inv_S, d0y = fmaf(-e0.y, fS, cy_num) * inv_S;
    const float d1x = fmaf(-e1.x, fS, cx_num) * inv_S, d1y = fmaf(-e1.y, fS, cy_num) * inv_S;
    const float d2x = fmaf(-e2.x, fS, cx_num) * inv_S, d2y = fmaf(-e2.y, fS, cy_num) * inv_S;
    int a0, a1, b0, b1;
    unpack_box(box, a0, a1, b0, b1);
    // (a list may name a face whose box misses the tile by a pixel: see raster_setup_lists_kernel)
    const bool misses = a1 < tc0 || a0 > tc0 + kTile - 1 || b1 < tr0 || b0 > tr0 + kTile - 1;
    const unsigned bits = misses ? kNoFace :
                          (unsigned)min(max(a0 - tc0, 0), 15) | ((unsigned)min(max(a1 - tc0, 0), 15) << 4) |
                          ((unsigned)min(max(b0 - tr0, 0), 15) << 8) | ((unsigned)min(max(b1 - tr0, 0), 15) << 12);
    // per edge: Sx, Sy, S0 = S.(centre - A), S1 = S.(centre - B), H0 = n.(centre - A), n = (-Sy, Sx)
    // per edge: Sx, Sy, C, L, H0 = n.(centre - A).  With s = S.(x - A) the pixel's place along the
    // edge, the part of its distance that lies along the edge is how far s is outside [0, |S.(B - A)|].
    // Over the tile s stays within R = (|Sx| + |Sy|) 15/S of its value at the centre, so an end of the
    // interval farther away than that may be moved up to that reach without changing any pixel of
    // the tile; the interval is then [centre value - da, centre value - db] with da, db below, both
    // small numbers formed from S.(centre - A) and S.(centre - B) -- each exact near its own end --
    // and the pixel's distance outside it is max(|s - middle| - L, 0): C = (da + db) / 2 puts the
    // middle at the origin, L = (da - db) / 2.
    const float reach = 15.f * inv_S;
    auto along = [&](float Sx, float Sy, float ax, float ay, float bx, float by, float &C, float &L) {
        const float R = (fabsf(Sx) + fabsf(Sy)) * reach;
        const float SA = fmaf(Sx, ax, Sy * ay), SB = fmaf(Sx, bx, Sy * by);
        const float da = fminf(SA, fmaxf(R, SB)), db = fminf(SA, fmaxf(-R, SB));
        C = 0.5f * (da + db);
        L = 0.5f * (da - db);
    };
    float C0, L0, C1, L1, C2, L2;
    along(e0.z, e0.w, d0x, d0y, d1x, d1y, C0, L0);
    along(e1.z, e1.w, d1x, d1y, d2x, d2y, C1, L1);
    along(e2.z, e2.w, d2x, d2y, d0x, d0y, C2, L2);
    o[0] = make_float4(e0.z, e0.w, C0, L0);
    o[1] = make_float4(fmaf(-e0.w, d0x, e0.z * d0y), e1.z, e1.w, C1);
    o[2] = make_float4(L1, fmaf(-e1.w, d1x, e1.z * d1y), e2.z, e2.w);
    o[3] = make_float4(C2, L2, fmaf(-e2.w, d2x, e2.z * d2y), __uint_as_float(bits));
}

// The silhouette forward over per-tile lists: one workgroup per 16x16 tile.  A tile without faces
// writes its background and is done after one scalar load; the others build their faces'
// tile-relative records in LDS (one thread per face: ids, then the faces' records -- two trips to
// memory -- and one barrier), and every wave tests the list against its 8x8 blocks lane-parallel
// and walks its hits.
// HALVES = 2 (NB = 1 only; the smallest launches, bound by their longest waves): eight waves, two per
// block, which take the block's hits in turn and multiply their products together through LDS.
// (From 1536 tiles on such launches take raster_forward_tiled_shared_kernel instead.)
// NB = 1: four waves, one block each.  NB = 2 (the largest launches): two waves, rows 0-7 and 8-15,
// each wave's lanes holding one pixel of the left and one of the right block; a visit's record is
// then read from LDS once for both blocks.  The four broadcast ds_read_b128 of a visit take 16
// cycles of the CU's LDS array, which four SIMDs running one-block visits (31 vector instructions,
// 62 cycles of VALU issue each) keep fully busy: the LDS bounds them as much as the VALU does.  (Fewer, longer waves: launches of fewer
// than some ten thousand tiles, bound by their longest waves, keep one block per wave.)
__device__ inline float tile_visit(const float4 r0, const float4 r1, const float4 r2, const float4 r3, float ox, float oy)
{
    // along each edge (shared part), across it, beyond its ends
    const float c0 = fmaf(r0.x, ox, fmaf(r0.y, oy, r0.z)), c1 = fmaf(r1.y, ox, fmaf(r1.z, oy, r1.w)),
                c2 = fmaf(r2.z, ox, fmaf(r2.w, oy, r3.x));
    const float h0 = fmaf(-r0.y, ox, fmaf(r0.x, oy, r1.x));
    const float h1 = fmaf(-r1.z, ox, fmaf(r1.y, oy, r2.y));
    const float h2 = fmaf(-r2.w, ox, fmaf(r2.z, oy, r3.z));
    const float o0 = fmaxf(fabsf(c0) - r0.w, 0.f), o1 = fmaxf(fabsf(c1) - r2.x, 0.f), o2 = fmaxf(fabsf(c2) - r3.y, 0.f);
    const float dd = fminf(fmaf(o0, o0, h0 * h0), fminf(fmaf(o1, o1, h1 * h1), fmaf(o2, o2, h2 * h2)));
    // 1 - D = 1 / (1 + exp(+-d2 / sigma)), + inside: see pair_one_minus_D.  The DENOMINATOR is
    // returned: a pixel's product of (1 - D) is the reciprocal of the product of these, so the
    // reciprocal (a quarter-rate instruction) is taken once per pixel instead of once per visit.
    // An overflowing product is +inf and its reciprocal 0, which is what the product of the
    // factors would have underflowed to; every denominator is >= 1, so inf never meets 0.
    // (the sign of the smallest h put on dd: one bit-field insert; on a boundary, where that h is zero, dd is zero too)
    return 1.f + ex2(copysignf(dd, fminf(h0, fminf(h1, h2))));
}

template <int NB, int HALVES>
__global__ __launch_bounds__(kThreads / NB * HALVES) void raster_forward_tiled_kernel(
    const int *__restrict__ tcount, const unsigned short *__restrict__ tlist, const float4 *__restrict__ srec,
    const uint2 *__restrict__ pbox, int num_faces, float *__restrict__ rgba, float *__restrict__ saved, DevParams P)
{
    static_assert(HALVES == 1 || NB == 1, "two waves per block: one block per wave");
    __shared__ float4 s_trec[kTileCap * kTRecPad4];
    __shared__ float s_half[HALVES == 2 ? kThreads : 1];
    const int tiles_x = (P.S + kTile - 1) / kTile;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // neighbouring waves sit on different SIMDs: a block's two halves are waves 2k and 2k + 1
    const int wave = HALVES == 2 ? wave_id >> 1 : wave_id, part = HALVES == 2 ? wave_id & 1 : 0;
    int tile_x, tile_y;
    tile_position(blockIdx.y, blockIdx.z, tiles_x, SV3D_SPIRAL && HALVES == 2, tile_x, tile_y);
    const size_t tile = ((size_t)b * tiles_x + tile_y) * tiles_x + tile_x;
    // the list's ids are requested alongside its length, not behind it
    SV3D_STAMP(0);
    const int my_id = (int)threadIdx.x < kTileCap ? (int)tlist[tile * kTileCap + threadIdx.x] : 0;
    const int n_list = tcount[tile];
#ifdef SV3D_TRACE
    if (threadIdx.x == 0 && SV3D_WG_ID < 16384) {
        g_phase[SV3D_WG_ID * 8 + 1] = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(my_id < 0 ? 1 : 0);
        g_phase[SV3D_WG_ID * 8 + 7] = (unsigned long long)n_list;
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_phase[SV3D_WG_ID * 8 + 5] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
    const int S = P.S, F = num_faces;
    const int tile_c0 = tile_x * kTile, tile_r0 = tile_y * kTile;
    const int quad = lane >> 2, sub = lane & 3;
    // lane -> pixel (lc, lr) of the wave's (left) block, and with NB = 2 also (lc + 8, lr) of the right one
    const int bc0 = NB == 2 ? 0 : (wave & 1) * 8, br0 = NB == 2 ? wave * 8 : (wave >> 1) * 8;
    const int lc = bc0 + (quad & 3) * 2 + (sub & 1), lr = br0 + (quad >> 2) * 2 + (sub >> 1);
    const int col = tile_c0 + lc, row = tile_r0 + lr;
    float A0 = 1.f, A1 = 1.f;                   // products of the visits' denominators, see tile_visit
    if (n_list > 0) {                                                   // uniform over the workgroup
        // pixel centres relative to the tile centre: odd multiples of 1/S, exact when S is a power of two
        // (otherwise rounded by 6e-8 of at most 15/S: a thirtieth of what rounding the centre itself costs)
        const float ox0 = (float)(2 * lc - (kTile - 1)) * P.inv_S, ox1 = (float)(2 * lc + 1) * P.inv_S;
        const float oy = -(float)(2 * lr - (kTile - 1)) * P.inv_S;
        // block centres (-+bcx, bcy) and half extent (between the outermost pixel centres)
        const float bcy = br0 ? -8.f * P.inv_S : 8.f * P.inv_S, bcx = 8.f * P.inv_S;
        const float bh = 7.f * P.inv_S;
        const float cx_num = (float)(2 * tile_c0 + kTile - S), cy_num = (float)(S - kTile - 2 * tile_r0);
        const unsigned long long lane_lt = (1ull << lane) - 1ull;
        int hits_seen = 0;                                              // hits of this block so far (wave-uniform)

        auto process_window = [&](int nw, bool last) {
            __syncthreads();
            SV3D_STAMP(2);
            for (int j0 = 0; j0 < nw; j0 += 64) {
                const int j = j0 + lane;
                bool hit0 = false, hit1 = false;
                if (j < nw) {
                    const float4 *r = &s_trec[j * kTRecPad4];
                    const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
                    const unsigned bits = __float_as_uint(r3.w);
                    const int fc0 = bits & 15, fc1 = (bits >> 4) & 15, fr0 = (bits >> 8) & 15, fr1 = (bits >> 12) & 15;
                    const bool rows = bits < kNoFace && !(fr1 < br0 || fr0 > br0 + 7);
                    // the largest h over a block, per edge: behind an edge line by more than the visit
                    // distance everywhere means that this face changes no pixel of the block
                    const float m0 = (fabsf(r0.x) + fabsf(r0.y)) * bh + fmaf(r0.x, bcy, r1.x);
                    const float m1 = (fabsf(r1.y) + fabsf(r1.z)) * bh + fmaf(r1.y, bcy, r2.y);
                    const float m2 = (fabsf(r2.z) + fabsf(r2.w)) * bh + fmaf(r2.z, bcy, r3.z);
                    // h changes by -Sy per unit of x: left block at -bcx, right block at +bcx
                    const float l = fminf(fmaf(r0.y, bcx, m0), fminf(fmaf(r1.z, bcx, m1), fmaf(r2.w, bcx, m2)));
                    const float g = fminf(fmaf(-r0.y, bcx, m0), fminf(fmaf(-r1.z, bcx, m1), fmaf(-r2.w, bcx, m2)));
                    if (NB == 2) {
                        hit0 = rows && fc0 <= 7 && l >= -P.visit_r;
                        hit1 = rows && fc1 >= 8 && g >= -P.visit_r;
                    } else {                    // the wave's one block: the left or the right one
                        hit0 = rows && (bc0 ? fc1 >= 8 && g >= -P.visit_r : fc0 <= 7 && l >= -P.visit_r);
                    }
                }
                unsigned long long mask0 = __ballot(hit0);
                const unsigned long long mask1 = __ballot(hit1);
                if (HALVES == 2) {              // the block's hits alternate between its two waves
                    const bool mine = hit0 && (((hits_seen + __popcll(mask0 & lane_lt)) & 1) == part);
                    hits_seen += __popcll(mask0);
                    mask0 = __ballot(mine);
                }
                unsigned long long mask = mask0 | mask1;
                while (mask) {
                    const int bit = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    const float4 *r = &s_trec[(j0 + bit) * kTRecPad4];
                    const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
                    if (NB == 1 || ((mask0 >> bit) & 1ull)) A0 *= tile_visit(r0, r1, r2, r3, ox0, oy);
                    if (NB == 2 && ((mask1 >> bit) & 1ull)) A1 *= tile_visit(r0, r1, r2, r3, ox1, oy);
                }
            }
            if (!last) __syncthreads();
        };

        if (n_list <= kTileCap) {
            if ((int)threadIdx.x < n_list) {    // one thread per face of the list builds its record in LDS
                // (the image's arrays from a scalar base, the face by a 32-bit offset)
                const float4 *q = reinterpret_cast<const float4 *>(
                    reinterpret_cast<const char *>(srec + (size_t)b * F * 3) + __umul24((unsigned)my_id, 48u));
                make_tile_record(q[0], q[1], q[2], (pbox + (size_t)b * F)[my_id], tile_c0, tile_r0, cx_num, cy_num, (float)S, P.inv_S,
                                 &s_trec[threadIdx.x * kTRecPad4]);
            }
            process_window(n_list, true);
        } else {
            // More faces than a list holds (a dense mesh drawn small): the tile goes through all of
            // the image's faces itself, a window of slots at a time, in face order.
            const int tc1 = tile_c0 + kTile - 1, tr1 = tile_r0 + kTile - 1;
            for (int f0 = 0; f0 < F; f0 += kTileCap) {                  // uniform over the workgroup
                const int nw = min(kTileCap, F - f0);
                if ((int)threadIdx.x < nw) {
                    const size_t f = (size_t)b * F + f0 + threadIdx.x;
                    const uint2 box = pbox[f];
                    int a0, a1, b0, b1;
                    unpack_box(box, a0, a1, b0, b1);
                    float4 *o = &s_trec[threadIdx.x * kTRecPad4];
                    if (!(a1 < tile_c0 || a0 > tc1 || b1 < tile_r0 || b0 > tr1)) {
                        const float4 *q = srec + f * 3;
                        make_tile_record(q[0], q[1], q[2], box, tile_c0, tile_r0, cx_num, cy_num, (float)S, P.inv_S, o);
                    } else {
                        o[3].w = __uint_as_float(kNoFace);
                    }
                }
                process_window(nw, f0 + kTileCap >= F);
            }
        }
        SV3D_STAMP(3);
        if (HALVES == 2) {                      // n_list is uniform over the workgroup: every wave is here
            if (part == 1) s_half[wave * 64 + lane] = A0;
            __syncthreads();
            if (part == 1) return;
            A0 *= s_half[wave * 64 + lane];
        }
    } else if (HALVES == 2 && part == 1) {
        return;
    }
    // (per image a base in scalar registers; per lane a 24-bit index: S <= 4096)
    const bool only_alpha = P.alpha_only;
    const int planes = only_alpha ? 1 : 4;
    float *saved_b = saved + (size_t)b * S * S;
    const int H = __builtin_amdgcn_readfirstlane(P.aa == 2 ? S >> 1 : S);
    const unsigned plane = (unsigned)H * (unsigned)H;
    float *rgba_b = rgba + (size_t)b * planes * plane;
    float *alpha_b = rgba_b + (size_t)(planes - 1) * plane;
#pragma unroll
    for (int side = 0; side < NB; ++side) {
        const float A = fminf(rcp(side ? A1 : A0), 1.f);
        const float alpha = 1.f - A;
        const int c = col + 8 * side;
        const bool inimg = c < S && row < S;
        if (inimg) saved_b[__umul24((unsigned)row, (unsigned)S) + (unsigned)c] = A;
        if (P.aa == 2) {
            const unsigned ob = __umul24((unsigned)(row >> 1), (unsigned)H) + (unsigned)(c >> 1);
            float v = inimg ? alpha : 0.f;
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            if (sub == 0 && inimg) {
                alpha_b[ob] = 0.25f * v;
                if (!only_alpha) { rgba_b[ob] = P.bg0; rgba_b[ob + plane] = P.bg1; rgba_b[ob + 2 * plane] = P.bg2; }
            }
        } else if (inimg) {
            const unsigned ob = __umul24((unsigned)row, (unsigned)S) + (unsigned)c;
            alpha_b[ob] = alpha;
            if (!only_alpha) { rgba_b[ob] = P.bg0; rgba_b[ob + plane] = P.bg1; rgba_b[ob + 2 * plane] = P.bg2; }
        }
    }
    SV3D_STAMP(4);
}

// Small launches, bound by their longest waves: the tile's four waves take its hits IN TURN, each
// hit for all of the tile's four 8x8 blocks that it touches (a lane holds one pixel of every block).
// A wave's work is then a quarter of the tile's whatever the hits' spread over the blocks, a record
// is read from LDS once per hit, and four waves per tile keep launches of up to 2048 tiles
// resident at once (the eight waves of HALVES = 2 above are two rounds of what the chip holds from
// 1024 tiles on, and their visits keep the CU's LDS as busy as its four SIMDs).  The waves' partial
// products meet in LDS, and wave w finishes block w.  Measured against the eight-wave form: forward
// -8 % at 8 images of 256^2, -13 % at 12; +5..16 % at 4 and 5 images (1024 and 1280 tiles: kept there).
__global__ __launch_bounds__(kThreads) void raster_forward_tiled_shared_kernel(
    const int *__restrict__ tcount, const unsigned short *__restrict__ tlist, const float4 *__restrict__ srec,
    const uint2 *__restrict__ pbox, int num_faces, float *__restrict__ rgba, float *__restrict__ saved, DevParams P)
{
    __shared__ float4 s_trec[kTileCap * kTRecPad4];
    __shared__ float s_part[4][4][64];          // [wave][block][lane]
    const int tiles_x = (P.S + kTile - 1) / kTile;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile_x, tile_y;
    tile_position(blockIdx.y, blockIdx.z, tiles_x, SV3D_SPIRAL, tile_x, tile_y);
    const size_t tile = ((size_t)b * tiles_x + tile_y) * tiles_x + tile_x;
    SV3D_STAMP(0);
    const int my_id = (int)threadIdx.x < kTileCap ? (int)tlist[tile * kTileCap + threadIdx.x] : 0;
    const int n_list = tcount[tile];
#ifdef SV3D_TRACE
    if (threadIdx.x == 0 && SV3D_WG_ID < 16384) {
        g_phase[SV3D_WG_ID * 8 + 1] = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(my_id < 0 ? 1 : 0);
        g_phase[SV3D_WG_ID * 8 + 7] = (unsigned long long)n_list;
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_phase[SV3D_WG_ID * 8 + 5] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
    const int S = P.S, F = num_faces;
    const int tile_c0 = tile_x * kTile, tile_r0 = tile_y * kTile;
    const int quad = lane >> 2, sub = lane & 3;
    // lane -> pixel (lc, lr) of block 0; block k (0 .. 3) lies 8 (k & 1) columns and 8 (k >> 1) rows further
    const int lc = (quad & 3) * 2 + (sub & 1), lr = (quad >> 2) * 2 + (sub >> 1);
    float Afin = 1.f;                           // the product of this wave's block (block `wave`)
    if (n_list > 0) {                                                   // uniform over the workgroup
        float A[4] = {1.f, 1.f, 1.f, 1.f};      // products of this wave's visits' denominators, per block
        const float ox[2] = {(float)(2 * lc - (kTile - 1)) * P.inv_S, (float)(2 * lc + 1) * P.inv_S};
        const float oy[2] = {-(float)(2 * lr - (kTile - 1)) * P.inv_S, -(float)(2 * lr + 1) * P.inv_S};
        const float bc = 8.f * P.inv_S, bh = 7.f * P.inv_S;             // block centres (-+bc, +-bc), half extent
        const float cx_num = (float)(2 * tile_c0 + kTile - S), cy_num = (float)(S - kTile - 2 * tile_r0);
        const unsigned long long lane_lt = (1ull << lane) - 1ull;
        int hits_seen = 0;                                              // hits of the tile so far (wave-uniform)

        auto process_window = [&](int nw, bool last) {
            __syncthreads();
            SV3D_STAMP(2);
            for (int j0 = 0; j0 < nw; j0 += 64) {
                const int j = j0 + lane;
                bool h0 = false, h1 = false, h2 = false, h3 = false;
                if (j < nw) {
                    const float4 *r = &s_trec[j * kTRecPad4];
                    const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
                    const unsigned bits = __float_as_uint(r3.w);
                    const int fc0 = bits & 15, fc1 = (bits >> 4) & 15, fr0 = (bits >> 8) & 15, fr1 = (bits >> 12) & 15;
                    const bool face = bits < kNoFace;
                    // the largest h over a block, per edge (see raster_forward_tiled_kernel): at the
                    // tile's centre, then moved to the upper / lower and left / right block centres
                    const float e0 = (fabsf(r0.x) + fabsf(r0.y)) * bh + r1.x;
                    const float e1 = (fabsf(r1.y) + fabsf(r1.z)) * bh + r2.y;
                    const float e2 = (fabsf(r2.z) + fabsf(r2.w)) * bh + r3.z;
                    const float u0 = fmaf(r0.x, bc, e0), u1 = fmaf(r1.y, bc, e1), u2 = fmaf(r2.z, bc, e2);       // upper
                    const float d0 = fmaf(-r0.x, bc, e0), d1 = fmaf(-r1.y, bc, e1), d2 = fmaf(-r2.z, bc, e2);    // lower
                    const float ul = fminf(fmaf(r0.y, bc, u0), fminf(fmaf(r1.z, bc, u1), fmaf(r2.w, bc, u2)));
                    const float ur = fminf(fmaf(-r0.y, bc, u0), fminf(fmaf(-r1.z, bc, u1), fmaf(-r2.w, bc, u2)));
                    const float dl = fminf(fmaf(r0.y, bc, d0), fminf(fmaf(r1.z, bc, d1), fmaf(r2.w, bc, d2)));
                    const float dr = fminf(fmaf(-r0.y, bc, d0), fminf(fmaf(-r1.z, bc, d1), fmaf(-r2.w, bc, d2)));
                    const bool up = face && fr0 <= 7, down = face && fr1 >= 8, left = fc0 <= 7, right = fc1 >= 8;
                    h0 = up && left && ul >= -P.visit_r;
                    h1 = up && right && ur >= -P.visit_r;
                    h2 = down && left && dl >= -P.visit_r;
                    h3 = down && right && dr >= -P.visit_r;
                }
                const bool any = h0 || h1 || h2 || h3;
                const unsigned long long m_any = __ballot(any);
                const bool mine = any && (((hits_seen + __popcll(m_any & lane_lt)) & 3) == wave);
                hits_seen += __popcll(m_any);
                const unsigned long long m0 = __ballot(mine && h0), m1 = __ballot(mine && h1),
                                         m2 = __ballot(mine && h2), m3 = __ballot(mine && h3);
                unsigned long long mask = m0 | m1 | m2 | m3;
                while (mask) {
                    const int bit = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    const float4 *r = &s_trec[(j0 + bit) * kTRecPad4];
                    const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
                    if ((m0 >> bit) & 1ull) A[0] *= tile_visit(r0, r1, r2, r3, ox[0], oy[0]);
                    if ((m1 >> bit) & 1ull) A[1] *= tile_visit(r0, r1, r2, r3, ox[1], oy[0]);
                    if ((m2 >> bit) & 1ull) A[2] *= tile_visit(r0, r1, r2, r3, ox[0], oy[1]);
                    if ((m3 >> bit) & 1ull) A[3] *= tile_visit(r0, r1, r2, r3, ox[1], oy[1]);
                }
            }
            if (!last) __syncthreads();
        };

        if (n_list <= kTileCap) {
            if ((int)threadIdx.x < n_list) {    // one thread per face of the list builds its record in LDS
                const float4 *q = reinterpret_cast<const float4 *>(
                    reinterpret_cast<const char *>(srec + (size_t)b * F * 3) + __umul24((unsigned)my_id, 48u));
                make_tile_record(q[0], q[1], q[2], (pbox + (size_t)b * F)[my_id], tile_c0, tile_r0, cx_num, cy_num, (float)S, P.inv_S,
                                 &s_trec[threadIdx.x * kTRecPad4]);
            }
            process_window(n_list, true);
        } else {
            // more faces than a list holds: all of the image's faces, a window of slots at a time
            const int tc1 = tile_c0 + kTile - 1, tr1 = tile_r0 + kTile - 1;
            for (int f0 = 0; f0 < F; f0 += kTileCap) {                  // uniform over the workgroup
                const int nw = min(kTileCap, F - f0);
                if ((int)threadIdx.x < nw) {
                    const size_t f = (size_t)b * F + f0 + threadIdx.x;
                    const uint2 box = pbox[f];
                    int a0, a1, b0, b1;
                    unpack_box(box, a0, a1, b0, b1);
                    float4 *o = &s_trec[threadIdx.x * kTRecPad4];
                    if (!(a1 < tile_c0 || a0 > tc1 || b1 < tile_r0 || b0 > tr1)) {
                        const float4 *q = srec + f * 3;
                        make_tile_record(q[0], q[1], q[2], box, tile_c0, tile_r0, cx_num, cy_num, (float)S, P.inv_S, o);
                    } else {
                        o[3].w = __uint_as_float(kNoFace);
                    }
                }
                process_window(nw, f0 + kTileCap >= F);
            }
        }
        SV3D_STAMP(3);
#pragma unroll
        for (int k = 0; k < 4; ++k) s_part[wave][k][lane] = A[k];
        __syncthreads();
        Afin = (s_part[0][wave][lane] * s_part[1][wave][lane]) * (s_part[2][wave][lane] * s_part[3][wave][lane]);
    }
    const bool only_alpha = P.alpha_only;
    const int planes = only_alpha ? 1 : 4;
    float *saved_b = saved + (size_t)b * S * S;
    const int H = __builtin_amdgcn_readfirstlane(P.aa == 2 ? S >> 1 : S);
    const unsigned plane = (unsigned)H * (unsigned)H;
    float *rgba_b = rgba + (size_t)b * planes * plane;
    float *alpha_b = rgba_b + (size_t)(planes - 1) * plane;
    const float A = fminf(rcp(Afin), 1.f);
    const float alpha = 1.f - A;
    const int c = tile_c0 + lc + 8 * (wave & 1), row = tile_r0 + lr + 8 * (wave >> 1);
    const bool inimg = c < S && row < S;
    if (inimg) saved_b[__umul24((unsigned)row, (unsigned)S) + (unsigned)c] = A;
    if (P.aa == 2) {
        const unsigned ob = __umul24((unsigned)(row >> 1), (unsigned)H) + (unsigned)(c >> 1);
        float v = inimg ? alpha : 0.f;
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        if (sub == 0 && inimg) {
            alpha_b[ob] = 0.25f * v;
            if (!only_alpha) { rgba_b[ob] = P.bg0; rgba_b[ob + plane] = P.bg1; rgba_b[ob + 2 * plane] = P.bg2; }
        }
    } else if (inimg) {
        const unsigned ob = __umul24((unsigned)row, (unsigned)S) + (unsigned)c;
        alpha_b[ob] = alpha;
        if (!only_alpha) { rgba_b[ob] = P.bg0; rgba_b[ob + plane] = P.bg1; rgba_b[ob + 2 * plane] = P.bg2; }
    }
    SV3D_STAMP(4);
}

// ------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------
// LDS plan of the surface-texture backward, per wave (one face): `tex_tables` gradient tables of
// Ts*3 floats and, when staged, Ts float4 texel values.
__host__ __device__ inline int tex_tables(int Ts) { return Ts <= 64 ? 4 : 1; }
__host__ __device__ inline bool tex_staged(int Ts) { return Ts <= 256; }
__host__ __device__ inline int tex_grad_floats(int Ts) { return (tex_tables(Ts) * Ts * 3 + 3) & ~3; }
__host__ __device__ inline int tex_lds_floats(int Ts)
{
    return tex_grad_floats(Ts) + (tex_staged(Ts) ? Ts * 4 : 0);      // both parts 16-byte multiples
}

template <int CTRL, int ROW_MASK>
__device__ inline float dpp_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(moved);
}

// One merge step of the texel-gradient combine: lanes whose `bit` of the lane id is clear take
// the values of lane + bit when both hold the same texel; the giving lane then holds none.
// TAKE / GIVE are the DPP controls that read lane + bit / lane - bit (identical for the
// symmetric quad permutations).  A source lane outside the row reads as "no texel".
template <int TAKE, int GIVE, int BIT>
__device__ inline void merge_texel_step(int &tid, float &g0, float &g1, float &g2, int lane)
{
    const bool taker = (lane & BIT) == 0;
    const int up = __builtin_amdgcn_update_dpp(-1, tid, TAKE, 0xf, 0xf, false);
    const int dn = TAKE == GIVE ? up : __builtin_amdgcn_update_dpp(-1, tid, GIVE, 0xf, 0xf, false);
    // (a source lane outside the row reads 0: with that the partner's value can ride on the
    // multiply-add itself as its cross-lane operand, instead of a move of its own)
    const float p0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(g0), TAKE, 0xf, 0xf, true));
    const float p1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(g1), TAKE, 0xf, 0xf, true));
    const float p2 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(g2), TAKE, 0xf, 0xf, true));
    const bool take = taker && up == tid;
    const bool give = !taker && dn == tid;
    const float takef = take ? 1.f : 0.f;
    g0 = fmaf(p0, takef, g0);
    g1 = fmaf(p1, takef, g1);
    g2 = fmaf(p2, takef, g2);
    tid = give ? -2 - lane : tid;
}

// Sum over the 64 lanes with a fixed tree; the total is valid in lane 63 only.
__device__ inline float wave_sum_lane63(float v)
{
    v = dpp_add<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);     // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);     // row_mirror: every lane of a row holds the row sum
    v = dpp_add<0x142, 0xa>(v);     // row_bcast:15: rows 1 and 3 add the row before them
    v = dpp_add<0x143, 0xc>(v);     // row_bcast:31: rows 2 and 3 add rows 0+1
    return v;
}

// `saved` are the forward's per-raster-pixel aggregates (1 float in silhouette mode, 6 in the
// colour modes) and `grad_rgba` the incoming gradient at output resolution; both are folded per
// visit, so there is no per-pixel preparation pass and no scratch buffer.
// HALVES = 2 (small launches): two waves per face, each taking every other 64-pixel step and
// the first of them adding the second's totals, exchanged through LDS, before it writes.  With
// few faces the launch is bound by its longest waves, not by issue slots, and this halves them.
// SPANS: which pixels of the face's box the lanes walk.  false: all of them, row-major.  true: per
// pixel row only the interval of columns that can hold an active pixel (below).  Finding a lane's
// row then costs a short search per step, which pays once the faces are large enough in pixels
// (the host decides from raster size and face count).
// The backward kernel's leading arguments, listed ONCE: the kernel's parameter list and the struct
// that mirrors its argument block are both generated from this list, so they cannot part ways.
// (The argument block of a kernel is laid out like a struct: every argument at its natural
// alignment, in order.  The kernel reads its by-value parameter block through the block's address,
// at the offset this struct gives.)
#define SV3D_BWD_ARGS(X)                                                                             \
    X(const float4 *__restrict__, rec) X(int, nf) X(int, order_batch) X(const float *__restrict__, saved) \
    X(const float *__restrict__, grad_rgba) X(int, S_) X(int, aa_) X(int, alpha_only_) X(int, F_)     \
    X(unsigned, inv_F_) X(float, inv_S_) X(const float *__restrict__, tex)                          \
    X(const uint2 *__restrict__, pbox) X(float *__restrict__, gfv) X(float *__restrict__, gtex)
#define SV3D_AS_FIELD(type, name) type name;
#define SV3D_AS_PARAM(type, name) type name,
struct BackwardArgsLayout {
    SV3D_BWD_ARGS(SV3D_AS_FIELD)
    DevParams P;
};
static_assert(offsetof(BackwardArgsLayout, tex) == 56, "the preloaded arguments are the first 14 dwords");

template <int MODE, bool LO, int HALVES, bool SPANS>
__global__ __launch_bounds__(kThreads) void raster_backward_kernel(
    // The leading 14 dwords are what a wave needs to request its face's record and, once that is
    // there, its first pixels; built with kernel argument preloading they arrive in registers with
    // the wave: the record request goes out at once instead of after a first trip to memory for
    // the argument block.  nf = B * F; order_batch = B; S_ .. inv_S_ repeat fields of the
    // parameter block, which is requested together with the first pixels (below).
    SV3D_BWD_ARGS(SV3D_AS_PARAM) DevParams P_block)
{
    extern __shared__ float s_dyn[];            // SURFACE: per wave, texel gradient tables + texel values
#ifdef SV3D_TRACE
    const unsigned long long trace_t0 = __builtin_amdgcn_s_memrealtime();
#endif

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Workgroups are handed to the 8 XCDs round-robin and each XCD has its own L2.  Consecutive
    // faces read overlapping pixels of one image, so the launch order is permuted to give every
    // XCD a contiguous range of faces: an image's pixels then miss in one L2 instead of eight.
    // Small batches keep the plain order: with one or two images per XCD the images' unequal
    // work would unbalance the XCDs (measured: +1 us at batch 8, -2..5 % from batch 64 on).
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, per = nwg >> 3, rem = nwg & 7;
    const int wg = order_batch >= 32 ? xcd * per + min(xcd, rem) + (blockIdx.x >> 3) : blockIdx.x;
    // global face index b*F + f, and which share of its steps this wave takes
    const int half = HALVES == 2 ? (wave & 1) : 0;
    const int gf_raw = HALVES == 2 ? wg * 2 + (wave >> 1) : wg * 4 + wave;
    const bool live = gf_raw < nf;
    if (HALVES == 1 && !live) return;           // whole wave; HALVES == 1 uses no block-level barrier
    const int gf = live ? gf_raw : nf - 1;      // HALVES == 2: idle waves still reach the barrier
    // The record is requested first, from the preloaded arguments alone, and waited for.
    // (The eight-channel mode, whose registers leave no room for a record in scalar registers across
    // the arithmetic, reads it with vector loads: measured +7 % otherwise.)
    constexpr bool kRequestAhead = MODE != SV3D_TEX_VERTEX8;
    const FaceRec r = kRequestAhead ? load_rec_uniform(rec + (size_t)gf * kRec4) : load_rec(rec + (size_t)gf * kRec4);
    // The parameter block is read through the address of the argument block.  Left to the
    // compiler, its loads are issued ahead of the record request and, a register of theirs being
    // reused, waited for before it: a trip to memory in front of the record's.  The address is
    // therefore made to depend on the record (an empty statement the compiler cannot see
    // through): the block's loads then go out behind the record, alongside the first pixel loads,
    // which need none of it -- what they need repeats in the preloaded arguments.
    typedef const char __attribute__((address_space(4))) *ConstantBytes;
    ConstantBytes kargs = (ConstantBytes)__builtin_amdgcn_kernarg_segment_ptr();
    if (kRequestAhead) asm volatile("" : "+s"(kargs) : "s"(r.flags));
    const DevParams P = copy_params((ConstantParams)(kargs + offsetof(BackwardArgsLayout, P)));
    (void)P_block;
    // The pixel box is taken from the record (floats 28..29), not from the compact box array: one
    // request, one wait, then the pixel loads can go out.
    const int b = image_of_face(gf, F_, inv_F_);
    const int tgf = P.TB == P.B ? gf : gf - (b - b % P.TB) * F_;      // the face in its texture set
    const int S = S_;
    constexpr int NC = mode_channels(MODE);     // colour channels
    constexpr int PP = MODE == SV3D_TEX_NONE ? 1 : (NC == 3 ? 6 : 12);   // saved floats per pixel
    const float step = 2.f * inv_S_;
    const float *saved_b = saved + (size_t)b * S * S * PP;
    const int H = aa_ == 2 ? S >> 1 : S;
    const float gscale = aa_ == 2 ? 0.25f : 1.f;
    // d(+-d2 / sigma) / d(d2), doubled for d(d2) / d(foot - pixel)
    const float dist_slope = (MODE == SV3D_TEX_NONE ? 2.f * gscale : 2.f) * P.inv_sigma;
    const size_t plane = (size_t)H * H;
    const bool only_alpha = MODE == SV3D_TEX_NONE && alpha_only_;
    // colour + alpha planes of image b, or its alpha plane alone
    const float *grad_b = grad_rgba + (size_t)b * (only_alpha ? 1 : NC + 1) * plane;
    const float *grad_alpha = grad_b + (only_alpha ? 0 : NC * plane);      // alpha plane of image b
    const float fS = (float)S;
    int col0, col1, row0, row1;
    unpack_box(make_uint2(r.box_x, r.box_y), col0, col1, row0, row1);
    const int bw = col1 - col0 + 1, bh = row1 - row0 + 1;
    const int npix = (live && bw > 0 && bh > 0) ? bw * bh : 0;
#ifdef SV3D_TRACE
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the record and the box are here
    const unsigned long long trace_ta = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(r.p0x != 12345.f ? 0 : 1);
#endif

    // SURFACE: the face's texel gradients accumulate in wave-private LDS tables.  Lanes of one
    // step mostly hit the same few texels, so small textures get four tables (lane & 3 picks one,
    // summed in a fixed order at the end) to cut the serialisation of same-address adds; the
    // face's texel values are staged beside them as float4 when they fit.
    const int Ts = P.R * P.R;
    const int ncopy = tex_tables(Ts);
    const bool staged = tex_staged(Ts);
    float *s_tex = s_dyn + wave * tex_lds_floats(Ts);
    float *s_val = s_tex + tex_grad_floats(Ts);
    float tv[3 * NC];                           // VERTEX modes: the face's per-corner colours
    if (MODE == SV3D_TEX_SURFACE) {
        for (int i = lane; i < ncopy * Ts * 3; i += 64) s_tex[i] = 0.f;
        if (staged) {
            for (int i = lane; i < Ts; i += 64) {
                const float *tp = tex + ((size_t)tgf * Ts + i) * 3;
                reinterpret_cast<float4 *>(s_val)[i] = make_float4(tp[0], tp[1], tp[2], 0.f);
            }
        }
    } else if (mode_vertex(MODE)) {
#pragma unroll
        for (int i = 0; i < 3 * NC; ++i) tv[i] = tex[(size_t)tgf * (3 * NC) + i];
    }

    float acc[9];
    float tacc[3 * NC];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 3 * NC; ++i) tacc[i] = 0.f;

    // SPANS: lanes enumerate, row by row, the pixels of the box that can be active at all, 64 per
    // step.  A pixel is active only if it is inside the face or nearer to it than the cull distance,
    // so it lies on the inner side of every edge line moved outwards by that distance: on each pixel
    // row that is one interval of columns (the box is the bounding box of this set grown by the
    // same distance, and for typical faces a third of it lies outside).  Rows are handled 64 at a
    // time, one per lane: the lane works out its row's interval [lo, lo + len), a prefix sum of the
    // lengths numbers the pixels, and a step takes 64 consecutive numbers.  The intervals err on
    // the wide side; whether a pixel counts is still decided by pair_geom alone.
    // Otherwise lanes enumerate the whole box row-major, 64 pixels per step: (rx, ry) = position
    // inside the box, kept as exact small integers in fp32 and advanced by kStride pixels per step
    // with at most one row wrap per lane.
    // Either way the per-pixel inputs of step i+1 are requested before step i is evaluated, so that
    // their latency overlaps the arithmetic.
    const float bw_f = (float)bw, bh_f = (float)bh;
    float xp0, yp0;
    pixel_centre(col0, row0, S, inv_S_, xp0, yp0);
    constexpr int kStride = 64 * HALVES;        // pixel numbers between two steps of one wave
    const int rows_live = npix > 0 ? bh : 0;
    float adv_x = 0.f, adv_y = 0.f, rx_f = 0.f, ry_f = 0.f;
    // without anti-aliasing the lane's pixel index (as a byte offset: times four) is carried along with
    // its position: a step adds one of two wave-uniform increments (with or without a row wrap)
    int pix_run = 0, pix_adv = 0, pix_adv_wrap = 0;
    if (!SPANS) {
        const float inv_bw = bw > 0 ? rcp((float)bw) : 0.f;     // the fix-ups below absorb its last-bit error
        // kStride = adv_y * bw + adv_x, by the reciprocal and the same fix-ups as the lanes' start below
        adv_y = floorf(((float)kStride + 0.5f) * inv_bw);
        adv_x = fmaf(-adv_y, bw_f, (float)kStride);
        if (adv_x < 0.f) { adv_y -= 1.f; adv_x += bw_f; }
        if (adv_x >= bw_f) { adv_y += 1.f; adv_x -= bw_f; }
        if (bw <= 0) adv_x = adv_y = 0.f;
        const float lane0 = (float)(half * 64 + lane);
        ry_f = floorf((lane0 + 0.5f) * inv_bw);
        rx_f = fmaf(-ry_f, bw_f, lane0);
        if (rx_f < 0.f) { ry_f -= 1.f; rx_f += bw_f; }
        if (rx_f >= bw_f) { ry_f += 1.f; rx_f -= bw_f; }
        pix_run = ((row0 + (int)ry_f) * S + col0 + (int)rx_f) * 4;
        pix_adv = ((int)adv_y * S + (int)adv_x) * 4;
        pix_adv_wrap = pix_adv + (S - bw) * 4;
    }

    // Raw per-pixel inputs: the forward's saved aggregates and the incoming gradient.  They are
    // combined where they are used, so that the loads stay in flight across a step.
    // integers 2c + 1 - S of the box's first column and row (exact in fp32)
    const float kx0 = (float)(2 * col0 + 1 - S), ky0 = (float)(2 * (S - 1 - row0) + 1 - S);
    struct PixIn {
        float xp, yp, xl, yl;
        float A, ssum, m, col[NC];          // saved: 1 - alpha product, softmax sum, max, colours
        float ga, gch[NC];                  // dL/d(alpha, colours) of the output pixel
        bool valid;
    };
    // per-pixel loads below take a wave-uniform base and a 32-bit byte offset (planes and the saved
    // array of one image stay far below 4 GB), which the hardware adds itself
    auto at32 = [](const float *base, unsigned index4) -> float {       // index4: four times the float's index
        return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + index4);
    };
    // SPANS, of the current 64 rows: inclusive prefix sum of the interval lengths, start << 12 | lo
    // per row, pixels in all, first row.  Otherwise span_total is the box's pixel count.
    int span_cum = 0, span_pack = 0, span_total = SPANS ? 0 : npix, span_row0 = 0;
    auto fetch = [&](int base) {                    // inputs of pixel number base + lane
        PixIn in;
        float cx, cy;
        unsigned pix = 0, gpix = 0;             // four times the pixel's index in the raster / the output image
        if (SPANS) {
            const int pnum = base + lane;
            in.valid = pnum < span_total;
            // The rows this step touches run from the one holding number `base` to the one holding
            // the step's last number; a lane's row is the last of them that starts at or before its
            // number.  span_pack is ordered like the starts, so it is compared whole.
            const int last = min(base + 63, span_total - 1);
            const int r_first = __builtin_ctzll(__ballot(span_cum > base) | (1ull << 63));
            const int r_last = __builtin_ctzll(__ballot(span_cum > last) | (1ull << 63));
            const int key = (pnum << 12) | 0xfff;
            int cur = __builtin_amdgcn_readlane(span_pack, r_first);
            int rowv = r_first;
            if (MODE == SV3D_TEX_SURFACE) {
                // this mode keeps the LDS pipe busy with its texel tables: the rows are walked with
                // scalar reads of their lanes instead
                for (int rr = r_first + 1; rr <= r_last; ++rr) {        // wave-uniform
                    const int pk = __builtin_amdgcn_readlane(span_pack, rr);
                    const bool ok = pk <= key;
                    rowv = ok ? rr : rowv;
                    cur = ok ? pk : cur;
                }
            } else {
                // a descent by halving strides over the row lanes, read through the cross-lane network
                int pos4 = r_first * 4;                                  // byte index of the row's lane
                const int more = r_last - r_first;                       // wave-uniform
                if (more > 0) {
                    for (int stride4 = 4 << (31 - __builtin_clz(more)); stride4 >= 4; stride4 >>= 1) {
                        const int cand4 = min(pos4 + stride4, 252);
                        const int pk = __builtin_amdgcn_ds_bpermute(cand4, span_pack);
                        const bool ok = pk <= key;
                        pos4 = ok ? cand4 : pos4;
                        cur = ok ? pk : cur;
                    }
                }
                rowv = pos4 >> 2;
            }
            const int colv = pnum - (cur >> 12) + (cur & 0xfff);
            cx = in.valid ? (float)colv : 0.f;
            cy = in.valid ? (float)(span_row0 + rowv) : 0.f;
        } else {
            in.valid = ry_f < bh_f;
            // a lane past the box keeps its (finite) position: nothing of it is used but the pixel
            // indices, which are made valid ones
            cx = rx_f;
            cy = ry_f;
            pix = gpix = in.valid ? (unsigned)pix_run : 0u;
            if (aa_ == 2) {                     // wave-uniform
                const unsigned prow = (unsigned)(row0 + (int)cy), pcol = (unsigned)(col0 + (int)cx);
                gpix = in.valid ? (__umul24(prow >> 1, (unsigned)H) + (pcol >> 1)) << 2 : 0u;
            }
            // advance to the pixel kStride positions further
            rx_f += adv_x;
            ry_f += adv_y;
            const bool wrap = rx_f >= bw_f;
            rx_f -= wrap ? bw_f : 0.f;
            ry_f += wrap ? 1.f : 0.f;
            pix_run += wrap ? pix_adv_wrap : pix_adv;
        }
        in.xp = fmaf(cx, step, xp0);
        in.yp = fmaf(-cy, step, yp0);
        if (LO) {           // remainder of the exact centre over the rounded one, see struct Pixel
            in.xl = fmaf(-in.xp, fS, fmaf(2.f, cx, kx0)) * inv_S_;
            in.yl = fmaf(-in.yp, fS, fmaf(-2.f, cy, ky0)) * inv_S_;
        } else {
            in.xl = in.yl = 0.f;
        }
        if (SPANS) {
            const unsigned prow = (unsigned)(row0 + (int)cy), pcol = (unsigned)(col0 + (int)cx);
            // rows, columns and sizes are at most 4096: 24-bit multiplies (one v_mad_u32_u24 each)
            pix = (__umul24(prow, (unsigned)S) + pcol) << 2;
            gpix = aa_ == 2 ? (__umul24(prow >> 1, (unsigned)H) + (pcol >> 1)) << 2 : pix;
        }
        in.ssum = in.m = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) in.col[c] = in.gch[c] = 0.f;
        in.ga = at32(grad_alpha, gpix);
        if (MODE == SV3D_TEX_NONE) {
            in.A = at32(saved_b, pix);
        } else if (NC == 3) {
            const float2 *sp = reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(saved_b) + pix * 6u);
            const float2 a0 = sp[0], a1 = sp[1], a2 = sp[2];
            in.A = a0.x; in.ssum = a0.y; in.m = a1.x; in.col[0] = a1.y; in.col[1] = a2.x; in.col[2] = a2.y;
        } else {
            const float4 *sp = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(saved_b) + pix * 12u);
            const float4 a0 = sp[0], a1 = sp[1], a2 = sp[2];
            in.A = a0.x; in.ssum = a0.y; in.m = a0.z;
            const float cs[8] = {a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z};
#pragma unroll
            for (int c = 0; c < NC; ++c) in.col[c] = cs[c < 8 ? c : 7];
        }
        if (MODE != SV3D_TEX_NONE) {
#pragma unroll
            for (int c = 0; c < NC; ++c) in.gch[c] = at32(grad_b + c * plane, gpix);
        }
        return in;
    };

#ifdef SV3D_TRACE
    unsigned long long trace_tb = trace_ta;
#endif
    // SPANS: 64 rows of the box at a time (mostly all of them); otherwise one pass over the box
    for (int rg = 0; rg < (SPANS ? rows_live : 1); rg += 64) {
    if (SPANS) {
        // Edge k's scaled edge function at column i of a row is W_k - i * g_k (W_k its value at the
        // box's first column); the moved line is W_k - i * g_k >= -c_k, c_k = margin |e_k| / |det|.
        // An edge closer to the row direction than 1:16 bounds nothing here (its crossing is too
        // sensitive to rounding); the slack covers the rounding of the others up to the largest S.
        const float grow = P.margin * fabsf(r.inv_det) * (1.f + 1.f / 1024.f);
        const float c_e0 = grow * __builtin_amdgcn_rsqf(r.il0), c_e1 = grow * __builtin_amdgcn_rsqf(r.il1),
                    c_e2 = grow * __builtin_amdgcn_rsqf(r.il2);
        const float g_e0 = r.f0y * step, g_e1 = r.f1y * step, g_e2 = r.f2y * step;
        const bool use_e0 = fabsf(r.e0y) * 16.f > fabsf(r.e0x) && g_e0 != 0.f,
                   use_e1 = fabsf(r.e1y) * 16.f > fabsf(r.e1x) && g_e1 != 0.f,
                   use_e2 = fabsf(r.e2y) * 16.f > fabsf(r.e2x) && g_e2 != 0.f;
        const float ig_e0 = rcp(g_e0), ig_e1 = rcp(g_e1), ig_e2 = rcp(g_e2);
        const float span_slack = 0.02f + 1.6e-5f * fS;
        const float ax_e0 = r.p0x - xp0, ax_e1 = r.p1x - xp0, ax_e2 = r.p2x - xp0;
        const int row_in_box = rg + lane;
        const float yrow = fmaf(-(float)row_in_box, step, yp0);
        const float ay0 = r.p0y - yrow, ay1 = r.p1y - yrow, ay2 = r.p2y - yrow;
        // edge k's function at the first column, plus its allowance, over its change per column
        const float q0 = (fmaf(r.f0y, ax_e0, -(r.f0x * ay0)) + c_e0) * ig_e0;
        const float q1 = (fmaf(r.f1y, ax_e1, -(r.f1x * ay1)) + c_e1) * ig_e1;
        const float q2 = (fmaf(r.f2y, ax_e2, -(r.f2x * ay2)) + c_e2) * ig_e2;
        float lo_f = 0.f, hi_f = bw_f - 1.f;
        // g > 0: columns up to q; g < 0: columns from q on (a NaN or an unused edge bounds nothing)
        if (use_e0) { if (g_e0 > 0.f) hi_f = fminf(hi_f, floorf(q0 + span_slack)); else lo_f = fmaxf(lo_f, ceilf(q0 - span_slack)); }
        if (use_e1) { if (g_e1 > 0.f) hi_f = fminf(hi_f, floorf(q1 + span_slack)); else lo_f = fmaxf(lo_f, ceilf(q1 - span_slack)); }
        if (use_e2) { if (g_e2 > 0.f) hi_f = fminf(hi_f, floorf(q2 + span_slack)); else lo_f = fmaxf(lo_f, ceilf(q2 - span_slack)); }
        lo_f = fminf(lo_f, bw_f - 1.f);             // 12 bits hold it; an empty row may gain its last pixel
        hi_f = fmaxf(hi_f, -1.f);
        const int lo_i = (int)lo_f;
        const int len = row_in_box < rows_live ? max((int)hi_f - lo_i + 1, 0) : 0;
        span_cum = wave_inclusive_scan(len);
        span_total = __builtin_amdgcn_readlane(span_cum, 63);
        span_pack = ((span_cum - len) << 12) | lo_i;            // 64 rows of at most 4096 pixels: 18 + 12 bits
        span_row0 = rg;
    }
    PixIn nxt;
    if (span_total > half * 64) nxt = fetch(half * 64);
    // The parameter block read through the argument block's address must be the one the preloaded
    // arguments repeat.  Both are generated from one list (SV3D_BWD_ARGS), so it is; should a
    // compiler ever lay the argument block out differently, no wave works on a block it cannot
    // trust, and the face's gradient is written as NaN rather than left as stale memory.
    if (P.S != S_ || P.F != F_ || P.inv_F != inv_F_) {
        if (live && half == 0 && lane < 9) gfv[(size_t)gf * 9 + lane] = __int_as_float(0x7fc00000);
        return;
    }
#ifdef SV3D_TRACE
    if (rg == 0) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the first pixels are here
        trace_tb = __builtin_amdgcn_s_memrealtime();
    }
#endif
    auto evaluate = [&](const PixIn &in) {
        const float xp = in.xp, yp = in.yp;
        const bool valid = in.valid;
        // fold the gradient with the aggregates: g_alpha A and g_k / ssum (silhouettes: the
        // anti-aliasing scale rides on the constant of the distance's chain rule below)
        const float P0 = MODE == SV3D_TEX_NONE ? in.A * in.ga : in.A * (in.ga * gscale);
        float G[NC];
        const float mfin = in.m;
        if (MODE != SV3D_TEX_NONE) {
            float is = rcp(in.ssum);
            is = is * (2.f - in.ssum * is) * gscale;
#pragma unroll
            for (int c = 0; c < NC; ++c) G[c] = in.gch[c] * is;
        }

        Pixel px;
        px.x = xp; px.y = yp; px.xl = in.xl; px.yl = in.yl;
        PairGeom g = pair_geom<LO, MODE == SV3D_TEX_NONE>(r, px, P);
        const bool act = valid && g.act;
        // dL/dx, x = +-d2/sigma, through alpha
        float gx = act ? P0 * (MODE == SV3D_TEX_NONE ? g.D_any : g.D) : 0.f;

        if (MODE != SV3D_TEX_NONE) {
            const PairDepth d = pair_depth(r, g, P);
            const bool contrib = valid && d.contrib;
            // SURFACE: this lane's texel and colour gradient, handed to the merge below
            int ttid = -2 - lane;                   // negative ids are unique per lane: "no texel"
            float tg0 = 0.f, tg1 = 0.f, tg2 = 0.f;
            if (contrib) {
                const float ez = ex2((d.zn - mfin) * P.l2e_over_gamma);
                float kc[NC];
                int tid = 0;
                if (MODE == SV3D_TEX_SURFACE) {
                    tid = texel_index(d.n0, d.n1, d.n2, P.R);
                    if (staged) {
                        const float4 tv4 = reinterpret_cast<const float4 *>(s_val)[tid];
                        kc[0] = tv4.x; kc[1] = tv4.y; kc[2] = tv4.z;
                    } else {
                        const float *tp = tex + ((size_t)tgf * Ts + tid) * 3;
                        kc[0] = tp[0]; kc[1] = tp[1]; kc[2] = tp[2];
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < NC; ++c) kc[c] = d.n0 * tv[c] + d.n1 * tv[NC + c] + d.n2 * tv[2 * NC + c];
                }
                // sum_k g_k (colour_k - pixel_k) / ssum, differences first: where this face
                // dominates the pixel the two colours nearly cancel
                float S1 = 0.f;
#pragma unroll
                for (int c = 0; c < NC; ++c) S1 = fmaf(G[c], kc[c] - in.col[c], S1);
                const float De = g.D * ez;
                gx += De * g.omD * S1;                               // through D of the softmax weight
                const float gz = -(De * S1) * P.inv_gamma * P.inv_range;   // dL/dzp
                float gc[NC];                                        // dL/dcolour
#pragma unroll
                for (int c = 0; c < NC; ++c) gc[c] = G[c] * De;
                const float zp2 = d.zp * d.zp;
                acc[2] += gz * zp2 * d.n0 * r.iz0 * r.iz0;
                acc[5] += gz * zp2 * d.n1 * r.iz1 * r.iz1;
                acc[8] += gz * zp2 * d.n2 * r.iz2 * r.iz2;
                float gn0 = -gz * zp2 * r.iz0, gn1 = -gz * zp2 * r.iz1, gn2 = -gz * zp2 * r.iz2;
                if (MODE == SV3D_TEX_SURFACE) {
                    ttid = tid; tg0 = gc[0]; tg1 = gc[1]; tg2 = gc[2];
                } else {
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        tacc[c] = fmaf(d.n0, gc[c], tacc[c]);
                        tacc[NC + c] = fmaf(d.n1, gc[c], tacc[NC + c]);
                        tacc[2 * NC + c] = fmaf(d.n2, gc[c], tacc[2 * NC + c]);
                        gn0 = fmaf(gc[c], tv[c], gn0);
                        gn1 = fmaf(gc[c], tv[NC + c], gn1);
                        gn2 = fmaf(gc[c], tv[2 * NC + c], gn2);
                    }
                }
                // through the normalisation n = clamp(w)/sum and the clamp
                const float dotn = gn0 * d.n0 + gn1 * d.n1 + gn2 * d.n2;
                const float gw0 = (g.w0 > 0.f && g.w0 < 1.f) ? (gn0 - dotn) * d.rs : 0.f;
                const float gw1 = (g.w1 > 0.f && g.w1 < 1.f) ? (gn1 - dotn) * d.rs : 0.f;
                const float gw2 = (g.w2 > 0.f && g.w2 < 1.f) ? (gn2 - dotn) * d.rs : 0.f;
                // grad_w_j = (-e_{j+1}.y, e_{j+1}.x) / det; moving vertex i shifts the pixel's
                // barycentrics like moving the pixel by -w_i
                const float Vx = -(gw0 * r.f1y + gw1 * r.f2y + gw2 * r.f0y);
                const float Vy = gw0 * r.f1x + gw1 * r.f2x + gw2 * r.f0x;
                acc[0] -= g.w0 * Vx; acc[1] -= g.w0 * Vy;
                acc[3] -= g.w1 * Vx; acc[4] -= g.w1 * Vy;
                acc[6] -= g.w2 * Vx; acc[7] -= g.w2 * Vy;
            }
            if (MODE == SV3D_TEX_SURFACE && gtex != nullptr) {
                // LDS float atomics retire about one lane per clock, and neighbouring lanes
                // (neighbouring pixels of a row) mostly hit the same texel: lanes of an aligned
                // group of 8 with equal texels are first merged into the lowest of them (fixed
                // order), then only the lanes left holding a texel issue the adds.
                merge_texel_step<0xB1, 0xB1, 1>(ttid, tg0, tg1, tg2, lane);     // quad_perm [1,0,3,2]
                merge_texel_step<0x4E, 0x4E, 2>(ttid, tg0, tg1, tg2, lane);     // quad_perm [2,3,0,1]
                merge_texel_step<0x104, 0x114, 4>(ttid, tg0, tg1, tg2, lane);   // row_shl:4 / row_shr:4
                // (Moving the 3 x n values of the n lanes left to the first 3 n lanes, six pushes
                // through the cross-lane network, and adding them with one instruction instead of
                // three was measured: +4..8 %; a push costs the LDS as much as the add it saves.)
                if (ttid >= 0) {
                    // the table by the lane's group of eight as well as its place in it: whole
                    // groups mostly end up in their first lane
                    // (24-bit index arithmetic: the 32-bit product would be a quarter-rate instruction)
                    const unsigned table = (unsigned)(((lane >> 3) + lane) & (ncopy - 1));
                    float *slot = s_tex + __umul24(__umul24(table, (unsigned)Ts) + (unsigned)ttid, 3u);
                    atomicAdd(slot + 0, tg0);
                    atomicAdd(slot + 1, tg1);
                    atomicAdd(slot + 2, tg2);
                }
            }
        }
        // through the distance: d2 = |foot - pixel|^2, foot = (1-t) p_k + t p_{k+1}
        const float gd = gx * copysignf(dist_slope, g.wmin);
        const float cA = gd * (1.f - g.t), cB = gd * g.t;
        const float f0 = g.k0 ? cA : (g.k2 ? cB : 0.f);
        const float f1 = g.k1 ? cA : (g.k0 ? cB : 0.f);
        const float f2 = g.k2 ? cA : (g.k1 ? cB : 0.f);
        acc[0] = fmaf(f0, g.dx, acc[0]); acc[1] = fmaf(f0, g.dy, acc[1]);
        acc[3] = fmaf(f1, g.dx, acc[3]); acc[4] = fmaf(f1, g.dy, acc[4]);
        acc[6] = fmaf(f2, g.dx, acc[6]); acc[7] = fmaf(f2, g.dy, acc[7]);
    };
    if (SV3D_BWD_UNROLL && MODE == SV3D_TEX_NONE) {
        // Two steps per trip, each with its own set of input registers: with one set the inputs
        // requested during a step have to be copied into place for the next one (the silhouette
        // step is short enough for those copies to show).
        int base = half * 64;
        if (base < span_total) {
            PixIn other;
            for (;;) {                                                  // every test is wave-uniform
                const bool more = base + kStride < span_total;
                if (more) other = fetch(base + kStride);
                evaluate(nxt);
                if (!more) break;
                base += kStride;
                const bool more2 = base + kStride < span_total;
                if (more2) nxt = fetch(base + kStride);
                evaluate(other);
                if (!more2) break;
                base += kStride;
            }
        }
    } else {
        for (int base = half * 64; base < span_total; base += kStride) {
            const PixIn in = nxt;
            if (base + kStride < span_total) nxt = fetch(base + kStride);   // wave-uniform
            evaluate(in);
        }
    }
    }

#ifdef SV3D_TRACE
    if (lane == 0 && blockIdx.x * 4 + wave < 16384) {       // stamp at the end of the pixel loop
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long *t = g_trace + ((size_t)blockIdx.x * 4 + wave) * 4;
        t[0] = trace_t0;
        t[1] = __builtin_amdgcn_s_memrealtime();
        t[2] = ((unsigned long long)xcc << 32) | hwid;
        // pixels of the box, and when the record and the first pixels had arrived (10 ns units)
        t[3] = (unsigned long long)npix | (((trace_ta - trace_t0) & 0xffffull) << 32) | (((trace_tb - trace_t0) & 0xffffull) << 48);
    }
#endif
    float tot[9];
#pragma unroll
    for (int i = 0; i < 9; ++i)                     // silhouettes carry no depth gradient
        tot[i] = (MODE == SV3D_TEX_NONE && i % 3 == 2) ? 0.f : wave_sum_lane63(acc[i]);
    float ttot[3 * NC];
    if (mode_vertex(MODE) && gtex != nullptr) {
#pragma unroll
        for (int i = 0; i < 3 * NC; ++i) ttot[i] = wave_sum_lane63(tacc[i]);
    }
    if (HALVES == 2) {
        // the odd wave of a face hands its totals to the even one (its texel tables stay where
        // they are, in its own LDS region, and are read from there below)
        __shared__ float s_half[2][9 + 3 * NC];
        if (half == 1 && lane == 63) {
#pragma unroll
            for (int i = 0; i < 9; ++i) s_half[wave >> 1][i] = tot[i];
            if (mode_vertex(MODE) && gtex != nullptr) {
#pragma unroll
                for (int i = 0; i < 3 * NC; ++i) s_half[wave >> 1][9 + i] = ttot[i];
            }
        }
        __syncthreads();
        if (half == 1 || !live) return;
        if (lane == 63) {
#pragma unroll
            for (int i = 0; i < 9; ++i) tot[i] += s_half[wave >> 1][i];
            if (mode_vertex(MODE) && gtex != nullptr) {
#pragma unroll
                for (int i = 0; i < 3 * NC; ++i) ttot[i] += s_half[wave >> 1][9 + i];
            }
        }
    }
    if (lane == 63) {
        float *o = gfv + (size_t)gf * 9;
#pragma unroll
        for (int i = 0; i < 9; ++i) o[i] = tot[i];
    }
    if (gtex != nullptr) {
        if (MODE == SV3D_TEX_SURFACE) {
            const float *s_other = s_tex + tex_lds_floats(Ts);      // HALVES == 2: the odd wave's tables
            for (int i = lane; i < Ts * 3; i += 64) {
                float t = s_tex[i];
                for (int c = 1; c < ncopy; ++c) t += s_tex[c * Ts * 3 + i];
                if (HALVES == 2) {
                    for (int c = 0; c < ncopy; ++c) t += s_other[c * Ts * 3 + i];
                }
                gtex[(size_t)gf * Ts * 3 + i] = t;
            }
        } else if (mode_vertex(MODE)) {
            if (lane == 63) {
#pragma unroll
                for (int i = 0; i < 3 * NC; ++i) gtex[(size_t)gf * (3 * NC) + i] = ttot[i];
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
#ifndef SV3D_BWD_HALVES_MAX_FACES
#define SV3D_BWD_HALVES_MAX_FACES 12288         // (measured: 10240 faces -2 % with two waves per face, 15360 faces +5 %)
#endif
constexpr long kBackwardHalvesMaxFaces = SV3D_BWD_HALVES_MAX_FACES;
int make_dev_params(const sv3d_raster_params *p, DevParams &d)
{
    if (!p) return SV3D_ERR_INVALID_ARG;
    if (p->batch <= 0 || p->num_faces <= 0 || p->raster_size <= 0) return SV3D_ERR_INVALID_ARG;
    if (p->aa != 1 && p->aa != 2) return SV3D_ERR_INVALID_ARG;
    if (p->aa == 2 && (p->raster_size & 1)) return SV3D_ERR_INVALID_ARG;
    if (p->texture_mode < SV3D_TEX_NONE || p->texture_mode > SV3D_TEX_VERTEX8) return SV3D_ERR_INVALID_ARG;
    if (p->texture_mode == SV3D_TEX_SURFACE && p->texture_res <= 0) return SV3D_ERR_INVALID_ARG;
    if (!(p->sigma > 0.f) || !(p->gamma > 0.f) || !(p->dist_eps > 0.f) || !(p->dist_eps < 1.f))
        return SV3D_ERR_INVALID_ARG;
    if (!(p->far_plane > p->near_plane)) return SV3D_ERR_INVALID_ARG;
    if (p->texture_batch < 0 || (p->texture_batch > 0 && p->batch % p->texture_batch != 0))
        return SV3D_ERR_INVALID_ARG;
    if ((p->alpha_only != 0 && p->alpha_only != 1) || (p->alpha_only && p->texture_mode != SV3D_TEX_NONE))
        return SV3D_ERR_INVALID_ARG;
    if (p->backward_enumeration < SV3D_BWD_AUTO || p->backward_enumeration > SV3D_BWD_INTERVALS)
        return SV3D_ERR_INVALID_ARG;
    if (p->raster_size > 4096 || p->batch > 65535) return SV3D_ERR_UNSUPPORTED;
    if ((long)p->batch * p->num_faces > 0x7fffffffL / 4) return SV3D_ERR_UNSUPPORTED;
    if (p->texture_mode == SV3D_TEX_SURFACE && p->texture_res > 36) return SV3D_ERR_UNSUPPORTED;
    const double l2e = 1.4426950408889634;
    d.B = p->batch; d.F = p->num_faces; d.S = p->raster_size; d.aa = p->aa;
    d.TB = p->texture_batch > 0 ? p->texture_batch : p->batch;
    d.mode = p->texture_mode; d.R = p->texture_mode == SV3D_TEX_SURFACE ? p->texture_res : 1;
    d.double_side = p->double_side;
    d.bg0 = p->background[0]; d.bg1 = p->background[1]; d.bg2 = p->background[2];
    d.near_plane = p->near_plane; d.far_plane = p->far_plane; d.eps = p->eps;
    d.neg_l2e_over_sigma = (float)(-l2e / (double)p->sigma);
    d.inv_sigma = (float)(1.0 / (double)p->sigma);
    const double thr = log(1.0 / (double)p->dist_eps - 1.0) * (double)p->sigma;
    d.threshold = (float)thr;
    d.margin = (float)sqrt(thr > 0.0 ? thr : 0.0);
    d.l2e_over_gamma = (float)(l2e / (double)p->gamma);
    d.inv_gamma = (float)(1.0 / (double)p->gamma);
    d.inv_range = (float)(1.0 / ((double)p->far_plane - (double)p->near_plane));
    d.ssum0 = (float)exp((double)p->eps / (double)p->gamma);
    d.inv_S = (float)(1.0 / (double)p->raster_size);
    d.alpha_fast = (double)p->dist_eps / (1.0 - (double)p->dist_eps) < 2.9e-8;   // below half an ulp of 1
    // The silhouette forward multiplies 1 - D = 1 / (1 + exp(-d2/sigma)) of the outside pairs into a
    // pixel's product.  From d2 = 17.5 sigma on, exp(-d2/sigma) <= 2.5e-8 is below half an ulp of 1
    // and the factor is exactly 1.0f: a block that far beyond an edge line of a face is not visited,
    // although the formula's cull distance (dist_eps = 1e-10: 23 sigma) is larger.  Every pixel still
    // multiplies the same factors (bit-equal images; where two waves share a block's hits they deal
    // them out differently, and the product's last bit may move).  The colour modes and the backward
    // keep the formula's distance: their weights D and gradients D (1 - D) of such pairs are tiny
    // but not zero.
    const double exact_one = 17.5 * (double)p->sigma;
    d.visit_threshold = (d.mode == SV3D_TEX_NONE && d.alpha_fast && exact_one < thr) ? (float)exact_one : d.threshold;
    d.alpha_only = p->alpha_only;
    d.c_scale = (float)sqrt(l2e / (double)p->sigma);
    d.visit_r = (float)sqrt((double)d.visit_threshold * l2e / (double)p->sigma);
    d.inv_F = p->num_faces == 1 ? 0xffffffffu : (unsigned)(0x100000000ull / (unsigned long long)p->num_faces);
    return SV3D_OK;
}

inline size_t rec_floats(const DevParams &d) { return (size_t)d.B * d.F * kRec; }
inline size_t group_boxes(const DevParams &d) { return (size_t)d.B * face_groups(d.F); }
inline int saved_per_pixel(const DevParams &d)
{
    return d.mode == SV3D_TEX_NONE ? 1 : (d.mode == SV3D_TEX_VERTEX8 ? 12 : 6);
}
inline size_t backward_ws_need(const DevParams &)
{
    // the backward folds the incoming gradient with the saved aggregates on the fly
    return 0;
}

// Coarse bins are built for meshes with many faces (ids must fit 16 bits); their storage follows the
// pixel boxes in the face_rec buffer: counts [B][nbins] int32, then lists [B][nbins][cap] uint16.
#ifndef SV3D_COARSE_MIN_FACES
#define SV3D_COARSE_MIN_FACES 2048
#endif
inline bool coarse_enabled(const DevParams &d) { return d.F >= SV3D_COARSE_MIN_FACES && d.F <= 65535 && d.S > kBinPix; }
inline int coarse_bins_x(const DevParams &d) { return (d.S + kBinPix - 1) / kBinPix; }
inline int coarse_cap(const DevParams &d) { return (d.F + 7) & ~7; }
inline size_t coarse_bytes(const DevParams &d)
{
    if (!coarse_enabled(d)) return 0;
    const size_t nb = (size_t)d.B * coarse_bins_x(d) * coarse_bins_x(d);
    return nb * sizeof(int) + nb * coarse_cap(d) * sizeof(unsigned short);
}

// Per-tile lists: large silhouette launches whose tile, scaled by sqrt(log2(e)/sigma), is small enough
// for the tile-relative evaluation to stay within the rounding the absolute one has.
inline long launch_tiles(const DevParams &d)
{
    const long tiles = (d.S + kTile - 1) / kTile;
    return tiles * tiles * d.B;
}
// small launches: setup and lists in one kernel, two waves per block in the forward
inline bool tiled_fused(const DevParams &d) { return launch_tiles(d) <= SV3D_TILED_FUSED_MAX_TILES; }
// Lists hold 128 faces; a tile with more goes through all of the image's faces instead, so a mesh
// with many faces per tile (a coarse raster) keeps the binning forward.
inline bool lists_fit(const DevParams &d)
{
    const long tiles = (d.S + kTile - 1) / kTile;
    return SV3D_TILED && d.F <= 6000 && (long)d.F <= SV3D_TILED_MAX_FACES_PER_TILE * tiles * tiles;
}
// the silhouette forward over tile-relative records
inline bool tiled_enabled(const DevParams &d)
{
    return lists_fit(d) && d.mode == SV3D_TEX_NONE && d.alpha_fast &&
           (launch_tiles(d) > SV3D_TILED_MIN_TILES || tiled_fused(d)) &&
           d.c_scale * (float)kTile * d.inv_S <= 64.f;
}
// Every other large launch (the colour modes: their depth softmax needs the barycentrics the
// corner-relative records give) takes the lists in place of its tiles' own binning, nothing else.
inline bool binning_lists_enabled(const DevParams &d)
{
    return lists_fit(d) && !tiled_enabled(d) && launch_tiles(d) > SV3D_BINNING_LISTS_MIN_TILES;
}
// Which form of the tiled forward a launch takes (all measured, tools/kbench.py):
//   two_blocks: the largest launches; two waves per tile, each two 8x8 blocks.
//   shared:     the tile's four waves share all its hits (raster_forward_tiled_shared_kernel): from 1536
//               tiles up to 12287 for meshes of four faces per tile or more (8 .. 40 images of 1280
//               faces at 256^2: -6 .. -13 % against the next two forms), 6144 for two, 3072 for one
//               (2 / 3 images of 1280 faces at 512^2: -8 / -4 %; 6: none; 8: +3 %).
//   pairs:      below 1536 tiles, meshes of two faces per tile or more: two waves per block taking its
//               hits in turn (a mesh of fewer leaves a block's second wave next to nothing to share).
//   blocks:     everything else: four waves per tile, one block each.
enum class TiledForm { two_blocks, shared, pairs, blocks };
inline TiledForm tiled_form(const DevParams &d)
{
    const long tiles = (d.S + kTile - 1) / kTile, t2 = tiles * tiles, n = launch_tiles(d);
    if (n >= SV3D_TILED_TWO_BLOCKS_MIN_TILES) return TiledForm::two_blocks;
    const long shared_max = d.F >= 4 * t2 ? SV3D_TILED_SHARED_MAX_TILES : d.F >= 2 * t2 ? 6144 : d.F >= t2 ? 3072 : 0;
    if (n >= SV3D_TILED_SHARED_MIN_TILES && n <= shared_max) return TiledForm::shared;
    if (n < SV3D_TILED_SHARED_MIN_TILES && d.F >= 2 * t2) return TiledForm::pairs;
    return TiledForm::blocks;
}
inline size_t tiled_bytes(const DevParams &d)
{
    const bool tiled = tiled_enabled(d);
    if (!tiled && !binning_lists_enabled(d)) return 0;
    const size_t tiles = (size_t)((d.S + kTile - 1) / kTile);
    const size_t nt = (size_t)d.B * tiles * tiles;
    return nt * sizeof(int) + 16 + (tiled ? (size_t)d.B * d.F * 3 * sizeof(float4) : 0) +
           nt * kTileCap * sizeof(unsigned short);
}

// Which kernels a forward call runs.
struct ForwardRoute {
    bool fused;     // setup and lists in one launch (small silhouette launches from face_vertices)
    bool tiled;     // the silhouette forward over per-tile lists and tile-relative records
    bool blists;    // the general forward with per-tile lists in place of its binning
    bool coarse;    // the general forward over coarse bins
};
// (small launches take the lists from face_vertices only: a list builder would have to project the
// mesh's vertices once more per bin, and such calls come through the operators, bound by the host)
inline ForwardRoute forward_route(const DevParams &d, bool from_mesh)
{
    ForwardRoute r;
    r.fused = tiled_enabled(d) && tiled_fused(d) && !from_mesh;
    r.tiled = tiled_enabled(d) && (r.fused || launch_tiles(d) > SV3D_TILED_MIN_TILES);
    r.blists = binning_lists_enabled(d);
    r.coarse = coarse_enabled(d) && !r.tiled && !r.blists;
    return r;
}

inline int launch_status()
{
    return hipGetLastError() == hipSuccess ? SV3D_OK : SV3D_ERR_LAUNCH;
}

}  // namespace

extern "C" int sv3d_raster_sizes(const sv3d_raster_params *p, size_t *face_rec_bytes,
                                 size_t *pixel_saved_bytes, size_t *backward_ws_bytes)
{
    DevParams d;
    const int st = make_dev_params(p, d);
    if (st != SV3D_OK) return st;
    const size_t pix = (size_t)d.B * d.S * d.S * saved_per_pixel(d) * sizeof(float);
    if (face_rec_bytes)
        *face_rec_bytes = rec_floats(d) * sizeof(float) + ((size_t)d.B * d.F + group_boxes(d)) * sizeof(uint2) +
                          coarse_bytes(d) + tiled_bytes(d);
    if (pixel_saved_bytes) *pixel_saved_bytes = pix;
    if (backward_ws_bytes) *backward_ws_bytes = backward_ws_need(d);
    return SV3D_OK;
}

namespace {

// Either face_vertices (mesh == nullptr) or the object-space mesh is the geometry source; with a
// mesh, face_vertices is an optional output.
int raster_forward_impl(const sv3d_raster_params *p, const float *face_vertices, const float *textures,
                        float *rgba_out, void *face_rec, void *pixel_saved, void *stream,
                        hipEvent_t ev_start, hipEvent_t ev_stop, const sv3d_mesh_input *mesh = nullptr,
                        int repeats = 1)
{
    DevParams d;
    const int st = make_dev_params(p, d);
    if (st != SV3D_OK) return st;
    if (mesh) {
        if (!mesh->vertices || !mesh->cams || !mesh->faces || mesh->num_vertices <= 0)
            return SV3D_ERR_INVALID_ARG;
        if (mesh->faces_batch != 1 && mesh->faces_batch != d.B) return SV3D_ERR_INVALID_ARG;
    } else if (!face_vertices) {
        return SV3D_ERR_INVALID_ARG;
    }
    if (!rgba_out || !face_rec || !pixel_saved) return SV3D_ERR_INVALID_ARG;
    if (d.mode != SV3D_TEX_NONE && !textures) return SV3D_ERR_INVALID_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    float4 *rec = static_cast<float4 *>(face_rec);
    uint2 *pbox = reinterpret_cast<uint2 *>(static_cast<float *>(face_rec) + rec_floats(d));
    const int nf = d.B * d.F;
    uint2 *gbox = pbox + nf;                        // [B][face_groups(F)], then the coarse bins
    const dim3 sgrid((d.F + kThreads - 1) / kThreads, d.B);
    const int tiles = (d.S + kTile - 1) / kTile;
    // per-tile lists: counts [B][tiles][tiles], then (16-byte aligned) the faces' oriented silhouette
    // records, then the tiles' lists of face ids
    const ForwardRoute route = forward_route(d, mesh != nullptr);
    const bool fused = route.fused, tiled = route.tiled, blists = route.blists, coarse = route.coarse;
    // two waves per block while the grid is small enough that single waves set the kernel's span
    // (measured: silhouettes of 12 and 16 images of 256^2 are faster unsplit; the colour modes, with
    // their longer visits, still gain at 12 images -- forward -13 % -- and lose at 16)
    const bool split = (long)tiles * tiles * d.B <= (d.mode == SV3D_TEX_NONE ? SV3D_SPLIT_MAX_TILES : SV3D_SPLIT_MAX_TILES_COLOUR);
    // group boxes are read by the large launches without coarse bins or per-tile lists only
    uint2 *gbox_setup = (!coarse && !tiled && !(SV3D_DIRECT_BOXES && split)) ? gbox : nullptr;
    int *tcount = nullptr;
    float4 *srec = nullptr;
    unsigned short *tlist = nullptr;
    if (tiled || blists) {
        const size_t nt = (size_t)d.B * tiles * tiles;
        char *tbase = reinterpret_cast<char *>(gbox + group_boxes(d)) + coarse_bytes(d);
        tcount = reinterpret_cast<int *>(tbase);
        float4 *after = reinterpret_cast<float4 *>((reinterpret_cast<uintptr_t>(tbase + nt * sizeof(int)) + 15) & ~(uintptr_t)15);
        if (tiled) srec = after;
        tlist = reinterpret_cast<unsigned short *>(after + (tiled ? (size_t)d.B * d.F * 3 : 0));
    }
    MeshArgs m = {};
    if (mesh) {
        m.vertices = mesh->vertices; m.cams = mesh->cams; m.faces = mesh->faces;
        m.V = mesh->num_vertices; m.faces_batch = mesh->faces_batch; m.offset_z = mesh->offset_z;
        m.a = {mesh->scale[0], mesh->scale[1], mesh->scale[2], mesh->shift[0], mesh->shift[1], mesh->shift[2]};
    }
    if (fused) {
        const int bx = coarse_bins_x(d), nbins = bx * bx, setup_wgs = (d.F + kThreads - 1) / kThreads;
        const dim3 fgrid((unsigned)(setup_wgs + nbins), (unsigned)d.B);
        hipLaunchKernelGGL(raster_setup_lists_kernel, fgrid, dim3(kFusedThreads),
                           lists_lds_bytes(d.F) + kStageFloats * sizeof(float), s, d, face_vertices, rec, pbox, srec,
                           tcount, tlist, bx, tiles, setup_wgs);
    } else if (mesh) {
        hipLaunchKernelGGL(raster_setup_mesh_kernel, sgrid, dim3(kThreads), 0, s, d, m,
                           const_cast<float *>(face_vertices), rec, pbox, gbox_setup, srec);
    } else {
        hipLaunchKernelGGL(raster_setup_kernel, sgrid, dim3(kThreads), 0, s, d, face_vertices, rec, pbox, gbox_setup, srec);
    }
    float *saved = static_cast<float *>(pixel_saved);
    CoarseBins cb = {nullptr, nullptr, 0, 0, 0, nullptr, nullptr};
    if (coarse) {
        const int bx = coarse_bins_x(d), nbins = bx * bx, cap = coarse_cap(d);
        int *count = reinterpret_cast<int *>(gbox + group_boxes(d));
        unsigned short *list = reinterpret_cast<unsigned short *>(count + (size_t)d.B * nbins);
        hipLaunchKernelGGL(raster_coarse_kernel, dim3((unsigned)(d.B * nbins)), dim3(kThreads), 0, s, d, pbox,
                           list, count, bx, nbins, cap);
        cb = {list, count, bx, nbins, cap, nullptr, nullptr};
    }
    if (blists) {
        const int bx = coarse_bins_x(d);
        hipLaunchKernelGGL(raster_tile_lists_kernel, dim3((unsigned)(bx * bx), (unsigned)d.B), dim3(kListThreads),
                           lists_lds_bytes(d.F), s, d, pbox, tcount, tlist, bx, tiles);
        cb.tile_count = tcount;
        cb.tile_list = tlist;
    }
    if (tiled) {
        const int bx = coarse_bins_x(d);
        if (!fused)
            hipLaunchKernelGGL(raster_tile_lists_kernel, dim3((unsigned)(bx * bx), (unsigned)d.B), dim3(kListThreads),
                               lists_lds_bytes(d.F), s, d, pbox, tcount, tlist, bx, tiles);
        if (ev_start) (void)hipEventRecord(ev_start, s);
        const dim3 tgrid((unsigned)d.B, (unsigned)tiles, (unsigned)tiles);
        for (int rep = 0; rep < repeats; ++rep) {
            const TiledForm form = tiled_form(d);
            if (form == TiledForm::two_blocks)
                hipLaunchKernelGGL((raster_forward_tiled_kernel<2, 1>), tgrid, dim3(kThreads / 2), 0, s, tcount, tlist,
                                   srec, pbox, d.F, rgba_out, saved, d);
            else if (form == TiledForm::shared)
                hipLaunchKernelGGL(raster_forward_tiled_shared_kernel, tgrid, dim3(kThreads), 0, s, tcount, tlist,
                                   srec, pbox, d.F, rgba_out, saved, d);
            else if (form == TiledForm::pairs)
                hipLaunchKernelGGL((raster_forward_tiled_kernel<1, 2>), tgrid, dim3(kThreads * 2), 0, s, tcount, tlist,
                                   srec, pbox, d.F, rgba_out, saved, d);
            else
                hipLaunchKernelGGL((raster_forward_tiled_kernel<1, 1>), tgrid, dim3(kThreads), 0, s, tcount, tlist,
                                   srec, pbox, d.F, rgba_out, saved, d);
        }
        if (ev_stop) (void)hipEventRecord(ev_stop, s);
        return launch_status();
    }
    if (ev_start) (void)hipEventRecord(ev_start, s);
    // the remainder-carrying variants when S is not a power of two
    const bool lo = (d.S & (d.S - 1)) != 0;
    const dim3 grid3((unsigned)d.B, (unsigned)tiles, (unsigned)tiles);  // tiles <= 256, batch <= 65535
#define SV3D_LAUNCH_FWD2(M, SP, LO_, CO_)                                                              \
    hipLaunchKernelGGL((raster_forward_kernel<M, SP, LO_, CO_>), grid3, dim3(SP * kThreads), 0, s, pbox,   \
                       gbox, d.F, rec, textures, rgba_out, saved, d, cb)
#define SV3D_LAUNCH_FWD(M)                                                                             \
    do {                                                                                               \
        if (coarse && lo) SV3D_LAUNCH_FWD2(M, 1, true, true);                                          \
        else if (coarse) SV3D_LAUNCH_FWD2(M, 1, false, true);                                          \
        else if (split && lo) SV3D_LAUNCH_FWD2(M, 2, true, false);                                     \
        else if (split) SV3D_LAUNCH_FWD2(M, 2, false, false);                                          \
        else if (lo) SV3D_LAUNCH_FWD2(M, 1, true, false);                                              \
        else SV3D_LAUNCH_FWD2(M, 1, false, false);                                                     \
    } while (0)
    for (int rep = 0; rep < repeats; ++rep) {       // > 1: measurement only, every launch writes the same values
    if (d.mode == SV3D_TEX_NONE) SV3D_LAUNCH_FWD(SV3D_TEX_NONE);
    else if (d.mode == SV3D_TEX_SURFACE) SV3D_LAUNCH_FWD(SV3D_TEX_SURFACE);
    else if (d.mode == SV3D_TEX_VERTEX) SV3D_LAUNCH_FWD(SV3D_TEX_VERTEX);
    else SV3D_LAUNCH_FWD(SV3D_TEX_VERTEX8);
    }
#undef SV3D_LAUNCH_FWD2
#undef SV3D_LAUNCH_FWD
    if (ev_stop) (void)hipEventRecord(ev_stop, s);
    return launch_status();
}

int raster_backward_impl(const sv3d_raster_params *p, const float *textures, const void *face_rec,
                         const void *pixel_saved, const float *grad_rgba, float *grad_face_vertices,
                         float *grad_textures, void *workspace, size_t workspace_bytes, void *stream,
                         hipEvent_t ev_start, hipEvent_t ev_stop, int repeats = 1)
{
    DevParams d;
    const int st = make_dev_params(p, d);
    if (st != SV3D_OK) return st;
    if (!face_rec || !pixel_saved || !grad_rgba || !grad_face_vertices) return SV3D_ERR_INVALID_ARG;
    if (d.mode != SV3D_TEX_NONE && !textures) return SV3D_ERR_INVALID_ARG;
    (void)workspace; (void)workspace_bytes;      // no mode needs scratch (sv3d_raster_sizes reports 0)
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float4 *rec = static_cast<const float4 *>(face_rec);
    const uint2 *pbox = reinterpret_cast<const uint2 *>(static_cast<const float *>(face_rec) + rec_floats(d));
    const float *saved = static_cast<const float *>(pixel_saved);
    // two waves per face while the launch is small: below two residency rounds of whole-face waves
    const bool halves = (long)d.B * d.F <= kBackwardHalvesMaxFaces;
    const dim3 fgrid(halves ? (d.B * d.F + 1) / 2 : (d.B * d.F + 3) / 4), block(kThreads);
    if (ev_start) (void)hipEventRecord(ev_start, s);
    const bool lo = (d.S & (d.S - 1)) != 0;         // raster size not a power of two: see struct Pixel
    const size_t lds = d.mode == SV3D_TEX_SURFACE ? (size_t)4 * tex_lds_floats(d.R * d.R) * sizeof(float) : 0;
    float *gt = d.mode == SV3D_TEX_NONE ? nullptr : grad_textures;
    // Interval enumeration once a face covers enough pixels for it to pay: the expected box is a
    // few times S^2 / F pixels (measured with 1280 faces: 320^2 -8 %, 384^2 -6..12 %, 512^2
    // -13..28 %, but 256^2 +2..3 %; 5120 faces: 1024^2 -15 %, 512^2 +7 %; 320 faces at 256^2 -11 %)
    // The colour modes' steps are three times as long as the silhouette's, so the enumeration's own
    // cost weighs a third as much: they gain from 24 pixels per face on (5120 faces at 512^2: surface
    // -7 %, vertex colours -13 %, eight channels -15 %; 1280 faces at 256^2, 64 images: -5..7 %; 5120
    // faces at 384^2: -2.5 %; 8 images at 256^2: no change).
    const long pixels_per_face_min = d.mode == SV3D_TEX_NONE ? 64L : 24L;
    const bool spans = p->backward_enumeration == SV3D_BWD_AUTO ? (long)d.S * d.S >= pixels_per_face_min * d.F
                                                                : p->backward_enumeration == SV3D_BWD_INTERVALS;
#define SV3D_BWD_CASE(M, LO_, H_, SP_)                                                                  \
    hipLaunchKernelGGL((raster_backward_kernel<M, LO_, H_, SP_>), fgrid, block, lds, s, rec, d.B * d.F,  \
                       d.B, saved, grad_rgba, d.S, d.aa, d.alpha_only, d.F, d.inv_F, d.inv_S, textures, pbox,  \
                       grad_face_vertices, gt, d)
#define SV3D_BWD_SPANS(M, SP_)                                                                          \
    do {                                                                                                \
        if (lo && halves) SV3D_BWD_CASE(M, true, 2, SP_);                                               \
        else if (lo) SV3D_BWD_CASE(M, true, 1, SP_);                                                    \
        else if (halves) SV3D_BWD_CASE(M, false, 2, SP_);                                               \
        else SV3D_BWD_CASE(M, false, 1, SP_);                                                           \
    } while (0)
#define SV3D_LAUNCH_BWD(M)                                                                              \
    do {                                                                                                \
        if (spans) SV3D_BWD_SPANS(M, true);                                                             \
        else SV3D_BWD_SPANS(M, false);                                                                  \
    } while (0)
    for (int rep = 0; rep < repeats; ++rep) {       // > 1: measurement only, every launch writes the same values
    if (d.mode == SV3D_TEX_NONE) SV3D_LAUNCH_BWD(SV3D_TEX_NONE);
    else if (d.mode == SV3D_TEX_SURFACE) SV3D_LAUNCH_BWD(SV3D_TEX_SURFACE);
    else if (d.mode == SV3D_TEX_VERTEX) SV3D_LAUNCH_BWD(SV3D_TEX_VERTEX);
    else SV3D_LAUNCH_BWD(SV3D_TEX_VERTEX8);
    }
#undef SV3D_LAUNCH_BWD
#undef SV3D_BWD_SPANS
#undef SV3D_BWD_CASE
    if (ev_stop) (void)hipEventRecord(ev_stop, s);
    return launch_status();
}

}  // namespace

extern "C" int sv3d_raster_forward(const sv3d_raster_params *p, const float *face_vertices,
                                   const float *textures, float *rgba_out, void *face_rec,
                                   void *pixel_saved, void *stream)
{
    return raster_forward_impl(p, face_vertices, textures, rgba_out, face_rec, pixel_saved, stream,
                               nullptr, nullptr);
}

extern "C" int sv3d_render_forward(const sv3d_raster_params *p, const sv3d_mesh_input *mesh,
                                   const float *textures, float *face_vertices_out, float *rgba_out,
                                   void *face_rec, void *pixel_saved, void *stream)
{
    if (!mesh) return SV3D_ERR_INVALID_ARG;
    return raster_forward_impl(p, face_vertices_out, textures, rgba_out, face_rec, pixel_saved, stream,
                               nullptr, nullptr, mesh);
}

extern "C" int sv3d_raster_forward_timed(const sv3d_raster_params *p, const float *face_vertices,
                                         const float *textures, float *rgba_out, void *face_rec,
                                         void *pixel_saved, void *stream, void *ev_start, void *ev_stop)
{
    return raster_forward_impl(p, face_vertices, textures, rgba_out, face_rec, pixel_saved, stream,
                               static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop));
}

extern "C" int sv3d_raster_backward(const sv3d_raster_params *p, const float *textures,
                                    const void *face_rec, const void *pixel_saved,
                                    const float *grad_rgba, float *grad_face_vertices,
                                    float *grad_textures, void *workspace, size_t workspace_bytes,
                                    void *stream)
{
    return raster_backward_impl(p, textures, face_rec, pixel_saved, grad_rgba, grad_face_vertices,
                                grad_textures, workspace, workspace_bytes, stream, nullptr, nullptr);
}

extern "C" int sv3d_raster_backward_timed(const sv3d_raster_params *p, const float *textures,
                                          const void *face_rec, const void *pixel_saved,
                                          const float *grad_rgba, float *grad_face_vertices,
                                          float *grad_textures, void *workspace,
                                          size_t workspace_bytes, void *stream, void *ev_start,
                                          void *ev_stop)
{
    return raster_backward_impl(p, textures, face_rec, pixel_saved, grad_rgba, grad_face_vertices,
                                grad_textures, workspace, workspace_bytes, stream,
                                static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop));
}

extern "C" const char *sv3d_raster_forward_kernel_name(const sv3d_raster_params *p, int32_t from_mesh)
{
    DevParams d;
    if (make_dev_params(p, d) != SV3D_OK) return "";
    if (!forward_route(d, from_mesh != 0).tiled) return "raster_forward_kernel";
    return tiled_form(d) == TiledForm::shared ? "raster_forward_tiled_shared_kernel" : "raster_forward_tiled_kernel";
}

extern "C" int sv3d_raster_forward_timed_n(const sv3d_raster_params *p, const float *face_vertices,
                                           const float *textures, float *rgba_out, void *face_rec,
                                           void *pixel_saved, void *stream, void *ev_start, void *ev_stop,
                                           int32_t repeats)
{
    if (repeats < 1 || repeats > 1000) return SV3D_ERR_INVALID_ARG;
    return raster_forward_impl(p, face_vertices, textures, rgba_out, face_rec, pixel_saved, stream,
                               static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop), nullptr, repeats);
}

extern "C" int sv3d_raster_backward_timed_n(const sv3d_raster_params *p, const float *textures,
                                            const void *face_rec, const void *pixel_saved,
                                            const float *grad_rgba, float *grad_face_vertices,
                                            float *grad_textures, void *workspace,
                                            size_t workspace_bytes, void *stream, void *ev_start,
                                            void *ev_stop, int32_t repeats)
{
    if (repeats < 1 || repeats > 1000) return SV3D_ERR_INVALID_ARG;
    return raster_backward_impl(p, textures, face_rec, pixel_saved, grad_rgba, grad_face_vertices,
                                grad_textures, workspace, workspace_bytes, stream,
                                static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop), repeats);
}
