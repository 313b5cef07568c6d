/* sv3d_hip.h -- C ABI of libsv3d_hip.so, the MI355X (gfx950) hot path of the
 * CMR-style single-view reconstruction pipeline: soft rasterization forward /
 * backward and the brute-force Chamfer nearest-neighbour search.
 *
 * The reference (pure Python) reaches this arithmetic through two third-party
 * CUDA extensions.  Each entry point below names the reference call site whose
 * native dependency it replaces.
 *
 * Conventions
 *   - All pointers are DEVICE pointers to contiguous row-major fp32 / int32
 *     buffers owned by the caller and borrowed for the call.
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work
 *     on it (no allocation, no host synchronisation), so calls are graph-capture
 *     safe and re-entrant: the library keeps no mutable state between calls (the one
 *     exception is named at sv3d_distance_transform_static).  The current HIP device
 *     must own the buffers.
 *   - Return value: SV3D_OK or a negative sv3d_status; sv3d_strerror() names it.
 *   - Results are bitwise repeatable for fixed inputs: no float atomics to global
 *     memory anywhere, and every sum but one runs in an order the code fixes (register
 *     trees, ordered LDS passes).  The exception: grad_textures of SV3D_TEX_SURFACE.
 *     There one wave per face adds its lanes' texel gradients into wave-private LDS
 *     tables with ds_add_f32; lanes of one instruction that hit the same address are
 *     serialised by the LDS hardware, in an order the ISA does not spell out.  On
 *     gfx950 that order is the same from run to run (tested: bit-equal repeats), so the
 *     result is repeatable on this target; it is an observed property of the chip,
 *     not a guarantee of the instruction set.
 */
#ifndef SV3D_HIP_H
#define SV3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sv3d_status {
    SV3D_OK = 0,
    SV3D_ERR_INVALID_ARG = -1,   /* null pointer, non-positive size, bad enum */
    SV3D_ERR_UNSUPPORTED = -2,   /* size outside what the kernels are built for */
    SV3D_ERR_WORKSPACE = -3,     /* caller's buffer smaller than sv3d_raster_sizes() says */
    SV3D_ERR_LAUNCH = -4,        /* hipGetLastError() != hipSuccess after a launch */
    SV3D_ERR_NOT_READY = -5      /* a *_static call's first use on this GPU came under stream capture */
} sv3d_status;

/* SV3D_TEX_VERTEX8: eight scalar per-corner channels rendered in one pass (the reference renders
 * its part maps one at a time, each copied into three colour channels of which one is kept,
 * main.py:422-426; channel c of this mode equals channel 0 of such a render of map c).
 * textures [B,F,3,8], rgba_out / grad_rgba [B,9,S/aa,S/aa] (eight channels, then alpha), every
 * channel's background is background[0].                                                   */
enum { SV3D_TEX_NONE = 0, SV3D_TEX_SURFACE = 1, SV3D_TEX_VERTEX = 2, SV3D_TEX_VERTEX8 = 3 };

/* Rasterizer configuration: the arguments sr.SoftRenderer(...) receives at
 * nnutils/softras.py:33-37 plus the SoftRas defaults it leaves untouched. */
typedef struct sv3d_raster_params {
    int32_t batch;         /* B: images in the render batch                                   */
    int32_t num_faces;     /* F                                                               */
    int32_t raster_size;   /* S: resolution the fragments are evaluated at (2*image_size when  */
                           /*    anti-aliasing)                                               */
    int32_t aa;            /* 1 or 2: side of the fused average pool; output is S/aa           */
    int32_t texture_mode;  /* SV3D_TEX_*                                                      */
    int32_t texture_res;   /* R: surface textures hold R*R texels per face                    */
    int32_t double_side;   /* SoftRas fill_back: back faces take part in the colour softmax   */
    int32_t texture_batch; /* TB: texture sets in `textures`; 0 or B = one per image.  Otherwise  */
                           /*     TB divides B and image b reads set b % TB -- the sets of       */
                           /*     `torch.cat([textures] * num_cams)` (main.py:415) without the   */
                           /*     copies.  grad_textures stays per image, [B, ...].              */
    float background[3];
    float near_plane;      /* 1   */
    float far_plane;       /* 100 */
    float eps;             /* 1e-3: initial softmax maximum                                   */
    float sigma;           /* sigma_val */
    float dist_eps;        /* dist_eps as passed by the caller (1e-10), NOT its logit          */
    float gamma;           /* gamma_val */
    int32_t alpha_only;    /* 1 (SV3D_TEX_NONE only): the caller takes the silhouette alone, as     */
                           /*   `self.renderer(vs, fs)[:, -1]` does at nnutils/softras.py:64:        */
                           /*   rgba_out and grad_rgba are [B,1,S/aa,S/aa], the alpha plane; the     */
                           /*   three constant background planes are neither written nor read        */
    int32_t backward_enumeration; /* SV3D_BWD_*: which pixels of a face's box the backward's waves   */
                           /*   walk.  AUTO (0) lets the library pick per launch from raster size    */
                           /*   and face count; BOXES / INTERVALS force one form (tests, tuning).    */
                           /*   Both compute the same sums in a different order (equal to rounding). */
} sv3d_raster_params;

/* SV3D_BWD_BOXES: every pixel of the face's pixel box; SV3D_BWD_INTERVALS: per pixel row only the
 * interval of columns that can hold an active pixel. */
enum { SV3D_BWD_AUTO = 0, SV3D_BWD_BOXES = 1, SV3D_BWD_INTERVALS = 2 };

const char *sv3d_strerror(int status);
int sv3d_abi_version(void);

/* Byte sizes of the buffers the rasterizer calls need:
 *   face_rec_bytes    per-face setup records, pixel boxes, face-group boxes and what the forward
 *                     keeps per tile or bin for the configuration (coarse-bin face lists for
 *                     meshes from 2048 faces; per-tile face lists, and for silhouettes oriented
 *                     edge records, where the forward takes lists): written by forward, the
 *                     per-face part read by backward.  Opaque: its layout may change with the
 *                     parameters and between versions of the library, so forward and backward
 *                     of one call pair must come from one library and one parameter block.
 *   pixel_saved_bytes per-raster-pixel aggregates written by forward, read by backward
 *   backward_ws_bytes scratch for backward; always 0 (the backward folds its inputs on
 *                     the fly), kept so that callers sized by this query stay valid;
 *                     `workspace` may be NULL                                           */
int sv3d_raster_sizes(const sv3d_raster_params *p, size_t *face_rec_bytes,
                      size_t *pixel_saved_bytes, size_t *backward_ws_bytes);

/* Replaces soft_renderer's `soft_rasterize` forward as reached from
 * nnutils/softras.py:64,75,77 (sr.SoftRenderer.__call__ / .rasterizer).
 *   face_vertices [B,F,3,3]  (x, y, z) of the three corners, camera space, y up
 *   textures      NONE: ignored (may be NULL); SURFACE [B,F,R*R,3]; VERTEX [B,F,3,3];
 *                 VERTEX8 [B,F,3,8]
 *   rgba_out      [B,4,S/aa,S/aa]   channels R,G,B,alpha, row 0 = top of the image
 *                 (VERTEX8: [B,9,S/aa,S/aa])
 *   face_rec, pixel_saved: see sv3d_raster_sizes                                   */
int sv3d_raster_forward(const sv3d_raster_params *p, const float *face_vertices,
                        const float *textures, float *rgba_out, void *face_rec,
                        void *pixel_saved, void *stream);

/* Replaces soft_renderer's `soft_rasterize` backward (autograd of the call above).
 *   grad_rgba          [B,4,S/aa,S/aa]
 *   grad_face_vertices [B,F,3,3]  fully overwritten
 *   grad_textures      same shape as textures, fully overwritten (NULL for NONE, or
 *                      when the caller does not need it)                           */
int sv3d_raster_backward(const sv3d_raster_params *p, const float *textures,
                         const void *face_rec, const void *pixel_saved,
                         const float *grad_rgba, float *grad_face_vertices,
                         float *grad_textures, void *workspace, size_t workspace_bytes,
                         void *stream);

/* Measurement variants of the two calls above.  They have no counterpart in the reference
 * (which has no timing harness); they exist for benchmarks that must time one kernel on
 * the stream it runs on.  Same work and results as the plain calls; in addition the two
 * hipEvent_t (passed as void*, either may be NULL) are recorded on `stream` immediately
 * before and after the call's dominant kernel: the per-tile fragment kernel for forward,
 * the per-face gradient kernel for backward. */
int sv3d_raster_forward_timed(const sv3d_raster_params *p, const float *face_vertices,
                              const float *textures, float *rgba_out, void *face_rec,
                              void *pixel_saved, void *stream, void *ev_start, void *ev_stop);
int sv3d_raster_backward_timed(const sv3d_raster_params *p, const float *textures,
                               const void *face_rec, const void *pixel_saved,
                               const float *grad_rgba, float *grad_face_vertices,
                               float *grad_textures, void *workspace, size_t workspace_bytes,
                               void *stream, void *ev_start, void *ev_stop);

/* The same with the dominant kernel launched `repeats` times (1..1000) back to back between the
 * two events: an event pair costs microseconds of stream time, which a kernel of some tens of
 * microseconds cannot carry alone; elapsed / repeats is its average launch duration.  Every launch
 * writes the same values, so the outputs are those of the plain call. */
int sv3d_raster_forward_timed_n(const sv3d_raster_params *p, const float *face_vertices,
                                const float *textures, float *rgba_out, void *face_rec,
                                void *pixel_saved, void *stream, void *ev_start, void *ev_stop,
                                int32_t repeats);
int sv3d_raster_backward_timed_n(const sv3d_raster_params *p, const float *textures,
                                 const void *face_rec, const void *pixel_saved,
                                 const float *grad_rgba, float *grad_face_vertices,
                                 float *grad_textures, void *workspace, size_t workspace_bytes,
                                 void *stream, void *ev_start, void *ev_stop, int32_t repeats);

/* Name of the kernel the events of sv3d_raster_forward_timed(_n) bracket for this configuration
 * ("raster_forward_kernel", "raster_forward_tiled_kernel" or "raster_forward_tiled_shared_kernel";
 * from_mesh != 0: as called through
 * sv3d_render_forward), so that a benchmark can match its timings with a profiler's rows.  ""
 * for parameters the other entry points refuse.  No counterpart in the reference. */
const char *sv3d_raster_forward_kernel_name(const sv3d_raster_params *p, int32_t from_mesh);

/* Replaces chamferdist's forward as called at nnutils/loss_utils.py:85-86
 * (`ChamferDistance()(xyz1, xyz2)`).
 *   xyz1 [B,N,3], xyz2 [B,M,3] -> dist1 [B,N], dist2 [B,M] (SQUARED distances),
 *   idx1 [B,N], idx2 [B,M] (int32 index of the nearest point of the other cloud;
 *   the lowest index wins ties).
 *   d = fmaf(dz,dz,fmaf(dy,dy,dx*dx)) with dx = other - own, each in fp32.         */
int sv3d_chamfer_forward(const float *xyz1, const float *xyz2, int32_t B, int32_t N, int32_t M,
                         float *dist1, float *dist2, int32_t *idx1, int32_t *idx2, void *stream);

/* Replaces chamferdist's backward (autograd of the call above; only dist2 carries a
 * gradient in the reference, nnutils/loss_utils.py:87-88).
 *   grad_dist1 [B,N] / grad_dist2 [B,M]: either may be NULL (treated as zero)
 *   grad_xyz1 [B,N,3] / grad_xyz2 [B,M,3]: either may be NULL (not computed);
 *   the ones given are fully overwritten.                                          */
int sv3d_chamfer_backward(const float *xyz1, const float *xyz2, int32_t B, int32_t N, int32_t M,
                          const float *grad_dist1, const float *grad_dist2,
                          const int32_t *idx1, const int32_t *idx2,
                          float *grad_xyz1, float *grad_xyz2, void *stream);

/* Replaces sr.Mesh's `vertices[faces]` gather (soft_renderer Mesh.face_vertices,
 * reached from nnutils/softras.py:64,72,77) fused with the renderer's affine
 * pre-stage: out[b,f,k,c] = values[b, faces[fb,f,k], c] * scale[c] + shift[c]
 * (the y flip of softras.py:60 and the look_at z shift are a scale / shift).
 *   values [B,V,3]; faces [FB,F,3] int32 with FB == B or FB == 1 (shared topology);
 *   scale, shift: 3 floats each, HOST pointers (read during the call), NULL = identity */
int sv3d_face_gather(const float *values, const int32_t *faces, int32_t B, int32_t V, int32_t F,
                     int32_t faces_batch, const float *scale, const float *shift,
                     float *out, void *stream);

/* Backward of sv3d_face_gather (autograd of the `vertices[faces]` gather):
 * grad_values[b,v,c] = scale[c] * sum of grad_out[b,f,k,c] over the corners (f,k) with
 * faces[fb,f,k] == v, summed in ascending (f,k) order.  Gather form: no atomics, no
 * index to prepare, bitwise repeatable.  grad_values [B,V,3] is fully overwritten.   */
int sv3d_face_scatter(const float *grad_out, const int32_t *faces, int32_t B, int32_t V, int32_t F,
                      int32_t faces_batch, const float *scale, float *grad_values, void *stream);

/* `vertices[faces]`-style gather of C <= 8 scalar channels per vertex and its backward, for the
 * per-vertex part maps of main.py:422-426 rendered with SV3D_TEX_VERTEX8:
 *   out[b,f,k,:] = values[b, faces[fb,f,k], :]         values [B,V,C] -> out [B,F,3,C]
 *   grad_values[b,v,:] = sum of grad_out[b,f,k,:] over corners (f,k) with faces[fb,f,k] == v,
 *   in ascending (f,k) order (no atomics); meshes whose sort tables exceed the LDS
 *   (2V + 3F ints > 150 KB) return SV3D_ERR_UNSUPPORTED.                                   */
int sv3d_face_gather_n(const float *values, const int32_t *faces, int32_t B, int32_t V, int32_t F,
                       int32_t faces_batch, int32_t C, float *out, void *stream);
int sv3d_face_scatter_n(const float *grad_out, const int32_t *faces, int32_t B, int32_t V, int32_t F,
                        int32_t faces_batch, int32_t C, float *grad_values, void *stream);

/* The renderer call's input stage, nnutils/softras.py:58-61: the weak-perspective camera
 * `orthographic_proj_withz` (nnutils/geom_utils.py:63-80, quat_rotate :143-161) followed by
 * a per-axis scale and shift (the y flip of softras.py:60; the look_at transform of
 * softras.py:33-37, which for an eye on the -z axis is a z shift):
 *   P   = cam_scale * (q (0,X) q*) + (tx, ty, offset_z)      q is used as given, un-normalised
 *   out = scale * P + shift                                                              */
typedef struct sv3d_mesh_input {
    const float *vertices;   /* [B,V,3] object space                                      */
    const float *cams;       /* [B,7] cam_scale, tx, ty, qw, qx, qy, qz                   */
    const int32_t *faces;    /* [FB,F,3] int32                                            */
    int32_t num_vertices;    /* V                                                         */
    int32_t faces_batch;     /* FB: B, or 1 for a topology shared by the batch            */
    float offset_z;
    float scale[3];
    float shift[3];
} sv3d_mesh_input;

/* Replaces `proj_fn(vertices, cams, offset_z)` + flip + `vertices[faces]` of
 * nnutils/softras.py:58-61,64 (about twenty elementwise launches in the reference) by one
 * kernel: face_vertices [B,F,3,3] = out[b, faces[fb,f,k], :] without materialising `out`.  */
int sv3d_project_gather(const sv3d_mesh_input *mesh, int32_t B, int32_t F, float *face_vertices,
                        void *stream);

/* Backward of sv3d_project_gather (autograd of the lines cited there):
 *   grad_vertices [B,V,3] and grad_cams [B,7], both fully overwritten.  Sums run in a fixed
 *   order (corner slots ascending per vertex, then a fixed tree over vertices).            */
int sv3d_project_scatter(const sv3d_mesh_input *mesh, int32_t B, int32_t F,
                         const float *grad_face_vertices, float *grad_vertices, float *grad_cams,
                         void *stream);

/* sv3d_project_gather + sv3d_raster_forward with the camera stage folded into the
 * rasterizer's per-face setup kernel (one launch fewer, no round trip of face_vertices):
 * the whole of `SoftNeuralRenderer.forward`, nnutils/softras.py:57-77, for textures that are
 * already per face.  face_vertices_out [B,F,3,3] may be NULL when the caller has no use for
 * it; the other arguments are those of sv3d_raster_forward (p->batch, p->num_faces give B, F).
 * Its backward is sv3d_raster_backward followed by sv3d_project_scatter.                    */
int sv3d_render_forward(const sv3d_raster_params *p, const sv3d_mesh_input *mesh,
                        const float *textures, float *face_vertices_out, float *rgba_out,
                        void *face_rec, void *pixel_saved, void *stream);

/* ---- mesh regularisers of the training step ------------------------------------------------
 * Each call returns the per-mesh loss [B] and, when grad_vertices is not NULL, d loss[b] /
 * d vertices[b] as [B,V,3] in the same launch; the caller scales that by the incoming gradient.
 * Index arrays are shared by the batch; entries outside their range are clamped.             */

/* `sr.LaplacianLoss(vertex, faces)(x)` as built at main.py:202 and called at main.py:515: the
 * uniform graph Laplacian with rows divided by the vertex degree,
 *   loss[b] = sum_v | x[b,v] - mean of x[b,n] over the neighbours n of v |^2.
 * The neighbour lists are CSR: nbr[row_start[v] .. row_start[v+1]), without v itself; the
 * relation must be symmetric (it is for the edges of a face list).  The reference holds the
 * operator as a dense V x V matrix.  SV3D_ERR_UNSUPPORTED when 16 V bytes exceed 150 KB of LDS. */
int sv3d_laplacian_loss(const float *vertices, const int32_t *row_start, const int32_t *nbr, int32_t B,
                        int32_t V, float *loss, float *grad_vertices, void *stream);

/* `sr.FlattenLoss(faces)(x)` (main.py:203,516): over E edges, quads[e] = (v0, v1, v2, v3) with
 * (v0,v1) the edge and v2, v3 the opposite vertices of its two faces (which edges of a mesh are
 * listed is the caller's choice; upstream lists those found at corner positions (0,1), (1,2)),
 *   loss[b] = sum_e (cos_e + 1)^2,  cos_e between the two in-face altitudes onto the edge,
 * with the eps placement of the upstream module (eps = 1e-6 there).  inc[inc_start[v] ..
 * inc_start[v+1]) lists, ascending, the entries 4 e + k for which quads[e][k] == v; it is only
 * read when grad_vertices is not NULL.                                                        */
int sv3d_flatten_loss(const float *vertices, const int32_t *quads, const int32_t *inc_start,
                      const int32_t *inc, int32_t B, int32_t V, int32_t E, float eps, float *loss,
                      float *grad_vertices, void *stream);

/* y[b,v,:] = sum_k values[vb,k] * x[b, col[k], :] over the CSR row v (k ascending): the product
 * of `Laplacian.forward` / `.backward`, nnutils/laplacian.py:63-103 (scipy on the host there,
 * with a device -> host -> device copy per call).  x, y [B,V,3]; values [values_batch, nnz] with
 * values_batch = B, or 1 for weights shared by the batch.                                     */
int sv3d_mesh_spmv(const float *values, const int32_t *row_start, const int32_t *col, const float *x,
                   int32_t B, int32_t V, int32_t nnz, int32_t values_batch, float *y, void *stream);

/* ---- texture path either side of the rasterizer -------------------------------------------- */

/* `geom_utils.sample_textures(texture_flow, images)`, nnutils/geom_utils.py:11-26, and the same
 * sampling of a distance map in `texture_dt_loss_per_instance`, nnutils/loss_utils.py:242-256:
 * `grid_sample(images, flow, align_corners=True)` (bilinear, zero padding) followed by the permute
 * to channels-last, written directly as the rasterizer's texel layout:
 *   images [B,C,H,W], C <= 4;  flow [B,N,2] in [-1,1] (x, y), N = F*T*T;  out [B,N,C].           */
int sv3d_texture_sample(const float *images, const float *flow, int32_t B, int32_t C, int32_t H,
                        int32_t W, int32_t N, float *out, void *stream);

/* Its backward to the flow: grad_out [B,N,C] -> grad_flow [B,N,2] (fully overwritten).  The
 * gradient to the images is not provided: the reference samples dataset images and distance
 * maps, which carry no gradient (main.py:241-265).                                              */
int sv3d_texture_sample_backward(const float *images, const float *flow, const float *grad_out,
                                 int32_t B, int32_t C, int32_t H, int32_t W, int32_t N,
                                 float *grad_flow, void *stream);

/* `compute_dt_barrier(mask, k)` (barrier != 0) and `compute_dt(mask)` (barrier == 0),
 * utils/image.py:94-113, for a batch of masks [B,H,W] on the device (scipy's exact Euclidean
 * distance transform on the host per sample there, main.py:259-260):
 *   dist_out = EDT(1 - mask), dist_in = EDT(mask)      (distance of a non-zero pixel to the
 *                                                        nearest zero pixel, 0 at zero pixels)
 *   barrier: 1 / (1 + exp(-k (dist_out - dist_in) / max(H,W)));   plain: dist_out / max(H,W).
 * A field without any zero pixel gets scipy's values for that case (distances to row -1,
 * column 0).  H, W <= 4096.  `workspace` holds the per-row distances between the two launches:
 * sv3d_distance_transform_workspace() bytes (4 per pixel), caller-owned like every buffer.      */
int sv3d_distance_transform_workspace(int32_t B, int32_t H, int32_t W, size_t *bytes);
int sv3d_distance_transform(const float *masks, int32_t B, int32_t H, int32_t W, int32_t barrier,
                            float k, float *out, void *workspace, size_t workspace_bytes, void *stream);

/* sv3d_distance_transform for callers that cannot allocate device memory: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the
 * reference.  Same arguments, launches and results as sv3d_distance_transform without the
 * workspace pair: the per-row distances live in an array of SV3D_DT_STATIC_MAX_PIXELS words
 * (16 MiB) inside the library's device image, one copy per GPU, held until the process ends.
 *   B*H*W > SV3D_DT_STATIC_MAX_PIXELS -> SV3D_ERR_WORKSPACE (nothing is launched).
 *   The call allocates and frees nothing.  The first call on a GPU looks up the address of that
 *   GPU's copy of the array (which may load the library's device code there): make it outside
 *   stream capture -- any size will do; a first call on a capturing stream returns
 *   SV3D_ERR_NOT_READY and enqueues nothing.  Every call on one GPU uses that GPU's array: calls
 *   whose execution can overlap on one GPU (different streams, no ordering between them) race;
 *   calls ordered on one stream, or on different GPUs, do not.                                 */
#define SV3D_DT_STATIC_MAX_PIXELS (4 * 1024 * 1024)   /* e.g. 64 masks of 256 x 256 */
int sv3d_distance_transform_static(const float *masks, int32_t B, int32_t H, int32_t W,
                                   int32_t barrier, float k, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SV3D_HIP_H */
