// PyTorch operators `sv3d::*` over the C ABI of libsv3d_hip.so (include/sv3d_hip.h), registered
// from C++ so that a call costs a dispatcher hop instead of a Python-level op definition:
//
//   soft_rasterize / soft_rasterize_backward   soft_renderer's `soft_rasterize`, reached from
//                                              nnutils/softras.py:64,75,77
//   render / render_backward                   the whole renderer call of nnutils/softras.py:57-77
//                                              from the object-space mesh
//   chamfer_nn / chamfer_nn_backward           ChamferDistance as called at nnutils/loss_utils.py:85-86
//   face_gather / face_scatter                 sr.Mesh's `vertices[faces]` + affine pre-stage
//   project_gather / project_scatter           camera + flip + gather, nnutils/softras.py:58-61
//   face_gather_n / face_scatter_n             per-vertex part maps (main.py:422-426) per face corner
//   laplacian_loss / flatten_loss              sr.LaplacianLoss / sr.FlattenLoss (main.py:202-203,513,516):
//                                              loss and vertex gradient in one launch
//   mesh_spmv                                  CSR product of nnutils/laplacian.py:63-103
//   texture_sample / texture_sample_backward   geom_utils.sample_textures (nnutils/geom_utils.py:11-26)
//   distance_transform                         utils/image.py:94-113 on a batch of masks
//
// Every operator has a GPU kernel (key CUDA, which is HIP in a ROCm build of PyTorch), a Meta
// kernel (shapes only: fake-tensor tracing) and, for the differentiable ones, an Autograd kernel.
// There is no CPU kernel: CPU tensors fail in the dispatcher.  Inputs of other floating types
// (autocast regions hand over fp16) are computed in fp32 and gradients are returned in the
// inputs' types.  torch is used for device memory, streams and autograd plumbing only.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/autograd.h>
#include <torch/library.h>

#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "sv3d_hip.h"

namespace {

using at::Tensor;
using c10::optional;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;
using IntList = at::IntArrayRef;
using FloatList = at::ArrayRef<double>;

// ---- small helpers ---------------------------------------------------------------------------
void check_status(int st, const char *what)
{
    TORCH_CHECK(st == SV3D_OK, what, " failed: ", sv3d_strerror(st), " (status ", st, ")");
}

// fp32 contiguous form of a tensor: the tensor itself when it already is that (these kernels run
// below autograd, nothing they do is recorded), a converted copy otherwise
Tensor f32c(const Tensor &t)
{
    if (t.scalar_type() == at::kFloat && t.is_contiguous()) return t;
    return t.to(at::kFloat).contiguous();
}

struct Launch {
    c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard;
    void *stream;
    explicit Launch(const Tensor &t) : guard(t.device())
    {
        stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream();
    }
};

const float *fptr(const optional<Tensor> &t) { return t.has_value() && t->defined() && t->numel() > 0 ? t->data_ptr<float>() : nullptr; }
const float *fptr(const Tensor &t) { return t.defined() && t.numel() > 0 ? t.data_ptr<float>() : nullptr; }
float *fptr_mut(Tensor &t) { return t.defined() && t.numel() > 0 ? t.data_ptr<float>() : nullptr; }

void same_device(const Tensor &ref, const Tensor &t, const char *name)
{
    // a tensor on the host or on another GPU would reach the kernels as a foreign pointer
    TORCH_CHECK(t.device() == ref.device(), name, " is on ", t.device(), ", expected ", ref.device());
}

// icfg: raster_size, aa, texture_mode, texture_res, double_side, alpha_only[, backward_enumeration]
// fcfg: background r g b, near, far, eps, sigma, dist_eps, gamma
void check_icfg(IntList icfg)
{
    TORCH_CHECK_VALUE(icfg.size() == 6 || icfg.size() == 7, "raster configuration must hold 6 or 7 integers");
}

sv3d_raster_params make_params(int64_t B, int64_t F, IntList icfg, FloatList fcfg, const optional<Tensor> &tex)
{
    check_icfg(icfg);
    TORCH_CHECK_VALUE(fcfg.size() == 9, "raster configuration must hold 9 floats");
    sv3d_raster_params p;
    p.backward_enumeration = icfg.size() > 6 ? (int32_t)icfg[6] : SV3D_BWD_AUTO;
    p.batch = (int32_t)B;
    p.num_faces = (int32_t)F;
    p.raster_size = (int32_t)icfg[0];
    p.aa = (int32_t)icfg[1];
    p.texture_mode = (int32_t)icfg[2];
    p.texture_res = (int32_t)icfg[3];
    p.double_side = icfg[4] ? 1 : 0;
    p.alpha_only = (icfg[5] && icfg[2] == SV3D_TEX_NONE) ? 1 : 0;
    p.texture_batch = 0;
    if (tex.has_value() && tex->defined() && icfg[2] != SV3D_TEX_NONE && tex->size(0) != B)
        p.texture_batch = (int32_t)tex->size(0);
    for (int i = 0; i < 3; ++i) p.background[i] = (float)fcfg[i];
    p.near_plane = (float)fcfg[3];
    p.far_plane = (float)fcfg[4];
    p.eps = (float)fcfg[5];
    p.sigma = (float)fcfg[6];
    p.dist_eps = (float)fcfg[7];
    p.gamma = (float)fcfg[8];
    return p;
}

int64_t image_planes(const sv3d_raster_params &p)
{
    if (p.alpha_only) return 1;
    return p.texture_mode == SV3D_TEX_VERTEX8 ? 9 : 4;
}

void check_textures(const optional<Tensor> &tex, int64_t B, int64_t F, int64_t mode, int64_t res)
{
    if (mode == SV3D_TEX_NONE) return;
    TORCH_CHECK_VALUE(tex.has_value() && tex->defined(), "textures required for texture_mode ", mode);
    const int64_t d2 = mode == SV3D_TEX_SURFACE ? res * res : 3;
    const int64_t d3 = mode == SV3D_TEX_VERTEX8 ? 8 : 3;
    TORCH_CHECK_VALUE(tex->dim() == 4 && tex->size(1) == F && tex->size(2) == d2 && tex->size(3) == d3 &&
                          tex->size(0) >= 1 && B % tex->size(0) == 0,
                      "textures must be [", B, " or a divisor of it, ", F, ", ", d2, ", ", d3, "], got ", tex->sizes());
}

void check_mesh(const Tensor &vertices, const Tensor &cams, const Tensor &faces)
{
    TORCH_CHECK_VALUE(vertices.dim() == 3 && vertices.size(2) == 3 && vertices.size(0) > 0 && vertices.size(1) > 0,
                      "vertices must be [B,V,3], got ", vertices.sizes());
    TORCH_CHECK_VALUE(cams.dim() == 2 && cams.size(0) == vertices.size(0) && cams.size(1) == 7,
                      "cams must be [B,7], got ", cams.sizes());
    TORCH_CHECK_VALUE(faces.dim() == 3 && faces.size(2) == 3 && faces.scalar_type() == at::kInt &&
                          (faces.size(0) == 1 || faces.size(0) == vertices.size(0)) && faces.size(1) > 0,
                      "faces must be int32 [B or 1,F,3], got ", faces.scalar_type(), " ", faces.sizes());
    same_device(vertices, cams, "cams");
    same_device(vertices, faces, "faces");
}

sv3d_mesh_input mesh_input(const Tensor &v, const Tensor &c, const Tensor &f, FloatList mesh_cfg, bool with_shift)
{
    TORCH_CHECK_VALUE(mesh_cfg.size() == 7, "mesh configuration must hold offset_z, scale[3], shift[3]");
    sv3d_mesh_input m;
    m.vertices = v.data_ptr<float>();
    m.cams = c.data_ptr<float>();
    m.faces = f.data_ptr<int32_t>();
    m.num_vertices = (int32_t)v.size(1);
    m.faces_batch = (int32_t)f.size(0);
    m.offset_z = (float)mesh_cfg[0];
    for (int i = 0; i < 3; ++i) {
        m.scale[i] = (float)mesh_cfg[1 + i];
        m.shift[i] = with_shift ? (float)mesh_cfg[4 + i] : 0.f;
    }
    return m;
}

struct RasterBuffers {
    Tensor rgba, rec, saved;
};

RasterBuffers alloc_raster(const sv3d_raster_params &p, const at::TensorOptions &opt)
{
    size_t rec_b = 0, saved_b = 0, ws_b = 0;
    const int st = sv3d_raster_sizes(&p, &rec_b, &saved_b, &ws_b);
    TORCH_CHECK(st == SV3D_OK, "sv3d_raster_sizes failed: ", sv3d_strerror(st), " (status ", st, ")");
    const int64_t H = p.raster_size / p.aa;
    RasterBuffers b;
    b.rgba = at::empty({p.batch, image_planes(p), H, H}, opt);
    b.rec = at::empty({(int64_t)(rec_b / 4)}, opt);
    b.saved = at::empty({(int64_t)(saved_b / 4)}, opt);
    return b;
}

// The two buffers a forward call hands to its backward, as the backward operators receive them from
// a caller: they must be what sv3d_raster_sizes says for this configuration, or the kernels would
// read past their ends.
void check_saved_buffers(const sv3d_raster_params &p, const Tensor &ref, const Tensor &face_rec, const Tensor &pixel_saved)
{
    size_t rec_b = 0, saved_b = 0, ws_b = 0;
    check_status(sv3d_raster_sizes(&p, &rec_b, &saved_b, &ws_b), "sv3d_raster_sizes");
    for (const auto &pr : {std::make_pair(&face_rec, rec_b), std::make_pair(&pixel_saved, saved_b)}) {
        const Tensor &t = *pr.first;
        const char *name = pr.first == &face_rec ? "face_rec" : "pixel_saved";
        TORCH_CHECK_VALUE(t.defined() && t.scalar_type() == at::kFloat && t.is_contiguous(), name,
                          " must be a contiguous float32 tensor (an output of the forward operator)");
        same_device(ref, t, name);
        TORCH_CHECK_VALUE((size_t)t.numel() * 4 >= pr.second, name, " holds ", t.numel() * 4, " bytes, this configuration needs ",
                          pr.second);
    }
}

Tensor fold_grad_textures(const Tensor &gtex, const Tensor &tex)
{
    // per-image gradients summed over the images that shared a texture set
    if (gtex.size(0) == tex.size(0)) return gtex;
    std::vector<int64_t> shape = {gtex.size(0) / tex.size(0)};
    for (auto s : tex.sizes()) shape.push_back(s);
    return gtex.view(shape).sum(0);
}

// ---- soft_rasterize ---------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor> soft_rasterize_gpu(const Tensor &face_vertices, const optional<Tensor> &textures,
                                                      IntList icfg, FloatList fcfg)
{
    TORCH_CHECK_VALUE(face_vertices.dim() == 4 && face_vertices.size(2) == 3 && face_vertices.size(3) == 3 &&
                          face_vertices.size(0) > 0 && face_vertices.size(1) > 0,
                      "face_vertices must be [B,F,3,3], got ", face_vertices.sizes());
    const int64_t B = face_vertices.size(0), F = face_vertices.size(1);
    check_icfg(icfg);
    check_textures(textures, B, F, icfg[2], icfg[3]);
    const Tensor fv = f32c(face_vertices);
    Tensor tex;
    if (icfg[2] != SV3D_TEX_NONE) {
        same_device(fv, *textures, "textures");
        tex = f32c(*textures);
    }
    const sv3d_raster_params p = make_params(B, F, icfg, fcfg, textures);
    Launch L(fv);
    RasterBuffers b = alloc_raster(p, fv.options());
    check_status(sv3d_raster_forward(&p, fv.data_ptr<float>(), fptr(tex), b.rgba.data_ptr<float>(),
                                     b.rec.data_ptr<float>(), b.saved.data_ptr<float>(), L.stream),
                 "sv3d_raster_forward");
    return {b.rgba, b.rec, b.saved};
}

std::tuple<Tensor, Tensor, Tensor> raster_meta(const Tensor &like, int64_t B, int64_t F, IntList icfg, FloatList fcfg,
                                               const optional<Tensor> &textures)
{
    const sv3d_raster_params p = make_params(B, F, icfg, fcfg, textures);
    size_t rec_b = 0, saved_b = 0, ws_b = 0;
    check_status(sv3d_raster_sizes(&p, &rec_b, &saved_b, &ws_b), "sv3d_raster_sizes");
    const int64_t H = p.raster_size / p.aa;
    const auto opt = like.options().dtype(at::kFloat);
    return {at::empty({B, image_planes(p), H, H}, opt), at::empty({(int64_t)(rec_b / 4)}, opt),
            at::empty({(int64_t)(saved_b / 4)}, opt)};
}

std::tuple<Tensor, Tensor, Tensor> soft_rasterize_meta(const Tensor &face_vertices, const optional<Tensor> &textures,
                                                       IntList icfg, FloatList fcfg)
{
    return raster_meta(face_vertices, face_vertices.size(0), face_vertices.size(1), icfg, fcfg, textures);
}

// shape = (batch, num_faces)
std::tuple<Tensor, Tensor> soft_rasterize_backward_gpu(const Tensor &grad_rgba, const optional<Tensor> &textures,
                                                       const Tensor &face_rec, const Tensor &pixel_saved,
                                                       IntList shape, IntList icfg, FloatList fcfg, bool need_textures)
{
    check_icfg(icfg);
    TORCH_CHECK_VALUE(shape.size() == 2 && shape[0] > 0 && shape[1] > 0, "shape must be (batch, faces)");
    const int64_t B = shape[0], F = shape[1];
    const Tensor g = f32c(grad_rgba);
    Tensor tex;
    if (icfg[2] != SV3D_TEX_NONE) {
        check_textures(textures, B, F, icfg[2], icfg[3]);
        same_device(g, *textures, "textures");
        tex = f32c(*textures);
    }
    const sv3d_raster_params p = make_params(B, F, icfg, fcfg, textures);
    TORCH_CHECK_VALUE(g.numel() == B * image_planes(p) * (p.raster_size / p.aa) * (p.raster_size / p.aa),
                      "grad_rgba does not match the rendered image, got ", g.sizes());
    check_saved_buffers(p, g, face_rec, pixel_saved);
    Launch L(g);
    Tensor gfv = at::empty({B, F, 3, 3}, g.options());
    Tensor gtex = at::empty({0}, g.options());
    if (tex.defined() && need_textures) {
        std::vector<int64_t> s = tex.sizes().vec();
        s[0] = B;
        gtex = at::empty(s, g.options());
    }
    check_status(sv3d_raster_backward(&p, fptr(tex), face_rec.data_ptr<float>(), pixel_saved.data_ptr<float>(),
                                      g.data_ptr<float>(), gfv.data_ptr<float>(), fptr_mut(gtex), nullptr, 0,
                                      L.stream),
                 "sv3d_raster_backward");
    if (tex.defined() && need_textures) gtex = fold_grad_textures(gtex, tex);
    return {gfv, gtex};
}

std::tuple<Tensor, Tensor> soft_rasterize_backward_meta(const Tensor &grad_rgba, const optional<Tensor> &textures,
                                                        const Tensor &face_rec, const Tensor &pixel_saved,
                                                        IntList shape, IntList icfg, FloatList fcfg, bool need_textures)
{
    const auto opt = grad_rgba.options().dtype(at::kFloat);
    Tensor gtex = (textures.has_value() && textures->defined() && need_textures && icfg[2] != SV3D_TEX_NONE)
                      ? at::empty(textures->sizes(), opt)
                      : at::empty({0}, opt);
    return {at::empty({shape[0], shape[1], 3, 3}, opt), gtex};
}

// ---- render -------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor> render_gpu(const Tensor &vertices, const Tensor &cams, const Tensor &faces,
                                              const optional<Tensor> &textures, FloatList mesh_cfg, IntList icfg,
                                              FloatList fcfg)
{
    check_mesh(vertices, cams, faces);
    const int64_t B = vertices.size(0), F = faces.size(1);
    check_icfg(icfg);
    check_textures(textures, B, F, icfg[2], icfg[3]);
    const Tensor v = f32c(vertices), c = f32c(cams), f = faces.contiguous();
    Tensor tex;
    if (icfg[2] != SV3D_TEX_NONE) {
        same_device(v, *textures, "textures");
        tex = f32c(*textures);
    }
    const sv3d_raster_params p = make_params(B, F, icfg, fcfg, textures);
    const sv3d_mesh_input m = mesh_input(v, c, f, mesh_cfg, true);
    Launch L(v);
    RasterBuffers b = alloc_raster(p, v.options());
    check_status(sv3d_render_forward(&p, &m, fptr(tex), nullptr, b.rgba.data_ptr<float>(), b.rec.data_ptr<float>(),
                                     b.saved.data_ptr<float>(), L.stream),
                 "sv3d_render_forward");
    return {b.rgba, b.rec, b.saved};
}

std::tuple<Tensor, Tensor, Tensor> render_meta(const Tensor &vertices, const Tensor &cams, const Tensor &faces,
                                               const optional<Tensor> &textures, FloatList mesh_cfg, IntList icfg,
                                               FloatList fcfg)
{
    return raster_meta(vertices, vertices.size(0), faces.size(1), icfg, fcfg, textures);
}

std::tuple<Tensor, Tensor, Tensor> render_backward_gpu(const Tensor &grad_rgba, const Tensor &vertices,
                                                       const Tensor &cams, const Tensor &faces,
                                                       const optional<Tensor> &textures, const Tensor &face_rec,
                                                       const Tensor &pixel_saved, FloatList mesh_cfg, IntList icfg,
                                                       FloatList fcfg, bool need_textures)
{
    check_mesh(vertices, cams, faces);
    same_device(grad_rgba, vertices, "vertices");
    const Tensor g = f32c(grad_rgba), v = f32c(vertices), c = f32c(cams), f = faces.contiguous();
    const int64_t B = v.size(0), F = f.size(1);
    check_icfg(icfg);
    Tensor tex;
    if (icfg[2] != SV3D_TEX_NONE) {
        check_textures(textures, B, F, icfg[2], icfg[3]);
        same_device(g, *textures, "textures");
        tex = f32c(*textures);
    }
    const sv3d_raster_params p = make_params(B, F, icfg, fcfg, textures);
    TORCH_CHECK_VALUE(g.numel() == B * image_planes(p) * (p.raster_size / p.aa) * (p.raster_size / p.aa),
                      "grad_rgba does not match the rendered image, got ", g.sizes());
    check_saved_buffers(p, g, face_rec, pixel_saved);
    Launch L(g);
    Tensor gfv = at::empty({B, F, 3, 3}, g.options());
    Tensor gtex = at::empty({0}, g.options());
    if (tex.defined() && need_textures) {
        std::vector<int64_t> s = tex.sizes().vec();
        s[0] = B;
        gtex = at::empty(s, g.options());
    }
    Tensor gv = at::empty_like(v), gc = at::empty_like(c);
    check_status(sv3d_raster_backward(&p, fptr(tex), face_rec.data_ptr<float>(), pixel_saved.data_ptr<float>(),
                                      g.data_ptr<float>(), gfv.data_ptr<float>(), fptr_mut(gtex), nullptr, 0,
                                      L.stream),
                 "sv3d_raster_backward");
    const sv3d_mesh_input m = mesh_input(v, c, f, mesh_cfg, false);
    check_status(sv3d_project_scatter(&m, (int32_t)B, (int32_t)F, gfv.data_ptr<float>(), gv.data_ptr<float>(),
                                      gc.data_ptr<float>(), L.stream),
                 "sv3d_project_scatter");
    if (tex.defined() && need_textures) gtex = fold_grad_textures(gtex, tex);
    return {gv, gc, gtex};
}

std::tuple<Tensor, Tensor, Tensor> render_backward_meta(const Tensor &grad_rgba, const Tensor &vertices,
                                                        const Tensor &cams, const Tensor &faces,
                                                        const optional<Tensor> &textures, const Tensor &face_rec,
                                                        const Tensor &pixel_saved, FloatList mesh_cfg, IntList icfg,
                                                        FloatList fcfg, bool need_textures)
{
    const auto opt = grad_rgba.options().dtype(at::kFloat);
    Tensor gtex = (textures.has_value() && textures->defined() && need_textures && icfg[2] != SV3D_TEX_NONE)
                      ? at::empty(textures->sizes(), opt)
                      : at::empty({0}, opt);
    return {at::empty(vertices.sizes(), opt), at::empty(cams.sizes(), opt), gtex};
}

// ---- chamfer -----------------------------------------------------------------------------------
void check_clouds(const Tensor &a, const Tensor &b)
{
    TORCH_CHECK_VALUE(a.dim() == 3 && b.dim() == 3 && a.size(2) == 3 && b.size(2) == 3 && a.size(0) == b.size(0),
                      "point clouds must be [B,N,3] and [B,M,3], got ", a.sizes(), " and ", b.sizes());
    TORCH_CHECK_VALUE(a.size(0) > 0 && a.size(1) > 0 && b.size(1) > 0, "empty point cloud");
}

std::tuple<Tensor, Tensor, Tensor, Tensor> chamfer_nn_gpu(const Tensor &xyz1, const Tensor &xyz2)
{
    check_clouds(xyz1, xyz2);
    same_device(xyz1, xyz2, "xyz2");
    const Tensor a = f32c(xyz1), b = f32c(xyz2);
    const int64_t B = a.size(0), N = a.size(1), M = b.size(1);
    Launch L(a);
    Tensor d1 = at::empty({B, N}, a.options()), d2 = at::empty({B, M}, a.options());
    Tensor i1 = at::empty({B, N}, a.options().dtype(at::kInt)), i2 = at::empty({B, M}, a.options().dtype(at::kInt));
    check_status(sv3d_chamfer_forward(a.data_ptr<float>(), b.data_ptr<float>(), (int32_t)B, (int32_t)N, (int32_t)M,
                                      d1.data_ptr<float>(), d2.data_ptr<float>(), i1.data_ptr<int32_t>(),
                                      i2.data_ptr<int32_t>(), L.stream),
                 "sv3d_chamfer_forward");
    return {d1, d2, i1, i2};
}

std::tuple<Tensor, Tensor, Tensor, Tensor> chamfer_nn_meta(const Tensor &xyz1, const Tensor &xyz2)
{
    check_clouds(xyz1, xyz2);
    const int64_t B = xyz1.size(0), N = xyz1.size(1), M = xyz2.size(1);
    const auto f = xyz1.options().dtype(at::kFloat), i = xyz1.options().dtype(at::kInt);
    return {at::empty({B, N}, f), at::empty({B, M}, f), at::empty({B, N}, i), at::empty({B, M}, i)};
}

std::tuple<Tensor, Tensor> chamfer_nn_backward_gpu(const Tensor &xyz1, const Tensor &xyz2,
                                                   const optional<Tensor> &grad_dist1,
                                                   const optional<Tensor> &grad_dist2, const Tensor &idx1,
                                                   const Tensor &idx2, bool need1, bool need2)
{
    check_clouds(xyz1, xyz2);
    same_device(xyz1, xyz2, "xyz2");
    const Tensor a = f32c(xyz1), b = f32c(xyz2);
    const int64_t B = a.size(0), N = a.size(1), M = b.size(1);
    Tensor gd1, gd2;
    if (grad_dist1.has_value() && grad_dist1->defined()) gd1 = f32c(*grad_dist1);
    if (grad_dist2.has_value() && grad_dist2->defined()) gd2 = f32c(*grad_dist2);
    TORCH_CHECK_VALUE(!gd1.defined() || gd1.numel() == B * N, "grad_dist1 must be [B,N], got ", gd1.sizes());
    TORCH_CHECK_VALUE(!gd2.defined() || gd2.numel() == B * M, "grad_dist2 must be [B,M], got ", gd2.sizes());
    if (gd1.defined()) same_device(a, gd1, "grad_dist1");
    if (gd2.defined()) same_device(a, gd2, "grad_dist2");
    for (const auto &pr : {std::make_pair(&idx1, B * N), std::make_pair(&idx2, B * M)}) {
        const Tensor &t = *pr.first;
        const char *name = pr.first == &idx1 ? "idx1" : "idx2";
        TORCH_CHECK_VALUE(t.defined() && t.scalar_type() == at::kInt && t.is_contiguous() && t.numel() == pr.second, name,
                          " must be a contiguous int32 tensor with one entry per point");
        same_device(a, t, name);
    }
    Launch L(a);
    Tensor g1 = need1 ? at::empty_like(a) : at::empty({0}, a.options());
    Tensor g2 = need2 ? at::empty_like(b) : at::empty({0}, a.options());
    check_status(sv3d_chamfer_backward(a.data_ptr<float>(), b.data_ptr<float>(), (int32_t)B, (int32_t)N, (int32_t)M,
                                       fptr(gd1), fptr(gd2), idx1.data_ptr<int32_t>(), idx2.data_ptr<int32_t>(),
                                       fptr_mut(g1), fptr_mut(g2), L.stream),
                 "sv3d_chamfer_backward");
    return {g1, g2};
}

std::tuple<Tensor, Tensor> chamfer_nn_backward_meta(const Tensor &xyz1, const Tensor &xyz2,
                                                    const optional<Tensor> &, const optional<Tensor> &,
                                                    const Tensor &, const Tensor &, bool need1, bool need2)
{
    const auto opt = xyz1.options().dtype(at::kFloat);
    return {need1 ? at::empty(xyz1.sizes(), opt) : at::empty({0}, opt),
            need2 ? at::empty(xyz2.sizes(), opt) : at::empty({0}, opt)};
}

// ---- gathers ------------------------------------------------------------------------------------
void check_faces(const Tensor &values, const Tensor &faces)
{
    TORCH_CHECK_VALUE(faces.dim() == 3 && faces.size(2) == 3 && faces.scalar_type() == at::kInt &&
                          (faces.size(0) == 1 || faces.size(0) == values.size(0)) && faces.size(1) > 0,
                      "faces must be int32 [B or 1,F,3], got ", faces.scalar_type(), " ", faces.sizes());
    same_device(values, faces, "faces");
}

void float3(FloatList v, float out[3], const char *name)
{
    TORCH_CHECK_VALUE(v.size() == 3, name, " must hold 3 floats");
    for (int i = 0; i < 3; ++i) out[i] = (float)v[i];
}

Tensor face_gather_gpu(const Tensor &values, const Tensor &faces, FloatList scale, FloatList shift)
{
    TORCH_CHECK_VALUE(values.dim() == 3 && values.size(2) == 3 && values.size(0) > 0 && values.size(1) > 0,
                      "values must be [B,V,3], got ", values.sizes());
    check_faces(values, faces);
    const Tensor v = f32c(values), f = faces.contiguous();
    float sc[3], sh[3];
    float3(scale, sc, "scale");
    float3(shift, sh, "shift");
    Launch L(v);
    Tensor out = at::empty({v.size(0), f.size(1), 3, 3}, v.options());
    check_status(sv3d_face_gather(v.data_ptr<float>(), f.data_ptr<int32_t>(), (int32_t)v.size(0), (int32_t)v.size(1),
                                  (int32_t)f.size(1), (int32_t)f.size(0), sc, sh, out.data_ptr<float>(), L.stream),
                 "sv3d_face_gather");
    return out;
}

Tensor face_gather_meta(const Tensor &values, const Tensor &faces, FloatList, FloatList)
{
    return at::empty({values.size(0), faces.size(1), 3, 3}, values.options().dtype(at::kFloat));
}

Tensor face_scatter_gpu(const Tensor &grad_out, const Tensor &faces, int64_t num_vertices, FloatList scale)
{
    const Tensor g = f32c(grad_out), f = faces.contiguous();
    float sc[3];
    float3(scale, sc, "scale");
    Launch L(g);
    Tensor out = at::empty({g.size(0), num_vertices, 3}, g.options());
    check_status(sv3d_face_scatter(g.data_ptr<float>(), f.data_ptr<int32_t>(), (int32_t)g.size(0),
                                   (int32_t)num_vertices, (int32_t)g.size(1), (int32_t)f.size(0), sc,
                                   out.data_ptr<float>(), L.stream),
                 "sv3d_face_scatter");
    return out;
}

Tensor face_scatter_meta(const Tensor &grad_out, const Tensor &, int64_t num_vertices, FloatList)
{
    return at::empty({grad_out.size(0), num_vertices, 3}, grad_out.options().dtype(at::kFloat));
}

Tensor face_gather_n_gpu(const Tensor &values, const Tensor &faces)
{
    TORCH_CHECK_VALUE(values.dim() == 3 && values.size(2) >= 1 && values.size(2) <= 8 && values.size(0) > 0,
                      "values must be [B,V,C] with C <= 8, got ", values.sizes());
    check_faces(values, faces);
    const Tensor v = f32c(values), f = faces.contiguous();
    Launch L(v);
    Tensor out = at::empty({v.size(0), f.size(1), 3, v.size(2)}, v.options());
    check_status(sv3d_face_gather_n(v.data_ptr<float>(), f.data_ptr<int32_t>(), (int32_t)v.size(0), (int32_t)v.size(1),
                                    (int32_t)f.size(1), (int32_t)f.size(0), (int32_t)v.size(2), out.data_ptr<float>(),
                                    L.stream),
                 "sv3d_face_gather_n");
    return out;
}

Tensor face_gather_n_meta(const Tensor &values, const Tensor &faces)
{
    return at::empty({values.size(0), faces.size(1), 3, values.size(2)}, values.options().dtype(at::kFloat));
}

Tensor face_scatter_n_gpu(const Tensor &grad_out, const Tensor &faces, int64_t num_vertices)
{
    const Tensor g = f32c(grad_out), f = faces.contiguous();
    Launch L(g);
    Tensor out = at::empty({g.size(0), num_vertices, g.size(3)}, g.options());
    check_status(sv3d_face_scatter_n(g.data_ptr<float>(), f.data_ptr<int32_t>(), (int32_t)g.size(0),
                                     (int32_t)num_vertices, (int32_t)g.size(1), (int32_t)f.size(0), (int32_t)g.size(3),
                                     out.data_ptr<float>(), L.stream),
                 "sv3d_face_scatter_n");
    return out;
}

Tensor face_scatter_n_meta(const Tensor &grad_out, const Tensor &, int64_t num_vertices)
{
    return at::empty({grad_out.size(0), num_vertices, grad_out.size(3)}, grad_out.options().dtype(at::kFloat));
}

Tensor project_gather_gpu(const Tensor &vertices, const Tensor &cams, const Tensor &faces, FloatList mesh_cfg)
{
    check_mesh(vertices, cams, faces);
    const Tensor v = f32c(vertices), c = f32c(cams), f = faces.contiguous();
    const sv3d_mesh_input m = mesh_input(v, c, f, mesh_cfg, true);
    Launch L(v);
    Tensor out = at::empty({v.size(0), f.size(1), 3, 3}, v.options());
    check_status(sv3d_project_gather(&m, (int32_t)v.size(0), (int32_t)f.size(1), out.data_ptr<float>(), L.stream),
                 "sv3d_project_gather");
    return out;
}

Tensor project_gather_meta(const Tensor &vertices, const Tensor &, const Tensor &faces, FloatList)
{
    return at::empty({vertices.size(0), faces.size(1), 3, 3}, vertices.options().dtype(at::kFloat));
}

std::tuple<Tensor, Tensor> project_scatter_gpu(const Tensor &grad_out, const Tensor &vertices, const Tensor &cams,
                                               const Tensor &faces, FloatList mesh_cfg)
{
    const Tensor g = f32c(grad_out), v = f32c(vertices), c = f32c(cams), f = faces.contiguous();
    const sv3d_mesh_input m = mesh_input(v, c, f, mesh_cfg, false);
    Launch L(g);
    Tensor gv = at::empty_like(v), gc = at::empty_like(c);
    check_status(sv3d_project_scatter(&m, (int32_t)g.size(0), (int32_t)g.size(1), g.data_ptr<float>(),
                                      gv.data_ptr<float>(), gc.data_ptr<float>(), L.stream),
                 "sv3d_project_scatter");
    return {gv, gc};
}

std::tuple<Tensor, Tensor> project_scatter_meta(const Tensor &grad_out, const Tensor &vertices, const Tensor &cams,
                                                const Tensor &, FloatList)
{
    const auto opt = grad_out.options().dtype(at::kFloat);
    return {at::empty(vertices.sizes(), opt), at::empty(cams.sizes(), opt)};
}

// ---- mesh regularisers, CSR product, texture path ------------------------------------------------
void check_vertices(const Tensor &v, const char *name)
{
    TORCH_CHECK_VALUE(v.dim() == 3 && v.size(2) == 3 && v.size(0) > 0 && v.size(1) > 0, name,
                      " must be a non-empty [B,V,3] tensor, got ", v.sizes());
    TORCH_CHECK_VALUE(v.is_floating_point(), name, " must be a floating point tensor");
}

void check_table(const Tensor &ref, const Tensor &t, const char *name)
{
    TORCH_CHECK_VALUE(t.scalar_type() == at::kInt && t.is_contiguous(), name, " must be a contiguous int32 tensor");
    same_device(ref, t, name);      // a host table handed to a kernel is a GPU memory fault, not an exception
}

std::tuple<Tensor, Tensor> laplacian_loss_gpu(const Tensor &vertices, const Tensor &row_start, const Tensor &nbr)
{
    check_vertices(vertices, "vertices");
    check_table(vertices, row_start, "row_start");
    check_table(vertices, nbr, "nbr");
    const Tensor x = f32c(vertices);
    const int64_t B = x.size(0), V = x.size(1);
    TORCH_CHECK_VALUE(row_start.numel() == V + 1, "row_start must hold V + 1 = ", V + 1, " entries, got ", row_start.numel());
    Launch L(x);
    Tensor loss = at::empty({B}, x.options()), grad = at::empty_like(x);
    check_status(sv3d_laplacian_loss(x.data_ptr<float>(), row_start.data_ptr<int32_t>(), nbr.data_ptr<int32_t>(),
                                     (int32_t)B, (int32_t)V, loss.data_ptr<float>(), grad.data_ptr<float>(), L.stream),
                 "sv3d_laplacian_loss");
    return {loss, grad};
}

std::tuple<Tensor, Tensor> loss_and_grad_meta(const Tensor &vertices)
{
    const auto opt = vertices.options().dtype(at::kFloat);
    return {at::empty({vertices.size(0)}, opt), at::empty(vertices.sizes(), opt)};
}

std::tuple<Tensor, Tensor> laplacian_loss_meta(const Tensor &vertices, const Tensor &, const Tensor &)
{
    return loss_and_grad_meta(vertices);
}

std::tuple<Tensor, Tensor> flatten_loss_gpu(const Tensor &vertices, const Tensor &quads, const Tensor &inc_start,
                                            const Tensor &inc, double eps)
{
    check_vertices(vertices, "vertices");
    check_table(vertices, quads, "quads");
    check_table(vertices, inc_start, "inc_start");
    check_table(vertices, inc, "inc");
    const Tensor x = f32c(vertices);
    const int64_t B = x.size(0), V = x.size(1);
    TORCH_CHECK_VALUE(quads.dim() == 2 && quads.size(1) == 4 && quads.size(0) > 0,
                      "quads must be a non-empty [E,4] tensor, got ", quads.sizes());
    const int64_t E = quads.size(0);
    TORCH_CHECK_VALUE(inc_start.numel() == V + 1 && inc.numel() == 4 * E, "incidence tables must hold V + 1 and 4 E entries");
    Launch L(x);
    Tensor loss = at::empty({B}, x.options()), grad = at::empty_like(x);
    check_status(sv3d_flatten_loss(x.data_ptr<float>(), quads.data_ptr<int32_t>(), inc_start.data_ptr<int32_t>(),
                                   inc.data_ptr<int32_t>(), (int32_t)B, (int32_t)V, (int32_t)E, (float)eps,
                                   loss.data_ptr<float>(), grad.data_ptr<float>(), L.stream),
                 "sv3d_flatten_loss");
    return {loss, grad};
}

std::tuple<Tensor, Tensor> flatten_loss_meta(const Tensor &vertices, const Tensor &, const Tensor &, const Tensor &, double)
{
    return loss_and_grad_meta(vertices);
}

// values_t, row_start_t, col_t: the transposed operator, used by the backward only
Tensor mesh_spmv_gpu(const Tensor &x, const Tensor &values, const Tensor &row_start, const Tensor &col, const Tensor &,
                     const Tensor &, const Tensor &)
{
    check_vertices(x, "x");
    check_table(x, row_start, "row_start");
    check_table(x, col, "col");
    same_device(x, values, "values");
    const Tensor xs = f32c(x), vals = f32c(values);
    const int64_t B = xs.size(0), V = xs.size(1), nnz = col.numel();
    TORCH_CHECK_VALUE(vals.dim() == 2 && vals.size(1) == nnz && (vals.size(0) == 1 || vals.size(0) == B),
                      "values must be [B or 1, nnz=", nnz, "], got ", vals.sizes());
    TORCH_CHECK_VALUE(row_start.numel() == V + 1, "row_start must hold V + 1 = ", V + 1, " entries, got ", row_start.numel());
    Launch L(xs);
    Tensor y = at::empty_like(xs);
    check_status(sv3d_mesh_spmv(vals.data_ptr<float>(), row_start.data_ptr<int32_t>(), col.data_ptr<int32_t>(),
                                xs.data_ptr<float>(), (int32_t)B, (int32_t)V, (int32_t)nnz, (int32_t)vals.size(0),
                                y.data_ptr<float>(), L.stream),
                 "sv3d_mesh_spmv");
    return y;
}

Tensor mesh_spmv_meta(const Tensor &x, const Tensor &, const Tensor &, const Tensor &, const Tensor &, const Tensor &,
                      const Tensor &)
{
    return at::empty(x.sizes(), x.options().dtype(at::kFloat));
}

constexpr int64_t kMaxSampleChannels = 4;

void check_sample(const Tensor &images, const Tensor &flow)
{
    same_device(images, flow, "flow");
    TORCH_CHECK_VALUE(images.dim() == 4 && images.size(1) >= 1 && images.size(1) <= kMaxSampleChannels &&
                          images.size(2) >= 1 && images.size(3) >= 1,
                      "images must be [B,C,H,W] with C <= ", kMaxSampleChannels, ", got ", images.sizes());
    TORCH_CHECK_VALUE(flow.dim() == 3 && flow.size(2) == 2 && flow.size(0) == images.size(0) && flow.size(1) > 0,
                      "flow must be [B,N,2] with the images' batch size, got ", flow.sizes());
}

Tensor texture_sample_gpu(const Tensor &images, const Tensor &flow)
{
    check_sample(images, flow);
    const Tensor img = f32c(images), fl = f32c(flow);
    const int64_t B = img.size(0), C = img.size(1), H = img.size(2), W = img.size(3), N = fl.size(1);
    Launch L(img);
    Tensor out = at::empty({B, N, C}, img.options());
    check_status(sv3d_texture_sample(img.data_ptr<float>(), fl.data_ptr<float>(), (int32_t)B, (int32_t)C, (int32_t)H,
                                     (int32_t)W, (int32_t)N, out.data_ptr<float>(), L.stream),
                 "sv3d_texture_sample");
    return out;
}

Tensor texture_sample_meta(const Tensor &images, const Tensor &flow)
{
    return at::empty({flow.size(0), flow.size(1), images.size(1)}, images.options().dtype(at::kFloat));
}

Tensor texture_sample_backward_gpu(const Tensor &images, const Tensor &flow, const Tensor &grad_out)
{
    check_sample(images, flow);
    same_device(images, grad_out, "grad_out");
    const Tensor img = f32c(images), fl = f32c(flow), g = f32c(grad_out);
    const int64_t B = img.size(0), C = img.size(1), H = img.size(2), W = img.size(3), N = fl.size(1);
    TORCH_CHECK_VALUE(g.dim() == 3 && g.size(0) == B && g.size(1) == N && g.size(2) == C, "grad_out must be [B,N,C], got ",
                      g.sizes());
    Launch L(img);
    Tensor grad_flow = at::empty_like(fl);
    check_status(sv3d_texture_sample_backward(img.data_ptr<float>(), fl.data_ptr<float>(), g.data_ptr<float>(), (int32_t)B,
                                              (int32_t)C, (int32_t)H, (int32_t)W, (int32_t)N,
                                              grad_flow.data_ptr<float>(), L.stream),
                 "sv3d_texture_sample_backward");
    return grad_flow;
}

Tensor texture_sample_backward_meta(const Tensor &, const Tensor &flow, const Tensor &)
{
    return at::empty(flow.sizes(), flow.options().dtype(at::kFloat));
}

Tensor distance_transform_gpu(const Tensor &masks, bool barrier, double k)
{
    TORCH_CHECK_VALUE(masks.dim() == 3 && masks.numel() > 0, "masks must be a non-empty [B,H,W] tensor, got ", masks.sizes());
    const Tensor m = f32c(masks);
    const int64_t B = m.size(0), H = m.size(1), W = m.size(2);
    size_t bytes = 0;
    check_status(sv3d_distance_transform_workspace((int32_t)B, (int32_t)H, (int32_t)W, &bytes),
                 "sv3d_distance_transform_workspace");
    Launch L(m);
    Tensor out = at::empty_like(m);
    Tensor ws = at::empty({(int64_t)bytes}, m.options().dtype(at::kByte));
    check_status(sv3d_distance_transform(m.data_ptr<float>(), (int32_t)B, (int32_t)H, (int32_t)W, barrier ? 1 : 0, (float)k,
                                         out.data_ptr<float>(), ws.data_ptr(), bytes, L.stream),
                 "sv3d_distance_transform");
    return out;
}

Tensor distance_transform_meta(const Tensor &masks, bool, double)
{
    return at::empty(masks.sizes(), masks.options().dtype(at::kFloat));
}

// ---- dispatcher handles (calls from the autograd kernels go below autograd) ----------------------
template <class Sig>
c10::TypedOperatorHandle<Sig> find_op(const char *name)
{
    return c10::Dispatcher::singleton().findSchemaOrThrow(name, "").typed<Sig>();
}

using RasterOut = std::tuple<Tensor, Tensor, Tensor>;

RasterOut call_soft_rasterize(const Tensor &fv, const optional<Tensor> &tex, IntList icfg, FloatList fcfg)
{
    static auto op = find_op<RasterOut(const Tensor &, const optional<Tensor> &, IntList, FloatList)>("sv3d::soft_rasterize");
    return op.call(fv, tex, icfg, fcfg);
}

std::tuple<Tensor, Tensor> call_soft_rasterize_backward(const Tensor &g, const optional<Tensor> &tex, const Tensor &rec,
                                                        const Tensor &saved, IntList shape, IntList icfg,
                                                        FloatList fcfg, bool need_tex)
{
    static auto op = find_op<std::tuple<Tensor, Tensor>(const Tensor &, const optional<Tensor> &, const Tensor &,
                                                        const Tensor &, IntList, IntList, FloatList, bool)>(
        "sv3d::soft_rasterize_backward");
    return op.call(g, tex, rec, saved, shape, icfg, fcfg, need_tex);
}

RasterOut call_render(const Tensor &v, const Tensor &c, const Tensor &f, const optional<Tensor> &tex, FloatList mesh_cfg,
                      IntList icfg, FloatList fcfg)
{
    static auto op = find_op<RasterOut(const Tensor &, const Tensor &, const Tensor &, const optional<Tensor> &,
                                       FloatList, IntList, FloatList)>("sv3d::render");
    return op.call(v, c, f, tex, mesh_cfg, icfg, fcfg);
}

RasterOut call_render_backward(const Tensor &g, const Tensor &v, const Tensor &c, const Tensor &f,
                               const optional<Tensor> &tex, const Tensor &rec, const Tensor &saved, FloatList mesh_cfg,
                               IntList icfg, FloatList fcfg, bool need_tex)
{
    static auto op = find_op<RasterOut(const Tensor &, const Tensor &, const Tensor &, const Tensor &,
                                       const optional<Tensor> &, const Tensor &, const Tensor &, FloatList, IntList,
                                       FloatList, bool)>("sv3d::render_backward");
    return op.call(g, v, c, f, tex, rec, saved, mesh_cfg, icfg, fcfg, need_tex);
}

Tensor cast_like(const Tensor &g, at::ScalarType t) { return g.scalar_type() == t ? g : g.to(t); }

// ---- autograd kernels -----------------------------------------------------------------------------
struct SoftRasterizeFn : public torch::autograd::Function<SoftRasterizeFn> {
    static variable_list forward(AutogradContext *ctx, const Tensor &fv, const optional<Tensor> &tex, IntList icfg,
                                 FloatList fcfg)
    {
        at::AutoDispatchBelowADInplaceOrView guard;
        auto out = call_soft_rasterize(fv, tex, icfg, fcfg);
        const bool has_tex = tex.has_value() && tex->defined() && icfg[2] != SV3D_TEX_NONE;
        ctx->save_for_backward({has_tex ? *tex : Tensor(), std::get<1>(out), std::get<2>(out)});
        ctx->saved_data["icfg"] = icfg.vec();
        ctx->saved_data["fcfg"] = fcfg.vec();
        ctx->saved_data["shape"] = std::vector<int64_t>{fv.size(0), fv.size(1)};
        ctx->saved_data["fv_dtype"] = (int64_t)fv.scalar_type();
        ctx->saved_data["need_tex"] = has_tex && tex->requires_grad();
        ctx->mark_non_differentiable({std::get<1>(out), std::get<2>(out)});
        // no zero-filled stand-ins for the gradients of the two saved buffers (megabytes each)
        ctx->set_materialize_grads(false);
        return {std::get<0>(out), std::get<1>(out), std::get<2>(out)};
    }

    static variable_list backward(AutogradContext *ctx, variable_list grads)
    {
        if (!grads[0].defined()) return {Tensor(), Tensor(), Tensor(), Tensor()};
        const auto saved = ctx->get_saved_variables();
        const Tensor &tex = saved[0];
        const auto icfg = ctx->saved_data["icfg"].toIntVector();
        const auto fcfg = ctx->saved_data["fcfg"].toDoubleVector();
        const auto shape = ctx->saved_data["shape"].toIntVector();
        const bool need_tex = ctx->saved_data["need_tex"].toBool();
        optional<Tensor> otex;
        if (tex.defined()) otex = tex;
        auto g = call_soft_rasterize_backward(grads[0], otex, saved[1], saved[2], shape, icfg, fcfg, need_tex);
        Tensor gfv = cast_like(std::get<0>(g), (at::ScalarType)ctx->saved_data["fv_dtype"].toInt());
        Tensor gtex = need_tex ? cast_like(std::get<1>(g), tex.scalar_type()) : Tensor();
        return {gfv, gtex, Tensor(), Tensor()};
    }
};

RasterOut soft_rasterize_autograd(const Tensor &fv, const optional<Tensor> &tex, IntList icfg, FloatList fcfg)
{
    auto out = SoftRasterizeFn::apply(fv, tex, icfg, fcfg);
    return {out[0], out[1], out[2]};
}

struct RenderFn : public torch::autograd::Function<RenderFn> {
    static variable_list forward(AutogradContext *ctx, const Tensor &v, const Tensor &c, const Tensor &f,
                                 const optional<Tensor> &tex, FloatList mesh_cfg, IntList icfg, FloatList fcfg)
    {
        at::AutoDispatchBelowADInplaceOrView guard;
        auto out = call_render(v, c, f, tex, mesh_cfg, icfg, fcfg);
        const bool has_tex = tex.has_value() && tex->defined() && icfg[2] != SV3D_TEX_NONE;
        ctx->save_for_backward({v, c, f, has_tex ? *tex : Tensor(), std::get<1>(out), std::get<2>(out)});
        ctx->saved_data["mesh_cfg"] = mesh_cfg.vec();
        ctx->saved_data["icfg"] = icfg.vec();
        ctx->saved_data["fcfg"] = fcfg.vec();
        ctx->saved_data["need_tex"] = has_tex && tex->requires_grad();
        ctx->mark_non_differentiable({std::get<1>(out), std::get<2>(out)});
        ctx->set_materialize_grads(false);
        return {std::get<0>(out), std::get<1>(out), std::get<2>(out)};
    }

    static variable_list backward(AutogradContext *ctx, variable_list grads)
    {
        if (!grads[0].defined()) return {Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
        const auto saved = ctx->get_saved_variables();
        const Tensor &v = saved[0], &c = saved[1], &f = saved[2], &tex = saved[3];
        const auto mesh_cfg = ctx->saved_data["mesh_cfg"].toDoubleVector();
        const auto icfg = ctx->saved_data["icfg"].toIntVector();
        const auto fcfg = ctx->saved_data["fcfg"].toDoubleVector();
        const bool need_tex = ctx->saved_data["need_tex"].toBool();
        optional<Tensor> otex;
        if (tex.defined()) otex = tex;
        auto g = call_render_backward(grads[0], v, c, f, otex, saved[4], saved[5], mesh_cfg, icfg, fcfg, need_tex);
        Tensor gtex = need_tex ? cast_like(std::get<2>(g), tex.scalar_type()) : Tensor();
        return {cast_like(std::get<0>(g), v.scalar_type()), cast_like(std::get<1>(g), c.scalar_type()), Tensor(), gtex,
                Tensor(), Tensor(), Tensor()};
    }
};

RasterOut render_autograd(const Tensor &v, const Tensor &c, const Tensor &f, const optional<Tensor> &tex,
                          FloatList mesh_cfg, IntList icfg, FloatList fcfg)
{
    auto out = RenderFn::apply(v, c, f, tex, mesh_cfg, icfg, fcfg);
    return {out[0], out[1], out[2]};
}

struct ChamferFn : public torch::autograd::Function<ChamferFn> {
    static variable_list forward(AutogradContext *ctx, const Tensor &a, const Tensor &b)
    {
        at::AutoDispatchBelowADInplaceOrView guard;
        static auto op = find_op<std::tuple<Tensor, Tensor, Tensor, Tensor>(const Tensor &, const Tensor &)>("sv3d::chamfer_nn");
        auto out = op.call(a, b);
        ctx->save_for_backward({a, b, std::get<2>(out), std::get<3>(out)});
        ctx->mark_non_differentiable({std::get<2>(out), std::get<3>(out)});
        // a distance output that took no part in the loss keeps an undefined gradient: the backward
        // then skips that side (the reference uses dist2 alone, nnutils/loss_utils.py:87-88)
        ctx->set_materialize_grads(false);
        return {std::get<0>(out), std::get<1>(out), std::get<2>(out), std::get<3>(out)};
    }

    static variable_list backward(AutogradContext *ctx, variable_list grads)
    {
        const auto saved = ctx->get_saved_variables();
        const Tensor &a = saved[0], &b = saved[1];
        const bool need1 = ctx->needs_input_grad(0), need2 = ctx->needs_input_grad(1);
        if ((!need1 && !need2) || (!grads[0].defined() && !grads[1].defined())) return {Tensor(), Tensor()};
        static auto op = find_op<std::tuple<Tensor, Tensor>(const Tensor &, const Tensor &, const optional<Tensor> &,
                                                            const optional<Tensor> &, const Tensor &, const Tensor &,
                                                            bool, bool)>("sv3d::chamfer_nn_backward");
        optional<Tensor> g1, g2;
        if (grads[0].defined()) g1 = grads[0];
        if (grads[1].defined()) g2 = grads[1];
        auto g = op.call(a, b, g1, g2, saved[2], saved[3], need1, need2);
        return {need1 ? cast_like(std::get<0>(g), a.scalar_type()) : Tensor(),
                need2 ? cast_like(std::get<1>(g), b.scalar_type()) : Tensor()};
    }
};

std::tuple<Tensor, Tensor, Tensor, Tensor> chamfer_nn_autograd(const Tensor &a, const Tensor &b)
{
    auto out = ChamferFn::apply(a, b);
    return {out[0], out[1], out[2], out[3]};
}

struct FaceGatherFn : public torch::autograd::Function<FaceGatherFn> {
    static Tensor forward(AutogradContext *ctx, const Tensor &values, const Tensor &faces, FloatList scale,
                          FloatList shift)
    {
        at::AutoDispatchBelowADInplaceOrView guard;
        static auto op = find_op<Tensor(const Tensor &, const Tensor &, FloatList, FloatList)>("sv3d::face_gather");
        ctx->save_for_backward({faces});
        ctx->saved_data["scale"] = scale.vec();
        ctx->saved_data["nv"] = values.size(1);
        ctx->saved_data["dtype"] = (int64_t)values.scalar_type();
        return op.call(values, faces, scale, shift);
    }

    static variable_list backward(AutogradContext *ctx, variable_list grads)
    {
        static auto op = find_op<Tensor(const Tensor &, const Tensor &, int64_t, FloatList)>("sv3d::face_scatter");
        const auto saved = ctx->get_saved_variables();
        const auto scale = ctx->saved_data["scale"].toDoubleVector();
        Tensor g = op.call(grads[0], saved[0], ctx->saved_data["nv"].toInt(), scale);
        return {cast_like(g, (at::ScalarType)ctx->saved_data["dtype"].toInt()), Tensor(), Tensor(), Tensor()};
    }
};

Tensor face_gather_autograd(const Tensor &values, const Tensor &faces, FloatList scale, FloatList shift)
{
    return FaceGatherFn::apply(values, faces, scale, shift);
}

struct FaceGatherNFn : public torch::autograd::Function<FaceGatherNFn> {
    static Tensor forward(AutogradContext *ctx, const Tensor &values, const Tensor &faces)
    {
        at::AutoDispatchBelowADInplaceOrView guard;
        static auto op = find_op<Tensor(const Tensor &, const Tensor &)>("sv3d::face_gather_n");
        ctx->save_for_backward({faces});
        ctx->saved_data["nv"] = values.size(1);
        ctx->saved_data["dtype"] = (int64_t)values.scalar_type();
        return op.call(values, faces);
    }

    static variable_list backward(AutogradContext *ctx, variable_list grads)
    {
        static auto op = find_op<Tensor(const Tensor &, const Tensor &, int64_t)>("sv3d::face_scatter_n");
        const auto saved = ctx->get_saved_variables();
        Tensor g = op.call(grads[0], saved[0], ctx->saved_data["nv"].toInt());
        return {cast_like(g, (at::ScalarType)ctx->saved_data["dtype"].toInt()), Tensor()};
    }
};

Tensor face_gather_n_autograd(const Tensor &values, const Tensor &faces) { return FaceGatherNFn::apply(values, faces); }

struct ProjectGatherFn : public torch::autograd::Function<ProjectGatherFn> {
    static Tensor forward(AutogradContext *ctx, const Tensor &v, const Tensor &c, const Tensor &f, FloatList mesh_cfg)
    {
        at::AutoDispatchBelowADInplaceOrView guard;
        static auto op = find_op<Tensor(const Tensor &, const Tensor &, const Tensor &, FloatList)>("sv3d::project_gather");
        ctx->save_for_backward({v, c, f});
        ctx->saved_data["mesh_cfg"] = mesh_cfg.vec();
        return op.call(v, c, f, mesh_cfg);
    }

    static variable_list backward(AutogradContext *ctx, variable_list grads)
    {
        static auto op = find_op<std::tuple<Tensor, Tensor>(const Tensor &, const Tensor &, const Tensor &, const Tensor &,
                                                            FloatList)>("sv3d::project_scatter");
        const auto saved = ctx->get_saved_variables();
        const auto mesh_cfg = ctx->saved_data["mesh_cfg"].toDoubleVector();
        auto g = op.call(grads[0], saved[0], saved[1], saved[2], mesh_cfg);
        return {cast_like(std::get<0>(g), saved[0].scalar_type()), cast_like(std::get<1>(g), saved[1].scalar_type()),
                Tensor(), Tensor()};
    }
};

Tensor project_gather_autograd(const Tensor &v, const Tensor &c, const Tensor &f, FloatList mesh_cfg)
{
    return ProjectGatherFn::apply(v, c, f, mesh_cfg);
}


// loss [B] with its vertex gradient computed by the same launch: the backward scales that gradient
template <class Derived>
struct LossWithGradFn : public torch::autograd::Function<Derived> {
    static variable_list backward(AutogradContext *ctx, variable_list grads)
    {
        variable_list out(Derived::kInputs);
        if (!grads[0].defined()) return out;
        const auto saved = ctx->get_saved_variables();
        const Tensor g = grads[0].to(at::kFloat).view({-1, 1, 1}) * saved[0];
        out[0] = cast_like(g, (at::ScalarType)ctx->saved_data["dtype"].toInt());
        return out;
    }
};

struct LaplacianLossFn : public LossWithGradFn<LaplacianLossFn> {
    static constexpr int kInputs = 3;
    static variable_list forward(AutogradContext *ctx, const Tensor &v, const Tensor &row_start, const Tensor &nbr)
    {
        at::AutoDispatchBelowADInplaceOrView guard;
        static auto op = find_op<std::tuple<Tensor, Tensor>(const Tensor &, const Tensor &, const Tensor &)>("sv3d::laplacian_loss");
        auto out = op.call(v, row_start, nbr);
        ctx->save_for_backward({std::get<1>(out)});
        ctx->saved_data["dtype"] = (int64_t)v.scalar_type();
        ctx->mark_non_differentiable({std::get<1>(out)});
        ctx->set_materialize_grads(false);
        return {std::get<0>(out), std::get<1>(out)};
    }
};

std::tuple<Tensor, Tensor> laplacian_loss_autograd(const Tensor &v, const Tensor &row_start, const Tensor &nbr)
{
    auto out = LaplacianLossFn::apply(v, row_start, nbr);
    return {out[0], out[1]};
}

struct FlattenLossFn : public LossWithGradFn<FlattenLossFn> {
    static constexpr int kInputs = 5;
    static variable_list forward(AutogradContext *ctx, const Tensor &v, const Tensor &quads, const Tensor &inc_start,
                                 const Tensor &inc, double eps)
    {
        at::AutoDispatchBelowADInplaceOrView guard;
        static auto op = find_op<std::tuple<Tensor, Tensor>(const Tensor &, const Tensor &, const Tensor &, const Tensor &,
                                                            double)>("sv3d::flatten_loss");
        auto out = op.call(v, quads, inc_start, inc, eps);
        ctx->save_for_backward({std::get<1>(out)});
        ctx->saved_data["dtype"] = (int64_t)v.scalar_type();
        ctx->mark_non_differentiable({std::get<1>(out)});
        ctx->set_materialize_grads(false);
        return {std::get<0>(out), std::get<1>(out)};
    }
};

std::tuple<Tensor, Tensor> flatten_loss_autograd(const Tensor &v, const Tensor &quads, const Tensor &inc_start,
                                                 const Tensor &inc, double eps)
{
    auto out = FlattenLossFn::apply(v, quads, inc_start, inc, eps);
    return {out[0], out[1]};
}

using SpmvSig = Tensor(const Tensor &, const Tensor &, const Tensor &, const Tensor &, const Tensor &, const Tensor &,
                       const Tensor &);

struct MeshSpmvFn : public torch::autograd::Function<MeshSpmvFn> {
    static Tensor forward(AutogradContext *ctx, const Tensor &x, const Tensor &values, const Tensor &row_start,
                          const Tensor &col, const Tensor &values_t, const Tensor &row_start_t, const Tensor &col_t)
    {
        at::AutoDispatchBelowADInplaceOrView guard;
        static auto op = find_op<SpmvSig>("sv3d::mesh_spmv");
        ctx->save_for_backward({values, row_start, col, values_t, row_start_t, col_t});
        ctx->saved_data["dtype"] = (int64_t)x.scalar_type();
        return op.call(x, values, row_start, col, values_t, row_start_t, col_t);
    }

    static variable_list backward(AutogradContext *ctx, variable_list grads)
    {
        static auto op = find_op<SpmvSig>("sv3d::mesh_spmv");
        const auto s = ctx->get_saved_variables();
        // x -> A x is linear: the gradient is the transposed operator applied to the incoming one
        // (through the dispatcher, so that it is differentiable in turn)
        Tensor g = op.call(grads[0], s[3], s[4], s[5], s[0], s[1], s[2]);
        return {cast_like(g, (at::ScalarType)ctx->saved_data["dtype"].toInt()), Tensor(), Tensor(), Tensor(), Tensor(),
                Tensor(), Tensor()};
    }
};

Tensor mesh_spmv_autograd(const Tensor &x, const Tensor &values, const Tensor &row_start, const Tensor &col,
                          const Tensor &values_t, const Tensor &row_start_t, const Tensor &col_t)
{
    return MeshSpmvFn::apply(x, values, row_start, col, values_t, row_start_t, col_t);
}

struct TextureSampleFn : public torch::autograd::Function<TextureSampleFn> {
    static Tensor forward(AutogradContext *ctx, const Tensor &images, const Tensor &flow)
    {
        TORCH_CHECK(!images.requires_grad(), "sv3d::texture_sample has no gradient for the images");
        at::AutoDispatchBelowADInplaceOrView guard;
        static auto op = find_op<Tensor(const Tensor &, const Tensor &)>("sv3d::texture_sample");
        ctx->save_for_backward({images, flow});
        return op.call(images, flow);
    }

    static variable_list backward(AutogradContext *ctx, variable_list grads)
    {
        static auto op = find_op<Tensor(const Tensor &, const Tensor &, const Tensor &)>("sv3d::texture_sample_backward");
        const auto s = ctx->get_saved_variables();
        Tensor g = op.call(s[0], s[1], grads[0]);
        return {Tensor(), cast_like(g, s[1].scalar_type())};
    }
};

Tensor texture_sample_autograd(const Tensor &images, const Tensor &flow) { return TextureSampleFn::apply(images, flow); }

}  // namespace

TORCH_LIBRARY(sv3d, m)
{
    m.def("soft_rasterize(Tensor face_vertices, Tensor? textures, int[] icfg, float[] fcfg) -> (Tensor, Tensor, Tensor)");
    m.def("soft_rasterize_backward(Tensor grad_rgba, Tensor? textures, Tensor face_rec, Tensor pixel_saved, "
          "int[] shape, int[] icfg, float[] fcfg, bool need_textures) -> (Tensor, Tensor)");
    m.def("render(Tensor vertices, Tensor cams, Tensor faces, Tensor? textures, float[] mesh_cfg, int[] icfg, "
          "float[] fcfg) -> (Tensor, Tensor, Tensor)");
    m.def("render_backward(Tensor grad_rgba, Tensor vertices, Tensor cams, Tensor faces, Tensor? textures, "
          "Tensor face_rec, Tensor pixel_saved, float[] mesh_cfg, int[] icfg, float[] fcfg, bool need_textures) "
          "-> (Tensor, Tensor, Tensor)");
    m.def("chamfer_nn(Tensor xyz1, Tensor xyz2) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("chamfer_nn_backward(Tensor xyz1, Tensor xyz2, Tensor? grad_dist1, Tensor? grad_dist2, Tensor idx1, "
          "Tensor idx2, bool need1, bool need2) -> (Tensor, Tensor)");
    m.def("face_gather(Tensor values, Tensor faces, float[] scale, float[] shift) -> Tensor");
    m.def("face_scatter(Tensor grad_out, Tensor faces, int num_vertices, float[] scale) -> Tensor");
    m.def("face_gather_n(Tensor values, Tensor faces) -> Tensor");
    m.def("face_scatter_n(Tensor grad_out, Tensor faces, int num_vertices) -> Tensor");
    m.def("project_gather(Tensor vertices, Tensor cams, Tensor faces, float[] mesh_cfg) -> Tensor");
    m.def("project_scatter(Tensor grad_out, Tensor vertices, Tensor cams, Tensor faces, float[] mesh_cfg) -> (Tensor, Tensor)");
    m.def("laplacian_loss(Tensor vertices, Tensor row_start, Tensor nbr) -> (Tensor, Tensor)");
    m.def("flatten_loss(Tensor vertices, Tensor quads, Tensor inc_start, Tensor inc, float eps) -> (Tensor, Tensor)");
    m.def("mesh_spmv(Tensor x, Tensor values, Tensor row_start, Tensor col, Tensor values_t, Tensor row_start_t, "
          "Tensor col_t) -> Tensor");
    m.def("texture_sample(Tensor images, Tensor flow) -> Tensor");
    m.def("texture_sample_backward(Tensor images, Tensor flow, Tensor grad_out) -> Tensor");
    m.def("distance_transform(Tensor masks, bool barrier, float k) -> Tensor");
}

TORCH_LIBRARY_IMPL(sv3d, CUDA, m)
{
    m.impl("soft_rasterize", soft_rasterize_gpu);
    m.impl("soft_rasterize_backward", soft_rasterize_backward_gpu);
    m.impl("render", render_gpu);
    m.impl("render_backward", render_backward_gpu);
    m.impl("chamfer_nn", chamfer_nn_gpu);
    m.impl("chamfer_nn_backward", chamfer_nn_backward_gpu);
    m.impl("face_gather", face_gather_gpu);
    m.impl("face_scatter", face_scatter_gpu);
    m.impl("face_gather_n", face_gather_n_gpu);
    m.impl("face_scatter_n", face_scatter_n_gpu);
    m.impl("project_gather", project_gather_gpu);
    m.impl("project_scatter", project_scatter_gpu);
    m.impl("laplacian_loss", laplacian_loss_gpu);
    m.impl("flatten_loss", flatten_loss_gpu);
    m.impl("mesh_spmv", mesh_spmv_gpu);
    m.impl("texture_sample", texture_sample_gpu);
    m.impl("texture_sample_backward", texture_sample_backward_gpu);
    m.impl("distance_transform", distance_transform_gpu);
}

TORCH_LIBRARY_IMPL(sv3d, Meta, m)
{
    m.impl("soft_rasterize", soft_rasterize_meta);
    m.impl("soft_rasterize_backward", soft_rasterize_backward_meta);
    m.impl("render", render_meta);
    m.impl("render_backward", render_backward_meta);
    m.impl("chamfer_nn", chamfer_nn_meta);
    m.impl("chamfer_nn_backward", chamfer_nn_backward_meta);
    m.impl("face_gather", face_gather_meta);
    m.impl("face_scatter", face_scatter_meta);
    m.impl("face_gather_n", face_gather_n_meta);
    m.impl("face_scatter_n", face_scatter_n_meta);
    m.impl("project_gather", project_gather_meta);
    m.impl("project_scatter", project_scatter_meta);
    m.impl("laplacian_loss", laplacian_loss_meta);
    m.impl("flatten_loss", flatten_loss_meta);
    m.impl("mesh_spmv", mesh_spmv_meta);
    m.impl("texture_sample", texture_sample_meta);
    m.impl("texture_sample_backward", texture_sample_backward_meta);
    m.impl("distance_transform", distance_transform_meta);
}

TORCH_LIBRARY_IMPL(sv3d, Autograd, m)
{
    m.impl("soft_rasterize", soft_rasterize_autograd);
    m.impl("render", render_autograd);
    m.impl("chamfer_nn", chamfer_nn_autograd);
    m.impl("face_gather", face_gather_autograd);
    m.impl("face_gather_n", face_gather_n_autograd);
    m.impl("project_gather", project_gather_autograd);
    m.impl("laplacian_loss", laplacian_loss_autograd);
    m.impl("flatten_loss", flatten_loss_autograd);
    m.impl("mesh_spmv", mesh_spmv_autograd);
    m.impl("texture_sample", texture_sample_autograd);
}
