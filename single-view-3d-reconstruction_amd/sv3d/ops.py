"""PyTorch-ROCm custom ops over the libsv3d_hip.so C ABI.

Ops (namespace `sv3d`), all CUDA(HIP)-only -- a CPU tensor raises, there is no fallback:

  soft_rasterize / soft_rasterize_backward   the third-party `soft_rasterize` the reference
                                             reaches from nnutils/softras.py:64,75,77
  chamfer_nn / chamfer_nn_backward           the third-party ChamferDistance called at
                                             nnutils/loss_utils.py:85-86
  face_gather / face_scatter                 sr.Mesh's `vertices[faces]` + affine pre-stage
  project_gather / project_scatter           camera + flip + gather of nnutils/softras.py:58-61
  render / render_backward                   the whole renderer call of softras.py:57-77 from
                                             the object-space mesh (camera folded into setup)

torch is used for device memory, streams and autograd plumbing only.
"""
import ctypes
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import _lib as L


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t):
    """The current stream of the tensor's device as a raw handle."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(t.device.index))
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _f32c(t):
    """Detached fp32 contiguous view (a copy only when the tensor is not that already)."""
    t = t.detach()
    if t.dtype is not torch.float32:
        t = t.to(torch.float32)
    return t if t.is_contiguous() else t.contiguous()


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _on(device):
    """Device guard for allocations and launches; free when `device` is the current one."""
    return _NO_GUARD if device.index == torch.cuda.current_device() else torch.cuda.device(device)


def _params(fv, raster_size, aa, texture_mode, texture_res, double_side, background,
            near, far, eps, sigma, dist_eps, gamma):
    return L.make_params(fv.shape[0], fv.shape[1], raster_size, aa, texture_mode, texture_res,
                         double_side, background, near, far, eps, sigma, dist_eps, gamma)


def _channels(texture_mode):
    """Planes of the rendered image: colour channels + alpha."""
    return 9 if texture_mode == L.SV3D_TEX_VERTEX8 else 4


def _saved_per_pixel(texture_mode):
    return 1 if texture_mode == L.SV3D_TEX_NONE else (12 if texture_mode == L.SV3D_TEX_VERTEX8 else 6)


def _texture_shape(B, F, texture_mode, texture_res):
    if texture_mode == L.SV3D_TEX_SURFACE:
        return (B, F, texture_res * texture_res, 3)
    return (B, F, 3, 8) if texture_mode == L.SV3D_TEX_VERTEX8 else (B, F, 3, 3)


def _check_raster_inputs(fv, tex, texture_mode, texture_res):
    if tex is not None and tex.device != fv.device:
        raise RuntimeError("textures are on %s, face_vertices on %s" % (tex.device, fv.device))
    if fv.dim() != 4 or fv.shape[2:] != (3, 3):
        raise ValueError("face_vertices must be [B,F,3,3], got %s" % (tuple(fv.shape),))
    if texture_mode == L.SV3D_TEX_NONE:
        return
    if tex is None:
        raise ValueError("textures required for texture_mode %d" % texture_mode)
    _check_textures(tex, fv.shape[0], fv.shape[1], texture_mode, texture_res)


def _check_textures(tex, B, F, texture_mode, texture_res):
    """Textures are per image, or one set per group of images: a batch TB that divides B, image b
    reading set b % TB (the order of `torch.cat([textures] * num_cams)`, main.py:415)."""
    want = _texture_shape(B, F, texture_mode, texture_res)
    if tex is None or tex.dim() != 4 or tuple(tex.shape[1:]) != want[1:] or tex.shape[0] < 1 \
            or B % tex.shape[0] != 0:
        raise ValueError("textures must be %s (or a leading size that divides %d), got %s"
                         % (want, B, None if tex is None else tuple(tex.shape)))


def _with_texture_batch(p, tex):
    """Sets the struct's texture batch for this call (0 = one set per image)."""
    p.texture_batch = 0 if tex is None or tex.shape[0] == p.batch else tex.shape[0]
    return p


def _grad_textures_buffer(tex, B):
    """Per-image texture gradients [B, ...] for the kernel to fill."""
    return torch.empty((B,) + tuple(tex.shape[1:]), dtype=torch.float32, device=tex.device)


def _fold_grad_textures(gtex, tex):
    """Sums the per-image gradients over the images that shared a texture set."""
    if gtex.shape[0] == tex.shape[0]:
        return gtex
    return gtex.view((gtex.shape[0] // tex.shape[0],) + tuple(tex.shape)).sum(0)


# ---- soft rasterizer -----------------------------------------------------------------
@torch.library.custom_op("sv3d::soft_rasterize", mutates_args=(), device_types="cuda")
def soft_rasterize_op(face_vertices: Tensor, textures: Optional[Tensor], raster_size: int, aa: int,
                      texture_mode: int, texture_res: int, double_side: bool,
                      background: List[float], near: float, far: float, eps: float, sigma: float,
                      dist_eps: float, gamma: float) -> Tuple[Tensor, Tensor, Tensor]:
    _check_raster_inputs(face_vertices, textures, texture_mode, texture_res)
    fv = _f32c(face_vertices)
    tex = _f32c(textures) if texture_mode != L.SV3D_TEX_NONE else None
    p = _with_texture_batch(_params(fv, raster_size, aa, texture_mode, texture_res, double_side, background,
                                    near, far, eps, sigma, dist_eps, gamma), tex)
    rec_b, saved_b, _ = L.raster_sizes(p)
    H = raster_size // aa
    with _on(fv.device):
        rgba = torch.empty((fv.shape[0], _channels(texture_mode), H, H), dtype=torch.float32, device=fv.device)
        rec = torch.empty(rec_b // 4, dtype=torch.float32, device=fv.device)
        saved = torch.empty(saved_b // 4, dtype=torch.float32, device=fv.device)
        L.check(L.lib().sv3d_raster_forward(ctypes.byref(p), _ptr(fv), _ptr(tex), _ptr(rgba),
                                            _ptr(rec), _ptr(saved), _stream(fv)),
                "sv3d_raster_forward")
    return rgba, rec, saved


@soft_rasterize_op.register_fake
def _(face_vertices, textures, raster_size, aa, texture_mode, texture_res, double_side,
      background, near, far, eps, sigma, dist_eps, gamma):
    B, F = face_vertices.shape[0], face_vertices.shape[1]
    H = raster_size // aa
    per = _saved_per_pixel(texture_mode)
    rec_b = L.raster_sizes(L.make_params(B, F, raster_size, aa, texture_mode, texture_res, double_side,
                                         background, near, far, eps, sigma, dist_eps, gamma))[0]
    return (face_vertices.new_empty((B, _channels(texture_mode), H, H), dtype=torch.float32),
            face_vertices.new_empty((rec_b // 4,), dtype=torch.float32),
            face_vertices.new_empty((B * raster_size * raster_size * per,), dtype=torch.float32))


@torch.library.custom_op("sv3d::soft_rasterize_backward", mutates_args=(), device_types="cuda")
def soft_rasterize_backward_op(grad_rgba: Tensor, textures: Optional[Tensor], face_rec: Tensor,
                               pixel_saved: Tensor, batch: int, num_faces: int, raster_size: int,
                               aa: int, texture_mode: int, texture_res: int, double_side: bool,
                               background: List[float], near: float, far: float, eps: float,
                               sigma: float, dist_eps: float, gamma: float,
                               need_textures: bool) -> Tuple[Tensor, Tensor]:
    g = _f32c(grad_rgba)
    tex = _f32c(textures) if texture_mode != L.SV3D_TEX_NONE else None
    p = _with_texture_batch(L.make_params(batch, num_faces, raster_size, aa, texture_mode, texture_res,
                                          double_side, background, near, far, eps, sigma, dist_eps, gamma), tex)
    _, _, ws_b = L.raster_sizes(p)
    with _on(g.device):
        gfv = torch.empty((batch, num_faces, 3, 3), dtype=torch.float32, device=g.device)
        gtex = _grad_textures_buffer(tex, batch) if (tex is not None and need_textures) else \
            torch.empty(0, dtype=torch.float32, device=g.device)
        ws = torch.empty(ws_b // 4, dtype=torch.float32, device=g.device)
        L.check(L.lib().sv3d_raster_backward(ctypes.byref(p), _ptr(tex), _ptr(face_rec),
                                             _ptr(pixel_saved), _ptr(g), _ptr(gfv), _ptr(gtex),
                                             _ptr(ws), ctypes.c_size_t(ws_b), _stream(g)),
                "sv3d_raster_backward")
        if tex is not None and need_textures:
            gtex = _fold_grad_textures(gtex, tex)
    return gfv, gtex


@soft_rasterize_backward_op.register_fake
def _(grad_rgba, textures, face_rec, pixel_saved, batch, num_faces, raster_size, aa, texture_mode,
      texture_res, double_side, background, near, far, eps, sigma, dist_eps, gamma, need_textures):
    gfv = grad_rgba.new_empty((batch, num_faces, 3, 3), dtype=torch.float32)
    gtex = torch.empty_like(textures, dtype=torch.float32) if (textures is not None and need_textures) \
        else grad_rgba.new_empty((0,), dtype=torch.float32)
    return gfv, gtex


def _raster_setup_context(ctx, inputs, output):
    fv, tex = inputs[0], inputs[1]
    _, rec, saved = output
    ctx.scalars = tuple(inputs[2:])
    ctx.shape = (fv.shape[0], fv.shape[1])
    ctx.has_tex = tex is not None and inputs[4] != L.SV3D_TEX_NONE
    ctx.tex_dtype = tex.dtype if tex is not None else None
    ctx.fv_dtype = fv.dtype
    if ctx.has_tex:
        ctx.save_for_backward(tex, rec, saved)
    else:
        ctx.save_for_backward(rec, saved)


def _raster_backward(ctx, grad_rgba, grad_rec, grad_saved):
    if ctx.has_tex:
        tex, rec, saved = ctx.saved_tensors
    else:
        tex = None
        rec, saved = ctx.saved_tensors
    need_tex = ctx.has_tex and ctx.needs_input_grad[1]
    gfv, gtex = soft_rasterize_backward_op(grad_rgba, tex, rec, saved, ctx.shape[0], ctx.shape[1],
                                           *ctx.scalars, need_tex)
    gfv = gfv.to(ctx.fv_dtype)
    gtex = gtex.to(ctx.tex_dtype) if need_tex else None
    return (gfv, gtex) + (None,) * len(ctx.scalars)


soft_rasterize_op.register_autograd(_raster_backward, setup_context=_raster_setup_context)


def soft_rasterize(face_vertices, textures=None, image_size=256, *, anti_aliasing=False,
                   texture_type="surface", background_color=(0.0, 0.0, 0.0), near=1.0, far=100.0,
                   eps=1e-3, sigma_val=1e-5, dist_eps=1e-10, gamma_val=1e-4, fill_back=True):
    """Soft rasterization of `face_vertices` [B,F,3,3] -> RGBA [B,4,image_size,image_size].

    `textures`: None renders the silhouette only (RGB = background); [B,F,R*R,3] for
    texture_type 'surface'; [B,F,3,3] per-corner colours for 'vertex'; [B,F,3,8] eight scalar
    per-corner channels for 'vertex8' (output [B,9,H,W]: the eight channels, then alpha).  With
    `anti_aliasing` fragments are evaluated at twice the resolution and average-pooled
    inside the kernel.
    """
    aa = 2 if anti_aliasing else 1
    if textures is None:
        mode, res = L.SV3D_TEX_NONE, 1
    elif texture_type == "surface":
        mode = L.SV3D_TEX_SURFACE
        res = int(round(textures.shape[2] ** 0.5))
        if res * res != textures.shape[2]:
            raise ValueError("surface textures need a square texel count, got %d" % textures.shape[2])
    elif texture_type == "vertex":
        mode, res = L.SV3D_TEX_VERTEX, 1
    elif texture_type == "vertex8":
        mode, res = L.SV3D_TEX_VERTEX8, 1
    else:
        raise ValueError("unknown texture_type %r" % (texture_type,))
    if _plain(face_vertices, textures) and face_vertices.is_cuda:
        _check_raster_inputs(face_vertices, textures, mode, res)
        key = (face_vertices.shape[0], face_vertices.shape[1], int(image_size) * aa, aa, mode, res,
               bool(fill_back), tuple(float(c) for c in background_color), float(near), float(far),
               float(eps), float(sigma_val), float(dist_eps), float(gamma_val))
        return _RenderFn.apply(face_vertices, None, None, textures if mode != L.SV3D_TEX_NONE else None,
                               None, key, False)
    rgba, _, _ = soft_rasterize_op(face_vertices, textures, int(image_size) * aa, aa, mode, res,
                                   bool(fill_back), [float(c) for c in background_color],
                                   float(near), float(far), float(eps), float(sigma_val),
                                   float(dist_eps), float(gamma_val))
    return rgba


# ---- chamfer --------------------------------------------------------------------------
def _check_clouds(xyz1, xyz2):
    if xyz1.device != xyz2.device:
        raise RuntimeError("point clouds are on %s and %s" % (xyz1.device, xyz2.device))
    if xyz1.dim() != 3 or xyz2.dim() != 3 or xyz1.shape[2] != 3 or xyz2.shape[2] != 3 \
            or xyz1.shape[0] != xyz2.shape[0]:
        raise ValueError("point clouds must be [B,N,3] and [B,M,3], got %s and %s"
                         % (tuple(xyz1.shape), tuple(xyz2.shape)))
    if xyz1.shape[1] == 0 or xyz2.shape[1] == 0 or xyz1.shape[0] == 0:
        raise ValueError("empty point cloud")


@torch.library.custom_op("sv3d::chamfer_nn", mutates_args=(), device_types="cuda")
def chamfer_nn_op(xyz1: Tensor, xyz2: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    _check_clouds(xyz1, xyz2)
    a, b = _f32c(xyz1), _f32c(xyz2)
    B, N, M = a.shape[0], a.shape[1], b.shape[1]
    with _on(a.device):
        d1 = torch.empty((B, N), dtype=torch.float32, device=a.device)
        d2 = torch.empty((B, M), dtype=torch.float32, device=a.device)
        i1 = torch.empty((B, N), dtype=torch.int32, device=a.device)
        i2 = torch.empty((B, M), dtype=torch.int32, device=a.device)
        L.check(L.lib().sv3d_chamfer_forward(_ptr(a), _ptr(b), B, N, M, _ptr(d1), _ptr(d2),
                                             _ptr(i1), _ptr(i2), _stream(a)), "sv3d_chamfer_forward")
    return d1, d2, i1, i2


@chamfer_nn_op.register_fake
def _(xyz1, xyz2):
    B, N, M = xyz1.shape[0], xyz1.shape[1], xyz2.shape[1]
    return (xyz1.new_empty((B, N), dtype=torch.float32), xyz1.new_empty((B, M), dtype=torch.float32),
            xyz1.new_empty((B, N), dtype=torch.int32), xyz1.new_empty((B, M), dtype=torch.int32))


@torch.library.custom_op("sv3d::chamfer_nn_backward", mutates_args=(), device_types="cuda")
def chamfer_nn_backward_op(xyz1: Tensor, xyz2: Tensor, grad_dist1: Optional[Tensor],
                           grad_dist2: Optional[Tensor], idx1: Tensor, idx2: Tensor,
                           need1: bool, need2: bool) -> Tuple[Tensor, Tensor]:
    a, b = _f32c(xyz1), _f32c(xyz2)
    B, N, M = a.shape[0], a.shape[1], b.shape[1]
    gd1 = _f32c(grad_dist1) if grad_dist1 is not None else None
    gd2 = _f32c(grad_dist2) if grad_dist2 is not None else None
    with _on(a.device):
        g1 = torch.empty_like(a) if need1 else torch.empty(0, dtype=torch.float32, device=a.device)
        g2 = torch.empty_like(b) if need2 else torch.empty(0, dtype=torch.float32, device=a.device)
        L.check(L.lib().sv3d_chamfer_backward(_ptr(a), _ptr(b), B, N, M, _ptr(gd1), _ptr(gd2),
                                              _ptr(idx1), _ptr(idx2), _ptr(g1), _ptr(g2), _stream(a)),
                "sv3d_chamfer_backward")
    return g1, g2


@chamfer_nn_backward_op.register_fake
def _(xyz1, xyz2, grad_dist1, grad_dist2, idx1, idx2, need1, need2):
    e = xyz1.new_empty((0,), dtype=torch.float32)
    return (torch.empty_like(xyz1, dtype=torch.float32) if need1 else e,
            torch.empty_like(xyz2, dtype=torch.float32) if need2 else e)


def _chamfer_setup_context(ctx, inputs, output):
    xyz1, xyz2 = inputs
    _, _, i1, i2 = output
    ctx.save_for_backward(xyz1, xyz2, i1, i2)


def _chamfer_backward(ctx, gd1, gd2, gi1, gi2):
    xyz1, xyz2, i1, i2 = ctx.saved_tensors
    need1, need2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    if not (need1 or need2):
        return None, None
    g1, g2 = chamfer_nn_backward_op(xyz1, xyz2, gd1, gd2, i1, i2, need1, need2)
    return (g1.to(xyz1.dtype) if need1 else None, g2.to(xyz2.dtype) if need2 else None)


chamfer_nn_op.register_autograd(_chamfer_backward, setup_context=_chamfer_setup_context)


def chamfer_nn(xyz1, xyz2):
    """(dist1 [B,N], dist2 [B,M], idx1 [B,N] int32, idx2 [B,M] int32); squared distances."""
    if _plain(xyz1, xyz2) and xyz1.is_cuda and xyz2.is_cuda:
        return _ChamferFn.apply(xyz1, xyz2)
    return chamfer_nn_op(xyz1, xyz2)


# ---- mesh gather / scatter ------------------------------------------------------------
@torch.library.custom_op("sv3d::face_gather", mutates_args=(), device_types="cuda")
def face_gather_op(values: Tensor, faces: Tensor, scale: List[float], shift: List[float]) -> Tensor:
    if values.dim() != 3 or values.shape[2] != 3:
        raise ValueError("values must be [B,V,3], got %s" % (tuple(values.shape),))
    if faces.dim() != 3 or faces.shape[2] != 3 or faces.dtype != torch.int32 \
            or faces.shape[0] not in (1, values.shape[0]):
        raise ValueError("faces must be int32 [B or 1,F,3], got %s %s" % (faces.dtype, tuple(faces.shape)))
    if faces.device != values.device:
        raise RuntimeError("faces are on %s, values on %s" % (faces.device, values.device))
    v = _f32c(values)
    f = faces.contiguous()
    B, V, F = v.shape[0], v.shape[1], f.shape[1]
    with _on(v.device):
        out = torch.empty((B, F, 3, 3), dtype=torch.float32, device=v.device)
        L.check(L.lib().sv3d_face_gather(_ptr(v), _ptr(f), B, V, F, f.shape[0], L.float3(scale),
                                         L.float3(shift), _ptr(out), _stream(v)), "sv3d_face_gather")
    return out


@face_gather_op.register_fake
def _(values, faces, scale, shift):
    return values.new_empty((values.shape[0], faces.shape[1], 3, 3), dtype=torch.float32)


@torch.library.custom_op("sv3d::face_scatter", mutates_args=(), device_types="cuda")
def face_scatter_op(grad_out: Tensor, faces: Tensor, num_vertices: int, scale: List[float]) -> Tensor:
    g = _f32c(grad_out)
    B, F = g.shape[0], g.shape[1]
    with _on(g.device):
        out = torch.empty((B, num_vertices, 3), dtype=torch.float32, device=g.device)
        L.check(L.lib().sv3d_face_scatter(_ptr(g), _ptr(faces), B, num_vertices, F, faces.shape[0],
                                          L.float3(scale), _ptr(out), _stream(g)), "sv3d_face_scatter")
    return out


@face_scatter_op.register_fake
def _(grad_out, faces, num_vertices, scale):
    return grad_out.new_empty((grad_out.shape[0], num_vertices, 3), dtype=torch.float32)


def _gather_setup_context(ctx, inputs, output):
    values, faces, scale, _ = inputs
    ctx.save_for_backward(faces)
    ctx.num_vertices = values.shape[1]
    ctx.scale = list(scale)
    ctx.dtype = values.dtype


def _gather_backward(ctx, grad_out):
    (faces,) = ctx.saved_tensors
    g = face_scatter_op(grad_out, faces, ctx.num_vertices, ctx.scale)
    return g.to(ctx.dtype), None, None, None


face_gather_op.register_autograd(_gather_backward, setup_context=_gather_setup_context)


def face_gather(values, faces, scale=(1.0, 1.0, 1.0), shift=(0.0, 0.0, 0.0)):
    """out[b,f,k,:] = values[b, faces[b or 0, f, k], :] * scale + shift -> [B,F,3,3]."""
    if faces.dtype != torch.int32:
        faces = faces.to(torch.int32)
    return face_gather_op(values, faces.contiguous(), [float(s) for s in scale],
                          [float(s) for s in shift])


# ---- camera + gather, and the renderer call from the object-space mesh -----------------
def _check_same_device(ref, **others):
    """A tensor on another device (or on the host) would reach the kernels as a foreign pointer,
    which faults the GPU instead of raising: refuse it here."""
    for name, t in others.items():
        if t is not None and t.device != ref.device:
            raise RuntimeError("%s is on %s, expected %s" % (name, t.device, ref.device))


def _check_mesh(vertices, cams, faces):
    _check_same_device(vertices, cams=cams, faces=faces)
    if vertices.dim() != 3 or vertices.shape[2] != 3:
        raise ValueError("vertices must be [B,V,3], got %s" % (tuple(vertices.shape),))
    if cams.dim() != 2 or tuple(cams.shape) != (vertices.shape[0], 7):
        raise ValueError("cams must be [B,7], got %s" % (tuple(cams.shape),))
    if faces.dim() != 3 or faces.shape[2] != 3 or faces.dtype != torch.int32 \
            or faces.shape[0] not in (1, vertices.shape[0]):
        raise ValueError("faces must be int32 [B or 1,F,3], got %s %s" % (faces.dtype, tuple(faces.shape)))


@torch.library.custom_op("sv3d::project_gather", mutates_args=(), device_types="cuda")
def project_gather_op(vertices: Tensor, cams: Tensor, faces: Tensor, offset_z: float,
                      scale: List[float], shift: List[float]) -> Tensor:
    _check_mesh(vertices, cams, faces)
    v, c, f = _f32c(vertices), _f32c(cams), faces.contiguous()
    B, F = v.shape[0], f.shape[1]
    with _on(v.device):
        out = torch.empty((B, F, 3, 3), dtype=torch.float32, device=v.device)
        m = L.make_mesh_input(v, c, f, offset_z, scale, shift)
        L.check(L.lib().sv3d_project_gather(ctypes.byref(m), B, F, _ptr(out), _stream(v)),
                "sv3d_project_gather")
    return out


@project_gather_op.register_fake
def _(vertices, cams, faces, offset_z, scale, shift):
    return vertices.new_empty((vertices.shape[0], faces.shape[1], 3, 3), dtype=torch.float32)


@torch.library.custom_op("sv3d::project_scatter", mutates_args=(), device_types="cuda")
def project_scatter_op(grad_out: Tensor, vertices: Tensor, cams: Tensor, faces: Tensor,
                       offset_z: float, scale: List[float]) -> Tuple[Tensor, Tensor]:
    g, v, c = _f32c(grad_out), _f32c(vertices), _f32c(cams)
    B, F = g.shape[0], g.shape[1]
    with _on(g.device):
        gv = torch.empty_like(v)
        gc = torch.empty_like(c)
        m = L.make_mesh_input(v, c, faces, offset_z, scale, (0.0, 0.0, 0.0))
        L.check(L.lib().sv3d_project_scatter(ctypes.byref(m), B, F, _ptr(g), _ptr(gv), _ptr(gc),
                                             _stream(g)), "sv3d_project_scatter")
    return gv, gc


@project_scatter_op.register_fake
def _(grad_out, vertices, cams, faces, offset_z, scale):
    return (torch.empty_like(vertices, dtype=torch.float32), torch.empty_like(cams, dtype=torch.float32))


def _project_setup_context(ctx, inputs, output):
    vertices, cams, faces, offset_z, scale, _ = inputs
    ctx.save_for_backward(vertices, cams, faces)
    ctx.offset_z, ctx.scale = offset_z, list(scale)


def _project_backward(ctx, grad_out):
    vertices, cams, faces = ctx.saved_tensors
    gv, gc = project_scatter_op(grad_out, vertices, cams, faces, ctx.offset_z, ctx.scale)
    return gv.to(vertices.dtype), gc.to(cams.dtype), None, None, None, None


project_gather_op.register_autograd(_project_backward, setup_context=_project_setup_context)


def _faces_i32(faces):
    if faces.dtype != torch.int32:
        faces = faces.to(torch.int32)
    return faces.contiguous()


def project_gather(vertices, cams, faces, offset_z=0.0, scale=(1.0, 1.0, 1.0), shift=(0.0, 0.0, 0.0)):
    """face_vertices [B,F,3,3] of the mesh seen through the weak-perspective cameras `cams`
    [B,7] = (s, tx, ty, quaternion): out = scale * (s * rot(q, X) + (tx, ty, offset_z)) + shift
    gathered per face corner (nnutils/softras.py:58-61 + `vertices[faces]`); differentiable
    with respect to vertices and cams."""
    return project_gather_op(vertices, cams, _faces_i32(faces), float(offset_z),
                             [float(s) for s in scale], [float(s) for s in shift])


@torch.library.custom_op("sv3d::render", mutates_args=(), device_types="cuda")
def render_op(vertices: Tensor, cams: Tensor, faces: Tensor, textures: Optional[Tensor],
              offset_z: float, scale: List[float], shift: List[float], raster_size: int, aa: int,
              texture_mode: int, texture_res: int, double_side: bool, background: List[float],
              near: float, far: float, eps: float, sigma: float, dist_eps: float,
              gamma: float) -> Tuple[Tensor, Tensor, Tensor]:
    _check_mesh(vertices, cams, faces)
    v, c, f = _f32c(vertices), _f32c(cams), faces.contiguous()
    B, F = v.shape[0], f.shape[1]
    if texture_mode != L.SV3D_TEX_NONE:
        _check_textures(textures, B, F, texture_mode, texture_res)
        _check_same_device(vertices, textures=textures)
    tex = _f32c(textures) if texture_mode != L.SV3D_TEX_NONE else None
    p = _with_texture_batch(L.make_params(B, F, raster_size, aa, texture_mode, texture_res, double_side,
                                          background, near, far, eps, sigma, dist_eps, gamma), tex)
    rec_b, saved_b, _ = L.raster_sizes(p)
    H = raster_size // aa
    with _on(v.device):
        rgba = torch.empty((B, _channels(texture_mode), H, H), dtype=torch.float32, device=v.device)
        rec = torch.empty(rec_b // 4, dtype=torch.float32, device=v.device)
        saved = torch.empty(saved_b // 4, dtype=torch.float32, device=v.device)
        m = L.make_mesh_input(v, c, f, offset_z, scale, shift)
        L.check(L.lib().sv3d_render_forward(ctypes.byref(p), ctypes.byref(m), _ptr(tex), None, _ptr(rgba),
                                            _ptr(rec), _ptr(saved), _stream(v)), "sv3d_render_forward")
    return rgba, rec, saved


@render_op.register_fake
def _(vertices, cams, faces, textures, offset_z, scale, shift, raster_size, aa, texture_mode, texture_res,
      double_side, background, near, far, eps, sigma, dist_eps, gamma):
    B, F = vertices.shape[0], faces.shape[1]
    H = raster_size // aa
    per = _saved_per_pixel(texture_mode)
    rec_b = L.raster_sizes(L.make_params(B, F, raster_size, aa, texture_mode, texture_res, double_side,
                                         background, near, far, eps, sigma, dist_eps, gamma))[0]
    return (vertices.new_empty((B, _channels(texture_mode), H, H), dtype=torch.float32),
            vertices.new_empty((rec_b // 4,), dtype=torch.float32),
            vertices.new_empty((B * raster_size * raster_size * per,), dtype=torch.float32))


@torch.library.custom_op("sv3d::render_backward", mutates_args=(), device_types="cuda")
def render_backward_op(grad_rgba: Tensor, vertices: Tensor, cams: Tensor, faces: Tensor,
                       textures: Optional[Tensor], face_rec: Tensor, pixel_saved: Tensor,
                       offset_z: float, scale: List[float], raster_size: int, aa: int,
                       texture_mode: int, texture_res: int, double_side: bool,
                       background: List[float], near: float, far: float, eps: float, sigma: float,
                       dist_eps: float, gamma: float,
                       need_textures: bool) -> Tuple[Tensor, Tensor, Tensor]:
    g, v, c = _f32c(grad_rgba), _f32c(vertices), _f32c(cams)
    tex = _f32c(textures) if texture_mode != L.SV3D_TEX_NONE else None
    B, F = v.shape[0], faces.shape[1]
    p = _with_texture_batch(L.make_params(B, F, raster_size, aa, texture_mode, texture_res, double_side,
                                          background, near, far, eps, sigma, dist_eps, gamma), tex)
    _, _, ws_b = L.raster_sizes(p)
    with _on(g.device):
        gfv = torch.empty((B, F, 3, 3), dtype=torch.float32, device=g.device)
        gtex = _grad_textures_buffer(tex, B) if (tex is not None and need_textures) else \
            torch.empty(0, dtype=torch.float32, device=g.device)
        ws = torch.empty(ws_b // 4, dtype=torch.float32, device=g.device)
        gv, gc = torch.empty_like(v), torch.empty_like(c)
        st = _stream(g)
        L.check(L.lib().sv3d_raster_backward(ctypes.byref(p), _ptr(tex), _ptr(face_rec), _ptr(pixel_saved),
                                             _ptr(g), _ptr(gfv), _ptr(gtex), _ptr(ws), ctypes.c_size_t(ws_b),
                                             st), "sv3d_raster_backward")
        m = L.make_mesh_input(v, c, faces, offset_z, scale, (0.0, 0.0, 0.0))
        L.check(L.lib().sv3d_project_scatter(ctypes.byref(m), B, F, _ptr(gfv), _ptr(gv), _ptr(gc), st),
                "sv3d_project_scatter")
        if tex is not None and need_textures:
            gtex = _fold_grad_textures(gtex, tex)
    return gv, gc, gtex


@render_backward_op.register_fake
def _(grad_rgba, vertices, cams, faces, textures, face_rec, pixel_saved, offset_z, scale, raster_size, aa,
      texture_mode, texture_res, double_side, background, near, far, eps, sigma, dist_eps, gamma,
      need_textures):
    gtex = torch.empty_like(textures, dtype=torch.float32) if (textures is not None and need_textures) \
        else grad_rgba.new_empty((0,), dtype=torch.float32)
    return (torch.empty_like(vertices, dtype=torch.float32), torch.empty_like(cams, dtype=torch.float32), gtex)


def _render_setup_context(ctx, inputs, output):
    vertices, cams, faces, tex = inputs[0], inputs[1], inputs[2], inputs[3]
    _, rec, saved = output
    ctx.offset_z, ctx.scale = inputs[4], list(inputs[5])
    ctx.scalars = tuple(inputs[7:])
    ctx.has_tex = tex is not None and inputs[9] != L.SV3D_TEX_NONE
    if ctx.has_tex:
        ctx.save_for_backward(vertices, cams, faces, rec, saved, tex)
    else:
        ctx.save_for_backward(vertices, cams, faces, rec, saved)


def _render_backward(ctx, grad_rgba, grad_rec, grad_saved):
    vertices, cams, faces, rec, saved = ctx.saved_tensors[:5]
    tex = ctx.saved_tensors[5] if ctx.has_tex else None
    need_tex = ctx.has_tex and ctx.needs_input_grad[3]
    gv, gc, gtex = render_backward_op(grad_rgba, vertices, cams, faces, tex, rec, saved, ctx.offset_z,
                                      ctx.scale, *ctx.scalars, need_tex)
    return (gv.to(vertices.dtype), gc.to(cams.dtype), None, gtex.to(tex.dtype) if need_tex else None) + \
        (None,) * 15


render_op.register_autograd(_render_backward, setup_context=_render_setup_context)


def render(vertices, cams, faces, textures=None, image_size=256, *, offset_z=0.0,
           scale=(1.0, 1.0, 1.0), shift=(0.0, 0.0, 0.0), anti_aliasing=False, texture_type="surface",
           background_color=(0.0, 0.0, 0.0), near=1.0, far=100.0, eps=1e-3, sigma_val=1e-5,
           dist_eps=1e-10, gamma_val=1e-4, fill_back=True, alpha_only=False):
    """`project_gather` + `soft_rasterize` as one op: RGBA [B,4,image_size,image_size] of the mesh
    (vertices [B,V,3] object space, faces [B or 1,F,3]) seen through `cams` [B,7].  `textures` are
    per face as for `soft_rasterize` ([B,F,R*R,3] or per-corner [B,F,3,3]).  Differentiable with
    respect to vertices, cams and textures."""
    aa = 2 if anti_aliasing else 1
    if textures is None:
        mode, res = L.SV3D_TEX_NONE, 1
    elif texture_type == "surface":
        mode = L.SV3D_TEX_SURFACE
        res = int(round(textures.shape[2] ** 0.5))
        if res * res != textures.shape[2]:
            raise ValueError("surface textures need a square texel count, got %d" % textures.shape[2])
    elif texture_type == "vertex":
        mode, res = L.SV3D_TEX_VERTEX, 1
    elif texture_type == "vertex8":
        mode, res = L.SV3D_TEX_VERTEX8, 1
    else:
        raise ValueError("unknown texture_type %r" % (texture_type,))
    f32 = _faces_i32(faces)
    if _plain(vertices, cams, textures) and vertices.is_cuda:
        _check_mesh(vertices, cams, f32)
        B, F = vertices.shape[0], f32.shape[1]
        if mode != L.SV3D_TEX_NONE:
            _check_textures(textures, B, F, mode, res)
            _check_same_device(vertices, textures=textures)
        key = (B, F, int(image_size) * aa, aa, mode, res, bool(fill_back),
               tuple(float(c) for c in background_color), float(near), float(far), float(eps),
               float(sigma_val), float(dist_eps), float(gamma_val))
        mesh_args = (float(offset_z), tuple(float(v) for v in scale), tuple(float(v) for v in shift))
        return _RenderFn.apply(vertices, cams, f32, textures if mode != L.SV3D_TEX_NONE else None,
                               mesh_args, key, bool(alpha_only))
    rgba, _, _ = render_op(vertices, cams, f32, textures, float(offset_z),
                           [float(s) for s in scale], [float(s) for s in shift], int(image_size) * aa, aa,
                           mode, res, bool(fill_back), [float(c) for c in background_color], float(near),
                           float(far), float(eps), float(sigma_val), float(dist_eps), float(gamma_val))
    return rgba[:, -1] if alpha_only else rgba


# ---- eager fast path ------------------------------------------------------------------------
# torch.library's Python dispatch costs 100-150 us per op call (schema handling, functionalisation
# checks), more than the kernels themselves at small batches.  Outside of tracing the public
# functions therefore go through plain autograd.Functions over the same C ABI calls; the registered
# ops above stay the route for torch.compile / fake-tensor tracing.  Both routes run the same
# kernels with the same arguments.
def _plain(*tensors):
    """True when every tensor is an ordinary eager tensor (no fake / functional wrapper)."""
    if torch.compiler.is_compiling():
        return False
    for t in tensors:
        if t is not None and type(t) is not Tensor and type(t) is not torch.nn.Parameter:
            return False
    return True


_PARAM_CACHE = {}


def _cached_params(key, tex, alpha_only):
    """(params struct, face_rec_bytes, pixel_saved_bytes) per configuration tuple, texture batch and
    output form.  A cached struct is never written to again: forward and backward calls of
    different graphs (autograd runs backward on its own threads) may hold the same one."""
    tb = 0 if tex is None or tex.shape[0] == key[0] else tex.shape[0]
    full = (key, tb, alpha_only)
    hit = _PARAM_CACHE.get(full)
    if hit is None:
        p = L.make_params(*key, alpha_only=alpha_only)
        p.texture_batch = tb
        rec_b, saved_b, _ = L.raster_sizes(p)
        hit = (p, rec_b, saved_b)
        if len(_PARAM_CACHE) > 256:
            _PARAM_CACHE.clear()
        _PARAM_CACHE[full] = hit
    return hit


def _raster_backward_call(p, g, tex, rec, saved, need_tex, B, F):
    gfv = torch.empty((B, F, 3, 3), dtype=torch.float32, device=g.device)
    gtex = _grad_textures_buffer(tex, B) if (tex is not None and need_tex) else None
    L.check(L.lib().sv3d_raster_backward(ctypes.byref(p), _ptr(tex), _ptr(rec),
                                         _ptr(saved), _ptr(g), _ptr(gfv), _ptr(gtex), None, ctypes.c_size_t(0),
                                         _stream(g)), "sv3d_raster_backward")
    return gfv, (_fold_grad_textures(gtex, tex) if gtex is not None else None)


class _RenderFn(torch.autograd.Function):
    """`render_op` / `soft_rasterize_op` without the dispatcher: mesh given -> sv3d_render_forward,
    else face_vertices -> sv3d_raster_forward."""

    @staticmethod
    def forward(ctx, geom, cams, faces, textures, mesh_args, key, alpha_only):
        # geom: vertices [B,V,3] when cams is given, else face_vertices [B,F,3,3]
        x = _f32c(geom)
        tex = _f32c(textures) if key[4] != L.SV3D_TEX_NONE else None
        # alpha_only without colours: the kernels write (and later read the gradient of) the alpha
        # plane alone; with colours the full image is rendered and its last plane handed out
        plane_only = alpha_only and key[4] == L.SV3D_TEX_NONE
        p, rec_b, saved_b = _cached_params(key, tex, plane_only)
        B, S, aa = key[0], key[2], key[3]
        dev = x.device
        with _on(dev):
            rgba = torch.empty((B, 1 if plane_only else _channels(key[4]), S // aa, S // aa),
                               dtype=torch.float32, device=dev)
            rec = torch.empty(rec_b // 4, dtype=torch.float32, device=dev)
            saved = torch.empty(saved_b // 4, dtype=torch.float32, device=dev)
            if cams is not None:
                c = _f32c(cams)
                m = L.make_mesh_input(x, c, faces, *mesh_args)
                L.check(L.lib().sv3d_render_forward(ctypes.byref(p), ctypes.byref(m), _ptr(tex), None,
                                                    _ptr(rgba), _ptr(rec), _ptr(saved), _stream(x)),
                        "sv3d_render_forward")
                ctx.save_for_backward(x, c, faces, rec, saved, tex)
            else:
                L.check(L.lib().sv3d_raster_forward(ctypes.byref(p), _ptr(x), _ptr(tex), _ptr(rgba), _ptr(rec),
                                                    _ptr(saved), _stream(x)), "sv3d_raster_forward")
                ctx.save_for_backward(rec, saved, tex)
        ctx.key, ctx.mesh_args, ctx.has_mesh = key, mesh_args, cams is not None
        ctx.alpha_only, ctx.plane_only = alpha_only, plane_only
        ctx.dtypes = (geom.dtype, cams.dtype if cams is not None else None,
                      textures.dtype if textures is not None else None)
        # alpha_only: the silhouette plane itself is the output, so that autograd hands its gradient
        # straight back instead of building a zero-filled RGBA gradient around it
        return rgba[:, -1] if alpha_only else rgba

    @staticmethod
    def backward(ctx, grad_rgba):
        key = ctx.key
        tex_saved = ctx.saved_tensors[-1]
        p, _, _ = _cached_params(key, tex_saved, ctx.plane_only)
        if ctx.plane_only:
            g = _f32c(grad_rgba)                # [B,H,W] is the [B,1,H,W] the kernel reads
        elif ctx.alpha_only:
            full = torch.zeros((key[0], _channels(key[4]), key[2] // key[3], key[2] // key[3]),
                               dtype=torch.float32, device=grad_rgba.device)
            full[:, -1].copy_(grad_rgba)
            g = full
        else:
            g = _f32c(grad_rgba)
        need_tex = ctx.needs_input_grad[3]
        with _on(g.device):
            if ctx.has_mesh:
                x, c, faces, rec, saved, tex = ctx.saved_tensors
                gfv, gtex = _raster_backward_call(p, g, tex, rec, saved, need_tex, key[0], key[1])
                gv, gc = torch.empty_like(x), torch.empty_like(c)
                m = L.make_mesh_input(x, c, faces, ctx.mesh_args[0], ctx.mesh_args[1], (0.0, 0.0, 0.0))
                L.check(L.lib().sv3d_project_scatter(ctypes.byref(m), key[0], key[1], _ptr(gfv), _ptr(gv),
                                                     _ptr(gc), _stream(g)), "sv3d_project_scatter")
                ggeom, gcams = gv.to(ctx.dtypes[0]), gc.to(ctx.dtypes[1])
            else:
                rec, saved, tex = ctx.saved_tensors
                gfv, gtex = _raster_backward_call(p, g, tex, rec, saved, need_tex, key[0], key[1])
                ggeom, gcams = gfv.to(ctx.dtypes[0]), None
        return ggeom, gcams, None, (gtex.to(ctx.dtypes[2]) if gtex is not None else None), None, None, None


class _ChamferFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        _check_clouds(xyz1, xyz2)
        a, b = _f32c(xyz1), _f32c(xyz2)
        B, N, M = a.shape[0], a.shape[1], b.shape[1]
        with _on(a.device):
            d1 = torch.empty((B, N), dtype=torch.float32, device=a.device)
            d2 = torch.empty((B, M), dtype=torch.float32, device=a.device)
            i1 = torch.empty((B, N), dtype=torch.int32, device=a.device)
            i2 = torch.empty((B, M), dtype=torch.int32, device=a.device)
            L.check(L.lib().sv3d_chamfer_forward(_ptr(a), _ptr(b), B, N, M, _ptr(d1), _ptr(d2), _ptr(i1),
                                                 _ptr(i2), _stream(a)), "sv3d_chamfer_forward")
        ctx.save_for_backward(a, b, i1, i2)
        ctx.dtypes = (xyz1.dtype, xyz2.dtype)
        ctx.mark_non_differentiable(i1, i2)
        return d1, d2, i1, i2

    @staticmethod
    def backward(ctx, gd1, gd2, gi1, gi2):
        a, b, i1, i2 = ctx.saved_tensors
        need1, need2 = ctx.needs_input_grad
        if not (need1 or need2):
            return None, None
        B, N, M = a.shape[0], a.shape[1], b.shape[1]
        gd1 = _f32c(gd1) if gd1 is not None else None
        gd2 = _f32c(gd2) if gd2 is not None else None
        with _on(a.device):
            g1 = torch.empty_like(a) if need1 else None
            g2 = torch.empty_like(b) if need2 else None
            L.check(L.lib().sv3d_chamfer_backward(_ptr(a), _ptr(b), B, N, M, _ptr(gd1), _ptr(gd2), _ptr(i1),
                                                  _ptr(i2), _ptr(g1), _ptr(g2), _stream(a)),
                    "sv3d_chamfer_backward")
        return (g1.to(ctx.dtypes[0]) if need1 else None, g2.to(ctx.dtypes[1]) if need2 else None)


# ---- per-vertex channel gather (part maps) ---------------------------------------------------
@torch.library.custom_op("sv3d::face_gather_n", mutates_args=(), device_types="cuda")
def face_gather_n_op(values: Tensor, faces: Tensor) -> Tensor:
    if values.dim() != 3 or not 1 <= values.shape[2] <= 8:
        raise ValueError("values must be [B,V,C] with C <= 8, got %s" % (tuple(values.shape),))
    if faces.dim() != 3 or faces.shape[2] != 3 or faces.dtype != torch.int32 \
            or faces.shape[0] not in (1, values.shape[0]):
        raise ValueError("faces must be int32 [B or 1,F,3], got %s %s" % (faces.dtype, tuple(faces.shape)))
    if faces.device != values.device:
        raise RuntimeError("faces are on %s, values on %s" % (faces.device, values.device))
    v, f = _f32c(values), faces.contiguous()
    B, V, C, F = v.shape[0], v.shape[1], v.shape[2], f.shape[1]
    with _on(v.device):
        out = torch.empty((B, F, 3, C), dtype=torch.float32, device=v.device)
        L.check(L.lib().sv3d_face_gather_n(_ptr(v), _ptr(f), B, V, F, f.shape[0], C, _ptr(out), _stream(v)),
                "sv3d_face_gather_n")
    return out


@face_gather_n_op.register_fake
def _(values, faces):
    return values.new_empty((values.shape[0], faces.shape[1], 3, values.shape[2]), dtype=torch.float32)


@torch.library.custom_op("sv3d::face_scatter_n", mutates_args=(), device_types="cuda")
def face_scatter_n_op(grad_out: Tensor, faces: Tensor, num_vertices: int) -> Tensor:
    g = _f32c(grad_out)
    B, F, C = g.shape[0], g.shape[1], g.shape[3]
    with _on(g.device):
        out = torch.empty((B, num_vertices, C), dtype=torch.float32, device=g.device)
        L.check(L.lib().sv3d_face_scatter_n(_ptr(g), _ptr(faces), B, num_vertices, F, faces.shape[0], C,
                                            _ptr(out), _stream(g)), "sv3d_face_scatter_n")
    return out


@face_scatter_n_op.register_fake
def _(grad_out, faces, num_vertices):
    return grad_out.new_empty((grad_out.shape[0], num_vertices, grad_out.shape[3]), dtype=torch.float32)


def _gather_n_setup_context(ctx, inputs, output):
    values, faces = inputs
    ctx.save_for_backward(faces)
    ctx.num_vertices, ctx.dtype = values.shape[1], values.dtype


def _gather_n_backward(ctx, grad_out):
    (faces,) = ctx.saved_tensors
    return face_scatter_n_op(grad_out, faces, ctx.num_vertices).to(ctx.dtype), None


face_gather_n_op.register_autograd(_gather_n_backward, setup_context=_gather_n_setup_context)


def face_gather_n(values, faces):
    """out[b,f,k,:] = values[b, faces[b or 0, f, k], :] for C <= 8 channels: [B,V,C] -> [B,F,3,C]."""
    return face_gather_n_op(values, _faces_i32(faces))
