"""The `sv3d::*` PyTorch-ROCm operators of the hot path and their Python entry points.

The operators are registered from C++ (csrc/torch_ops.cpp -> libsv3d_torch.so: GPU, Meta and
Autograd kernels over the libsv3d_hip.so C ABI); this module loads that library and normalises
arguments.  All operators are GPU-only -- a CPU tensor fails in the dispatcher, there is no fallback:

  soft_rasterize / soft_rasterize_backward   the third-party `soft_rasterize` the reference
                                             reaches from nnutils/softras.py:64,75,77
  chamfer_nn / chamfer_nn_backward           the third-party ChamferDistance called at
                                             nnutils/loss_utils.py:85-86
  face_gather / face_scatter                 sr.Mesh's `vertices[faces]` + affine pre-stage
  project_gather / project_scatter           camera + flip + gather of nnutils/softras.py:58-61
  render / render_backward                   the whole renderer call of softras.py:57-77 from
                                             the object-space mesh (camera folded into setup)
  face_gather_n / face_scatter_n             per-vertex part maps gathered per face corner

The mesh regularisers, the CSR product, texture sampling and the distance transforms are operators
of the same library; their entry points are in `sv3d.meshreg` and `sv3d.texture`.
"""
import os

import torch

from . import _lib as L

TORCH_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsv3d_torch.so")


def _load_torch_ops():
    if not os.path.exists(TORCH_LIB_PATH):
        raise L.Sv3dError(
            "libsv3d_torch.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C single-view-3d-reconstruction_amd/csrc`; there is no fallback path" % TORCH_LIB_PATH)
    L.lib()                                    # libsv3d_hip.so first: the operator library links against it
    torch.ops.load_library(TORCH_LIB_PATH)
    return torch.ops.sv3d


_ops = _load_torch_ops()


# ---- argument packing --------------------------------------------------------------------------
_MODES = {"surface": L.SV3D_TEX_SURFACE, "vertex": L.SV3D_TEX_VERTEX, "vertex8": L.SV3D_TEX_VERTEX8}


def _texture_mode(textures, texture_type):
    """(SV3D_TEX_*, texels per face side) of a textures argument."""
    if textures is None:
        return L.SV3D_TEX_NONE, 1
    mode = _MODES.get(texture_type)
    if mode is None:
        raise ValueError("unknown texture_type %r" % (texture_type,))
    if mode != L.SV3D_TEX_SURFACE:
        return mode, 1
    if textures.dim() != 4:
        raise ValueError("surface textures must be [B,F,R*R,3], got %s" % (tuple(textures.shape),))
    res = int(round(textures.shape[2] ** 0.5))
    if res * res != textures.shape[2]:
        raise ValueError("surface textures need a square texel count, got %d" % textures.shape[2])
    return mode, res


# `sv3d_raster_params.backward_enumeration` of the calls made through this module: the library's
# own choice unless a test or a tuning run asks for one form (the C ABI itself keeps no such state)
_backward_enumeration = L.SV3D_BWD_AUTO


def set_backward_enumeration(mode):
    """"auto" (the library picks per launch), "boxes" or "intervals" for the rasterizer backward of
    every later call through this module; returns the previous setting."""
    global _backward_enumeration
    names = {"auto": L.SV3D_BWD_AUTO, "boxes": L.SV3D_BWD_BOXES, "intervals": L.SV3D_BWD_INTERVALS}
    prev = [k for k, v in names.items() if v == _backward_enumeration][0]
    _backward_enumeration = names[mode]
    return prev


def raster_config(image_size, anti_aliasing, mode, res, fill_back, alpha_only, background_color, near, far,
                  eps, sigma_val, dist_eps, gamma_val):
    """(icfg, fcfg): the integer and float halves of `sv3d_raster_params` as the operators take them
    (raster_size, aa, texture_mode, texture_res, double_side, alpha_only, backward_enumeration;
    background r g b, near, far, eps, sigma, dist_eps, gamma)."""
    aa = 2 if anti_aliasing else 1
    return ([int(image_size) * aa, aa, mode, res, int(bool(fill_back)), int(bool(alpha_only)), _backward_enumeration],
            [float(background_color[0]), float(background_color[1]), float(background_color[2]), float(near),
             float(far), float(eps), float(sigma_val), float(dist_eps), float(gamma_val)])


def _faces_i32(faces):
    if faces.dtype != torch.int32:
        faces = faces.to(torch.int32)
    return faces.contiguous()


# ---- soft rasterizer ---------------------------------------------------------------------------
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
    mode, res = _texture_mode(textures, texture_type)
    icfg, fcfg = raster_config(image_size, anti_aliasing, mode, res, fill_back, False, background_color, near,
                               far, eps, sigma_val, dist_eps, gamma_val)
    return _ops.soft_rasterize(face_vertices, textures, icfg, fcfg)[0]


# ---- chamfer -----------------------------------------------------------------------------------
def chamfer_nn(xyz1, xyz2):
    """(dist1 [B,N], dist2 [B,M], idx1 [B,N] int32, idx2 [B,M] int32); squared distances."""
    return _ops.chamfer_nn(xyz1, xyz2)


# ---- mesh gather / scatter ---------------------------------------------------------------------
def face_gather(values, faces, scale=(1.0, 1.0, 1.0), shift=(0.0, 0.0, 0.0)):
    """out[b,f,k,:] = values[b, faces[b or 0, f, k], :] * scale + shift -> [B,F,3,3]."""
    return _ops.face_gather(values, _faces_i32(faces), [float(s) for s in scale], [float(s) for s in shift])


def face_gather_n(values, faces):
    """out[b,f,k,:] = values[b, faces[b or 0, f, k], :] for C <= 8 channels: [B,V,C] -> [B,F,3,C]."""
    return _ops.face_gather_n(values, _faces_i32(faces))


def _mesh_cfg(offset_z, scale, shift):
    return [float(offset_z), float(scale[0]), float(scale[1]), float(scale[2]),
            float(shift[0]), float(shift[1]), float(shift[2])]


def project_gather(vertices, cams, faces, offset_z=0.0, scale=(1.0, 1.0, 1.0), shift=(0.0, 0.0, 0.0)):
    """face_vertices [B,F,3,3] of the mesh seen through the weak-perspective cameras `cams`
    [B,7] = (s, tx, ty, quaternion): out = scale * (s * rot(q, X) + (tx, ty, offset_z)) + shift
    gathered per face corner (nnutils/softras.py:58-61 + `vertices[faces]`); differentiable
    with respect to vertices and cams."""
    return _ops.project_gather(vertices, cams, _faces_i32(faces), _mesh_cfg(offset_z, scale, shift))


# ---- the renderer call from the object-space mesh ----------------------------------------------
def render(vertices, cams, faces, textures=None, image_size=256, *, offset_z=0.0,
           scale=(1.0, 1.0, 1.0), shift=(0.0, 0.0, 0.0), anti_aliasing=False, texture_type="surface",
           background_color=(0.0, 0.0, 0.0), near=1.0, far=100.0, eps=1e-3, sigma_val=1e-5,
           dist_eps=1e-10, gamma_val=1e-4, fill_back=True, alpha_only=False):
    """`project_gather` + `soft_rasterize` as one op: RGBA [B,4,image_size,image_size] of the mesh
    (vertices [B,V,3] object space, faces [B or 1,F,3]) seen through `cams` [B,7].  `textures` are
    per face as for `soft_rasterize` ([B,F,R*R,3] or per-corner [B,F,3,3]).  Differentiable with
    respect to vertices, cams and textures.  `alpha_only` returns the silhouette [B,H,W]; without
    textures the kernels then write, and read the gradient of, that one plane only."""
    mode, res = _texture_mode(textures, texture_type)
    icfg, fcfg = raster_config(image_size, anti_aliasing, mode, res, fill_back, alpha_only, background_color,
                               near, far, eps, sigma_val, dist_eps, gamma_val)
    return render_packed(vertices, cams, faces, textures, _mesh_cfg(offset_z, scale, shift), icfg, fcfg)


def render_packed(vertices, cams, faces, textures, mesh_cfg, icfg, fcfg):
    """`render` with the configuration already packed (`raster_config`, `_mesh_cfg`): callers that
    render with fixed settings pack them once instead of on every call."""
    rgba = _ops.render(vertices, cams, _faces_i32(faces), textures, mesh_cfg, icfg, fcfg)[0]
    mode, alpha_only = icfg[2], icfg[5]
    if not alpha_only:
        return rgba
    # without textures the image IS the alpha plane: a view, whose backward is a view too (selecting
    # a plane would make autograd build a zero-filled image around the incoming gradient)
    return rgba.squeeze(1) if mode == L.SV3D_TEX_NONE else rgba[:, -1]
