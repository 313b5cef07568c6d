"""CPU oracle for the renderer call of nnutils/softras.py -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  It restates, in torch (any float dtype, autograd-capable):

  * the weak-perspective projection of nnutils/geom_utils.py:49-80,108-161
    (`hamilton_product`, `quat_rotate`, `orthographic_proj`,
    `orthographic_proj_withz`) -- PINNED by tests/golden/proj_vectors.npz, which
    holds the reference functions' own outputs on seeded inputs;
  * `SoftNeuralRenderer.forward` of nnutils/softras.py:57-77 (projection with
    offset_z 5, y flip, the three modes: mask / lit RGB / unlit vertex colours);
  * the SoftRas python glue around the kernels (Mesh face gather, surface
    Lighting, look_at + orthogonal Transform, 2x supersampling + average pool),
    PARITY UNPINNED like raster_oracle.py (third-party `soft_renderer`, absent).
"""
import torch

from . import raster_oracle


# ---- nnutils/geom_utils.py ------------------------------------------------------------
def hamilton_product(qa, qb):
    """geom_utils.py:108-133."""
    a0, a1, a2, a3 = qa.unbind(-1)
    b0, b1, b2, b3 = qb.unbind(-1)
    return torch.stack([
        a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3,
        a0 * b1 + a1 * b0 + a2 * b3 - a3 * b2,
        a0 * b2 - a1 * b3 + a2 * b0 + a3 * b1,
        a0 * b3 + a1 * b2 - a2 * b1 + a3 * b0,
    ], -1)


def quat_rotate(X, q):
    """geom_utils.py:143-161: X_rot = q (0, X) q*."""
    q = q[:, None, :].expand(X.shape[0], X.shape[1], 4)
    qc = torch.cat([q[..., :1], -q[..., 1:]], -1)
    Xq = torch.cat([torch.zeros_like(X[..., :1]), X], -1)
    return hamilton_product(q, hamilton_product(Xq, qc))[..., 1:4]


def orthographic_proj(X, cam):
    """geom_utils.py:49-60."""
    rot = quat_rotate(X, cam[:, -4:])
    return cam[:, 0].reshape(-1, 1, 1) * rot[..., :2] + cam[:, 1:3].reshape(-1, 1, 2)


def orthographic_proj_withz(X, cam, offset_z=0.0):
    """geom_utils.py:63-80."""
    rot = quat_rotate(X, cam[:, -4:])
    proj = cam[:, 0].reshape(-1, 1, 1) * rot
    xy = proj[..., :2] + cam[:, 1:3].reshape(-1, 1, 2)
    z = proj[..., 2:3] + offset_z
    return torch.cat([xy, z], -1)


# ---- SoftRas python glue (third-party, restated) -------------------------------------
EYE_Z = -2.732          # softras.py:36


def face_gather(vertices, faces):
    """Mesh.face_vertices: vertices[b, faces[b]] -> [B,F,3,3]."""
    B = vertices.shape[0]
    bidx = torch.arange(B)[:, None, None]
    return vertices[bidx, faces.long()]


def surface_light(face_vertices, ambient, directional, direction=(0.0, 1.0, 0.0)):
    """Per-face light [B,F,3]: ambient + directional * relu(n . dir), white light."""
    v10 = face_vertices[:, :, 0] - face_vertices[:, :, 1]
    v12 = face_vertices[:, :, 2] - face_vertices[:, :, 1]
    n = torch.nn.functional.normalize(torch.cross(v12, v10, dim=-1), eps=1e-6, dim=-1)
    d = torch.tensor(direction, dtype=face_vertices.dtype)
    cos = torch.relu((n * d).sum(-1))
    return (ambient + directional * cos)[..., None].expand(-1, -1, 3)


def render(vertices, faces, cams, textures=None, *, img_size=256, texture_type="surface",
           anti_aliasing=True, sigma=1e-5, gamma=1e-4, dist_eps=1e-10,
           ambient=0.8, directional=0.5, light_dir=(0.0, 1.0, 0.0), background=(0.0, 0.0, 0.0),
           offset_z=5.0, stage_dtype=None):
    """softras.py:57-77.  Returns mask [B,H,W] when textures is None else RGB [B,3,H,W].

    `stage_dtype` (e.g. torch.float32): the camera stage -- projection, flip, look_at shift -- is
    evaluated in that type and its result handed to the rasterizer in the inputs' type.  With fp64
    inputs this isolates what storing the projected vertices in single precision costs, the floor
    under any fp32 implementation of the call, whatever its rasterizer does."""
    dt = vertices.dtype
    if stage_dtype is not None:
        vertices, cams = vertices.to(stage_dtype), cams.to(stage_dtype)
    verts = orthographic_proj_withz(vertices, cams, offset_z=offset_z)          # :58
    vs = verts * torch.tensor([1.0, -1.0, 1.0], dtype=verts.dtype)                # :59-60
    S = img_size * (2 if anti_aliasing else 1)
    fv_world = face_gather(vs, faces)
    shift = torch.tensor([0.0, 0.0, -EYE_Z], dtype=verts.dtype)
    fv = (fv_world + shift).to(dt)                                                # look_at, orthogonal
    fv_world = fv_world.to(dt)
    kw = dict(sigma=sigma, gamma=gamma, dist_eps=dist_eps, background=background)
    if textures is None:                                                          # :62-65
        rgba, _ = raster_oracle.soft_rasterize(fv, None, S, **kw)
        out = rgba[:, 3:4]
    elif texture_type == "vertex":                                                # :69-75, no lighting
        ft = face_gather(textures, faces)
        rgba, _ = raster_oracle.soft_rasterize(fv, ft, S, texture_mode="vertex", **kw)
        out = rgba[:, :3]
    else:                                                                         # :77
        light = surface_light(fv_world, ambient, directional, light_dir)
        rgba, _ = raster_oracle.soft_rasterize(fv, textures * light[:, :, None, :], S,
                                               texture_mode="surface", **kw)
        out = rgba[:, :3]
    if anti_aliasing:
        out = torch.nn.functional.avg_pool2d(out, kernel_size=2, stride=2)
    return out[:, 0] if textures is None else out
