"""Shared synthetic-input builders for the tests (seeded, CUB-shaped)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EYE_SHIFT = 2.732


def load_mesh(sub):
    m = np.load(os.path.join(GOLDEN, "mesh_sub%d.npz" % sub))
    return torch.from_numpy(m["verts"]).float(), torch.from_numpy(m["faces"]).int()


def random_cams(B, seed=1):
    """[s, tx, ty, quaternion]: s ~ U(0.6,0.9), t ~ U(-0.1,0.1), q random unit."""
    g = torch.Generator().manual_seed(seed)
    cams = torch.empty(B, 7)
    cams[:, 0] = 0.6 + 0.3 * torch.rand(B, generator=g)
    cams[:, 1:3] = 0.2 * torch.rand(B, 2, generator=g) - 0.1
    q = torch.randn(B, 4, generator=g)
    cams[:, 3:] = q / q.norm(dim=1, keepdim=True)
    return cams


def random_vertices(verts, B, seed=0, jitter=0.02):
    g = torch.Generator().manual_seed(seed)
    return verts[None].repeat(B, 1, 1) + jitter * torch.randn(B, verts.shape[0], 3, generator=g)


def camera_space_faces(verts, faces, B, seed=0):
    """Face vertices [B,F,3,3] (fp32) as the rasterizer sees them: projected by seeded weak
    perspective cameras (oracle restatement of geom_utils), y flipped, z shifted by the eye."""
    from oracle import softras_oracle as so
    v = random_vertices(verts, B, seed)
    cams = random_cams(B, seed + 1)
    proj = so.orthographic_proj_withz(v.double(), cams.double(), offset_z=5.0)
    proj = proj * torch.tensor([1.0, -1.0, 1.0], dtype=torch.float64)
    proj[..., 2] += EYE_SHIFT
    fv = so.face_gather(proj, faces[None].expand(B, -1, -1))
    return fv.float().contiguous()


def rel_err(a, b):
    """max |a-b| / max |b| (the tolerance convention used by the parity tests)."""
    a = a.double()
    b = b.double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def elementwise_rel_err(a, b, floor=1e-3, q=0.999):
    """q-quantile of |a-b| / |b| over the entries of b above `floor` times its largest one: the
    element-wise companion of `rel_err`, which small entries cannot hide in."""
    a = a.double().reshape(-1)
    b = b.double().reshape(-1)
    big = b.abs() > floor * b.abs().max()
    if not bool(big.any()):
        return 0.0
    r = ((a - b).abs() / b.abs())[big]
    k = min(r.numel() - 1, int(q * r.numel()))
    return float(torch.sort(r).values[k])


def sliver_faces(fv, ratio=1e4):
    """[..., F] bool: faces whose three projected corners are collinear to fp32 -- longest edge squared
    above `ratio` times the doubled area, so that rounding the corner differences alone moves the
    area by some 1e-3 of itself.  Their edges all but coincide; which of them is "nearest" to a pixel,
    hence which corners the distance gradient goes to, is decided by rounding in any precision, and
    the oracle freezes that choice as fp64 makes it.  Gradient comparisons leave such faces out (and
    count them): one of them, carrying a thousandth of the largest gradient, is enough to reach 1e-5
    of the largest entry."""
    p = fv[..., :2].double()
    e = p[..., [1, 2, 0], :] - p
    longest2 = (e * e).sum(-1).max(-1).values
    area2 = (e[..., 0, 0] * (p[..., 2, 1] - p[..., 0, 1]) - e[..., 0, 1] * (p[..., 2, 0] - p[..., 0, 0])).abs()
    return longest2 > ratio * area2
