"""CPU tests (no GPU): the oracle against the golden vectors captured from the reference,
self-consistency of the restated formulas, host logic, and the C-ABI library's symbols."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from helpers import GOLDEN, camera_space_faces, load_mesh
from oracle import chamfer_oracle, raster_oracle, softras_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- projection: pinned by the reference's own outputs -----------------------------------
def test_projection_oracle_matches_reference_vectors():
    g = np.load(os.path.join(GOLDEN, "proj_vectors.npz"))
    X, cams = torch.from_numpy(g["X"]), torch.from_numpy(g["cams"])
    np.testing.assert_allclose(softras_oracle.orthographic_proj_withz(X, cams, 5.0).numpy(),
                               g["withz_off5"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(softras_oracle.orthographic_proj_withz(X, cams).numpy(),
                               g["withz_off0"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(softras_oracle.orthographic_proj(X, cams).numpy(), g["ortho"],
                               rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(softras_oracle.quat_rotate(X, cams[:, 3:]).numpy(), g["quat_rotate"],
                               rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(softras_oracle.hamilton_product(torch.from_numpy(g["qa"]),
                                                               torch.from_numpy(g["qb"])).numpy(),
                               g["hamilton"], rtol=1e-6, atol=1e-6)


def test_product_geom_utils_matches_reference_vectors():
    from nnutils import geom_utils
    g = np.load(os.path.join(GOLDEN, "proj_vectors.npz"))
    X, cams = torch.from_numpy(g["X"]), torch.from_numpy(g["cams"])
    np.testing.assert_allclose(geom_utils.orthographic_proj_withz(X, cams, offset_z=5.).numpy(),
                               g["withz_off5"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(geom_utils.orthographic_proj(X, cams).numpy(), g["ortho"],
                               rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(geom_utils.quat_rotate(X, cams[:, 3:]).numpy(), g["quat_rotate"],
                               rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(geom_utils.hamilton_product(torch.from_numpy(g["qa"]),
                                                           torch.from_numpy(g["qb"])).numpy(),
                               g["hamilton"], rtol=1e-6, atol=1e-6)


def test_mesh_fixtures_shapes():
    for sub, (nv, nf) in {1: (42, 80), 2: (162, 320), 3: (642, 1280), 4: (2562, 5120)}.items():
        v, f = load_mesh(sub)
        assert v.shape == (nv, 3) and f.shape == (nf, 3)
        assert int(f.max()) == nv - 1 and int(f.min()) == 0
    s = np.load(os.path.join(GOLDEN, "mesh_sub3_sym.npz"))
    assert s["counts"].tolist() == [32, 305, 32, 624]       # mesh_net.py:119,640 hard-codes 624


# ---- chamfer oracle -----------------------------------------------------------------------
def test_chamfer_c_port_matches_numpy_restatement():
    rng = np.random.default_rng(5)
    a = rng.standard_normal((3, 157, 3)).astype(np.float32)
    b = rng.standard_normal((3, 211, 3)).astype(np.float32)
    c = chamfer_oracle.forward_c(a, b)
    n = chamfer_oracle.forward_np(a, b)
    for x, y in zip(c, n):
        np.testing.assert_array_equal(x, y)
    # squared distances, lowest index on ties, exact zeros
    lat = np.stack(np.meshgrid(np.arange(4), np.arange(4), np.arange(2), indexing="ij"), -1)
    lat = lat.reshape(1, -1, 3).astype(np.float32)
    q = np.array([[[0.5, 0.5, 0.5], [1.0, 2.0, 1.0], [3.5, 3.0, 0.0]]], np.float32)
    d1, d2, i1, i2 = chamfer_oracle.forward_c(q, lat)
    assert d1[0].tolist() == [0.75, 0.0, 0.25]
    assert i1[0].tolist() == [0, int(np.where((lat[0] == [1, 2, 1]).all(1))[0][0]),
                              int(np.where((lat[0] == [3, 3, 0]).all(1))[0][0])]
    dup = np.concatenate([lat, lat], 1)
    _, _, j1, _ = chamfer_oracle.forward_c(q, dup)
    np.testing.assert_array_equal(j1, i1)                   # duplicates: the first copy wins


def test_chamfer_backward_oracle_matches_autograd_of_dense_formula():
    rng = np.random.default_rng(6)
    a = rng.standard_normal((2, 33, 3)).astype(np.float32)
    b = rng.standard_normal((2, 41, 3)).astype(np.float32)
    gd1 = rng.standard_normal((2, 33)).astype(np.float32)
    gd2 = rng.standard_normal((2, 41)).astype(np.float32)
    _, _, i1, i2 = chamfer_oracle.forward_c(a, b)
    g1, g2 = chamfer_oracle.backward_c(a, b, gd1, gd2, i1, i2)
    ta = torch.tensor(a, dtype=torch.float64, requires_grad=True)
    tb = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    d = ((ta[:, :, None] - tb[:, None]) ** 2).sum(-1)
    loss = (d.min(2).values * torch.tensor(gd1)).sum() + (d.min(1).values * torch.tensor(gd2)).sum()
    loss.backward()
    np.testing.assert_allclose(g1, ta.grad.numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(g2, tb.grad.numpy(), rtol=1e-9, atol=1e-9)


# ---- raster oracle: self-consistency of the restated formulas -----------------------------
def _segment_d2(p, a, b):
    e = b - a
    t = ((p - a) * e).sum(-1) / (e * e).sum(-1).clamp(min=1e-300)
    q = a + t.clamp(0, 1)[..., None] * e
    return ((p - q) ** 2).sum(-1)


def test_barycentric_distance_equals_distance_to_boundary():
    """The `t - w` algebra with the region / obtuse-corner selection is the squared distance
    from the pixel centre to the triangle's boundary, + inside / - outside."""
    verts, faces = load_mesh(2)
    fv = camera_space_faces(verts, faces, 1, seed=3)[0].double()
    S, sigma = 32, 1e-2                                         # wide band so most pairs are active
    out = raster_oracle.rasterize_image(fv, None, S, sigma=sigma, texture_mode="none")
    xp, yp = raster_oracle.pixel_centres(S)
    p = torch.stack([xp, yp], -1)[:, None, :]
    a, b, c = fv[None, :, 0, :2], fv[None, :, 1, :2], fv[None, :, 2, :2]
    d2 = torch.minimum(torch.minimum(_segment_d2(p, a, b), _segment_d2(p, b, c)), _segment_d2(p, c, a))
    def cr(u, v):
        return u[..., 0] * v[..., 1] - u[..., 1] * v[..., 0]
    s0, s1, s2 = cr(b - a, p - a), cr(c - b, p - b), cr(a - c, p - c)
    inside = ((s0 > 0) & (s1 > 0) & (s2 > 0)) | ((s0 < 0) & (s1 < 0) & (s2 < 0))
    import math
    thr = math.log(1 / 1e-10 - 1) * sigma
    margin = math.sqrt(thr)
    keep = ~((p[..., 0] > fv[None, :, :, 0].max(-1).values + margin) | (p[..., 0] < fv[None, :, :, 0].min(-1).values - margin) |
             (p[..., 1] > fv[None, :, :, 1].max(-1).values + margin) | (p[..., 1] < fv[None, :, :, 1].min(-1).values - margin))
    act = keep & (inside | (d2 < thr))
    D = torch.where(act, torch.sigmoid(torch.where(inside, d2, -d2) / sigma), torch.zeros_like(d2))
    alpha = 1 - torch.prod(1 - D, 1)
    np.testing.assert_allclose(out["rgba"][3].reshape(-1).numpy(), alpha.numpy(), rtol=1e-9, atol=1e-12)


def test_raster_oracle_analytic_single_triangle():
    # right triangle; pixel centres of a 4x4 image are at +-0.25, +-0.75
    fv = torch.tensor([[[-0.5, -0.5, 5.0], [0.5, -0.5, 5.0], [-0.5, 0.5, 5.0]]], dtype=torch.float64)
    sigma = 1e-2
    out = raster_oracle.rasterize_image(fv, torch.tensor([[[0.2, 0.4, 0.6]]], dtype=torch.float64), 4,
                                        sigma=sigma, texture_mode="surface", background=(1.0, 0.0, 0.0))
    a = out["rgba"][3]
    sg = lambda x: 1 / (1 + np.exp(-x))
    # row 0 is the TOP: pixel (row 2, col 1) = (-0.25, -0.25): inside, nearest boundary at 0.25
    assert abs(a[2, 1].item() - sg(0.25 ** 2 / sigma)) < 1e-12
    # pixel (row 0, col 3) = (0.75, 0.75): outside; the hypotenuse x + y = 0 is at distance 1.5/sqrt(2)
    d2 = (1.5 / np.sqrt(2)) ** 2
    assert abs(a[0, 3].item() - sg(-d2 / sigma)) < 1e-12
    # pixel (row 3, col 3) = (0.75,-0.75): nearest point is the vertex (0.5,-0.5)
    assert abs(a[3, 3].item() - sg(-(2 * 0.25 ** 2) / sigma)) < 1e-12
    # any pixel touched by the face shows the face colour with no background share
    np.testing.assert_allclose(out["rgba"][:3, 2, 1].numpy(), [0.2, 0.4, 0.6], atol=1e-12)


def test_raster_oracle_background_only_and_depth_order():
    far_tri = torch.tensor([[[-0.9, -0.9, 9.0], [0.9, -0.9, 9.0], [0.0, 0.9, 9.0]]], dtype=torch.float64)
    near_tri = far_tri.clone()
    near_tri[..., 2] = 4.0
    fv = torch.cat([far_tri, near_tri], 0)
    tex = torch.tensor([[[1.0, 0.0, 0.0]], [[0.0, 0.0, 1.0]]], dtype=torch.float64)
    out = raster_oracle.rasterize_image(fv, tex, 8, texture_mode="surface", background=(0.0, 1.0, 0.0))
    centre = out["rgba"][:, 4, 4]
    np.testing.assert_allclose(centre.numpy(), [0.0, 0.0, 1.0, 1.0], atol=1e-9)     # nearer face wins
    corner = out["rgba"][:, 0, 0]
    np.testing.assert_allclose(corner.numpy(), [0.0, 1.0, 0.0, 0.0], atol=1e-12)    # background only


def test_raster_oracle_gradient_finite_difference():
    verts, faces = load_mesh(1)
    fv = camera_space_faces(verts, faces, 1, seed=7)[0].double()
    g = torch.rand(4, 16, 16, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    tex = torch.rand(80, 3, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    kw = dict(sigma=3e-3, gamma=3e-2, texture_mode="vertex")

    def loss(x):
        return (raster_oracle.rasterize_image(x, tex, 16, **kw)["rgba"] * g).sum()

    x = fv.clone().requires_grad_(True)
    loss(x).backward()
    worst = 0.0
    for idx in [(3, 0, 0), (10, 1, 1), (20, 2, 2), (5, 1, 2), (40, 0, 1), (66, 2, 0)]:
        h = 1e-6
        a, b = fv.clone(), fv.clone()
        a[idx] += h
        b[idx] -= h
        fd = float((loss(a) - loss(b)) / (2 * h))
        worst = max(worst, abs(fd - float(x.grad[idx])) / max(abs(fd), 1e-6))
    assert worst < 1e-4, worst


def test_count_active_pairs_matches_dense_cull():
    verts, faces = load_mesh(2)
    fv = camera_space_faces(verts, faces, 2, seed=11)
    dense = [raster_oracle.rasterize_image(fv[b].double(), None, 32, sigma=1e-4, texture_mode="none")["n_active"]
             for b in range(2)]
    assert raster_oracle.count_active_pairs(fv, 32, 1e-4, 1e-10) == dense


# ---- C ABI: the library loads and exports every declared symbol ---------------------------
def test_c_abi_exports_every_declared_symbol():
    from sv3d import _lib
    header = open(os.path.join(ROOT, "include", "sv3d_hip.h")).read()
    declared = set(re.findall(r"\b(sv3d_[a-z_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.lib().sv3d_abi_version() == 3
    assert b"workspace" in _lib.lib().sv3d_strerror(-3)


def test_c_abi_argument_validation_without_gpu():
    from sv3d import _lib
    p = _lib.make_params(8, 1280, 256, 1, 0, 1, True, (0, 0, 0), 1, 100, 1e-3, 1e-4, 1e-10, 1e-4)
    rec, saved, ws = _lib.raster_sizes(p)
    # per face a 128-byte record and an 8-byte pixel box, per group of 64 faces an 8-byte box; a
    # silhouette launch also holds per tile (8 images x 16 x 16) a count and 128 16-bit face ids, and
    # per face a 48-byte oriented record (16-byte aligned behind the counts)
    plain = 8 * 1280 * (32 * 4 + 8) + 8 * 20 * 8
    assert rec == plain + 8 * 256 * 4 + 16 + 8 * 1280 * 48 + 8 * 256 * 128 * 2
    assert saved == 8 * 256 * 256 * 4 and ws == 0
    # 2048 tiles of a mesh with five faces per tile: the form whose four waves share a tile's hits
    assert _lib.lib().sv3d_raster_forward_kernel_name(ctypes.byref(p), 0) == b"raster_forward_tiled_shared_kernel"
    assert _lib.lib().sv3d_raster_forward_kernel_name(ctypes.byref(p), 1) == b"raster_forward_kernel"
    p64 = _lib.make_params(64, 1280, 256, 1, 0, 1, True, (0, 0, 0), 1, 100, 1e-3, 1e-4, 1e-10, 1e-4)
    assert _lib.lib().sv3d_raster_forward_kernel_name(ctypes.byref(p64), 0) == b"raster_forward_tiled_kernel"
    p.texture_mode = 1
    p.texture_res = 6
    # a colour launch of this size keeps the plain layout (lists from 4096 tiles on)
    assert _lib.raster_sizes(p) == (plain, 8 * 256 * 256 * 6 * 4, 0)       # no backward scratch in any mode
    assert _lib.lib().sv3d_raster_forward_kernel_name(ctypes.byref(p), 0) == b"raster_forward_kernel"
    bad = _lib.make_params(0, 1280, 256, 1, 0, 1, True, (0, 0, 0), 1, 100, 1e-3, 1e-4, 1e-10, 1e-4)
    with pytest.raises(_lib.Sv3dError):
        _lib.raster_sizes(bad)
    bad = _lib.make_params(1, 10, 255, 2, 0, 1, True, (0, 0, 0), 1, 100, 1e-3, 1e-4, 1e-10, 1e-4)
    with pytest.raises(_lib.Sv3dError):
        _lib.raster_sizes(bad)
    L = _lib.lib()
    assert L.sv3d_chamfer_forward(None, None, 1, 1, 1, None, None, None, None, None) == -1
    assert L.sv3d_raster_forward(ctypes.byref(p), None, None, None, None, None, None) == -1


def test_ops_refuse_cpu_tensors():
    import sv3d
    with pytest.raises(NotImplementedError):
        sv3d.chamfer_nn(torch.zeros(1, 4, 3), torch.zeros(1, 4, 3))
    with pytest.raises(NotImplementedError):
        sv3d.soft_rasterize(torch.zeros(1, 4, 3, 3), None, 16)


def test_torch_operator_library_registers_every_operator():
    """libsv3d_torch.so loads next to libsv3d_hip.so and defines the sv3d::* operators with Meta
    kernels (shapes without a GPU); CPU tensors are refused by the dispatcher, not computed."""
    import torch
    import sv3d  # noqa: F401  (loads both libraries)
    names = ("soft_rasterize", "soft_rasterize_backward", "render", "render_backward", "chamfer_nn",
             "chamfer_nn_backward", "face_gather", "face_scatter", "face_gather_n", "face_scatter_n",
             "project_gather", "project_scatter", "laplacian_loss", "flatten_loss", "mesh_spmv", "texture_sample",
             "texture_sample_backward", "distance_transform")
    for n in names:
        assert hasattr(torch.ops.sv3d, n), n
    meta = dict(device="meta")
    v, cams = torch.empty(3, 642, 3, **meta), torch.empty(3, 7, **meta)
    f = torch.empty(1, 1280, 3, dtype=torch.int32, **meta)
    rgba, rec, saved = torch.ops.sv3d.render(v, cams, f, None, [5.0, 1, -1, 1, 0, 0, 2.732], [512, 2, 0, 1, 1, 0],
                                             [0, 0, 0, 1, 100, 1e-3, 1e-5, 1e-10, 1e-4])
    assert rgba.shape == (3, 4, 256, 256) and saved.numel() == 3 * 512 * 512
    d1, d2, i1, i2 = torch.ops.sv3d.chamfer_nn(torch.empty(2, 10, 3, **meta), torch.empty(2, 20, 3, **meta))
    assert d2.shape == (2, 20) and i1.dtype == torch.int32
    loss, grad = torch.ops.sv3d.flatten_loss(v, torch.empty(410, 4, dtype=torch.int32, **meta),
                                             torch.empty(643, dtype=torch.int32, **meta),
                                             torch.empty(1640, dtype=torch.int32, **meta), 1e-6)
    assert loss.shape == (3,) and grad.shape == v.shape
    tex = torch.ops.sv3d.texture_sample(torch.empty(3, 3, 64, 64, **meta), torch.empty(3, 1280 * 36, 2, **meta))
    assert tex.shape == (3, 1280 * 36, 3)
    assert torch.ops.sv3d.distance_transform(torch.empty(3, 64, 48, **meta), True, 50.0).shape == (3, 64, 48)
    with pytest.raises(NotImplementedError):
        torch.ops.sv3d.chamfer_nn(torch.zeros(1, 3, 3), torch.zeros(1, 4, 3))
    with pytest.raises(NotImplementedError):
        torch.ops.sv3d.laplacian_loss(torch.zeros(1, 3, 3), torch.zeros(4, dtype=torch.int32),
                                      torch.zeros(6, dtype=torch.int32))
