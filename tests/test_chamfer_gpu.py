"""GPU parity tests for the Chamfer nearest-neighbour path (through the C ABI).

Bar: indices bit-exact against the C oracle, squared distances within 1e-6 relative
(they are in fact bit-exact: same fmaf chain), gradients within 1e-6 relative to max.
"""
import numpy as np
import pytest
import torch

from helpers import rel_err
from oracle import chamfer_oracle

pytestmark = pytest.mark.gpu

DIST_RTOL = 1e-6
GRAD_RTOL = 1e-6


def _run(a, b):
    import sv3d
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    d1, d2, i1, i2 = sv3d.chamfer_nn(ta, tb)
    torch.cuda.synchronize()
    return d1.cpu().numpy(), d2.cpu().numpy(), i1.cpu().numpy(), i2.cpu().numpy()


def _check(a, b):
    got = _run(a, b)
    want = chamfer_oracle.forward_c(a, b)
    np.testing.assert_array_equal(got[2], want[2])
    np.testing.assert_array_equal(got[3], want[3])
    np.testing.assert_allclose(got[0], want[0], rtol=DIST_RTOL, atol=0)
    np.testing.assert_allclose(got[1], want[1], rtol=DIST_RTOL, atol=0)
    assert got[2].dtype == np.int32 and got[0].dtype == np.float32
    return got


@pytest.mark.parametrize("B,N,M", [(1, 1, 1), (2, 642, 642), (3, 157, 2051), (2, 2048, 2048),
                                   (1, 1025, 63), (5, 64, 4097),
                                   # calls served by the waves-work-alone kernel: every candidate slot
                                   # count (4, 8, 12, 16 per lane) and its edges, odd query counts
                                   (4, 256, 256), (3, 257, 512), (2, 768, 769), (1, 1024, 1023), (16, 642, 100),
                                   (7, 3, 65), (16, 1024, 1024), (17, 1024, 1024),
                                   # the tiled kernel's other forms: four waves with one query per lane
                                   # (above 512 workgroups), two per lane (from 65536 points per call),
                                   # four per lane (from 262144)
                                   (9, 2048, 2048), (52, 642, 642), (33, 2051, 1990), (64, 2048, 2100)])
def test_forward_random_clouds(B, N, M):
    rng = np.random.default_rng(B * 1000 + N + M)
    _check(rng.standard_normal((B, N, 3)).astype(np.float32),
           rng.standard_normal((B, M, 3)).astype(np.float32))


@pytest.mark.parametrize("N,M", [(300, 290), (642, 642), (2100, 2051)])      # waves-work-alone and tiled kernels
def test_forward_nan_and_overflowing_coordinates_keep_indices_inside_the_cloud(N, M):
    """A query no candidate compares below +inf against (a NaN coordinate, or coordinates so large
    that every squared distance overflows) reports distance +inf and index 0, whichever kernel the
    size picks; NaN candidates are never chosen.  Callers gather with these indices."""
    rng = np.random.default_rng(N + M)
    a = rng.standard_normal((2, N, 3)).astype(np.float32)
    b = rng.standard_normal((2, M, 3)).astype(np.float32)
    a[0, 5, 1] = np.nan                     # a NaN query / candidate
    a[1, N - 1, :] = 3e19                   # every distance from and to it overflows to +inf
    b[0, 7, 2] = np.nan
    b[1, 0, 0] = np.nan                     # candidate 0 itself is NaN: still the answer for hopeless queries
    d1, d2, i1, i2 = _check(a, b)
    assert i1.min() >= 0 and i1.max() < M and i2.min() >= 0 and i2.max() < N
    assert np.isinf(d1[0, 5]) and i1[0, 5] == 0 and np.isinf(d1[1, N - 1]) and i1[1, N - 1] == 0
    assert not np.any(i1[0] == 7) and not np.any(i2[0] == 5) and not np.isnan(d1).any() and not np.isnan(d2).any()


def test_forward_ties_duplicates_and_zero_distances():
    lat = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij"), -1)
    lat = lat.reshape(1, -1, 3).astype(np.float32)                      # 512 lattice points
    q = (lat[:, ::3] + 0.5).astype(np.float32)                          # 8-way ties everywhere
    d1, d2, i1, i2 = _check(q, lat)
    assert np.all(d1[0] == 0.75)
    dup = np.concatenate([lat, lat, lat], 1)                            # every point three times
    d1, _, i1, _ = _check(lat, dup)
    assert np.all(d1 == 0) and np.array_equal(i1[0], np.arange(512))    # first copy wins
    _check(dup, dup[:, ::-1].copy())
    # the same on clouds small enough for the kernel whose lanes split the candidates: the minimum
    # is attained in several lanes at once and the lowest index must win
    small = lat[:, :64]
    dup = np.concatenate([small, small, small], 1)
    d1, _, i1, _ = _check(np.repeat(small, 4, 0), np.repeat(dup, 4, 0))
    assert np.all(d1 == 0) and np.array_equal(i1[3], np.arange(64))
    _check(np.repeat(dup, 4, 0), np.repeat(dup[:, ::-1], 4, 0).copy())


def test_forward_mesh_vertices_and_mirror():
    """The reference's call: vertices against their x-mirrored copy (main.py:522-525)."""
    from helpers import load_mesh
    v, _ = load_mesh(3)
    rng = np.random.default_rng(0)
    a = (v.numpy()[None] + 0.02 * rng.standard_normal((4, 642, 3))).astype(np.float32)
    b = a * np.array([-1, 1, 1], np.float32)
    _check(a, b)


@pytest.mark.parametrize("use1,use2", [(True, True), (False, True), (True, False)])
def test_backward_matches_oracle(use1, use2):
    import sv3d
    rng = np.random.default_rng(9)
    a = rng.standard_normal((3, 300, 3)).astype(np.float32)
    b = rng.standard_normal((3, 1100, 3)).astype(np.float32)
    # add duplicates so that several points share a nearest neighbour (scatter side)
    b[:, 500:600] = b[:, 0:1]
    gd1 = rng.standard_normal((3, 300)).astype(np.float32)
    gd2 = rng.standard_normal((3, 1100)).astype(np.float32)
    ta = torch.from_numpy(a).cuda().requires_grad_(True)
    tb = torch.from_numpy(b).cuda().requires_grad_(True)
    d1, d2, i1, i2 = sv3d.chamfer_nn(ta, tb)
    loss = 0
    if use1:
        loss = loss + (d1 * torch.from_numpy(gd1).cuda()).sum()
    if use2:
        loss = loss + (d2 * torch.from_numpy(gd2).cuda()).sum()
    loss.backward()
    w1, w2 = chamfer_oracle.backward_c(a, b, gd1 if use1 else 0 * gd1, gd2 if use2 else 0 * gd2,
                                       i1.cpu().numpy(), i2.cpu().numpy())
    assert rel_err(ta.grad.cpu(), torch.from_numpy(w1)) < GRAD_RTOL
    assert rel_err(tb.grad.cpu(), torch.from_numpy(w2)) < GRAD_RTOL


@pytest.mark.parametrize("N,M", [(8192, 8192),      # sorted form, LDS tables above 64 KB
                                 (5460, 5460),      # tables 96 bytes short of 64 KB: with the kernel's
                                                    # static arrays the launch needs the raised limit
                                 (700, 14000)])     # tables beyond the LDS: scan form
def test_backward_both_sides_large_clouds(N, M):
    import sv3d
    rng = np.random.default_rng(21)
    a = rng.standard_normal((2, N, 3)).astype(np.float32)
    b = rng.standard_normal((2, M, 3)).astype(np.float32)
    b[:, 100:400] = b[:, 7:8]                      # one long bucket
    gd1 = rng.standard_normal((2, N)).astype(np.float32)
    gd2 = rng.standard_normal((2, M)).astype(np.float32)
    ta = torch.from_numpy(a).cuda().requires_grad_(True)
    tb = torch.from_numpy(b).cuda().requires_grad_(True)
    grads = []
    for _ in range(2):
        ta.grad = tb.grad = None
        d1, d2, i1, i2 = sv3d.chamfer_nn(ta, tb)
        ((d1 * torch.from_numpy(gd1).cuda()).sum() + (d2 * torch.from_numpy(gd2).cuda()).sum()).backward()
        grads.append((ta.grad.clone(), tb.grad.clone()))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])
    w1, w2 = chamfer_oracle.backward_c(a, b, gd1, gd2, i1.cpu().numpy(), i2.cpu().numpy())
    assert rel_err(grads[0][0].cpu(), torch.from_numpy(w1)) < GRAD_RTOL
    assert rel_err(grads[0][1].cpu(), torch.from_numpy(w2)) < GRAD_RTOL


def test_backward_collapsed_cloud_stays_fast_and_exact():
    """Every point of the second cloud has the same nearest neighbour (a cloud collapsed onto one
    point next to it): one bucket holds all 8192 indices.  The walk of such a bucket is bounded by
    a scan of the index array; the result equals the oracle's."""
    import time
    import sv3d
    rng = np.random.default_rng(4)
    a = rng.standard_normal((2, 8192, 3)).astype(np.float32)
    b = (a[:, 17:18] + 1e-3 * rng.standard_normal((2, 8192, 3))).astype(np.float32)
    gd1 = rng.standard_normal((2, 8192)).astype(np.float32)
    gd2 = rng.standard_normal((2, 8192)).astype(np.float32)
    ta = torch.from_numpy(a).cuda().requires_grad_(True)
    tb = torch.from_numpy(b).cuda().requires_grad_(True)
    d1, d2, i1, i2 = sv3d.chamfer_nn(ta, tb)
    assert int((i2 == 17).sum()) == 2 * 8192
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ((d1 * torch.from_numpy(gd1).cuda()).sum() + (d2 * torch.from_numpy(gd2).cuda()).sum()).backward()
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 0.5
    w1, w2 = chamfer_oracle.backward_c(a, b, gd1, gd2, i1.cpu().numpy(), i2.cpu().numpy())
    # points on the hull of the clump collect thousands of first-cloud points each: fp32 sums of
    # thousands of O(1) terms against the oracle's double accumulation
    assert rel_err(ta.grad.cpu(), torch.from_numpy(w1)) < 1e-5
    assert rel_err(tb.grad.cpu(), torch.from_numpy(w2)) < 1e-5


def test_reference_usage_only_dist2_and_detached_first_cloud():
    """loss_utils.py:75-88 keeps dist2.mean(); main.py:525 detaches the first cloud."""
    from nnutils import loss_utils
    rng = np.random.default_rng(11)
    a = rng.standard_normal((4, 642, 3)).astype(np.float32)
    b = (a + 0.05 * rng.standard_normal(a.shape)).astype(np.float32)
    ta = torch.from_numpy(a).cuda().requires_grad_(True)
    tb = torch.from_numpy(b).cuda().requires_grad_(True)
    loss = loss_utils.chamfer_loss(ta.detach(), tb)
    assert loss.dim() == 0
    assert abs(loss.item() - chamfer_oracle.chamfer_loss(a, b)) < 1e-6 * max(1.0, abs(loss.item()))
    loss.backward()
    assert ta.grad is None
    _, _, i1, i2 = chamfer_oracle.forward_c(a, b)
    gd2 = np.full((4, 642), 1.0 / (4 * 642), np.float32)
    _, w2 = chamfer_oracle.backward_c(a, b, 0 * gd2, gd2, i1, i2)
    # with the first cloud detached only the gather term 2/(BM) (x2 - x1[idx2]) is live
    j = torch.from_numpy(i2).long()
    pure = 2.0 / (4 * 642) * (torch.from_numpy(b) - torch.gather(torch.from_numpy(a), 1, j[..., None].expand(-1, -1, 3)))
    assert rel_err(tb.grad.cpu(), pure) < GRAD_RTOL
    assert rel_err(torch.from_numpy(w2), pure.double()) < 1e-6


def test_bitwise_repeatable():
    import sv3d
    rng = np.random.default_rng(3)
    a = torch.from_numpy(rng.standard_normal((2, 999, 3)).astype(np.float32)).cuda()
    b = torch.from_numpy(rng.standard_normal((2, 777, 3)).astype(np.float32)).cuda()
    b[:, 100:200] = b[:, :1]
    outs = []
    for _ in range(3):
        x, y = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        d1, d2, i1, i2 = sv3d.chamfer_nn(x, y)
        (d1.sum() + (d2 * d2).sum()).backward()
        outs.append([t.cpu() for t in (d1, d2, i1, i2, x.grad, y.grad)])
    for o in outs[1:]:
        for p, q in zip(outs[0], o):
            assert torch.equal(p, q)


def test_full_size_8192_properties():
    """The largest benchmark clouds, 8192 x 8192: too slow for the scalar oracle in full, so check
    size-independent properties plus the oracle on a slice of the queries."""
    import sv3d
    rng = np.random.default_rng(8192)
    a = rng.standard_normal((2, 8192, 3)).astype(np.float32)
    b = rng.standard_normal((2, 8192, 3)).astype(np.float32)
    d1, d2, i1, i2 = _run(a, b)
    # (1) the reported distance is the distance to the reported index, bit for bit
    for dist, idx, own, oth in ((d1, i1, a, b), (d2, i2, b, a)):
        sel = np.take_along_axis(oth, idx[..., None].astype(np.int64), 1)
        dd = (sel - own).astype(np.float32)
        f = lambda x, y, z: (x.astype(np.float64) * y.astype(np.float64) + z.astype(np.float64)).astype(np.float32)
        re = f(dd[..., 2], dd[..., 2], f(dd[..., 1], dd[..., 1], (dd[..., 0] * dd[..., 0]).astype(np.float32)))
        np.testing.assert_array_equal(dist, re)
    # (2) oracle on the first 256 queries of each direction
    w = chamfer_oracle.forward_c(a[:, :256], b)
    np.testing.assert_array_equal(i1[:, :256], w[2])
    np.testing.assert_array_equal(d1[:, :256], w[0])
    w = chamfer_oracle.forward_c(a, b[:, :256])
    np.testing.assert_array_equal(i2[:, :256], w[3])
    # (3) a cloud against itself: zero distance, identity index
    d1, d2, i1, i2 = _run(a, a.copy())
    assert not d1.any() and not d2.any()
    assert np.array_equal(i1[0], np.arange(8192)) and np.array_equal(i2[1], np.arange(8192))


def test_invalid_arguments_raise():
    import sv3d
    with pytest.raises(ValueError):
        sv3d.chamfer_nn(torch.zeros(1, 0, 3).cuda(), torch.zeros(1, 4, 3).cuda())
    with pytest.raises(ValueError):
        sv3d.chamfer_nn(torch.zeros(2, 4, 3).cuda(), torch.zeros(1, 4, 3).cuda())
    with pytest.raises(ValueError):
        sv3d.chamfer_nn(torch.zeros(1, 4, 2).cuda(), torch.zeros(1, 4, 2).cuda())
