"""Texture path: the HIP sampler behind `geom_utils.sample_textures` (geom_utils.py:11-46,
loss_utils.py:242-301) against torch's own `grid_sample` in float64, and the batched distance
transform behind `utils.image.compute_dt(_barrier)` (utils/image.py:94-113) against scipy's
`distance_transform_edt` -- the very library calls the reference makes, so parity is pinned.

Tolerances: samples and flow gradients 1e-5 of the largest reference value (fp32 against fp64:
the pixel coordinate of a 64-pixel image carries ~4e-6 of fp32 rounding, and random images have
unit slope; samples whose flow lands within 1e-4 of a pixel boundary are excluded from the
GRADIENT check, where the bilinear derivative jumps); distance maps 2e-6 absolute (values in [0,1], exact
integer squared distances, fp32 square root and sigmoid).
"""
import numpy as np
import pytest
import torch

from helpers import rel_err
from oracle import texture_oracle as to

pytestmark = pytest.mark.gpu


def _flow(B, F, T, seed, spread=1.15):
    g = torch.Generator().manual_seed(seed)
    return (2 * torch.rand(B, F, T, T, 2, generator=g) - 1) * spread        # some samples outside [-1,1]


def _away_from_pixel_edges(flow, H, W, margin=1e-4):
    ix = (flow[..., 0].double() + 1) / 2 * (W - 1)
    iy = (flow[..., 1].double() + 1) / 2 * (H - 1)
    fx, fy = ix - ix.floor(), iy - iy.floor()
    ok = (fx > margin) & (fx < 1 - margin) & (fy > margin) & (fy < 1 - margin)
    return ok.reshape(flow.shape[0], -1)


@pytest.mark.parametrize("B,C,H,W,F,T", [(2, 3, 64, 64, 50, 6), (3, 1, 37, 53, 31, 2), (1, 4, 8, 5, 7, 3),
                                         (2, 3, 1, 1, 5, 2)])
def test_sample_textures_matches_grid_sample(B, C, H, W, F, T):
    from nnutils import geom_utils
    g = torch.Generator().manual_seed(B * 100 + C)
    images = torch.rand(B, C, H, W, generator=g)
    flow = _flow(B, F, T, seed=H)
    flow[0, 0, 0, 0] = torch.tensor([-1.0, -1.0])                 # exact corners and the centre
    flow[0, 0, 0, 1] = torch.tensor([1.0, 1.0])
    flow[0, 1, 0, 0] = torch.tensor([0.0, 0.0])
    w = torch.rand(B, F, T * T, C, generator=g) - 0.5

    fd = flow.double().requires_grad_(True)
    want = to.sample_textures(fd, images.double())
    (want * w.double()).sum().backward()

    fg = flow.cuda().requires_grad_(True)
    got = geom_utils.sample_textures(fg, images.cuda())
    assert got.shape == (B, F, T * T, C) and got.is_contiguous()
    (got * w.cuda()).sum().backward()
    assert rel_err(got.cpu(), want.detach()) < 1e-5
    if H > 1 and W > 1:
        ok = _away_from_pixel_edges(flow, H, W)[..., None].expand(-1, -1, 2)
        gg = fg.grad.cpu().reshape(B, -1, 2)
        gw = fd.grad.reshape(B, -1, 2)
        assert float(((gg.double() - gw).abs() * ok).max() / gw.abs().max()) < 1e-5
    got2 = geom_utils.sample_textures_2(fg.detach(), images.cuda())
    assert got2.shape == (B, F, T, T, C) and torch.equal(got2.reshape(got.shape), got.detach())


def test_sampler_layout_feeds_the_rasterizer_and_dt_loss():
    """The sampler's output is what `SoftNeuralRenderer.forward(..., textures=)` takes
    (main.py:395,415-418), and the distance-map loss samples one channel (loss_utils.py:286-301)."""
    from nnutils import geom_utils, loss_utils
    from nnutils.softras import SoftNeuralRenderer
    from helpers import load_mesh, random_cams, random_vertices
    verts, faces = load_mesh(2)
    B, F, T = 2, faces.shape[0], 4
    images = torch.rand(B, 3, 32, 32, generator=torch.Generator().manual_seed(1)).cuda()
    flow = (0.9 * _flow(B, F, T, seed=3, spread=1.0)).cuda().requires_grad_(True)
    tex = geom_utils.sample_textures(flow, images)
    r = SoftNeuralRenderer(img_size=32)
    r.ambient_light_only()
    img = r.forward(random_vertices(verts, B).cuda(), faces[None].expand(B, -1, -1).cuda(), random_cams(B).cuda(),
                    textures=tex)
    img.sum().backward()
    assert torch.isfinite(flow.grad).all() and float(flow.grad.abs().max()) > 0

    dmap = torch.rand(2 * B, 1, 16, 16, generator=torch.Generator().manual_seed(2))
    fl = _flow(2 * B, 9, 3, seed=5)
    probs = torch.tensor([[0.3, 0.9], [0.7, 0.1]])
    want = (to.sample_textures(fl.double(), dmap.double()).mean([1, 2, 3]).view(2, -1) * probs.double()).sum(0).mean()
    got = loss_utils.texture_dt_loss(fl.cuda(), dmap.cuda(), probs=probs.cuda(), num_cams=2)
    assert abs(got.item() - want.item()) < 1e-6 * abs(want.item())
    uniform = loss_utils.texture_dt_loss(fl.cuda(), dmap.cuda(), num_cams=2)
    want_u = to.sample_textures(fl.double(), dmap.double()).mean([1, 2, 3]).view(2, -1).sum(0).mean() / 2
    assert abs(uniform.item() - want_u.item()) < 1e-6 * abs(want_u.item())


def _masks():
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:64, 0:64]
    disk = ((yy - 30) ** 2 + (xx - 35) ** 2 < 18 ** 2).astype(np.float32)
    blobs = (rng.random((64, 64)) < 0.02).astype(np.float32)
    two = np.zeros((64, 64), np.float32)
    two[5:20, 5:30] = 1
    two[40:60, 33:36] = 1
    soft = disk.copy()
    soft[28:32, 10:20] = 0.5                 # neither 0 nor 1: foreground of both fields
    return np.stack([disk, blobs, two, soft, np.zeros((64, 64), np.float32), np.ones((64, 64), np.float32)])


def test_distance_transforms_match_scipy():
    from utils import image as image_utils
    masks = _masks()
    got_b = image_utils.compute_dt_barrier(torch.from_numpy(masks).cuda()).cpu().numpy()
    got_d = image_utils.compute_dt(torch.from_numpy(masks).cuda()).cpu().numpy()
    for i, m in enumerate(masks):
        assert np.abs(got_b[i] - to.compute_dt_barrier(m)).max() < 2e-6, i
        assert np.abs(got_d[i] - to.compute_dt(m)).max() < 2e-6, i
    # other k, a single [H,W] array in and out, non-square sizes beyond one workgroup's width
    one = image_utils.compute_dt_barrier(masks[2], k=10)
    assert isinstance(one, np.ndarray) and np.abs(one - to.compute_dt_barrier(masks[2], k=10)).max() < 2e-6
    rng = np.random.default_rng(3)
    wide = (rng.random((2, 40, 300)) < 0.01).astype(np.float32)
    got = image_utils.compute_dt_barrier(torch.from_numpy(wide).cuda()).cpu().numpy()
    for i in range(2):
        assert np.abs(got[i] - to.compute_dt_barrier(wide[i])).max() < 2e-6
    # the reference's second call is on the mirrored mask (main.py:260): mirrored transform
    flipped = image_utils.compute_dt_barrier(torch.from_numpy(masks[:4, :, ::-1].copy()).cuda())
    assert torch.equal(flipped.cpu(), torch.from_numpy(got_b[:4, :, ::-1].copy()))


def test_full_size_masks_properties():
    """256 x 256 (the reference's image size), batch 16: inside the mask the barrier map is below
    1/2, outside above, exactly 1/2 nowhere; repeatable."""
    from utils import image as image_utils
    yy, xx = np.mgrid[0:256, 0:256]
    masks = np.stack([((yy - 128) ** 2 + (xx - 100 - 3 * i) ** 2 < (40 + 4 * i) ** 2).astype(np.float32)
                      for i in range(16)])
    m = torch.from_numpy(masks).cuda()
    a, b = image_utils.compute_dt_barrier(m), image_utils.compute_dt_barrier(m)
    assert torch.equal(a, b)
    assert bool((a[m == 1] < 0.5).all()) and bool((a[m == 0] > 0.5).all())
    assert np.abs(a[3].cpu().numpy() - to.compute_dt_barrier(masks[3])).max() < 2e-6


def test_invalid_arguments_raise():
    import sv3d.texture as tx
    with pytest.raises(ValueError):
        tx.texture_sample(torch.rand(2, 5, 4, 4).cuda(), torch.rand(2, 3, 2).cuda())       # 5 channels
    with pytest.raises(ValueError):
        tx.texture_sample(torch.rand(2, 3, 4, 4).cuda(), torch.rand(3, 3, 2).cuda())       # batch mismatch
    with pytest.raises(RuntimeError):
        tx.texture_sample(torch.rand(2, 3, 4, 4), torch.rand(2, 3, 2))                      # CPU tensors
    with pytest.raises(RuntimeError):
        tx.texture_sample(torch.rand(2, 3, 4, 4).cuda().requires_grad_(True), torch.rand(2, 3, 2).cuda())
    with pytest.raises(ValueError):
        tx.dt_barrier(torch.rand(4, 4).cuda())
