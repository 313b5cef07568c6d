"""The rasterizer at the sizes the training configurations run it at, against the fp64 oracle on
sampled pixels (the dense oracle cannot hold 512^2 x 1280..5120 pairs at once).

Per image a set of OUTPUT pixels is drawn, half uniformly and half from the object's own pixels
(alpha > 0 in the kernels' image, which only decides where to look).  The oracle evaluates the
raster pixels under them (four per output pixel with anti-aliasing) and averages; the incoming
gradient is non-zero on exactly those output pixels, so the oracle's face and texture gradients
are the complete ones.  Bars as everywhere: image and gradients 1e-5 of the largest reference
value.  In addition the element-wise relative error over entries above 1e-3 of the largest one is
recorded (`p999_*`: its 99.9th percentile) and asserted below 1e-3.  An absolute error of 1e-6 of
the largest entry -- a tenth of the bar above -- already is 1e-3 of an entry a thousand times
smaller, so this figure is bounded by (max-norm error) / 1e-3 and says that no class of small
entries is systematically wrong; measured 5e-5 .. 3.5e-4.  For scale, `fp32_upstream_*` records the
same figures for the oracle's own upstream-style formulas evaluated densely in fp32 on the same
inputs (what single precision costs with SoftRas's barycentric `t - w` algebra, before any kernel
is involved): max-norm 4e-2 .. 6e-2, p999 of order 1.

Sliver faces (`helpers.sliver_faces`: three projected corners collinear to fp32, so that which of
their all but coincident edges is "nearest" to a pixel is decided by rounding in any precision) are
left out of the gradient comparison and counted.

Sizes follow the reference's defaults (softras.py:33-37: sigma 1e-5, gamma 1e-4, anti-aliasing;
mesh_net.py:34 tex_size 6; utils/mesh.py:13-14 subdivide 3 / 4; main.py:56 part maps).
"""
import json
import os

import pytest
import torch

from helpers import camera_space_faces, elementwise_rel_err, load_mesh, rel_err, sliver_faces
from oracle import raster_oracle

pytestmark = pytest.mark.gpu

TOL = 1e-5
P999_TOL = 1e-3
_LOG = []


def _record(name, **kw):
    _LOG.append(dict(test=name, **{k: float(v) for k, v in kw.items()}))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "raster_parity_fullsize.json"), "w") as fh:
            json.dump(_LOG, fh, indent=1)


def _textures(mode, B, F, R, seed=2):
    g = torch.Generator().manual_seed(seed)
    if mode == "surface":
        return torch.rand(B, F, R * R, 3, generator=g)
    if mode == "vertex":
        return torch.rand(B, F, 3, 3, generator=g)
    if mode == "vertex8":
        return torch.rand(B, F, 3, 8, generator=g)
    return None


def _render(fv, tex, img, aa, mode, **kw):
    import sv3d
    return sv3d.soft_rasterize(fv, tex, img, anti_aliasing=aa, texture_type=mode if mode != "none" else "surface",
                               sigma_val=kw.get("sigma", 1e-5), gamma_val=kw.get("gamma", 1e-4),
                               dist_eps=kw.get("dist_eps", 1e-10),
                               background_color=kw.get("background", (0., 0., 0.)))


def _sample_output_pixels(alpha, n, gen):
    """n flat output-pixel indices: half uniform, half from the object's pixels."""
    hw = alpha.numel()
    uniform = torch.randperm(hw, generator=gen)[: n // 2]
    obj = torch.nonzero(alpha.reshape(-1) > 0)[:, 0]
    obj = obj[torch.randperm(obj.numel(), generator=gen)[: n - n // 2]]
    return torch.unique(torch.cat([uniform, obj]))


def _raster_pixels(out_pix, img, aa):
    """Raster-pixel indices under each output pixel: [n, aa*aa] (row-major in the S x S raster)."""
    k = 2 if aa else 1
    S = img * k
    r, c = out_pix // img, out_pix % img
    sub = [(k * r + i) * S + (k * c + j) for i in range(k) for j in range(k)]
    return torch.stack(sub, 1)


def sampled_compare(name, fv, tex, img, aa, mode, n_out, check_images=None, fp32_floor=False, grad_tol=None, **kw):
    """GPU forward + backward over the whole batch; oracle on `n_out` output pixels of every image
    in `check_images` (default: all).  Returns the worst errors.  `grad_tol`: bar of the vertex
    gradient where the case is beyond what fp32 corner-relative arithmetic holds (the caller then
    compares with the upstream arithmetic in fp32, `fp32_floor`)."""
    B, F = fv.shape[:2]
    S = img * (2 if aa else 1)
    channels = 9 if mode == "vertex8" else 4
    check = list(range(B)) if check_images is None else list(check_images)
    gen = torch.Generator().manual_seed(11)
    x = fv.cuda().requires_grad_(True)
    t = tex.cuda().requires_grad_(True) if tex is not None else None
    with torch.no_grad():
        alpha0 = _render(x, t, img, aa, mode, **kw)[:, -1].cpu()

    # oracle forward on the sampled pixels first: it also says which pixels to leave out
    per_image = {}
    for b in check:
        out_pix = _sample_output_pixels(alpha0[b], n_out, gen)
        rp = _raster_pixels(out_pix, img, aa)
        xb = fv[b].double().requires_grad_(True)
        tb = tex[b].double().requires_grad_(True) if tex is not None else None
        o = raster_oracle.rasterize_image(xb, tb, S, texture_mode=mode, pixels=rp.reshape(-1), **kw)
        want = o["rgba"].reshape(channels, -1, rp.shape[1]).mean(-1)              # [C, n]
        unstable = o["unstable"].reshape(-1, rp.shape[1]).any(-1)                 # [n]
        vals = (2 * torch.rand(channels, out_pix.numel(), generator=gen) - 1) * (~unstable)[None]
        per_image[b] = (out_pix, xb, tb, want, unstable, vals)

    # images the oracle does not look at still get a gradient, so the launch does its full work
    grad = 2 * torch.rand(B, channels, img * img, generator=gen) - 1
    grad[check] = 0
    for b, (out_pix, _, _, _, _, vals) in per_image.items():
        grad[b][:, out_pix] = vals
    got = _render(x, t, img, aa, mode, **kw)
    (got * grad.reshape(B, channels, img, img).cuda()).sum().backward()
    got = got.detach().cpu().reshape(B, channels, -1)
    gfv = x.grad.cpu()
    gtex = t.grad.cpu() if t is not None else None
    assert torch.isfinite(got).all() and torch.isfinite(gfv).all()

    worst = dict(img=0.0, grad_fv=0.0, grad_tex=0.0, p999_fv=0.0, p999_tex=0.0, unstable=0.0, sliver_faces=0.0)
    for b, (out_pix, xb, tb, want, unstable, vals) in per_image.items():
        (want * vals.double()).sum().backward()
        keep = ~unstable
        e_img = float((got[b][:, out_pix].double() - want.detach())[:, keep].abs().max() / want.detach().abs().max())
        worst["img"] = max(worst["img"], e_img)
        sliver = sliver_faces(fv[b])
        worst["sliver_faces"] += float(sliver.sum())
        err_fv = (gfv[b].double() - xb.grad).abs()[~sliver].max() / xb.grad.abs().max()
        worst["grad_fv"] = max(worst["grad_fv"], float(err_fv))
        worst["p999_fv"] = max(worst["p999_fv"], elementwise_rel_err(gfv[b][~sliver], xb.grad[~sliver]))
        worst["unstable"] = max(worst["unstable"], float(unstable.float().mean()))
        assert float(xb.grad.abs().max()) > 0
        if tb is not None:
            worst["grad_tex"] = max(worst["grad_tex"], rel_err(gtex[b], tb.grad))
            worst["p999_tex"] = max(worst["p999_tex"], elementwise_rel_err(gtex[b], tb.grad))
    if fp32_floor:
        b = check[0]
        out_pix, xb, tb, want, unstable, vals = per_image[b]
        x32 = fv[b].float().requires_grad_(True)
        t32 = tex[b].float().requires_grad_(True) if tex is not None else None
        o = raster_oracle.rasterize_image(x32, t32, S, texture_mode=mode,
                                          pixels=_raster_pixels(out_pix, img, aa).reshape(-1), **kw)
        (o["rgba"].reshape(channels, -1, 4 if aa else 1).mean(-1) * vals).sum().backward()
        worst["fp32_upstream_grad_fv"] = rel_err(x32.grad, xb.grad)
        worst["fp32_upstream_p999_fv"] = elementwise_rel_err(x32.grad, xb.grad)
    _record(name, **worst)
    assert worst["unstable"] < 2e-2 and worst["sliver_faces"] <= 2 * len(check), worst
    assert worst["img"] < TOL and worst["grad_fv"] < (grad_tol or TOL) and worst["grad_tex"] < TOL, worst
    assert worst["p999_fv"] < P999_TOL * (grad_tol or TOL) / TOL and worst["p999_tex"] < P999_TOL, worst
    return worst


def test_surface_rgb_at_the_reference_raster():
    """The training step's RGB render (main.py:418): 8 images, 512^2 raster pooled to 256^2,
    1280 faces, 36 texels per face, sigma 1e-5, gamma 1e-4."""
    verts, faces = load_mesh(3)
    B = 8
    fv = camera_space_faces(verts, faces, B, seed=0)
    tex = _textures("surface", B, faces.shape[0], 6)
    sampled_compare("c3_surface_512aa_T6", fv, tex, 256, True, "surface", 192, background=(0.2, 0.4, 0.6))


def test_mask_at_the_reference_raster():
    """The step's silhouette render (main.py:417) at the same size."""
    verts, faces = load_mesh(3)
    fv = camera_space_faces(verts, faces, 4, seed=2)
    w = sampled_compare("c3_mask_512aa", fv, None, 256, True, "none", 256, fp32_floor=True)
    assert w["grad_fv"] < w["fp32_upstream_grad_fv"] and w["p999_fv"] < w["fp32_upstream_p999_fv"]


@pytest.mark.parametrize("mode", ["none", "surface"])
def test_hd_mesh_at_the_reference_raster(mode):
    """Subdivide 4 (2562 v / 5120 f), 512^2 raster, tex_size 6: the coarse-bin path at size."""
    verts, faces = load_mesh(4)
    B = 2
    fv = camera_space_faces(verts, faces, B, seed=4)
    tex = _textures(mode, B, faces.shape[0], 6)
    sampled_compare("c4_hd_%s_512aa_T6" % mode, fv, tex, 256, True, mode, 160, fp32_floor=(mode == "none"))


def test_eight_part_maps_at_the_reference_raster():
    """The part-map pass (main.py:422-426 rendered as eight channels at once), 512^2 raster."""
    verts, faces = load_mesh(3)
    B = 2
    fv = camera_space_faces(verts, faces, B, seed=6)
    tex = torch.softmax(torch.randn(B, faces.shape[0], 3, 8, generator=torch.Generator().manual_seed(3)), -1)
    sampled_compare("c5_vertex8_512aa", fv, tex, 256, True, "vertex8", 192)


@pytest.mark.parametrize("mode", ["none", "surface"])
def test_batch_40_launch_against_the_oracle(mode):
    """40 images x 1280 faces in one launch: the backward's XCD-permuted face order (batch >= 32)
    and its one-wave-per-face kernels (more than 16384 faces), checked against the oracle on five
    of the images rather than against smaller launches of the same kernels."""
    verts, faces = load_mesh(3)
    B = 40
    fv = camera_space_faces(verts, faces, B, seed=8)
    tex = _textures(mode, B, faces.shape[0], 3)
    sampled_compare("b40_%s_128" % mode, fv, tex, 128, False, mode, 384, check_images=(0, 7, 19, 32, 39),
                    sigma=1e-4, gamma=1e-3)


@pytest.mark.parametrize("B", [2, 5])
def test_tiled_forward_with_more_faces_in_a_tile_than_a_list_holds(B):
    """Silhouette launches take per-tile face lists of at most 128 faces; a tile with more (here the
    5120-face mesh drawn a fifth of the screen wide on a 512^2 raster: a few dozen tiles share all
    its faces) goes through all faces itself.  Two images: the launch whose setup kernel also
    builds the lists; five: lists from a kernel of their own."""
    verts, faces = load_mesh(4)
    fv = camera_space_faces(verts, faces, B, seed=12)
    fv[..., :2] *= 0.2                                      # x, y: a small object, dense in faces
    fv[..., 0] += 0.21
    w = sampled_compare("tiled_overflow_hd_small_512_b%d" % B, fv, None, 512, False, "none", 256,
                        check_images=(0, B - 1), sigma=3e-5)
    assert w["img"] < TOL


def test_colour_launch_whose_tile_lists_overflow():
    """The colour modes take the per-tile lists in place of their tiles' own binning from 4096 tiles
    on; a tile with more faces than a list holds bins for itself as before.  Five images of the
    5120-face mesh drawn small: both kinds of tile in one launch."""
    verts, faces = load_mesh(4)
    B = 5
    fv = camera_space_faces(verts, faces, B, seed=13)
    fv[..., :2] *= 0.2
    fv[..., 1] -= 0.3
    tex = _textures("vertex", B, faces.shape[0], 1)
    sampled_compare("lists_overflow_vertex_hd_small_512", fv, tex, 512, False, "vertex", 192, check_images=(0, 4),
                    sigma=3e-5, gamma=1e-3)


@pytest.mark.parametrize("B,img,aa,sub,sigma", [
    (8, 256, False, 3, 1e-4),       # the benchmark configuration: 2048 tiles, two waves per block
    (12, 256, False, 3, 1e-4),      # 3072 tiles of a mesh with five faces per tile: still two waves per block
    (3, 256, True, 3, 1e-5),        # 3072 tiles, one wave per block
    (4, 200, False, 3, 3e-5),       # partial tiles at the border, pixel centres fp32 cannot hold exactly
    (1, 256, True, 4, 1e-5),        # 5120 faces: the list builders stage them 2048 at a time
    (6, 256, False, 3, 1e-5),       # the sharpest sigma on the coarsest raster the tile-relative records serve
    (2, 128, False, 2, 1e-5),       # ... and beyond it (sqrt(log2 e / sigma) * 16 / S = 47: still served)
])
def test_small_silhouette_launches_with_setup_and_lists_in_one_kernel(B, img, aa, sub, sigma):
    """Launches of at most 4096 tiles that come with face_vertices: one kernel does the setup and
    builds the per-tile lists (its list builders work the pixel boxes out for themselves, a pixel
    wider than the stored ones), the forward takes the lists."""
    verts, faces = load_mesh(sub)
    fv = camera_space_faces(verts, faces, B, seed=30 + B)
    sampled_compare("fused_lists_b%d_%d%s_sub%d" % (B, img, "aa" if aa else "", sub), fv, None, img, aa, "none",
                    256, check_images=sorted({0, B // 2, B - 1}), sigma=sigma)


@pytest.mark.parametrize("B", [2, 5, 12])
def test_tiled_forward_with_faces_many_tiles_long(B):
    """An 80-face mesh on a 512^2 raster at sigma 1e-5: edges of a hundred pixels, six tiles long.  A
    tile's record holds, per edge, the part of the edge's extent that the tile can reach, so its
    numbers stay small however long the edge is; the sharpest sigma makes any lost bits show.
    Two images: lists from the setup kernel; five: lists from their own kernel, one block per wave;
    twelve: two blocks per wave."""
    verts, faces = load_mesh(1)
    fv = camera_space_faces(verts, faces, B, seed=40 + B)
    w = sampled_compare("tiled_long_edges_512_b%d" % B, fv, None, 512, False, "none", 384,
                        check_images=sorted({0, B // 2, B - 1}), fp32_floor=True, grad_tol=3e-5, sigma=1e-5)
    # the image is what the tile records decide; the gradient of faces this long at this sigma is at
    # the fp32 floor of any corner-relative form (the backward's), and below the upstream arithmetic's
    assert w["img"] < 0.3 * TOL and w["grad_fv"] < w["fp32_upstream_grad_fv"], w


def test_four_waves_sharing_a_tiles_hits_in_a_launch_with_its_own_list_kernel():
    """20 images of 256^2 (5120 tiles of a mesh with five faces per tile): lists from the list kernel,
    and the forward whose four waves take a tile's hits in turn, each for all four 8x8 blocks
    (raster_forward_tiled_shared_kernel).  Launches of 1536 to 4096 tiles take the same forward behind
    the fused setup + list kernel: the 8- and 12-image cases of
    test_small_silhouette_launches_with_setup_and_lists_in_one_kernel; with lists that overflow,
    test_tiled_forward_with_more_faces_in_a_tile_than_a_list_holds[2]; anti-aliased,
    test_hd_mesh_at_the_reference_raster[none]; partial tiles,
    test_tiled_forward_on_a_raster_that_is_not_a_power_of_two."""
    verts, faces = load_mesh(3)
    B = 20
    fv = camera_space_faces(verts, faces, B, seed=15)
    sampled_compare("tiled_shared_hits_b20_256", fv, None, 256, False, "none", 256, check_images=(0, 11, 19),
                    sigma=1e-4)


def test_tiled_forward_on_a_raster_that_is_not_a_power_of_two():
    """Per-tile lists with partial tiles at the border and pixel centres that fp32 cannot hold
    exactly (S = 200: 13 x 13 tiles, the last 8 pixels wide); 16 images."""
    verts, faces = load_mesh(3)
    B = 16
    fv = camera_space_faces(verts, faces, B, seed=14)
    sampled_compare("tiled_mask_200", fv, None, 200, False, "none", 256, check_images=(0, 9, 15), sigma=3e-5)


@pytest.mark.parametrize("mode,sub", [("surface", 3), ("vertex", 3), ("vertex8", 3), ("surface", 4)])
def test_colour_modes_in_a_launch_of_five_images(mode, sub):
    """The colour modes in a launch of 5120 tiles (5 images at the 512^2 raster): the size from which
    the silhouette forward takes per-tile face lists.  The colour modes keep the per-tile binning
    of the general kernel at every size (their depth softmax, with gamma = 1e-4, needs barycentrics
    to 1e-7, which the tile-relative records of the list path do not give: measured 2.5 - 3 x the
    gradient error, 1.4e-5 on the 5120-face mesh); this pins them at that size."""
    verts, faces = load_mesh(sub)
    B = 5
    fv = camera_space_faces(verts, faces, B, seed=20 + sub)
    if mode == "vertex8":
        tex = torch.softmax(torch.randn(B, faces.shape[0], 3, 8, generator=torch.Generator().manual_seed(5)), -1)
    else:
        tex = _textures(mode, B, faces.shape[0], 6)
    sampled_compare("b5_%s_sub%d_512aa" % (mode, sub), fv, tex, 256, True, mode, 160, check_images=(0, 2, 4),
                    background=(0.1, 0.3, 0.5))

