"""GPU tests of the reference-shaped API: nnutils.softras.SoftNeuralRenderer (softras.py:25-77)
and the `soft_renderer` drop-in, against oracle/softras_oracle.py (fp64, same fp32 inputs).

These read like the checks the reference's own (assert-less) smoke scripts
softras.py:81-156 gesture at: render the three modes, differentiate to vertices / cameras /
textures, and fit a teapot silhouette to misc/examples/data/example2_ref.png.
"""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, load_mesh, random_cams, random_vertices, rel_err
from oracle import softras_oracle

pytestmark = pytest.mark.gpu

# End to end the camera stage hands fp32 vertex positions to the rasterizer (|x| ~ 1: 6e-8 of
# rounding), which the 1/sigma = 1e5 slope of the fragments amplifies.  That cost is measured per
# case -- the fp64 oracle with only its camera stage evaluated in fp32 (`stage_dtype`) against the
# all-fp64 oracle -- and recorded beside the kernels' error in gpurun_out/softras_parity.json; the
# kernels must stay within FLOOR_FACTOR times that floor (their own projection rounds differently
# from the oracle's fp32 one, so the two errors are of one size, not equal), and never beyond
# IMG_TOL / GRAD_TOL.  Measured: kernels 1e-6 .. 1.7e-5, floor 1e-6 .. 3.9e-5 -- the kernels sit at
# the floor, so the call's error is the camera stage's fp32 rounding, not the rasterizer's.
IMG_TOL = 2e-5
GRAD_TOL = 5e-5
FLOOR_FACTOR = 3.0
_LOG = []


def _record(name, **kw):
    import json
    _LOG.append(dict(test=name, **{k: float(v) for k, v in kw.items()}))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "softras_parity.json"), "w") as fh:
            json.dump(_LOG, fh, indent=1)


def _inputs(sub, B, seed):
    verts, faces = load_mesh(sub)
    v = random_vertices(verts, B, seed)
    cams = random_cams(B, seed + 1)
    f = faces.long()[None].repeat(B, 1, 1)
    return v, f, cams


@pytest.mark.parametrize("mode", ["mask", "rgb", "vertex"])
def test_soft_neural_renderer_modes_match_oracle(mode):
    from nnutils.softras import SoftNeuralRenderer
    B, img = 2, 32
    v, f, cams = _inputs(2, B, 0)
    g = torch.Generator().manual_seed(3)
    if mode == "rgb":
        tex = torch.rand(B, f.shape[1], 9, 3, generator=g)
    elif mode == "vertex":
        tex = torch.rand(B, v.shape[1], 3, generator=g)
    else:
        tex = None
    r = SoftNeuralRenderer(img_size=img, texture_type="vertex" if mode == "vertex" else "surface")
    r.set_bgcolor([0.2, 0.3, 0.4])
    vg = v.cuda().requires_grad_(True)
    cg = cams.cuda().requires_grad_(True)
    tg = tex.cuda().requires_grad_(True) if tex is not None else None
    out = r(vg, f.cuda(), cg, textures=tg)
    assert out.shape == ((B, img, img) if mode == "mask" else (B, 3, img, img))
    w = torch.rand(out.shape, generator=g)
    (out * w.cuda()).sum().backward()

    vo = v.double().requires_grad_(True)
    co = cams.double().requires_grad_(True)
    to = tex.double().requires_grad_(True) if tex is not None else None
    want = softras_oracle.render(vo, f, co, to, img_size=img, background=(0.2, 0.3, 0.4),
                                 texture_type="vertex" if mode == "vertex" else "surface")
    (want * w.double()).sum().backward()
    # the floor: same oracle, camera stage in fp32
    vf = v.double().requires_grad_(True)
    cf = cams.double().requires_grad_(True)
    tf = tex.double().requires_grad_(True) if tex is not None else None
    floor = softras_oracle.render(vf, f, cf, tf, img_size=img, background=(0.2, 0.3, 0.4),
                                  texture_type="vertex" if mode == "vertex" else "surface",
                                  stage_dtype=torch.float32)
    (floor * w.double()).sum().backward()
    err = dict(img=rel_err(out.detach().cpu(), want.detach()), grad_v=rel_err(vg.grad.cpu(), vo.grad),
               grad_cam=rel_err(cg.grad.cpu(), co.grad),
               grad_tex=rel_err(tg.grad.cpu(), to.grad) if tex is not None else 0.0)
    flo = dict(img=rel_err(floor.detach(), want.detach()), grad_v=rel_err(vf.grad, vo.grad),
               grad_cam=rel_err(cf.grad, co.grad), grad_tex=rel_err(tf.grad, to.grad) if tex is not None else 0.0)
    _record("softras_%s" % mode, **err, **{"floor_" + k: x for k, x in flo.items()})
    for k, cap in (("img", IMG_TOL), ("grad_v", GRAD_TOL), ("grad_cam", GRAD_TOL), ("grad_tex", GRAD_TOL)):
        assert err[k] < cap, (k, err[k])
        assert err[k] < max(FLOOR_FACTOR * flo[k], 1e-5), (k, err[k], flo[k])


def test_lighting_and_mutators():
    from nnutils.softras import SoftNeuralRenderer
    B, img = 2, 32
    v, f, cams = _inputs(1, B, 5)
    tex = torch.rand(B, f.shape[1], 4, 3, generator=torch.Generator().manual_seed(1))
    r = SoftNeuralRenderer(img_size=img)
    r.set_light_dir([0.3, 0.8, -0.5])
    out = r(v.cuda(), f.cuda(), cams.cuda(), textures=tex.cuda())
    want = softras_oracle.render(v.double(), f, cams.double(), tex.double(), img_size=img,
                                 light_dir=(0.3, 0.8, -0.5))
    assert rel_err(out.cpu(), want) < IMG_TOL
    r.ambient_light_only()
    out = r(v.cuda(), f.cuda(), cams.cuda(), textures=tex.cuda())
    want = softras_oracle.render(v.double(), f, cams.double(), tex.double(), img_size=img,
                                 ambient=1.0, directional=0.0)
    assert rel_err(out.cpu(), want) < IMG_TOL
    pts = r.project_points(v.cuda(), cams.cuda())
    assert rel_err(pts.cpu(), softras_oracle.orthographic_proj_withz(v.double(), cams.double())[..., :2]) < 1e-6
    assert r.offset_z == 5. and r.part_seg is False and r.mask_only is False


def test_soft_renderer_object_model():
    """The attribute paths softras.py:45,48,51,73-75 reach into."""
    import soft_renderer as sr
    v, f, _ = _inputs(1, 1, 2)
    v = v * 0.5
    r = sr.SoftRenderer(image_size=32, camera_mode='look_at', perspective=False, eye=[0, 0, -2.732],
                        sigma_val=1e-4, anti_aliasing=False)
    out = r(v.cuda(), f.cuda())
    assert out.shape == (1, 4, 32, 32)
    # default white surface, ambient 0.5 + directional 0.5: RGB in [0.5, 1] where covered
    cov = out[0, 3] > 0.99
    assert cov.any() and float(out[0, :3][:, cov].min()) >= 0.5 - 1e-5
    assert float(out[0, :3][:, cov].max()) <= 1.0 + 1e-5
    r.rasterizer.background_color = [1.0, 0.0, 0.0]
    r.lighting.directionals[0].light_direction = [0.0, 0.0, -1.0]
    out2 = r(v.cuda(), f.cuda())
    assert torch.allclose(out2[0, :3, 0, 0].cpu(), torch.tensor([1.0, 0.0, 0.0]))
    mesh = sr.Mesh(v.cuda(), f.cuda(), textures=torch.rand(1, v.shape[1], 3).cuda(), texture_type='vertex')
    r.set_texture_mode(mesh.texture_type)
    mesh = r.transform(mesh)
    img = r.rasterizer(mesh, None)
    assert img.shape == (1, 4, 32, 32) and torch.isfinite(img).all()
    with pytest.raises(NotImplementedError):
        sr.SoftRenderer(camera_mode='projection')
    with pytest.raises(NotImplementedError):
        sr.SoftRenderer(camera_mode='look_at', dist_func='barycentric')


def test_mesh_regularisers_and_obj_roundtrip(tmp_path):
    import soft_renderer as sr
    verts, faces = load_mesh(2)
    lap = sr.LaplacianLoss(verts, faces).cuda()
    flat = sr.FlattenLoss(faces).cuda()
    x = verts[None].clone().cuda()
    # a sphere is locally flat-ish and regular: both losses are small and grow under noise
    l0, f0 = float(lap(x)), float(flat(x))
    xn = x + 0.05 * torch.randn(x.shape, generator=torch.Generator().manual_seed(0)).cuda()
    assert float(lap(xn)) > l0 and float(flat(xn)) > f0
    # closed manifold: 3F/2 edges, each seen from both ends by the Laplacian; the flatten loss lists
    # the upstream module's subset (edges at corner positions (0,1) or (1,2) of some face)
    assert lap.nbr.numel() == 2 * (3 * faces.shape[0] // 2)
    assert flat.v0s.shape[0] == 410 < 3 * faces.shape[0] // 2
    path = os.path.join(str(tmp_path), "m.obj")
    sr.save_obj(path, verts, faces)
    v2, f2 = sr.load_obj(path)
    assert torch.allclose(v2.cpu(), verts, atol=1e-6) and torch.equal(f2.cpu(), faces)
    m = sr.Mesh.from_obj(path)
    assert m.vertices.shape == (1, 162, 3) and m.faces.shape == (1, 320, 3)


def test_teapot_silhouette_fit_descends():
    """The reference's teapot_deform_test (softras.py:114-156) as an asserting test: Adam on
    the vertices against misc/examples/data/example2_ref.png must lower the MSE."""
    from nnutils.softras import SoftNeuralRenderer
    t = np.load(os.path.join(GOLDEN, "teapot.npz"))
    verts = torch.from_numpy(t["verts"])
    verts = verts - verts.mean(0, keepdim=True)
    verts = (verts / verts.abs().max() * 0.6)[None].cuda()
    faces = torch.from_numpy(t["faces"]).long()[None].cuda()
    ref = torch.from_numpy(t["example2_ref"].astype(np.float32))[None].cuda()
    cams = torch.tensor([[1., 0, 0, 1, 0, 0, 0]]).cuda()
    renderer = SoftNeuralRenderer(img_size=256)
    param = torch.nn.Parameter(verts.clone())
    opt = torch.optim.Adam([param], lr=1e-3, betas=(0.9, 0.999))
    losses = []
    for _ in range(40):
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(renderer(param, faces, cams), ref)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert np.isfinite(losses).all()
    assert losses[-1] < 0.97 * losses[0], losses[::8]
    assert float((param - verts).abs().max()) > 1e-3
