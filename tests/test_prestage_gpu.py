"""GPU tests of the renderer call's input stage as HIP ops: camera + y flip + depth shift +
`vertices[faces]` gather (nnutils/softras.py:58-61, nnutils/geom_utils.py:63-80,143-161) forward
and backward, the sorted gather-form scatter, and the renderer call that folds the stage into
the rasterizer's setup kernel.

Pinned by tests/golden/proj_vectors.npz (outputs of the reference's own geom_utils, captured by
import) and by fp64 torch autograd of the same formulas.
"""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, load_mesh, random_cams, random_vertices, rel_err
from oracle import softras_oracle

pytestmark = pytest.mark.gpu

SCALE, SHIFT = (1.0, -1.0, 1.0), (0.0, 0.0, 2.732)


def _reference_stage(v64, cams64, faces, offset_z=5.0):
    proj = softras_oracle.orthographic_proj_withz(v64, cams64, offset_z=offset_z)
    proj = proj * torch.tensor(SCALE, dtype=torch.float64) + torch.tensor(SHIFT, dtype=torch.float64)
    B = v64.shape[0]
    return proj[torch.arange(B)[:, None, None], faces.long().expand(B, -1, -1)]


def test_project_gather_matches_reference_projection_vectors():
    """The camera against outputs of the reference's orthographic_proj_withz (offset_z 5 and 0)."""
    import sv3d
    z = np.load(os.path.join(GOLDEN, "proj_vectors.npz"))
    X, cams = torch.from_numpy(z["X"]), torch.from_numpy(z["cams"])
    idx = torch.arange(36, dtype=torch.int32).view(1, 12, 3)
    idx = torch.cat([idx, torch.tensor([[[36, 0, 35]]], dtype=torch.int32)], dim=1)
    for key, off in (("withz_off5", 5.0), ("withz_off0", 0.0)):
        out = sv3d.project_gather(X.cuda(), cams.cuda(), idx.cuda(), offset_z=off)
        want = torch.from_numpy(z[key])[:, idx[0].long()]
        assert rel_err(out.cpu(), want) < 1e-6


@pytest.mark.parametrize("shared_topology", [True, False])
def test_project_gather_and_backward_match_fp64_autograd(shared_topology):
    import sv3d
    verts, faces = load_mesh(2)
    B = 3
    v = random_vertices(verts, B, 0)
    cams = random_cams(B, 1)
    cams[:, 3:] *= torch.tensor([[1.0], [0.8], [1.3]])          # the quaternion is used un-normalised
    fb = faces[None] if shared_topology else torch.stack([faces, faces.flip(0), faces.roll(5, 0)])
    vg, cg = v.cuda().requires_grad_(True), cams.cuda().requires_grad_(True)
    out = sv3d.project_gather(vg, cg, fb.cuda(), 5.0, SCALE, SHIFT)
    g = torch.randn(out.shape, generator=torch.Generator().manual_seed(2))
    (out * g.cuda()).sum().backward()
    vo, co = v.double().requires_grad_(True), cams.double().requires_grad_(True)
    want = _reference_stage(vo, co, fb)
    (want * g.double()).sum().backward()
    assert rel_err(out.detach().cpu(), want.detach()) < 1e-6
    assert rel_err(vg.grad.cpu(), vo.grad) < 1e-5
    assert rel_err(cg.grad.cpu(), co.grad) < 1e-5


def _random_topology(V, F, seed, fan=0):
    g = torch.Generator().manual_seed(seed)
    f = torch.randint(0, V, (F, 3), generator=g, dtype=torch.int32)
    if fan:
        f[:fan, 0] = 7                                            # one vertex with a very high valence
    return f


@pytest.mark.parametrize("V,F,fan", [(2562, 5120, 0),            # tables above the default LDS grant
                                     (300, 2000, 1500),          # long bucket
                                     (20000, 40000, 0)])         # tables exceed the LDS: scan kernels
def test_scatter_variants_match_index_add_and_repeat_bitwise(V, F, fan):
    import sv3d
    B = 2
    if (V, F) == (2562, 5120):
        _, faces = load_mesh(4)
    else:
        faces = _random_topology(V, F, 3, fan)
    gen = torch.Generator().manual_seed(4)
    v = torch.randn(B, V, 3, generator=gen)
    cams = random_cams(B, 5)
    g = torch.randn(B, F, 3, 3, generator=gen)
    fb = faces[None].cuda()
    # plain gather backward
    x = v.cuda().requires_grad_(True)
    (sv3d.face_gather(x, fb, scale=SCALE, shift=SHIFT) * g.cuda()).sum().backward()
    want = torch.zeros(B, V, 3, dtype=torch.float64)
    want.index_add_(1, faces.long().reshape(-1), (g.double() * torch.tensor(SCALE, dtype=torch.float64)).reshape(B, -1, 3))
    assert rel_err(x.grad.cpu(), want) < 1e-6
    first = x.grad.clone()
    x.grad = None
    (sv3d.face_gather(x, fb, scale=SCALE, shift=SHIFT) * g.cuda()).sum().backward()
    assert torch.equal(first, x.grad)
    # camera + gather backward
    vg, cg = v.cuda().requires_grad_(True), cams.cuda().requires_grad_(True)
    (sv3d.project_gather(vg, cg, fb, 5.0, SCALE, SHIFT) * g.cuda()).sum().backward()
    vo, co = v.double().requires_grad_(True), cams.double().requires_grad_(True)
    (_reference_stage(vo, co, faces[None]) * g.double()).sum().backward()
    assert rel_err(vg.grad.cpu(), vo.grad) < 1e-5
    assert rel_err(cg.grad.cpu(), co.grad) < 1e-5
    g1, c1 = vg.grad.clone(), cg.grad.clone()
    vg.grad = cg.grad = None
    (sv3d.project_gather(vg, cg, fb, 5.0, SCALE, SHIFT) * g.cuda()).sum().backward()
    assert torch.equal(g1, vg.grad) and torch.equal(c1, cg.grad)


@pytest.mark.parametrize("V,F", [(300, 2000), (20000, 40000)])      # sorted scatter; scan-form fallback
def test_out_of_range_face_indices_are_clamped_the_same_way_forward_and_backward(V, F):
    """Indices outside [0, V-1] read vertex 0 or V-1 in the gathers; both scatter forms must send
    the gradient of such a corner to that same vertex (the backward is the gather's transpose)."""
    import sv3d
    faces = _random_topology(V, F, 3, 0)
    bad = faces.clone()
    bad[5, 1], bad[17, 0], bad[F - 1, 2] = V + 7, -3, 10 * V
    clamped = bad.clamp(0, V - 1)
    gen = torch.Generator().manual_seed(9)
    v = torch.randn(2, V, 3, generator=gen).cuda()
    g = torch.randn(2, F, 3, 3, generator=gen).cuda()
    outs = []
    for f in (bad, clamped):
        x = v.clone().requires_grad_(True)
        out = sv3d.face_gather(x, f[None].cuda(), scale=SCALE, shift=SHIFT)
        (out * g).sum().backward()
        outs.append((out.detach(), x.grad))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("mode", ["mask", "rgb_lit", "rgb_ambient", "vertex"])
def test_fused_renderer_call_matches_staged_call(mode):
    """SoftNeuralRenderer through the fused ops against the same call assembled from separate
    stages (torch projection, Mesh, Lighting, Transform, rasterizer op)."""
    from nnutils.softras import SoftNeuralRenderer
    B, img = 2, 32
    verts, faces = load_mesh(2)
    v, cams = random_vertices(verts, B, 0), random_cams(B, 1)
    f = faces.long()[None].repeat(B, 1, 1).cuda()
    gen = torch.Generator().manual_seed(3)
    tex = None
    if mode.startswith("rgb"):
        tex = torch.rand(B, f.shape[1], 9, 3, generator=gen)
    elif mode == "vertex":
        tex = torch.rand(B, v.shape[1], 3, generator=gen)
    r = SoftNeuralRenderer(img_size=img, texture_type="vertex" if mode == "vertex" else "surface")
    if mode == "rgb_ambient":
        r.ambient_light_only()
    assert r._camera_stage() is not None
    res = []
    for fused in (True, False):
        vg, cg = v.cuda().requires_grad_(True), cams.cuda().requires_grad_(True)
        tg = tex.cuda().requires_grad_(True) if tex is not None else None
        if fused:
            out = r._forward_fused(vg, f, cg, tg, *r._camera_stage())
        else:
            out = r._forward_staged(vg, f, cg, tg)
        w = torch.rand(out.shape, generator=torch.Generator().manual_seed(9)).cuda()
        (out * w).sum().backward()
        res.append((out.detach().cpu(), vg.grad.cpu(), cg.grad.cpu(), tg.grad.cpu() if tg is not None else None))
    a, b = res
    # the two differ by fp32 rounding of the projected vertices, amplified by the 1/sigma slope
    assert rel_err(a[0], b[0]) < 2e-4
    assert rel_err(a[1], b[1]) < 2e-4
    assert rel_err(a[2], b[2]) < 2e-4
    if tex is not None:
        assert rel_err(a[3], b[3]) < 2e-4


def test_other_camera_transforms_take_the_staged_path():
    from nnutils.softras import SoftNeuralRenderer
    r = SoftNeuralRenderer(img_size=16)
    r.renderer.transform.eye = [0.5, 0, -2.732]
    assert r._camera_stage() is None
    verts, faces = load_mesh(1)
    out = r(verts[None].cuda() * 0.5, faces.long()[None].cuda(), torch.tensor([[1., 0, 0, 1, 0, 0, 0]]).cuda())
    assert out.shape == (1, 16, 16) and torch.isfinite(out).all()


def test_render_forward_writes_face_vertices_and_validates():
    import ctypes
    from sv3d import _lib as L
    verts, faces = load_mesh(1)
    B, F = 2, faces.shape[0]
    v = random_vertices(verts, B, 0).cuda()
    cams = random_cams(B, 1).cuda()
    fb = faces[None].cuda().contiguous()
    p = L.make_params(B, F, 16, 1, 0, 1, True, (0, 0, 0), 1, 100, 1e-3, 1e-4, 1e-10, 1e-4)
    rec_b, saved_b, _ = L.raster_sizes(p)
    rgba = torch.empty(B, 4, 16, 16, device="cuda")
    rec = torch.empty(rec_b // 4, device="cuda")
    saved = torch.empty(saved_b // 4, device="cuda")
    fv = torch.empty(B, F, 3, 3, device="cuda")
    m = L.make_mesh_input(v, cams, fb, 5.0, SCALE, SHIFT)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(L.lib().sv3d_render_forward(ctypes.byref(p), ctypes.byref(m), None, vp(fv), vp(rgba), vp(rec),
                                        vp(saved), st), "render")
    want = _reference_stage(v.cpu().double(), cams.cpu().double(), faces[None])
    assert rel_err(fv.cpu(), want) < 1e-6
    import sv3d
    two_call = sv3d.soft_rasterize(fv, None, 16, sigma_val=1e-4)
    assert rel_err(rgba.cpu(), two_call.cpu()) < 1e-6
    assert L.lib().sv3d_render_forward(ctypes.byref(p), None, None, vp(fv), vp(rgba), vp(rec), vp(saved), st) == -1
    m.faces_batch = 5
    assert L.lib().sv3d_render_forward(ctypes.byref(p), ctypes.byref(m), None, vp(fv), vp(rgba), vp(rec),
                                       vp(saved), st) == -1
    assert L.lib().sv3d_project_gather(ctypes.byref(m), B, F, vp(fv), st) == -1


def test_shared_pass_renders_equal_the_separate_calls():
    """forward_with_mask / render_parts against the 2 + Np separate calls of main.py:417-426."""
    from nnutils.softras import SoftNeuralRenderer
    B, img, Np = 2, 32, 7
    verts, faces = load_mesh(2)
    v, cams = random_vertices(verts, B, 0).cuda(), random_cams(B, 1).cuda()
    f = faces.long()[None].repeat(B, 1, 1).cuda()
    gen = torch.Generator().manual_seed(3)
    tex = torch.rand(B, f.shape[1], 16, 3, generator=gen).cuda()
    parts = torch.softmax(torch.randn(B, v.shape[1], Np, generator=gen), dim=2).cuda()
    mask_r, tex_r = SoftNeuralRenderer(img), SoftNeuralRenderer(img)
    tex_r.ambient_light_only()
    seg_r = SoftNeuralRenderer(img, texture_type='vertex')

    def run(shared):
        vg, cg = v.clone().requires_grad_(True), cams.clone().requires_grad_(True)
        tg, pg = tex.clone().requires_grad_(True), parts.clone().requires_grad_(True)
        if shared:
            rgb, mask = tex_r.forward_with_mask(vg, f, cg, tg)
            seg = seg_r.render_parts(vg, f, cg, pg)
        else:
            mask = mask_r(vg, f, cg)
            rgb = tex_r(vg, f, cg, textures=tg)
            seg = torch.cat([seg_r(vg, f, cg, pg[:, :, k:k + 1].repeat(1, 1, 3), True)[:, 0:1] for k in range(Np)], 1)
        wg = torch.Generator().manual_seed(11)
        loss = (mask * torch.rand(mask.shape, generator=wg).cuda()).sum() \
            + (rgb * torch.rand(rgb.shape, generator=wg).cuda()).sum() \
            + (seg * torch.rand(seg.shape, generator=wg).cuda()).sum()
        loss.backward()
        return [t.detach().cpu() for t in (mask, rgb, seg, vg.grad, cg.grad, tg.grad, pg.grad)]

    a, b = run(True), run(False)
    assert a[2].shape == (B, Np, img, img)
    for x, y in zip(a[:3], b[:3]):
        assert rel_err(x, y) < 1e-6                 # same images
    for x, y in zip(a[3:], b[3:]):
        assert rel_err(x, y) < 1e-5                 # same gradients, summed in a different order
    with pytest.raises(ValueError):
        tex_r.render_parts(v, f, cams, parts)


@pytest.mark.gpu
@pytest.mark.parametrize("ambient", [True, False])
def test_texture_sets_shared_by_camera_hypotheses(ambient):
    """main.py:413-418 renders `num_cams` camera hypotheses of the same meshes with
    `torch.cat([textures] * num_cams)`; passing the TB texture sets once must give the same image
    and the gradient summed over the hypotheses."""
    from nnutils.softras import SoftNeuralRenderer
    verts, faces = load_mesh(3)
    TB, Nc, T = 2, 3, 4
    B = TB * Nc
    g = torch.Generator().manual_seed(5)
    v = random_vertices(verts, TB, seed=2).repeat(Nc, 1, 1).cuda()
    cams = random_cams(B, seed=4).cuda()
    f = faces[None].expand(B, -1, -1).cuda()
    tex = torch.rand(TB, faces.shape[0], T * T, 3, generator=g)
    w = torch.rand(B, 3, 64, 64, generator=g).cuda()
    r = SoftNeuralRenderer(img_size=64)
    if ambient:
        r.ambient_light_only()
    t_cat = tex.cuda().requires_grad_(True)
    img_cat = r.forward(v, f, cams, torch.cat([t_cat] * Nc))
    (img_cat * w).sum().backward()
    t_set = tex.cuda().requires_grad_(True)
    img_set = r.forward(v, f, cams, t_set)
    (img_set * w).sum().backward()
    assert torch.equal(img_set, img_cat)
    assert rel_err(t_set.grad, t_cat.grad) < 1e-6          # same terms, summed in another order
    with pytest.raises(ValueError):
        r.forward(v, f, cams, tex[:1].repeat(4, 1, 1, 1).cuda())     # 4 does not divide 6


@pytest.mark.gpu
def test_fan_mesh_with_a_hub_vertex_in_every_face():
    """A fan of 4000 faces around vertex 0: the hub's bucket holds a third of all corner slots, so
    its gradient is gathered by a scan of the face array instead of the quadratic bucket walk."""
    import sv3d
    F = 4000
    idx = torch.arange(F, dtype=torch.int32)
    faces = torch.stack([torch.zeros(F, dtype=torch.int32), 1 + idx, 1 + (idx + 1) % F], 1)[None].cuda()
    g = torch.Generator().manual_seed(0)
    values = torch.randn(2, F + 1, 3, generator=g).cuda().requires_grad_(True)
    w = torch.randn(2, F, 3, 3, generator=g).cuda()
    out = sv3d.face_gather(values, faces)
    (out * w).sum().backward()
    ref = values.detach().clone().requires_grad_(True)
    (ref[:, faces[0].long()] * w).sum().backward()
    assert torch.equal(out, ref[:, faces[0].long()])
    assert rel_err(values.grad.cpu(), ref.grad.cpu()) < 1e-6
