"""The `sv3d::*` PyTorch custom ops are registered consistently: schema, fake (meta) kernels and
autograd formulas agree with the real kernels (torch.library.opcheck), and the ops can be traced
with fake tensors (what torch.compile and DDP's graph inspection rely on)."""
import pytest
import torch

from helpers import load_mesh, random_cams, random_vertices

pytestmark = pytest.mark.gpu

TESTS = ("test_schema", "test_faketensor", "test_autograd_registration")


def _mesh(B=2):
    verts, faces = load_mesh(1)
    return random_vertices(verts, B, 0).cuda(), random_cams(B, 1).cuda(), faces[None].cuda().contiguous()


MESH_CFG = [5.0, 1.0, -1.0, 1.0, 0.0, 0.0, 2.732]        # offset_z, scale, shift
FCFG = [0.0, 0.0, 0.0, 1.0, 100.0, 1e-3, 1e-4, 1e-10, 1e-2]   # background, near, far, eps, sigma, dist_eps, gamma


def test_opcheck_raster_and_render_ops():
    from sv3d import ops
    v, cams, f = _mesh()
    fv = ops.project_gather(v, cams, f, 5.0, (1.0, -1.0, 1.0), (0.0, 0.0, 2.732))
    tex = torch.rand(2, f.shape[1], 4, 3, device="cuda")
    surface = [32, 1, 1, 2, 1, 0]                        # raster_size, aa, mode, res, double_side, alpha_only
    sv = torch.ops.sv3d
    torch.library.opcheck(sv.soft_rasterize.default, (fv.requires_grad_(True), tex.requires_grad_(True), surface, FCFG),
                          test_utils=TESTS)
    torch.library.opcheck(sv.soft_rasterize.default, (fv, None, [32, 1, 0, 1, 1, 0], FCFG), test_utils=TESTS)
    torch.library.opcheck(sv.soft_rasterize.default, (fv, None, [32, 2, 0, 1, 1, 1], FCFG), test_utils=TESTS)
    torch.library.opcheck(sv.render.default, (v.requires_grad_(True), cams.requires_grad_(True), f, tex, MESH_CFG,
                                              surface, FCFG), test_utils=TESTS)


def test_opcheck_mesh_and_chamfer_ops():
    v, cams, f = _mesh()
    sv = torch.ops.sv3d
    torch.library.opcheck(sv.face_gather.default, (v.requires_grad_(True), f, [1.0, -1.0, 1.0], [0.0, 0.0, 1.0]),
                          test_utils=TESTS)
    torch.library.opcheck(sv.project_gather.default, (v, cams.requires_grad_(True), f, MESH_CFG), test_utils=TESTS)
    parts = torch.rand(2, v.shape[1], 8, device="cuda", requires_grad=True)
    torch.library.opcheck(sv.face_gather_n.default, (parts, f), test_utils=TESTS)
    a = torch.randn(2, 50, 3, device="cuda", requires_grad=True)
    b = torch.randn(2, 70, 3, device="cuda", requires_grad=True)
    torch.library.opcheck(sv.chamfer_nn.default, (a, b), test_utils=TESTS)


def test_half_precision_inputs_compute_in_fp32_and_return_their_own_type():
    """Under autocast the trainer hands fp16 tensors to the renderer (train_utils.py:192): the
    kernels compute in fp32 and gradients come back in the inputs' types."""
    import sv3d
    v, cams, f = _mesh()
    vh = v.half().requires_grad_(True)
    out = sv3d.render(vh, cams, f, None, 32, offset_z=5.0, alpha_only=True)
    assert out.dtype == torch.float32
    out.sum().backward()
    assert vh.grad.dtype == torch.float16 and torch.isfinite(vh.grad).all()
    ref = sv3d.render(vh.detach().float(), cams, f, None, 32, offset_z=5.0, alpha_only=True)
    assert torch.equal(out.detach(), ref)


def test_ops_trace_with_fake_tensors():
    from torch._subclasses.fake_tensor import FakeTensorMode
    import sv3d
    with FakeTensorMode():
        v = torch.empty(3, 642, 3, device="cuda")
        cams = torch.empty(3, 7, device="cuda")
        f = torch.empty(1, 1280, 3, dtype=torch.int32, device="cuda")
        out = sv3d.render(v, cams, f, None, 64, anti_aliasing=True)
        assert out.shape == (3, 4, 64, 64)
        d1, d2, i1, i2 = sv3d.chamfer_nn(torch.empty(3, 10, 3, device="cuda"), torch.empty(3, 20, 3, device="cuda"))
        assert d2.shape == (3, 20) and i1.dtype == torch.int32


def test_opcheck_regulariser_and_texture_ops():
    from sv3d import meshreg, texture
    verts, faces = load_mesh(1)
    V = verts.shape[0]
    x = random_vertices(verts, 2, 3).cuda().requires_grad_(True)
    rs, nb = (torch.from_numpy(a).cuda() for a in meshreg.neighbour_csr(faces, V))
    torch.library.opcheck(torch.ops.sv3d.laplacian_loss.default, (x, rs, nb), test_utils=TESTS)
    q = meshreg.edge_quads(faces)
    inc_start, inc = (torch.from_numpy(a).cuda() for a in meshreg.incidence_csr(q, V))
    quads = torch.from_numpy(q).int().cuda()
    torch.library.opcheck(torch.ops.sv3d.flatten_loss.default, (x, quads, inc_start, inc, 1e-6), test_utils=TESTS)
    vals = torch.rand(1, nb.numel(), device="cuda")
    torch.library.opcheck(torch.ops.sv3d.mesh_spmv.default, (x, vals, rs, nb, vals, rs, nb), test_utils=TESTS)
    images = torch.rand(2, 3, 16, 16, device="cuda")
    flow = (2 * torch.rand(2, 40, 2, device="cuda") - 1).requires_grad_(True)
    torch.library.opcheck(torch.ops.sv3d.texture_sample.default, (images, flow), test_utils=TESTS)
    masks = (torch.rand(2, 24, 40, device="cuda") < 0.3).float()
    torch.library.opcheck(torch.ops.sv3d.distance_transform.default, (masks, True, 50.0), test_utils=TESTS)


def _eager_and_replayed(fn, inputs):
    """fn(*inputs) -> list of output tensors to which backward is applied through a fixed weight.
    Runs forward + backward eagerly, then captures the same call in a HIP graph on the same input
    tensors and replays it; returns (outputs, gradients) of both."""
    weights = None

    def step():
        for x in inputs:
            x.grad = None
        outs = fn(*inputs)
        nonlocal weights
        if weights is None:
            g = torch.Generator(device="cuda").manual_seed(7)
            weights = [torch.rand(o.shape, device="cuda", generator=g) - 0.5 for o in outs]
        torch.autograd.backward([(o * w).sum() for o, w in zip(outs, weights)])
        return outs

    outs = step()
    torch.cuda.synchronize()
    eager = ([o.detach().clone() for o in outs], [x.grad.detach().clone() for x in inputs])
    del outs
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    # Nothing allocated before the capture may be released inside it (this build of PyTorch-ROCm
    # ends such a capture in a segmentation fault, with plain torch operators as well): the
    # gradients and outputs of the passes above go now.
    for x in inputs:
        x.grad = None
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outs = step()
    for o in outs:                              # a capture runs nothing: what replay leaves must come from replay
        o.detach().fill_(float("nan"))
    for x in inputs:
        x.grad.fill_(float("nan"))
    graph.replay()
    torch.cuda.synchronize()
    replayed = ([o.detach().clone() for o in outs], [x.grad.detach().clone() for x in inputs])
    return eager, replayed


@pytest.mark.parametrize("case", ["render_silhouette", "raster_surface", "raster_vertex", "render_parts8", "chamfer"])
def test_operators_give_identical_bits_inside_and_outside_graph_capture(case):
    """The operators are deterministic and capture-safe by contract (no allocation outside torch's
    allocator, no host synchronisation, every sum in a fixed order): on identical inputs a call
    replayed from a HIP graph must return exactly the bits of the eager call, forward and backward."""
    import sv3d
    from sv3d import ops
    B = 4
    verts, faces = load_mesh(2)
    v = random_vertices(verts, B, 0).cuda().requires_grad_(True)
    cams = random_cams(B, 1).cuda().requires_grad_(True)
    f = faces[None].cuda().int().contiguous()
    F = f.shape[1]
    g = torch.Generator().manual_seed(3)
    kw = dict(offset_z=5.0, scale=(1.0, -1.0, 1.0), shift=(0.0, 0.0, 2.732))
    if case == "render_silhouette":
        inputs = [v, cams]
        fn = lambda v, c: [sv3d.render(v, c, f, None, 64, anti_aliasing=True, sigma_val=1e-5, alpha_only=True, **kw)]
    elif case == "raster_surface":
        tex = torch.rand(B, F, 16, 3, generator=g).cuda().requires_grad_(True)
        inputs = [v, cams, tex]
        fn = lambda v, c, t: [sv3d.soft_rasterize(ops.project_gather(v, c, f, 5.0, kw["scale"], kw["shift"]), t, 64,
                                                  anti_aliasing=True, sigma_val=1e-5, gamma_val=1e-4)]
    elif case == "raster_vertex":
        tex = torch.rand(B, F, 3, 3, generator=g).cuda().requires_grad_(True)
        inputs = [v, cams, tex]
        fn = lambda v, c, t: [sv3d.render(v, c, f, t, 64, texture_type="vertex", sigma_val=1e-4, gamma_val=1e-3, **kw)]
    elif case == "render_parts8":
        parts = torch.rand(B, verts.shape[0], 8, generator=g).cuda().requires_grad_(True)
        inputs = [v, cams, parts]
        fn = lambda v, c, p: [sv3d.render(v, c, f, ops.face_gather_n(p, f), 48, texture_type="vertex8",
                                          anti_aliasing=True, sigma_val=1e-5, **kw)]
    else:
        a = torch.randn(B, 642, 3, generator=g).cuda().requires_grad_(True)
        b = torch.randn(B, 700, 3, generator=g).cuda().requires_grad_(True)
        inputs = [a, b]
        fn = lambda a, b: list(sv3d.chamfer_nn(a, b)[:2])
    (eo, eg), (ro, rg) = _eager_and_replayed(fn, inputs)
    for x, y in zip(eo + eg, ro + rg):
        assert torch.isfinite(y).all() and x.abs().max() > 0
        assert torch.equal(x, y), float((x - y).abs().max())


def test_backward_operators_refuse_buffers_that_do_not_fit_the_configuration():
    """The *_backward operators are reachable as torch.ops.sv3d.*: buffers that are not what the
    forward of the same configuration produced must raise, not reach the kernels."""
    from sv3d import ops
    v, cams, f = _mesh()
    fv = ops.project_gather(v, cams, f, 5.0, (1.0, -1.0, 1.0), (0.0, 0.0, 2.732))
    B, F = fv.shape[0], fv.shape[1]
    icfg = [32, 1, 1, 2, 1, 0]
    tex = torch.rand(B, F, 4, 3, device="cuda")
    sv = torch.ops.sv3d
    rgba, rec, saved = sv.soft_rasterize(fv, tex, icfg, FCFG)
    g = torch.rand_like(rgba)
    gfv, gtex = sv.soft_rasterize_backward(g, tex, rec, saved, [B, F], icfg, FCFG, True)
    assert gfv.shape == fv.shape and gtex.shape == tex.shape and torch.isfinite(gfv).all()
    with pytest.raises((ValueError, RuntimeError), match="face_rec"):
        sv.soft_rasterize_backward(g, tex, rec[: rec.numel() // 2], saved, [B, F], icfg, FCFG, True)
    with pytest.raises((ValueError, RuntimeError), match="pixel_saved"):
        sv.soft_rasterize_backward(g, tex, rec, saved[:100], [B, F], icfg, FCFG, True)
    with pytest.raises((ValueError, RuntimeError), match="pixel_saved"):
        sv.soft_rasterize_backward(g, tex, rec, saved.double(), [B, F], icfg, FCFG, True)
    with pytest.raises((ValueError, RuntimeError), match="textures"):
        sv.soft_rasterize_backward(g, None, rec, saved, [B, F], icfg, FCFG, True)
    with pytest.raises((ValueError, RuntimeError), match="face_rec"):
        sv.soft_rasterize_backward(g, tex, rec.cpu(), saved, [B, F], icfg, FCFG, True)
    with pytest.raises((ValueError, RuntimeError), match="face_rec"):            # a larger batch than the buffers hold
        sv.soft_rasterize_backward(torch.rand(4 * B, 4, 32, 32, device="cuda"), tex.repeat(4, 1, 1, 1), rec, saved,
                                   [4 * B, F], icfg, FCFG, True)
    a, b = torch.randn(2, 50, 3, device="cuda"), torch.randn(2, 70, 3, device="cuda")
    d1, d2, i1, i2 = sv.chamfer_nn(a, b)
    with pytest.raises((ValueError, RuntimeError), match="idx2"):
        sv.chamfer_nn_backward(a, b, None, torch.ones_like(d2), i1, i2[:, :10].contiguous(), True, True)
    with pytest.raises((ValueError, RuntimeError), match="idx1"):
        sv.chamfer_nn_backward(a, b, None, torch.ones_like(d2), i1.long(), i2, True, True)
