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
