"""`sv3d_distance_transform_static`, the entry point that allocates nothing, called through the
C ABI and compared with `sv3d_distance_transform` (caller's workspace): bit-equal outputs
(integer squared distances, the same launches), on two calls in a row, the second smaller, so
that what the first call left in the library's per-GPU array cannot leak into the second.  Also
against scipy on the second call, and the size limit.
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import texture_oracle as to

pytestmark = pytest.mark.gpu

MAX_PIXELS = 4 * 1024 * 1024            # SV3D_DT_STATIC_MAX_PIXELS


def _masks(B, H, W, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    out = []
    for i in range(B):
        cy, cx = rng.uniform(0.3, 0.7) * H, rng.uniform(0.3, 0.7) * W
        r = rng.uniform(0.1, 0.35) * min(H, W)
        m = ((yy - cy) ** 2 + (xx - cx) ** 2 < r * r).astype(np.float32)
        m[rng.random((H, W)) < 0.002] = 1.0
        out.append(m)
    return torch.from_numpy(np.stack(out))


def _with_workspace(lib, m, barrier, k):
    B, H, W = m.shape
    out = torch.empty_like(m)
    n = ctypes.c_size_t()
    assert lib.sv3d_distance_transform_workspace(B, H, W, ctypes.byref(n)) == 0
    ws = torch.empty(n.value, dtype=torch.uint8, device=m.device)
    st = lib.sv3d_distance_transform(m.data_ptr(), B, H, W, barrier, k, out.data_ptr(), ws.data_ptr(), n.value,
                                     torch.cuda.current_stream().cuda_stream)
    assert st == 0
    return out


def _static(lib, m, barrier, k, out):
    B, H, W = m.shape
    return lib.sv3d_distance_transform_static(m.data_ptr(), B, H, W, barrier, k, out.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream)


def test_static_scratch_distance_transform_equals_the_workspace_one():
    from sv3d import _lib as L
    lib = L.lib()
    big = _masks(12, 256, 256, seed=1).cuda()
    small = _masks(3, 96, 72, seed=2).cuda()
    want_big, want_small = _with_workspace(lib, big, 1, 50.0), _with_workspace(lib, small, 1, 50.0)
    want_small_plain = _with_workspace(lib, small, 0, 0.0)
    out_big, out_small, out_small_plain = torch.empty_like(big), torch.empty_like(small), torch.empty_like(small)
    torch.cuda.synchronize()

    before = torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
    assert _static(lib, big, 1, 50.0, out_big) == 0
    assert _static(lib, small, 1, 50.0, out_small) == 0          # second call, smaller input, same array
    assert _static(lib, small, 0, 0.0, out_small_plain) == 0
    torch.cuda.synchronize()
    assert (torch.cuda.memory_allocated(), torch.cuda.memory_reserved()) == before

    assert torch.equal(out_big, want_big)
    assert torch.equal(out_small, want_small)
    assert torch.equal(out_small_plain, want_small_plain)
    got = out_small.cpu().numpy()
    for i in range(small.shape[0]):
        assert np.abs(got[i] - to.compute_dt_barrier(small[i].cpu().numpy())).max() < 2e-6

    # the same call replayed from a HIP graph: capture forbids allocation, and the call makes none
    g = torch.cuda.CUDAGraph()
    out_graph = torch.empty_like(small)
    with torch.cuda.graph(g):
        assert _static(lib, small, 1, 50.0, out_graph) == 0
    out_graph.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_graph, want_small)


def test_static_scratch_rejects_sizes_beyond_its_array():
    from sv3d import _lib as L
    lib = L.lib()
    m = torch.zeros(1, 8, 8, device="cuda")
    out = torch.empty_like(m)
    B = MAX_PIXELS // (256 * 256) + 1
    st = lib.sv3d_distance_transform_static(m.data_ptr(), B, 256, 256, 1, 50.0, out.data_ptr(), 0)
    assert st == -3 and b"small" in lib.sv3d_strerror(st)
    assert lib.sv3d_distance_transform_static(0, 1, 8, 8, 1, 50.0, out.data_ptr(), 0) == -1
    # exactly the limit is accepted
    full = torch.zeros(MAX_PIXELS // (256 * 256), 256, 256, device="cuda")
    full[:, 100:150, 80:200] = 1
    out_full = torch.empty_like(full)
    assert _static(lib, full, 1, 50.0, out_full) == 0
    assert torch.equal(out_full, _with_workspace(lib, full, 1, 50.0))
