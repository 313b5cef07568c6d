#!/usr/bin/env python3
"""Timing of the Chamfer entry points through the C ABI: python tools/chamfer_bench.py [other_build.so]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "single-view-3d-reconstruction_amd"))
import torch

from sv3d import _lib as L, workmodel

lib = ctypes.CDLL(os.path.abspath(sys.argv[1])) if len(sys.argv) > 1 else L.lib()
dev = torch.device("cuda:0")
vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for B, N, M in [(4, 642, 642), (8, 642, 642), (16, 642, 642), (32, 642, 642), (64, 642, 642), (4, 162, 162), (16, 1024, 1024), (8, 2048, 2048), (64, 2048, 2048), (2, 8192, 8192), (16, 8192, 8192),
                (64, 2562, 2562), (32, 642, 642), (4, 2048, 2048), (16, 2048, 2048), (4, 8192, 8192), (8, 2562, 2562)]:
    g = torch.Generator().manual_seed(5)
    a = torch.randn(B, N, 3, generator=g).to(dev)
    b = torch.randn(B, M, 3, generator=g).to(dev)
    d1, d2 = torch.empty(B, N, device=dev), torch.empty(B, M, device=dev)
    i1 = torch.empty(B, N, dtype=torch.int32, device=dev)
    i2 = torch.empty(B, M, dtype=torch.int32, device=dev)
    gd2 = torch.full((B, M), 1.0 / (B * M), device=dev)
    gd1 = torch.full((B, N), 1.0 / (B * N), device=dev)
    g1, g2 = torch.empty_like(a), torch.empty_like(b)
    iters = 20

    def fwd():
        assert lib.sv3d_chamfer_forward(vp(a), vp(b), B, N, M, vp(d1), vp(d2), vp(i1), vp(i2), stream) == 0

    def bwd_ref():        # the reference's use: only dist2, first cloud detached
        assert lib.sv3d_chamfer_backward(vp(a), vp(b), B, N, M, None, vp(gd2), vp(i1), vp(i2), None, vp(g2), stream) == 0

    def bwd_full():
        assert lib.sv3d_chamfer_backward(vp(a), vp(b), B, N, M, vp(gd1), vp(gd2), vp(i1), vp(i2), vp(g1), vp(g2), stream) == 0

    res = []
    for fn in (fwd, bwd_ref, bwd_full):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / iters * 1e3)
    fl = workmodel.chamfer_flops(B, N, M)
    print("B %3d N %5d M %5d : fwd %9.1f us (%6.2f TFLOP/s, %5.1f%% of FP32 peak)  bwd(dist2 only) %7.1f us  bwd(both) %8.1f us"
          % (B, N, M, res[0], fl / res[0] / 1e6, 100 * fl / res[0] / 1e6 / 157.3, res[1], res[2]), flush=True)
