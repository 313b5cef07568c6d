#!/usr/bin/env python3
"""One Chamfer forward size launched repeatedly through the C ABI (the program rocprofv3 wraps in
tools/profile_chamfer.sh):  python tools/chamfer_prof_run.py CLOUDS POINTS [LAUNCHES]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "single-view-3d-reconstruction_amd"))
import torch

from sv3d import _lib as L

B, N = int(sys.argv[1]), int(sys.argv[2])
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
a, b = torch.randn(B, N, 3, generator=g).to(dev), torch.randn(B, N, 3, generator=g).to(dev)
d1, d2 = torch.empty(B, N, device=dev), torch.empty(B, N, device=dev)
i1, i2 = (torch.empty(B, N, dtype=torch.int32, device=dev) for _ in range(2))
vp = lambda t: ctypes.c_void_p(t.data_ptr())
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for _ in range(launches):
    L.check(L.lib().sv3d_chamfer_forward(vp(a), vp(b), B, N, N, vp(d1), vp(d2), vp(i1), vp(i2), stream), "chamfer")
torch.cuda.synchronize()
