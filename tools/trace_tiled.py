#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase times of the tiled silhouette forward from a -DSV3D_TRACE build.

    python tools/trace_tiled.py gpurun_variants/trace.so [config]
Stamps are thread 0's (the first wave of the workgroup): start, list length here, records built
(after the barrier), its visits done, end."""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "single-view-3d-reconstruction_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

import bench
import kbench
from sv3d import _lib as L

lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
cname = sys.argv[2] if len(sys.argv) > 2 else "c2"
B, img, aa, sub, sigma, mode, T = kbench.CONFIGS[cname]
dev = torch.device("cuda:0")
ns = argparse.Namespace(batch=B, image_size=img, subdivide=sub, mode=mode, tex_size=T)
fv, tex, grad = bench.build_inputs(ns, dev, seed=1000)
S = img * (2 if aa else 1)
p = L.make_params(B, fv.shape[1], S, 2 if aa else 1, 0, T, True, (0, 0, 0), 1, 100, 1e-3, sigma, 1e-10, 1e-4)
p.alpha_only = 1
_a, _b, _c = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
lib.sv3d_raster_sizes.argtypes = [ctypes.POINTER(L.RasterParams)] + [ctypes.POINTER(ctypes.c_size_t)] * 3
assert lib.sv3d_raster_sizes(ctypes.byref(p), ctypes.byref(_a), ctypes.byref(_b), ctypes.byref(_c)) == 0
f32 = dict(dtype=torch.float32, device=dev)
rgba = torch.empty((B, 4, img, img), **f32)
rec, saved = (torch.empty(n // 4 + 4, **f32) for n in (_a.value, _b.value))
vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
lib.sv3d_raster_forward.argtypes = [ctypes.POINTER(L.RasterParams)] + [ctypes.c_void_p] * 6
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for _ in range(20):
    assert lib.sv3d_raster_forward(ctypes.byref(p), vp(fv), None, vp(rgba), vp(rec), vp(saved), stream) == 0
torch.cuda.synchronize()
n = min(((S + 15) // 16) ** 2 * B, 16384)
pb = np.zeros(8 * 16384, dtype=np.uint64)
lib.sv3d_debug_read_phase.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.sv3d_debug_read_phase(pb.ctypes.data_as(ctypes.c_void_p), pb.nbytes) == 0
ph = pb[: n * 8].reshape(n, 8).astype(np.int64)
nl = ph[:, 7]
t = ph[:, :5] / 100.0
base = t[:, 0].min()
full = nl > 0
print("workgroups traced %d, non-empty %d, list length mean %.1f max %d" % (n, full.sum(), nl[full].mean(), nl.max()))
print("kernel span (first start to last end) %.2f us" % (t[:, 4].max() - base))
st = np.sort(t[:, 0] - base)
print("workgroups started by t us: " + "  ".join("%g:%d" % (x, int((st <= x).sum())) for x in (0.5, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12)))
en = np.sort(t[:, 4] - base)
print("workgroups finished by t us: " + "  ".join("%g:%d" % (x, int((en <= x).sum())) for x in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14)))
span = t[:, 4].max() - base
if span > 16:
    grid = np.linspace(0, span, 13)[1:]
    print("started by:  " + "  ".join("%.0f:%d" % (x, int((st <= x).sum())) for x in grid))
    print("finished by: " + "  ".join("%.0f:%d" % (x, int((en <= x).sum())) for x in grid))
names = ["start -> list length here", "-> records built (barrier)", "-> first wave's visits done", "-> end"]
for i in range(4):
    d = t[full, i + 1] - t[full, i]
    print("%-32s non-empty: mean %.2f p50 %.2f p90 %.2f max %.2f us" % (names[i], d.mean(), np.median(d), np.percentile(d, 90), d.max()))
d = t[full, 4] - t[full, 0]
print("lifetime non-empty: mean %.2f p90 %.2f max %.2f us; empty: mean %.2f us" % (d.mean(), np.percentile(d, 90), d.max(), (t[~full, 4] - t[~full, 0]).mean() if (~full).any() else 0))

hw = pb[: n * 8].reshape(n, 8)[:, 5]
if hw.any():
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xf
    hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
    cu = (hwid >> 8) & 0xf
    sh = (hwid >> 12) & 0x1
    se = (hwid >> 13) & 0x7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    uk = np.unique(key)
    load = np.array([nl[key == k].sum() for k in uk], dtype=np.float64)
    cnt = np.array([(key == k).sum() for k in uk])
    fin = np.array([t[key == k, 4].max() - base for k in uk])
    print("CUs seen %d; workgroups per CU min %d mean %.1f max %d" % (len(uk), cnt.min(), cnt.mean(), cnt.max()))
    print("listed faces per CU (sum over its tiles): min %.0f mean %.0f p90 %.0f max %.0f" % (load.min(), load.mean(), np.percentile(load, 90), load.max()))
    print("CU finish time: min %.1f mean %.1f max %.1f us; correlation with its listed faces %.2f" % (fin.min(), fin.mean(), fin.max(), np.corrcoef(load, fin)[0, 1]))
    order = np.argsort(t[:, 0])
    first = order[:64]
    print("first 64 workgroups started (wg id: cu key): " + " ".join("%d:%d" % (i, key[i]) for i in first[:32]))
    if os.environ.get("SV3D_TRACE_PLACEMENT"):
        for k in uk[:6]:
            ids = np.nonzero(key == k)[0]
            print("cu key %d (xcc %d): workgroups %s  lists %s  starts %s" % (k, xcc[ids[0]], ids.tolist(), nl[ids].tolist(),
                  np.round(t[ids, 0] - base, 1).tolist()))
        print("xcc of workgroup id mod 8:", [int(np.bincount(xcc[np.arange(n) % 8 == r], minlength=8).argmax()) for r in range(8)])
