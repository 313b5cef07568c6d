#!/usr/bin/env python3
"""Diagnostic: per-wave timeline of the backward kernel from a -DSV3D_TRACE build (the stamp is
taken at the end of a wave's pixel loop, before its reduction).

    python tools/trace_bwd.py gpurun_variants/trace.so [config]
"""
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
m = {"none": 0, "surface": 1, "vertex": 2}[mode]
p = L.make_params(B, fv.shape[1], S, 2 if aa else 1, m, T, True, (0, 0, 0), 1, 100, 1e-3, sigma, 1e-10, 1e-4)
_a, _b, _c = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
lib.sv3d_raster_sizes.argtypes = [ctypes.POINTER(L.RasterParams)] + [ctypes.POINTER(ctypes.c_size_t)] * 3
assert lib.sv3d_raster_sizes(ctypes.byref(p), ctypes.byref(_a), ctypes.byref(_b), ctypes.byref(_c)) == 0
f32 = dict(dtype=torch.float32, device=dev)
rgba = torch.empty((B, 4, img, img), **f32)
rec, saved = (torch.empty(max(n, 4) // 4, **f32) for n in (_a.value, _b.value))
gfv = torch.empty_like(fv)
gtex = torch.empty_like(tex) if tex is not None else None
vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
lib.sv3d_raster_forward.argtypes = [ctypes.POINTER(L.RasterParams)] + [ctypes.c_void_p] * 6
lib.sv3d_raster_backward.argtypes = [ctypes.POINTER(L.RasterParams)] + [ctypes.c_void_p] * 7 + [ctypes.c_size_t, ctypes.c_void_p]
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for _ in range(5):
    assert lib.sv3d_raster_forward(ctypes.byref(p), vp(fv), vp(tex), vp(rgba), vp(rec), vp(saved), stream) == 0
    assert lib.sv3d_raster_backward(ctypes.byref(p), vp(tex), vp(rec), vp(saved), vp(grad), vp(gfv), vp(gtex), None, 0, stream) == 0
torch.cuda.synchronize()
nwave = min(B * fv.shape[1] * (2 if B * fv.shape[1] <= 16384 else 1), 16384)
buf = np.zeros(4 * 16384, dtype=np.uint64)
lib.sv3d_debug_read_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.sv3d_debug_read_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
t = buf[: nwave * 4].reshape(nwave, 4)
t0, t1 = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
base = t0.min()
start, end = (t0 - base) / 100.0, (t1 - base) / 100.0
dur = end - start
npix = (t[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
hw = t[:, 2]
xcc = (hw >> np.uint64(32)).astype(np.int64)
hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
cu, sh, se, simd = (hwid >> 8) & 0xf, (hwid >> 12) & 0x1, (hwid >> 13) & 0x7, (hwid >> 4) & 0x3
steps = (npix + 63) // 64
print("waves traced", nwave, "span %.1f us" % end.max())
print("wave start: p10 %.1f p50 %.1f p90 %.1f max %.1f us" % tuple(np.percentile(start, [10, 50, 90, 100])))
print("waves started by t us: " + "  ".join("%g:%d" % (x, int((start <= x).sum())) for x in (0.5, 1, 2, 3, 4, 6, 8, 10, 12, 14, 16)))
print("waves finished by t us: " + "  ".join("%g:%d" % (x, int((end <= x).sum())) for x in (2, 4, 6, 8, 10, 12, 14, 16, 18, 20)))
print("wave dur: mean %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f us" % (dur.mean(), *np.percentile(dur, [50, 90, 99, 100])))
print("pixels per face box: mean %.0f p50 %.0f p90 %.0f max %d ; 64-pixel steps per face mean %.1f max %d"
      % (npix.mean(), np.median(npix), np.percentile(npix, 90), npix.max(), steps.mean(), steps.max()))
fit = np.polyfit(steps, dur, 1)
print("linear fit: dur = %.3f us/step * steps(face) + %.2f us" % (fit[0], fit[1]))
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
uk = np.unique(key)
ps = np.array([steps[key == k].sum() for k in uk], dtype=np.float64)
pe = np.array([end[key == k].max() for k in uk])
print("SIMDs seen %d ; steps per SIMD mean %.0f max %.0f min %.0f ; SIMD finish mean %.1f max %.1f min %.1f us"
      % (len(uk), ps.mean(), ps.max(), ps.min(), pe.mean(), pe.max(), pe.min()))
late = np.argsort(-end)[:6]
for i in late:
    print("  late wave %d: start %.1f end %.1f npix %d" % (i, start[i], end[i], npix[i]))
ta = ((t[:, 3] >> np.uint64(32)) & np.uint64(0xffff)).astype(np.int64) / 100.0
tb = ((t[:, 3] >> np.uint64(48)) & np.uint64(0xffff)).astype(np.int64) / 100.0
for lo_, hi_ in ((0, min(8192, nwave)), (min(8192, nwave), nwave)):
    if hi_ > lo_:
        sl = slice(lo_, hi_)
        print("waves %d..%d: start -> record and box here %.2f us (p90 %.2f) ; -> first pixels here %.2f us (p90 %.2f) ; -> end of steps %.2f us"
              % (lo_, hi_, ta[sl].mean(), np.percentile(ta[sl], 90), tb[sl].mean(), np.percentile(tb[sl], 90), dur[sl].mean()))
