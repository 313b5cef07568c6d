#!/usr/bin/env python3
"""Diagnostic: per-workgroup timeline of the forward kernel from a -DSV3D_TRACE build.

    python tools/trace_fwd.py gpurun_variants/v5_trace.so [config]
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "single-view-3d-reconstruction_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import argparse

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
# buffer sizes from the library under test itself (a build may keep more in them than the in-tree one)
_a, _b, _c = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
lib.sv3d_raster_sizes.argtypes = [ctypes.POINTER(L.RasterParams)] + [ctypes.POINTER(ctypes.c_size_t)] * 3
assert lib.sv3d_raster_sizes(ctypes.byref(p), ctypes.byref(_a), ctypes.byref(_b), ctypes.byref(_c)) == 0
rec_b, saved_b, ws_b = max(_a.value, L.raster_sizes(p)[0]), _b.value, _c.value
f32 = dict(dtype=torch.float32, device=dev)
rgba = torch.empty((B, 4, img, img), **f32)
rec, saved = (torch.empty(n // 4, **f32) for n in (rec_b, saved_b))
vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
lib.sv3d_raster_forward.argtypes = [ctypes.POINTER(L.RasterParams)] + [ctypes.c_void_p] * 6
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for _ in range(5):
    assert lib.sv3d_raster_forward(ctypes.byref(p), vp(fv), vp(tex), vp(rgba), vp(rec), vp(saved), stream) == 0
torch.cuda.synchronize()
ntile = ((S + 15) // 16) ** 2 * B
n = min(ntile, 4096)
buf = np.zeros(4 * 16384, dtype=np.uint64)
lib.sv3d_debug_read_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.sv3d_debug_read_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
t = buf[: n * 16].reshape(n, 4, 4)
t0 = t[..., 0].astype(np.int64)
t1 = t[..., 1].astype(np.int64)
base = t0.min()
start = (t0 - base) / 100.0          # s_memrealtime ticks at 100 MHz -> microseconds
end = (t1 - base) / 100.0
dur = end - start
visits = t[..., 3].astype(np.int64)
hw = t[..., 2]
xcc = (hw >> np.uint64(32)).astype(np.int64)
hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
cu = (hwid >> 8) & 0xf
sh = (hwid >> 12) & 0x1
se = (hwid >> 13) & 0x7
simd = (hwid >> 4) & 0x3
print("workgroups traced", n, "kernel span %.1f us" % end.max())
print("wave start: min %.1f p50 %.1f p99 %.1f max %.1f us" % (start.min(), np.median(start), np.percentile(start, 99), start.max()))
wg_start = np.sort(start.min(1))
print("workgroups started by t us: " + "  ".join("%g:%d" % (x, int((wg_start <= x).sum())) for x in (0.5, 1, 1.5, 2, 3, 4, 5, 6, 8, 10, 12, 14)))
wg_end = np.sort(end.max(1))
print("workgroups finished by t us: " + "  ".join("%g:%d" % (x, int((wg_end <= x).sum())) for x in (2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22)))
print("wave dur  : mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f us" % (dur.mean(), np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
print("visits/wave: mean %.1f max %d ; correlation dur~visits %.2f" % (visits.mean(), visits.max(), np.corrcoef(dur.ravel(), visits.ravel())[0, 1]))
wg_sum, wg_max = visits.sum(1), visits.max(1)
busy = wg_max > 0
print("visits per workgroup: mean %.1f; sum / (waves x max) over non-empty workgroups: %.2f (1 = waves of a workgroup evenly loaded)"
      % (wg_sum.mean(), (wg_sum[busy] / (visits.shape[1] * wg_max[busy])).mean()))
print("sum over workgroups of waves x max visits vs sum of visits: %.2f" % ((visits.shape[1] * wg_max).sum() / max(wg_sum.sum(), 1)))
z = visits.ravel() == 0
print("waves with zero visits: %.0f%%, their mean dur %.2f us" % (100 * z.mean(), dur.ravel()[z].mean() if z.any() else 0))
fit = np.polyfit(visits.ravel(), dur.ravel(), 1)
print("linear fit: dur = %.3f us/visit * visits + %.2f us" % (fit[0], fit[1]))
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
uk = np.unique(key)
print("distinct (xcc,se,sh,cu) ids seen:", len(uk))
per = [(visits[key == k].sum(), end[key == k].max(), (key == k).sum()) for k in uk]
per = np.array(per, dtype=np.float64)
print("per-CU visits: mean %.0f max %.0f min %.0f ; per-CU finish time: mean %.1f max %.1f min %.1f us ; waves per CU mean %.1f max %.0f"
      % (per[:, 0].mean(), per[:, 0].max(), per[:, 0].min(), per[:, 1].mean(), per[:, 1].max(), per[:, 1].min(), per[:, 2].mean(), per[:, 2].max()))
ks = key * 4 + simd
us = np.unique(ks)
pv = np.array([visits[ks == k].sum() for k in us], dtype=np.float64)
print("per-SIMD visits (traced waves): mean %.0f max %.0f min %.0f p90 %.0f ; SIMDs seen %d ; waves per SIMD mean %.1f"
      % (pv.mean(), pv.max(), pv.min(), np.percentile(pv, 90), len(us), visits.size / len(us)))
for w in range(visits.shape[1]):
    print("  traced wave %d sits on SIMD: %s" % (w, np.bincount(simd[:, w], minlength=4).tolist()))
late = np.argsort(-end.max(1))[:5]
for i in late:
    print("  late wg %d: start %.1f end %.1f visits %s cu-key %d" % (i, start[i].min(), end[i].max(), visits[i].tolist(), key[i, 0]))
try:
    pb = np.zeros(8 * 4096, dtype=np.uint64)
    lib.sv3d_debug_read_phase.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert lib.sv3d_debug_read_phase(pb.ctypes.data_as(ctypes.c_void_p), pb.nbytes) == 0
    ph = pb[: n * 8].reshape(n, 8).astype(np.int64)
    names = ["start->binning begin", "box loads+ballots (to barrier)", "scan", "id write (first window)",
             "record staging", "pair loop", "epilogue"]
    prev = t0[:, 0]
    nz = visits.sum(1) > 0
    for i in range(7):
        cur = ph[:, i]
        d = (cur - prev) / 100.0
        ok = cur > 0
        print("phase %-32s all: mean %.2f us   non-empty tiles: mean %.2f p90 %.2f   empty: mean %.2f"
              % (names[i], d[ok].mean(), d[ok & nz].mean() if (ok & nz).any() else 0,
                 np.percentile(d[ok & nz], 90) if (ok & nz).any() else 0, d[ok & ~nz].mean() if (ok & ~nz).any() else 0))
        prev = np.where(ok, cur, prev)
except AttributeError:
    pass
if os.environ.get("SV3D_TRACE_PLACEMENT"):
    # where the dispatcher put each workgroup: (xcc, se, sh, cu) per workgroup index
    k0 = key[:, 0]
    print("placement of workgroups 0..95 (xcc.se.sh.cu):")
    for i0 in range(0, 96, 8):
        print("  " + "  ".join("%d.%d.%d.%02d" % (xcc[i, 0], se[i, 0], sh[i, 0], cu[i, 0]) for i in range(i0, i0 + 8)))
    # period: smallest d > 0 with key[i] == key[i + d] for most i among the first 1024
    for d in (8, 16, 32, 64, 128, 256, 512, 1024):
        if d < n:
            print("  same CU at distance %4d: %.0f%%" % (d, 100.0 * np.mean(k0[: n - d] == k0[d:n])))
    order = {}
    for i in range(n):
        order.setdefault(int(k0[i]), []).append(i)
    some = list(order.items())[:6]
    for kk, lst in some:
        print("  CU key %d got workgroups %s" % (kk, lst))
