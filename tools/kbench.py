#!/usr/bin/env python3
"""Kernel-level timing of the raster entry points for one or more builds of libsv3d_hip.so.

    python tools/kbench.py [--libs a.so,b.so] [--configs small,mid,...] [--rounds 5] [--iters 50]

For every (library, config) prints the median / min time of the dominant forward and backward
kernels (HIP events around them via the *_timed entry points) and of a whole fwd+bwd step.
Variants are interleaved round-robin inside one process so that A/B deltas are comparable.
"""
import argparse
import ctypes
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "single-view-3d-reconstruction_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

CONFIGS = {
    # name: (batch, image_size, aa, subdivide, sigma, mode, tex_size)
    "c2": (8, 256, False, 3, 1e-4, "none", 1),               # the headline benchmark configuration
    "c2b64": (64, 256, False, 3, 1e-4, "none", 1),
    "ref_mask": (64, 256, True, 3, 1e-5, "none", 1),          # reference defaults, Bn = 16*B, B = 4
    "ref_rgb": (64, 256, True, 3, 1e-5, "surface", 6),
    "ref_vert": (64, 256, True, 3, 1e-5, "vertex", 1),
    "hd_rgb": (64, 256, True, 4, 1e-5, "surface", 6),
    "c2_rgb": (8, 256, False, 3, 1e-4, "surface", 6),
    "ref_vert8": (64, 256, True, 3, 1e-5, "vertex8", 1),    # eight part maps in one pass
    "ref_mask_b8": (8, 256, True, 3, 1e-5, "none", 1),
    "ref_mask_b3": (3, 256, True, 3, 1e-5, "none", 1),
    "ref_mask_b4": (4, 256, True, 3, 1e-5, "none", 1),
    "ref_mask_b12": (12, 256, True, 3, 1e-5, "none", 1),
    "ref_rgb_b12": (12, 256, True, 3, 1e-5, "surface", 6),
    "hd_mask_b3": (3, 256, True, 4, 1e-5, "none", 1),
    "ref_mask_b6": (6, 256, True, 3, 1e-5, "none", 1),
    "hd_mask_b8": (8, 256, True, 4, 1e-5, "none", 1),
    "c2b40": (40, 256, False, 3, 1e-4, "none", 1),
    "ref_mask_b2": (2, 256, True, 3, 1e-5, "none", 1),
    "hd_mask_b2": (2, 256, True, 4, 1e-5, "none", 1),         # 5120 faces in a small launch
    # colour modes at 51 and 26 raster pixels per face: where the backward's row intervals start to pay
    "c2_rgb_b16": (16, 256, False, 3, 1e-4, "surface", 6),
    "c2_rgb_b24": (24, 256, False, 3, 1e-4, "surface", 6),
    "c2_vert_b16": (16, 256, False, 3, 1e-4, "vertex", 1),
    "c2b64_rgb": (64, 256, False, 3, 1e-4, "surface", 6),
    "c2b64_vert": (64, 256, False, 3, 1e-4, "vertex", 1),
    "hd_vert": (64, 256, True, 4, 1e-5, "vertex", 1),
    "hd_vert8": (64, 256, True, 4, 1e-5, "vertex8", 1),
    "hd_rgb_s384": (64, 192, True, 4, 1e-5, "surface", 6),
    "c2b64_rgb_sharp": (64, 256, False, 3, 1e-5, "surface", 6),
    "ref_rgb_b2": (2, 256, True, 3, 1e-5, "surface", 6),
    "s128_b32": (32, 128, False, 3, 1e-4, "none", 1),
    "s64_b64": (64, 64, False, 2, 1e-4, "none", 1),
    "s128_b128": (128, 128, False, 3, 1e-4, "none", 1),     # many faces per tile in a large launch
    "ref_rgb_b3": (3, 256, True, 3, 1e-5, "surface", 6),
    "c2_rgb_b12": (12, 256, False, 3, 1e-4, "surface", 6),
    "ref_rgb_b8": (8, 256, True, 3, 1e-5, "surface", 6),
    "ref_rgb_b16": (16, 256, True, 3, 1e-5, "surface", 6),
    # batch sweep at the benchmark configuration: either side of the kernels' launch-size thresholds
    # face size sweep for the backward's enumeration choice (pixels per face ~ S^2 / F)
    "s384": (32, 384, False, 3, 1e-4, "none", 1),
    "s384_rgb": (32, 384, False, 3, 1e-5, "surface", 6),
    "s256_f320": (64, 256, False, 2, 1e-4, "none", 1),
    "s1024_f5120": (16, 512, True, 4, 1e-5, "none", 1),
    "s320_rgb": (32, 320, False, 3, 1e-5, "surface", 6),
    "c2b2": (2, 256, False, 3, 1e-4, "none", 1),
    "c2b4": (4, 256, False, 3, 1e-4, "none", 1),
    "c2b5": (5, 256, False, 3, 1e-4, "none", 1),
    "c2b6": (6, 256, False, 3, 1e-4, "none", 1),
    "c2b12": (12, 256, False, 3, 1e-4, "none", 1),
    "c2b16": (16, 256, False, 3, 1e-4, "none", 1),
    "c2b20": (20, 256, False, 3, 1e-4, "none", 1),
    "c2b24": (24, 256, False, 3, 1e-4, "none", 1),
    "c2b28": (28, 256, False, 3, 1e-4, "none", 1),
    "c2b32": (32, 256, False, 3, 1e-4, "none", 1),
    "hd_mask_b4": (4, 256, True, 4, 1e-5, "none", 1),
    "hd_mask_b6": (6, 256, True, 4, 1e-5, "none", 1),
    "c2b512": (512, 256, False, 3, 1e-4, "none", 1),
    # per-GPU shares of a global batch of 512 at the HD mesh (5120 faces) and of the part-map renders
    "hd_rgb_b256": (256, 256, True, 4, 1e-5, "surface", 6),
    "hd_mask_b256": (256, 256, True, 4, 1e-5, "none", 1),
    "ref_vert8_b256": (256, 256, True, 3, 1e-5, "vertex8", 1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="")
    ap.add_argument("--configs", default="c2,c2b64,ref_mask,ref_rgb")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--json", default="")
    ap.add_argument("--alpha-only", type=int, default=0, help="silhouette configs: alpha plane alone")
    ap.add_argument("--bwd-spans", type=int, default=-1,
                    help="backward pixel enumeration: -1 the library's choice, 0 whole boxes, 1 row intervals")
    args = ap.parse_args()

    from sv3d import _lib as L, workmodel
    import bench
    libs = [p for p in args.libs.split(",") if p] or [L.LIB_PATH]
    handles = []
    for path in libs:
        h = ctypes.CDLL(os.path.abspath(path))
        vp, sz = ctypes.c_void_p, ctypes.c_size_t
        pp = ctypes.POINTER(L.RasterParams)
        h.sv3d_raster_sizes.argtypes = [pp, ctypes.POINTER(sz), ctypes.POINTER(sz), ctypes.POINTER(sz)]
        h.sv3d_raster_forward_timed.argtypes = [pp, vp, vp, vp, vp, vp, vp, vp, vp]
        h.sv3d_raster_backward_timed.argtypes = [pp, vp, vp, vp, vp, vp, vp, vp, sz, vp, vp, vp]
        handles.append(h)
    ev = bench.HipEvents()
    dev = torch.device("cuda:0")
    results = []
    for cname in args.configs.split(","):
        B, img, aa, sub, sigma, mode, T = CONFIGS[cname]
        ns = argparse.Namespace(batch=B, image_size=img, subdivide=sub, mode=mode, tex_size=T)
        fv, tex, grad = bench.build_inputs(ns, dev, seed=1000)
        S = img * (2 if aa else 1)
        m = {"none": 0, "surface": 1, "vertex": 2, "vertex8": 3}[mode]
        p = L.make_params(B, fv.shape[1], S, 2 if aa else 1, m, T, True, (0, 0, 0), 1, 100, 1e-3, sigma, 1e-10, 1e-4,
                          alpha_only=bool(args.alpha_only) and mode == "none",
                          backward_enumeration=args.bwd_spans + 1)
        # every library sizes its own buffers (builds may differ in what they keep there): take the
        # largest requirement over the libraries under test, never the in-tree library's alone
        rec_b = saved_b = ws_b = 0
        for h in handles:
            a, b_, c = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
            st = h.sv3d_raster_sizes(ctypes.byref(p), ctypes.byref(a), ctypes.byref(b_), ctypes.byref(c))
            assert st == 0, st
            rec_b, saved_b, ws_b = max(rec_b, a.value), max(saved_b, b_.value), max(ws_b, c.value)
        f32 = dict(dtype=torch.float32, device=dev)
        rgba = torch.empty((B, 9 if mode == "vertex8" else 4, img, img), **f32)
        ws_b = max(ws_b, saved_b)                 # older builds want scratch in every mode
        rec, saved, ws = (torch.empty(n // 4, **f32) for n in (rec_b, saved_b, ws_b))
        gfv = torch.empty_like(fv)
        gtex = torch.empty_like(tex) if tex is not None else None
        vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        pairs = sum(workmodel.count_active_pairs(fv, S, sigma, 1e-10))
        fl_f, fl_b = workmodel.raster_flops(pairs, mode != "none")
        n = args.iters
        e = ev.create(4 * n)
        acc = {i: {"fwd": [], "bwd": [], "step": []} for i in range(len(handles))}
        for r in range(args.rounds + 1):
            for li, h in enumerate(handles):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n):
                    st = h.sv3d_raster_forward_timed(ctypes.byref(p), vp(fv), vp(tex), vp(rgba), vp(rec), vp(saved),
                                                     stream, e[4 * i], e[4 * i + 1])
                    st |= h.sv3d_raster_backward_timed(ctypes.byref(p), vp(tex), vp(rec), vp(saved), vp(grad), vp(gfv),
                                                       vp(gtex), vp(ws), ctypes.c_size_t(ws_b), stream,
                                                       e[4 * i + 2], e[4 * i + 3])
                    assert st == 0, st
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                if r == 0:
                    continue                      # warm-up round
                acc[li]["fwd"].append(sum(ev.elapsed_ms(e[4 * i], e[4 * i + 1]) for i in range(n)) / n * 1e3)
                acc[li]["bwd"].append(sum(ev.elapsed_ms(e[4 * i + 2], e[4 * i + 3]) for i in range(n)) / n * 1e3)
                acc[li]["step"].append(dt / n * 1e6)
        for li, path in enumerate(libs):
            a = acc[li]
            f, b, s = (statistics.median(a[k]) for k in ("fwd", "bwd", "step"))
            row = dict(lib=os.path.basename(path), config=cname, pairs=pairs, fwd_us=f, bwd_us=b, step_us=s,
                       fwd_min=min(a["fwd"]), bwd_min=min(a["bwd"]),
                       fwd_tflops=fl_f / f / 1e6, bwd_tflops=fl_b / b / 1e6, img_per_s=B / s * 1e6)
            results.append(row)
            print("%-22s %-9s pairs %9d  fwd %8.1f us (%5.2f TF)  bwd %8.1f us (%5.2f TF)  step %8.1f us  %9.0f img/s"
                  % (row["lib"], cname, pairs, f, row["fwd_tflops"], b, row["bwd_tflops"], s, row["img_per_s"]), flush=True)
    if args.json:
        with open(args.json, "w") as fh:
            json.dump(results, fh, indent=1)


if __name__ == "__main__":
    main()
