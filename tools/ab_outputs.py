#!/usr/bin/env python3
"""Do two builds of libsv3d_hip.so produce the same bits?

    python tools/ab_outputs.py --libs a.so,b.so [--configs c2,ref_mask,...]

Runs the raster forward and backward of every build on the inputs of tools/kbench.py's configs and
compares image, saved per-pixel state and gradients with the first build's: prints `equal` or the
largest difference relative to the tensor's largest magnitude.  Exit status 1 if any differ.
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "single-view-3d-reconstruction_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", required=True)
    ap.add_argument("--configs", default="c2,c2b64,ref_mask,ref_mask_b8")
    args = ap.parse_args()
    from sv3d import _lib as L
    import bench
    from kbench import CONFIGS
    dev = torch.device("cuda:0")
    vp, sz = ctypes.c_void_p, ctypes.c_size_t
    pp = ctypes.POINTER(L.RasterParams)
    handles = []
    for path in args.libs.split(","):
        h = ctypes.CDLL(os.path.abspath(path))
        h.sv3d_raster_sizes.argtypes = [pp, ctypes.POINTER(sz), ctypes.POINTER(sz), ctypes.POINTER(sz)]
        h.sv3d_raster_forward.argtypes = [pp, vp, vp, vp, vp, vp, vp]
        h.sv3d_raster_backward.argtypes = [pp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
        handles.append((os.path.basename(path), h))
    bad = 0
    for cname in args.configs.split(","):
        B, img, aa, sub, sigma, mode, T = CONFIGS[cname]
        ns = argparse.Namespace(batch=B, image_size=img, subdivide=sub, mode=mode, tex_size=T)
        fv, tex, grad = bench.build_inputs(ns, dev, seed=1000)
        S = img * (2 if aa else 1)
        m = {"none": 0, "surface": 1, "vertex": 2, "vertex8": 3}[mode]
        p = L.make_params(B, fv.shape[1], S, 2 if aa else 1, m, T, True, (0, 0, 0), 1, 100, 1e-3, sigma, 1e-10, 1e-4)
        ptr = lambda t: vp(t.data_ptr()) if t is not None else None
        stream = vp(torch.cuda.current_stream(dev).cuda_stream)
        first = None
        for name, h in handles:
            a, b_, c = sz(), sz(), sz()
            assert h.sv3d_raster_sizes(ctypes.byref(p), ctypes.byref(a), ctypes.byref(b_), ctypes.byref(c)) == 0
            f32 = dict(dtype=torch.float32, device=dev)
            rgba = torch.zeros((B, 9 if mode == "vertex8" else 4, img, img), **f32)
            rec = torch.zeros(a.value // 4, **f32)
            saved = torch.zeros(b_.value // 4, **f32)
            ws = torch.zeros(max(c.value, 4) // 4, **f32)
            gfv = torch.zeros_like(fv)
            gtex = torch.zeros_like(tex) if tex is not None else None
            st = h.sv3d_raster_forward(ctypes.byref(p), ptr(fv), ptr(tex), ptr(rgba), ptr(rec), ptr(saved), stream)
            st |= h.sv3d_raster_backward(ctypes.byref(p), ptr(tex), ptr(rec), ptr(saved), ptr(grad), ptr(gfv),
                                         ptr(gtex), ptr(ws), sz(c.value), stream)
            assert st == 0, st
            torch.cuda.synchronize()
            outs = {"image": rgba, "saved": saved, "grad_faces": gfv}
            if gtex is not None:
                outs["grad_textures"] = gtex
            outs = {k: v.cpu() for k, v in outs.items()}
            if first is None:
                first = outs
                continue
            for k, v in outs.items():
                if torch.equal(v, first[k]):
                    print("%-10s %-14s %-14s equal" % (cname, name, k))
                else:
                    bad += 1
                    d = float((v - first[k]).abs().max() / first[k].abs().max())
                    print("%-10s %-14s %-14s DIFFERS, max |d| / max |x| = %.3g" % (cname, name, k, d))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
