#!/usr/bin/env python3
"""Step time of raster forward + backward launched eagerly vs replayed from a captured HIP graph
(default bench workload: batch 8, 256^2, 1280 faces, silhouette)."""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "single-view-3d-reconstruction_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    import bench
    from sv3d import _lib as L
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps_per_graph", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    ns = argparse.Namespace(batch=a.batch, image_size=256, subdivide=3, mode="none", tex_size=1)
    fv, tex, grad = bench.build_inputs(ns, dev, seed=1000)
    B, F = fv.shape[0], fv.shape[1]
    p = L.make_params(B, F, 256, 1, 0, 1, True, (0, 0, 0), 1, 100, 1e-3, 1e-4, 1e-10, 1e-4)
    rec_b, saved_b, ws_b = L.raster_sizes(p)
    f32 = dict(dtype=torch.float32, device=dev)
    rgba = torch.empty((B, 4, 256, 256), **f32)
    rec, saved = torch.empty(rec_b // 4, **f32), torch.empty(saved_b // 4, **f32)
    gfv = torch.empty_like(fv)
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    lib = L.lib()

    def step():
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        L.check(lib.sv3d_raster_forward(ctypes.byref(p), vp(fv), None, vp(rgba), vp(rec), vp(saved), st), "f")
        L.check(lib.sv3d_raster_backward(ctypes.byref(p), None, vp(rec), vp(saved), vp(grad), vp(gfv), None, None,
                                         ctypes.c_size_t(0), st), "b")

    def timeit(fn, n):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    eager = timeit(step, 500)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(a.steps_per_graph):
            step()
    graph = timeit(g.replay, 500) / a.steps_per_graph
    print("batch %d: eager %.1f us/step, graph (%d steps per replay) %.1f us/step" % (B, eager, a.steps_per_graph, graph))


if __name__ == "__main__":
    main()
