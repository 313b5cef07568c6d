#!/usr/bin/env python3
"""End-to-end timing of one `SoftNeuralRenderer` call (forward + backward to vertices, cameras
and textures) -- what the training step pays per render, launch overheads included.

    python tools/render_bench.py [--batches 8,64] [--modes mask,rgb,rgb_lit,vertex] [--iters 50]

For every (batch, mode) prints microseconds per call for
  staged  the call assembled from separate stages (torch projection, Mesh, Transform, raster op)
  fused   camera + flip + gather + setup inside the HIP ops
  graph   the fused call captured once in a HIP graph (torch.cuda.CUDAGraph) and replayed
"""
import argparse
import os
import sys
import gc
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "single-view-3d-reconstruction_amd"))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="8,64")
    ap.add_argument("--modes", default="mask,rgb,rgb_lit,vertex")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--image_size", type=int, default=256)
    ap.add_argument("--subdivide", type=int, default=3)
    ap.add_argument("--profile", action="store_true", help="cProfile the eager fused call (host overhead)")
    args = ap.parse_args()
    from nnutils.softras import SoftNeuralRenderer
    from sv3d import meshgen
    verts, faces = meshgen.icosphere(args.subdivide)
    verts, faces = torch.from_numpy(verts).float(), torch.from_numpy(faces).long()
    dev = torch.device("cuda:0")
    for B in (int(b) for b in args.batches.split(",")):
        g = torch.Generator().manual_seed(B)
        v0 = (verts[None] + 0.02 * torch.randn(B, verts.shape[0], 3, generator=g)).to(dev)
        cams0 = torch.empty(B, 7)
        cams0[:, 0] = 0.6 + 0.3 * torch.rand(B, generator=g)
        cams0[:, 1:3] = 0.2 * torch.rand(B, 2, generator=g) - 0.1
        q = torch.randn(B, 4, generator=g)
        cams0[:, 3:] = q / q.norm(dim=1, keepdim=True)
        cams0 = cams0.to(dev)
        f = faces[None].repeat(B, 1, 1).to(dev)
        f32 = f.to(torch.int32)
        for mode in args.modes.split(","):
            r = SoftNeuralRenderer(args.image_size, texture_type="vertex" if mode == "vertex" else "surface")
            if mode == "rgb":
                r.ambient_light_only()
            tex0 = None
            if mode.startswith("rgb"):
                tex0 = torch.rand(B, f.shape[1], 36, 3, generator=g).to(dev)
            elif mode == "vertex":
                tex0 = torch.rand(B, verts.shape[0], 3, generator=g).to(dev)
            vg, cg = v0.clone().requires_grad_(True), cams0.clone().requires_grad_(True)
            tg = tex0.clone().requires_grad_(True) if tex0 is not None else None
            shape = (B, args.image_size, args.image_size) if mode == "mask" else (B, 3, args.image_size, args.image_size)
            w = torch.rand(shape, generator=g).to(dev)

            def call(fused, faces_t):
                if fused:
                    out = r._forward_fused(vg, faces_t, cg, tg, *r._camera_stage())
                else:
                    out = r._forward_staged(vg, faces_t, cg, tg)
                (out * w).sum().backward()

            def timeit(fn):
                for _ in range(25):                 # the allocator settles on the new sizes meanwhile
                    fn()
                torch.cuda.synchronize()
                gc.collect()
                gc.disable()                        # no collector pause inside the timed loop
                t0 = time.perf_counter()
                worst = 0.0
                for _ in range(args.iters):
                    t1 = time.perf_counter()
                    fn()
                    worst = max(worst, time.perf_counter() - t1)
                torch.cuda.synchronize()
                total = time.perf_counter() - t0
                gc.enable()
                if worst > 0.2 * total and args.iters >= 10:
                    print("  (one call took %.0f us of the loop's %.0f us on the host)" % (worst * 1e6, total * 1e6))
                return total / args.iters * 1e6

            if args.profile:
                import cProfile
                import pstats
                timeit(lambda: call(True, f32))
                pr = cProfile.Profile()
                pr.enable()
                for _ in range(200):
                    call(True, f32)
                torch.cuda.synchronize()
                pr.disable()
                pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
                continue
            t_staged = timeit(lambda: call(False, f))
            t_fused = timeit(lambda: call(True, f32))
            # capture: static input/grad buffers, side-stream warm-up as torch requires
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    call(True, f32)
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                call(True, f32)
            t_graph = timeit(graph.replay)
            print("B %3d %-8s staged %8.1f us  fused %8.1f us  graph %8.1f us   (%.0f / %.0f / %.0f img/s)"
                  % (B, mode, t_staged, t_fused, t_graph, B / t_staged * 1e6, B / t_fused * 1e6,
                     B / t_graph * 1e6), flush=True)
            # the graph's private memory pool would otherwise stay with the allocator and disturb
            # the eager timings of the configurations that follow
            del graph
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
