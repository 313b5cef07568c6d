"""Times the texture path: the HIP texture-flow sampler against torch's grid_sample + permute (+ the
contiguous copy the rasterizer needs) on the same GPU, forward + backward to the flow; and the
batched GPU distance transform against the reference's per-mask scipy calls on the host
(main.py:259-260: two barrier transforms per sample and iteration).

    python tools/texture_bench.py [--iters 100]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "single-view-3d-reconstruction_amd"))

import sv3d.texture as tx  # noqa: E402


def timed(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for B, F, T in ((8, 1280, 6), (32, 1280, 6), (32, 5120, 6)):
        images = torch.rand(B, 3, 256, 256, device=dev)
        flow = (2 * torch.rand(B, F * T * T, 2, device=dev) - 1).requires_grad_(True)
        w = torch.rand(B, F * T * T, 3, device=dev)

        def hip():
            flow.grad = None
            (tx.texture_sample(images, flow) * w).sum().backward()

        def ref():
            flow.grad = None
            s = torch.nn.functional.grid_sample(images, flow.view(B, F, T * T, 2), align_corners=True)
            (s.permute(0, 2, 3, 1).reshape(B, F * T * T, 3).contiguous() * w).sum().backward()

        t_hip, t_ref = timed(hip, args.iters), timed(ref, args.iters)
        print("sample B %3d F %5d T %d : HIP %8.1f us   grid_sample+permute %8.1f us   (x%.1f)"
              % (B, F, T, t_hip, t_ref, t_ref / t_hip), flush=True)

    from scipy.ndimage import distance_transform_edt
    yy, xx = np.mgrid[0:256, 0:256]
    for B in (4, 16, 64):
        masks = np.stack([((yy - 128) ** 2 + (xx - 100 - i % 40) ** 2 < (40 + i % 30) ** 2).astype(np.float32)
                          for i in range(B)])
        m = torch.from_numpy(masks).to(dev)
        t_gpu = timed(lambda: (tx.dt_barrier(m), None)[1], args.iters)

        def host():
            out = []
            for mk in masks:
                for mm in (mk, mk[:, ::-1]):
                    d = (distance_transform_edt(1 - mm) - distance_transform_edt(mm)) / 256
                    out.append(1.0 / (1 + np.exp(-50 * d)))
            return torch.from_numpy(np.stack(out)).float().to(dev)

        t0 = time.perf_counter()
        for _ in range(3):
            host()
        t_host = (time.perf_counter() - t0) / 3 * 1e6
        print("dt_barrier B %3d 256x256 : GPU (one call + flip) %8.1f us   scipy on the host, 2B masks %10.1f us   (x%.0f)"
              % (B, t_gpu, t_host, t_host / t_gpu), flush=True)


if __name__ == "__main__":
    main()
