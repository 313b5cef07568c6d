#!/usr/bin/env python3
"""Benchmark of the hot path: SoftRas forward + backward images/s on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (the headline benchmark configuration): silhouette-only soft rasterization forward + backward,
batch 8, 256x256, icosphere 642 vertices / 1280 faces, sigma 1e-4, dist_eps 1e-10, inputs
(camera-space face vertices, incoming gradient) resident in HBM.  One step = one
sv3d_raster_forward + one sv3d_raster_backward through the C ABI on the batch.  With N > 1
(launched by torch.distributed.run, one rank per GPU) every rank renders its own batch of
the same size -- the path shards by image with no data-path collective -- so scaling is
weak and `value` is the sum over ranks.

Prints ONE JSON line (rank 0).  `roofline` is for the kernel with the longest average
launch (FP32 VALU bound; timed with HIP events recorded on the launch stream around that
kernel on every 10th timed step).  `cpu_baseline` is the dense fp32 torch evaluation of the same
formulas (oracle port) on the host cores, on a bounded sample.
"""
import argparse
import ctypes
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "single-view-3d-reconstruction_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--batch", type=int, default=8, help="images per render batch per GPU")
    ap.add_argument("--image-size", type=int, default=256)
    ap.add_argument("--subdivide", type=int, default=3)
    ap.add_argument("--sigma", type=float, default=1e-4)
    ap.add_argument("--event_stride", type=int, default=10,
                    help="record HIP events around the kernels on every n-th timed step")
    ap.add_argument("--gamma", type=float, default=1e-4)
    ap.add_argument("--mode", choices=["none", "surface", "vertex"], default="none")
    ap.add_argument("--tex-size", type=int, default=6)
    ap.add_argument("--anti-aliasing", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


class HipEvents:
    """Raw hipEvent_t handles (the C ABI takes them as void*)."""

    def __init__(self):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]

    def create(self, n):
        out = []
        for _ in range(n):
            e = ctypes.c_void_p()
            if self.hip.hipEventCreate(ctypes.byref(e)) != 0:
                raise RuntimeError("hipEventCreate failed")
            out.append(e)
        return out

    def elapsed_ms(self, a, b):
        ms = ctypes.c_float()
        if self.hip.hipEventElapsedTime(ctypes.byref(ms), a, b) != 0:
            raise RuntimeError("hipEventElapsedTime failed")
        return ms.value


def build_inputs(args, device, seed):
    """Seeded CUB-shaped synthetic render batch -> camera-space face vertices [B,F,3,3]."""
    import sv3d
    from sv3d import meshgen
    from nnutils import geom_utils
    verts, faces = meshgen.icosphere(args.subdivide)
    g = torch.Generator().manual_seed(seed)
    B = args.batch
    v = torch.from_numpy(verts)[None].repeat(B, 1, 1) + 0.02 * torch.randn(B, verts.shape[0], 3, generator=g)
    cams = torch.empty(B, 7)
    cams[:, 0] = 0.6 + 0.3 * torch.rand(B, generator=g)
    cams[:, 1:3] = 0.2 * torch.rand(B, 2, generator=g) - 0.1
    q = torch.randn(B, 4, generator=g)
    cams[:, 3:] = q / q.norm(dim=1, keepdim=True)
    proj = geom_utils.orthographic_proj_withz(v.to(device), cams.to(device), offset_z=5.0)
    fv = sv3d.face_gather(proj, torch.from_numpy(faces)[None].to(device), scale=(1.0, -1.0, 1.0),
                          shift=(0.0, 0.0, 2.732))
    H = args.image_size
    grad = (2 * torch.rand(B, 4, H, H, generator=g) - 1).to(device)
    tex = None
    if args.mode == "surface":
        tex = torch.rand(B, faces.shape[0], args.tex_size ** 2, 3, generator=g).to(device)
    elif args.mode == "vertex":
        tex = torch.rand(B, faces.shape[0], 3, 3, generator=g).to(device)
    elif args.mode == "vertex8":
        tex = torch.rand(B, faces.shape[0], 3, 8, generator=g).to(device)
        grad = (2 * torch.rand(B, 9, H, H, generator=g) - 1).to(device)
    return fv.contiguous(), tex, grad.contiguous()


def cpu_baseline(args, fv0, seconds):
    """Dense fp32 torch-CPU forward+backward of ONE image of the batch, a slice of pixel rows
    at a time, until `seconds` of work are done; images/s extrapolated from the pixel share
    (the dense evaluation costs the same for every pixel)."""
    from oracle import raster_oracle
    cores = os.cpu_count() or 1
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    torch.set_num_threads(cores)
    S = args.image_size * (2 if args.anti_aliasing else 1)
    fv = fv0.float().cpu()
    rows_per_chunk = max(1, 2048 // S)
    done_px, t0 = 0, time.perf_counter()
    x = fv.clone().requires_grad_(True)
    row = 0
    while row < S:
        pix = torch.arange(row * S, min(S, row + rows_per_chunk) * S)
        out = raster_oracle.rasterize_image(x, None, S, sigma=args.sigma, texture_mode="none", pixels=pix)
        out["rgba"][3].sum().backward()
        done_px += pix.numel()
        row += rows_per_chunk
        if time.perf_counter() - t0 > seconds:
            break
    dt = time.perf_counter() - t0
    img_per_s = (done_px / float(S * S)) / dt
    return {"value": img_per_s, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "dense fp32 torch-CPU silhouette fwd+bwd (all pixels x all %d faces, oracle "
                      "port) on %d of %d pixel rows of image 0, %.1f s, scaled to a full image"
                      % (fv.shape[0], done_px // S, S, dt)}


def pmc_traffic(args, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 counter passes of this same
    command (profiles/pmc_default_workload.json: FETCH_SIZE and WRITE_SIZE in separate passes,
    1024-byte units, reads doubled as gfx950 tallies wide coalesced reads at half size).  None when
    the workload is not the default one or the summary is absent."""
    default = (args.batch, args.image_size, args.subdivide, args.mode, args.sigma, args.anti_aliasing) == \
        (8, 256, 3, "none", 1e-4, False)
    path = os.path.join(ROOT, "profiles", "pmc_default_workload.json")
    if not default or not os.path.exists(path):
        return None
    with open(path) as fh:
        row = json.load(fh).get(kernel, {})
    if "hbm_read_bytes_x2" not in row or "hbm_write_bytes" not in row:
        return None
    return row["hbm_read_bytes_x2"] + row["hbm_write_bytes"]


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    import sv3d
    from sv3d import _lib as L, workmodel

    fv, tex, grad = build_inputs(args, device, seed=1000 + rank)
    B, F = fv.shape[0], fv.shape[1]
    aa = 2 if args.anti_aliasing else 1
    S = args.image_size * aa
    mode = {"none": L.SV3D_TEX_NONE, "surface": L.SV3D_TEX_SURFACE, "vertex": L.SV3D_TEX_VERTEX}[args.mode]
    p = L.make_params(B, F, S, aa, mode, args.tex_size if args.mode == "surface" else 1, True,
                      (0.0, 0.0, 0.0), 1.0, 100.0, 1e-3, args.sigma, 1e-10, args.gamma)
    rec_b, saved_b, ws_b = L.raster_sizes(p)
    f32 = dict(dtype=torch.float32, device=device)
    rgba = torch.empty((B, 4, args.image_size, args.image_size), **f32)
    rec = torch.empty(rec_b // 4, **f32)
    saved = torch.empty(saved_b // 4, **f32)
    ws = torch.empty(ws_b // 4, **f32)
    gfv = torch.empty_like(fv)
    gtex = torch.empty_like(tex) if tex is not None else None
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    lib = L.lib()
    ev = HipEvents()
    K, W = args.steps, args.warmup
    # Kernel durations come from HIP events recorded on the kernels' stream inside the timed
    # region.  An event pair costs a few microseconds of stream time, a fifth of this step, so
    # only every `stride`-th step carries them; the other steps use the plain entry points.
    stride = max(1, args.event_stride)
    timed_steps = [i for i in range(K) if i % stride == 0]
    fe = ev.create(2 * len(timed_steps))
    be = ev.create(2 * len(timed_steps))
    slot = {i: j for j, i in enumerate(timed_steps)}
    pp = ctypes.byref(p)

    # the buffers are the same on every step: their pointer arguments are built once
    a_fv, a_tex, a_rgba, a_rec, a_saved = vp(fv), vp(tex), vp(rgba), vp(rec), vp(saved)
    a_grad, a_gfv, a_gtex, a_ws, a_wsb = vp(grad), vp(gfv), vp(gtex), vp(ws), ctypes.c_size_t(ws_b)

    def step(i=None):
        j = slot.get(i)
        if j is None:
            L.check(lib.sv3d_raster_forward(pp, a_fv, a_tex, a_rgba, a_rec, a_saved, stream), "fwd")
            L.check(lib.sv3d_raster_backward(pp, a_tex, a_rec, a_saved, a_grad, a_gfv, a_gtex, a_ws, a_wsb,
                                             stream), "bwd")
        else:
            L.check(lib.sv3d_raster_forward_timed(pp, a_fv, a_tex, a_rgba, a_rec, a_saved, stream,
                                                  fe[2 * j], fe[2 * j + 1]), "fwd")
            L.check(lib.sv3d_raster_backward_timed(pp, a_tex, a_rec, a_saved, a_grad, a_gfv, a_gtex, a_ws,
                                                   a_wsb, stream, be[2 * j], be[2 * j + 1]), "bwd")

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    for _ in range(W):
        step()
    # no pause of Python's cyclic collector inside the timed region: a full collection of the
    # process's objects takes longer than the GPU's launch queue can cover
    gc.collect()
    gc.disable()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        step(i)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    n_ev = len(timed_steps)
    fwd_ms = sum(ev.elapsed_ms(fe[2 * j], fe[2 * j + 1]) for j in range(n_ev)) / n_ev
    bwd_ms = sum(ev.elapsed_ms(be[2 * j], be[2 * j + 1]) for j in range(n_ev)) / n_ev
    pairs = sum(workmodel.count_active_pairs(fv, S, args.sigma, 1e-10))
    f_flop, b_flop = workmodel.raster_flops(pairs, colour=args.mode != "none")
    if bwd_ms >= fwd_ms:
        name, flop, ms = "raster_backward_kernel", b_flop, bwd_ms
    else:
        name, flop, ms = "raster_forward_kernel", f_flop, fwd_ms
    achieved = flop / (ms * 1e-3) / 1e12
    roofline = {"bound": "valu", "kernel": name, "achieved": achieved,
                "peak": workmodel.FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / workmodel.FP32_VALU_PEAK_TFLOPS,
                "traffic": pmc_traffic(args, name),
                "avg_kernel_ms": ms, "kernel_time_samples": n_ev, "active_pairs_per_launch": pairs,
                "flop_per_pair": flop / max(pairs, 1),
                "other_kernel": {"fwd_ms": fwd_ms, "bwd_ms": bwd_ms,
                                 "fwd_tflops": f_flop / (fwd_ms * 1e-3) / 1e12,
                                 "bwd_tflops": b_flop / (bwd_ms * 1e-3) / 1e12}}

    if rank != 0:
        return
    out = {
        "metric": "SoftRas fwd+bwd images/sec @256^2,1280 faces",
        "value": world * B * K / elapsed,
        "unit": "images/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": 1e3 * elapsed / K,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "SoftRas %s fwd+bwd, batch %d per GPU, %dx%d%s, %d verts/%d faces, sigma=%g"
                               % ("silhouette-only" if args.mode == "none" else args.mode + "-texture RGB",
                                  B, args.image_size, args.image_size, " (2x AA)" if aa == 2 else "",
                                  {1: 42, 2: 162, 3: 642, 4: 2562}.get(args.subdivide, -1), F, args.sigma),
                   "batch_per_gpu": B, "image_size": args.image_size, "faces": F,
                   "parallelism": "image-sharded x%d" % world},
        "roofline": roofline,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, fv[0], args.cpu_seconds)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
