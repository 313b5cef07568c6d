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
launch (FP32 VALU bound).  Kernel durations come from a separate pass AFTER the timed region:
60 launches per kernel, HIP events recorded on the launch stream around batches of 10 launches
of that kernel alone (an event pair costs microseconds of stream time, which a 17 us kernel
cannot carry: the timed region itself carries no events).  `cpu_baseline`
is the dense fp32 torch evaluation of the same formulas (oracle port) on the host cores, on a
bounded sample, with the other legs of the CPU reference configuration (network forward on one
256x256 image, dense Chamfer 642 x 642, dense Laplacian) beside it.

Four more objects ride on the same line (their own timings, their own warm-up).  With one rank
they run BEFORE the headline's timed region, so the headline starts on a GPU that is already busy;
with several ranks they run after it, under a time limit, and a leg that fails or hangs costs the
line that leg, not the headline (`extras_error` then says what happened):
  `api`         the same batch through the Python API (`sv3d.render`: camera stage + rasterizer,
                forward + backward to vertices and cameras), and one `SoftNeuralRenderer` call at
                the reference's own constants (anti-aliased 512^2 raster, sigma 1e-5)
  `chamfer`     `sv3d_chamfer_forward` at 64 x 642^2, 64 x 2048^2 and 16 x 8192^2: TFLOP/s and
                fraction of the FP32 vector peak (16 B N M flop per call)
  `train_step`  the synthetic CMR-style training step of examples/cmr_synthetic.py at the
                reference's default batch (4 dataset images -> 64 rendered images per pass):
                dataset images/s per rank and summed; with N > 1 (or --force-dist) the step runs
                under DistributedDataParallel, so its gradients are all-reduced by RCCL.  One
                rank also runs the step on the 5120-face mesh and the batch-64 reading of the
                full-step configuration (64 dataset images, one camera hypothesis, no mirror pair)
  `configs`     the larger configurations through the C ABI, forward + backward, 64 images each:
                surface RGB on the 5120-face mesh (T = 6, anti-aliased 512^2 raster), eight part
                maps in one pass at the reference defaults, and Chamfer 16 x 8192^2 forward +
                backward; images/s (clouds/s) and the dominant kernel's fraction of the FP32 peak
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
    ap.add_argument("--preroll-ms", dest="preroll_ms", type=float, default=80.0,
                    help="untimed repetition of the step before the warm-up steps, to bring the clocks up")
    ap.add_argument("--kernel-launches", dest="kernel_launches", type=int, default=60,
                    help="launches per kernel in the kernel-duration pass after the timed region (batches of 10)")
    ap.add_argument("--force-dist", dest="force_dist", action="store_true",
                    help="one rank: open the RCCL process group and wrap the train step in DDP anyway")
    ap.add_argument("--gamma", type=float, default=1e-4)
    ap.add_argument("--mode", choices=["none", "surface", "vertex"], default="none")
    ap.add_argument("--tex-size", type=int, default=6)
    ap.add_argument("--anti-aliasing", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=3.0, help="budget of the dense silhouette leg")
    ap.add_argument("--no-extras", action="store_true", help="headline only: no api / chamfer / train_step legs")
    ap.add_argument("--train-steps", type=int, default=12, help="timed steps of the train_step leg")
    ap.add_argument("--extras-limit-s", dest="extras_limit_s", type=float, default=300.0,
                    help="several ranks: time limit of the extra legs, after which the line is printed without them")
    ap.add_argument("--train-args", default="", help="extra examples/cmr_synthetic.py flags for the train_step leg "
                                                     "(rehearsals with a smaller network)")
    return ap.parse_args()


class HipEvents:
    """Raw hipEvent_t handles (the C ABI takes them as void*)."""

    def __init__(self):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]

    def create(self, n):
        out = []
        for _ in range(n):
            e = ctypes.c_void_p()
            # hipEventDisableSystemFence: the events only time kernels on their own stream, they
            # publish nothing to the host, and the system-scope fence of a default event adds
            # microseconds of stream time to what it brackets
            if self.hip.hipEventCreateWithFlags(ctypes.byref(e), int(os.environ.get("SV3D_EVENT_FLAGS", "0x20000000"), 0)) != 0:
                raise RuntimeError("hipEventCreateWithFlags failed")
            out.append(e)
        return out

    def destroy(self, events):
        for e in events:
            self.hip.hipEventDestroy(e)

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
    if args.mode == "none" and getattr(args, "alpha_only", False):
        grad = grad[:, 3:4]                     # the silhouette caller's gradient: the alpha plane alone
    tex = None
    if args.mode == "surface":
        tex = torch.rand(B, faces.shape[0], args.tex_size ** 2, 3, generator=g).to(device)
    elif args.mode == "vertex":
        tex = torch.rand(B, faces.shape[0], 3, 3, generator=g).to(device)
    elif args.mode == "vertex8":
        tex = torch.rand(B, faces.shape[0], 3, 8, generator=g).to(device)
        grad = (2 * torch.rand(B, 9, H, H, generator=g) - 1).to(device)
    return fv.contiguous(), tex, grad.contiguous()


class RasterCase:
    """One rasterizer configuration through the C ABI: inputs and buffers resident on the device,
    `step()` = sv3d_raster_forward + sv3d_raster_backward, `kernel_times()` = average launch
    duration of the two dominant kernels from HIP events on the launch stream."""

    def __init__(self, device, batch, image_size, subdivide, mode, sigma, gamma=1e-4, tex_size=6,
                 anti_aliasing=False, alpha_only=False, seed=1000):
        from sv3d import _lib as L, workmodel
        self.L, self.workmodel = L, workmodel
        ns = argparse.Namespace(batch=batch, image_size=image_size, subdivide=subdivide, mode=mode,
                                tex_size=tex_size, alpha_only=alpha_only)
        self.fv, self.tex, self.grad = build_inputs(ns, device, seed)
        self.mode, self.sigma = mode, sigma
        B, F = self.fv.shape[0], self.fv.shape[1]
        self.B, self.F = B, F
        aa = 2 if anti_aliasing else 1
        self.S = image_size * aa
        m = {"none": L.SV3D_TEX_NONE, "surface": L.SV3D_TEX_SURFACE, "vertex": L.SV3D_TEX_VERTEX,
             "vertex8": L.SV3D_TEX_VERTEX8}[mode]
        self.p = L.make_params(B, F, self.S, aa, m, tex_size if mode == "surface" else 1, True, (0.0, 0.0, 0.0),
                               1.0, 100.0, 1e-3, sigma, 1e-10, gamma, alpha_only=alpha_only and mode == "none")
        rec_b, saved_b, ws_b = L.raster_sizes(self.p)
        f32 = dict(dtype=torch.float32, device=device)
        planes = 1 if self.p.alpha_only else (9 if mode == "vertex8" else 4)
        self.rgba = torch.empty((B, planes, image_size, image_size), **f32)
        assert self.grad.shape == self.rgba.shape, (self.grad.shape, self.rgba.shape)
        self.rec = torch.empty(rec_b // 4, **f32)
        self.saved = torch.empty(saved_b // 4, **f32)
        self.gfv = torch.empty_like(self.fv)
        self.gtex = torch.empty_like(self.tex) if self.tex is not None else None
        vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        self.stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        self.lib = L.lib()
        pp = ctypes.byref(self.p)
        # the buffers are the same on every step: their pointer arguments are built once
        self.fwd_args = (pp, vp(self.fv), vp(self.tex), vp(self.rgba), vp(self.rec), vp(self.saved), self.stream)
        self.bwd_args = (pp, vp(self.tex), vp(self.rec), vp(self.saved), vp(self.grad), vp(self.gfv), vp(self.gtex),
                         None, ctypes.c_size_t(0), self.stream)

    def step(self):
        st = self.lib.sv3d_raster_forward(*self.fwd_args)
        if st == 0:
            st = self.lib.sv3d_raster_backward(*self.bwd_args)
        if st != 0:
            self.L.check(st, "sv3d_raster_forward / sv3d_raster_backward")

    def kernel_times(self, ev, launches=60, per_batch=10):
        """(forward ms, backward ms, launches per kernel): HIP events around batches of `per_batch`
        back-to-back launches of the one kernel (sv3d_raster_*_timed_n), so that the events' own
        stream time is spread over the batch."""
        nb = max(1, launches // per_batch)
        fe, be = ev.create(2 * nb), ev.create(2 * nb)
        n = ctypes.c_int32(per_batch)
        for j in range(nb):
            self.L.check(self.lib.sv3d_raster_forward_timed_n(*self.fwd_args, fe[2 * j], fe[2 * j + 1], n), "fwd")
        for j in range(nb):
            self.L.check(self.lib.sv3d_raster_backward_timed_n(*self.bwd_args, be[2 * j], be[2 * j + 1], n), "bwd")
        torch.cuda.synchronize()
        fwd = sum(ev.elapsed_ms(fe[2 * j], fe[2 * j + 1]) for j in range(nb)) / (nb * per_batch)
        bwd = sum(ev.elapsed_ms(be[2 * j], be[2 * j + 1]) for j in range(nb)) / (nb * per_batch)
        ev.destroy(fe + be)
        return fwd, bwd, nb * per_batch

    def forward_kernel_name(self):
        """The kernel `kernel_times` brackets for the forward of this configuration."""
        return self.lib.sv3d_raster_forward_kernel_name(self.fwd_args[0], 0).decode()

    def roofline(self, fwd_ms, bwd_ms, launches, traffic=None):
        wm = self.workmodel
        pairs = sum(wm.count_active_pairs(self.fv, self.S, self.sigma, 1e-10))
        f_flop, b_flop = wm.raster_flops(pairs, colour=self.mode != "none")
        if bwd_ms >= fwd_ms:
            name, flop, ms = "raster_backward_kernel", b_flop, bwd_ms
        else:
            name, flop, ms = self.forward_kernel_name(), f_flop, fwd_ms
        achieved = flop / (ms * 1e-3) / 1e12
        return {"bound": "valu", "kernel": name, "achieved": achieved, "peak": wm.FP32_VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": achieved / wm.FP32_VALU_PEAK_TFLOPS, "traffic": traffic,
                "avg_kernel_ms": ms, "kernel_time_samples": launches, "active_pairs_per_launch": pairs,
                "flop_per_pair": flop / max(pairs, 1),
                "other_kernel": {"fwd_ms": fwd_ms, "bwd_ms": bwd_ms,
                                 "fwd_tflops": f_flop / (fwd_ms * 1e-3) / 1e12,
                                 "bwd_tflops": b_flop / (bwd_ms * 1e-3) / 1e12,
                                 "fwd_frac": f_flop / (fwd_ms * 1e-3) / 1e12 / wm.FP32_VALU_PEAK_TFLOPS,
                                 "bwd_frac": b_flop / (bwd_ms * 1e-3) / 1e12 / wm.FP32_VALU_PEAK_TFLOPS}}


def configs_leg(device, ev):
    """The larger configurations, forward + backward through the C ABI: 64 images each (one GPU's
    share of the reference recipe at batch 4: 8 camera hypotheses x (4 + 4 mirrored) images) of
    surface RGB on the 5120-face mesh and of eight part maps in one pass; the headline's silhouette
    workload at 64 and 512 images per launch; Chamfer 16 x 8192^2."""
    from sv3d import _lib as L, workmodel
    out = {}
    for name, batch, kw in (("hd_surface_rgb", 64, dict(subdivide=4, mode="surface", tex_size=6, sigma=1e-5, anti_aliasing=True)),
                            ("part_maps_8_per_pass", 64, dict(subdivide=3, mode="vertex8", sigma=1e-5, anti_aliasing=True)),
                            ("silhouette_batch_64", 64, dict(subdivide=3, mode="none", sigma=1e-4, alpha_only=True)),
                            ("silhouette_batch_512", 512, dict(subdivide=3, mode="none", sigma=1e-4, alpha_only=True))):
        case = RasterCase(device, batch, 256, gamma=1e-4, **kw)
        # 80 ms of the same work first: the card's clocks fall while the host builds the next case
        # and only come back after tens of milliseconds (the headline's pre-roll, for every case)
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) < 0.08:
            for _ in range(8):
                case.step()
            torch.cuda.synchronize()
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            case.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        fwd_ms, bwd_ms, launches = case.kernel_times(ev, launches=20, per_batch=5)
        r = case.roofline(fwd_ms, bwd_ms, launches)
        what = {"surface": "surface-texture RGB (T=6)", "vertex8": "eight per-vertex part maps in one pass",
                "none": "silhouette (the headline workload at a larger batch)"}[kw["mode"]]
        out[name] = {"workload": "%s fwd+bwd, %d images, 256x256%s, %d faces, sigma=%g%s"
                                 % (what, batch, " (2x AA raster)" if kw.get("anti_aliasing") else "", case.F, kw["sigma"],
                                    " (x8 maps per image)" if kw["mode"] == "vertex8" else ""),
                     "ms_per_step": 1e3 * dt, "images_per_s": batch / dt, "fwd_kernel_ms": fwd_ms, "bwd_kernel_ms": bwd_ms,
                     "dominant_kernel": r["kernel"], "frac": r["frac"], "tflops": r["achieved"],
                     "fwd_frac": r["other_kernel"]["fwd_frac"], "bwd_frac": r["other_kernel"]["bwd_frac"],
                     "active_pairs_per_launch": r["active_pairs_per_launch"], "flop_per_pair": r["flop_per_pair"]}
        del case
        torch.cuda.empty_cache()
    # Chamfer 16 x 8192^2, forward + backward with both distance outputs carrying a gradient
    lib = L.lib()
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    B, N = 16, 8192
    g = torch.Generator().manual_seed(5)
    a, b = torch.randn(B, N, 3, generator=g).to(device), torch.randn(B, N, 3, generator=g).to(device)
    d1, d2 = torch.empty(B, N, device=device), torch.empty(B, N, device=device)
    i1 = torch.empty(B, N, dtype=torch.int32, device=device)
    i2 = torch.empty(B, N, dtype=torch.int32, device=device)
    gd1, gd2 = torch.rand(B, N, device=device), torch.rand(B, N, device=device)
    ga, gb = torch.empty_like(a), torch.empty_like(b)

    def fwd():
        L.check(lib.sv3d_chamfer_forward(vp(a), vp(b), B, N, N, vp(d1), vp(d2), vp(i1), vp(i2), stream), "chamfer")

    def bwd():
        L.check(lib.sv3d_chamfer_backward(vp(a), vp(b), B, N, N, vp(gd1), vp(gd2), vp(i1), vp(i2), vp(ga), vp(gb),
                                          stream), "chamfer backward")

    def timed(fn, iters=20):
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) < 0.05:          # clocks up first
            for _ in range(4):
                fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    f_ms, b_ms = timed(fwd), timed(bwd)
    tf = workmodel.chamfer_flops(B, N, N) / (f_ms * 1e-3) / 1e12
    out["chamfer_16x8192"] = {"workload": "Chamfer nearest neighbours, 16 cloud pairs of 8192 x 8192 points, forward + "
                                          "backward (both distance outputs carry a gradient)",
                              "forward_ms": f_ms, "backward_ms": b_ms, "cloud_pairs_per_s": B / ((f_ms + b_ms) * 1e-3),
                              "dominant_kernel": "chamfer_nn_kernel", "tflops": tf,
                              "frac": tf / workmodel.FP32_VALU_PEAK_TFLOPS}
    return out


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
    out = {"value": img_per_s, "unit": "images/s", "cores": cores, "kind": "port",
           "sample": "dense fp32 torch-CPU silhouette fwd+bwd (all pixels x all %d faces, oracle "
                     "port) on %d of %d pixel rows of image 0, %.1f s, scaled to a full image"
                     % (fv.shape[0], done_px // S, S, dt)}
    out.update(cpu_reference_legs(args, 1.0 / img_per_s))
    return out


def _median_ms(fn, repeats=5, warm=1):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return 1e3 * sorted(ts)[len(ts) // 2]


def cpu_reference_legs(args, silhouette_s):
    """The other legs of the CPU reference configuration (one synthetic 256x256 image, 642 vertices /
    1280 faces), each timed on the host cores with dense torch: the network forward (ResNet-18-shaped
    encoder + shape / camera / texture heads of examples/cmr_synthetic.py, random weights, inference
    mode), Chamfer 642 x 642 forward + backward, the uniform Laplacian loss forward + backward
    (dense V x V operator, as the upstream module holds it)."""
    from oracle import meshreg_oracle
    from sv3d import meshgen
    cmr = _load_cmr_module()
    verts, faces = meshgen.icosphere(3)
    vt, ft = torch.from_numpy(verts), torch.from_numpy(faces).long()
    net = cmr.MeshNetLite(vt, ft, 8, 6).eval()
    img = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(0))

    def net_forward():
        with torch.no_grad():
            net(img)

    a = (vt[None] + 0.02 * torch.randn(1, vt.shape[0], 3, generator=torch.Generator().manual_seed(1)))
    b = (a * torch.tensor([-1.0, 1.0, 1.0])).clone().requires_grad_(True)

    def chamfer():
        d = ((a[:, :, None] - b[:, None]) ** 2).sum(-1)            # dense [1,N,M] squared distances
        d.min(1).values.mean().backward()                            # dist2.mean(), loss_utils.py:87-88

    x = a.clone().requires_grad_(True)

    def laplacian():
        meshreg_oracle.laplacian_loss(x, faces).sum().backward()

    net_ms, ch_ms, lap_ms = _median_ms(net_forward, 3), _median_ms(chamfer), _median_ms(laplacian)
    return {"reference_config": {
        "what": "CPU reference configuration: network forward on one 256x256 image + dense silhouette "
                "fwd+bwd + Chamfer 642x642 fwd+bwd + dense Laplacian fwd+bwd, torch on the host cores",
        "network_forward_ms": net_ms, "silhouette_fwd_bwd_ms": 1e3 * silhouette_s,
        "chamfer_642_fwd_bwd_ms": ch_ms, "laplacian_fwd_bwd_ms": lap_ms,
        "total_ms": net_ms + 1e3 * silhouette_s + ch_ms + lap_ms}}


def _load_cmr_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("cmr_synthetic", os.path.join(ROOT, "examples", "cmr_synthetic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def api_leg(args, device, rank):
    """The headline batch through the Python API, and one SoftNeuralRenderer call at the reference's
    constants; images/s of forward + backward (gradients to vertices and cameras)."""
    import sv3d
    from nnutils.softras import SoftNeuralRenderer
    from sv3d import meshgen
    verts, faces = meshgen.icosphere(args.subdivide)
    g = torch.Generator().manual_seed(2000 + rank)
    B, H = args.batch, args.image_size
    v = (torch.from_numpy(verts)[None] + 0.02 * torch.randn(B, verts.shape[0], 3, generator=g)).to(device)
    cams = torch.empty(B, 7)
    cams[:, 0] = 0.6 + 0.3 * torch.rand(B, generator=g)
    cams[:, 1:3] = 0.2 * torch.rand(B, 2, generator=g) - 0.1
    q = torch.randn(B, 4, generator=g)
    cams[:, 3:] = q / q.norm(dim=1, keepdim=True)
    cams = cams.to(device)
    f = torch.from_numpy(faces)[None].to(device).to(torch.int32).contiguous()
    grad = (2 * torch.rand(B, H, H, generator=g) - 1).to(device)
    vg, cg = v.requires_grad_(True), cams.requires_grad_(True)
    renderer = SoftNeuralRenderer(H)
    f_all = f.expand(B, -1, -1).contiguous()

    def headline_config():
        vg.grad = cg.grad = None               # as the trainer's zero_grad(set_to_none=True): no accumulation pass
        out = sv3d.render(vg, cg, f, None, H, offset_z=5.0, scale=(1.0, -1.0, 1.0), shift=(0.0, 0.0, 2.732),
                          anti_aliasing=args.anti_aliasing, sigma_val=args.sigma, alpha_only=True)
        out.backward(grad)

    def reference_defaults():
        vg.grad = cg.grad = None
        renderer(vg, f_all, cg).backward(grad)

    res = {}
    for name, fn in (("headline_config", headline_config), ("reference_defaults", reference_defaults)):
        for _ in range(100):                     # library start-up, allocator, clocks
            fn()
        torch.cuda.synchronize()
        gc.collect()
        gc.disable()
        n, dt = 150, float("inf")
        for _ in range(3):                       # best of three rounds: the leg is host-bound and the host is shared
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            dt = min(dt, time.perf_counter() - t0)
        gc.enable()
        res[name] = {"us_per_call": 1e6 * dt / n, "images_per_s": B * n / dt}
    res["what"] = ("forward + backward of one renderer call on the headline batch through the Python API "
                   "(150 calls, best of three rounds): "
                   "headline_config = sv3d.render (camera stage + rasterizer at the headline's sigma and raster), "
                   "reference_defaults = nnutils.softras.SoftNeuralRenderer(img_size)(vertices, faces, cams) "
                   "(anti-aliased 2x raster, sigma 1e-5)")
    return res


def chamfer_leg(device):
    """sv3d_chamfer_forward at the reference's size and the two sizes the configurations name."""
    from sv3d import _lib as L, workmodel
    lib = L.lib()
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    rows = []
    for B, N in ((4, 642), (64, 642), (64, 2048), (16, 8192)):
        g = torch.Generator().manual_seed(5)
        a, b = torch.randn(B, N, 3, generator=g).to(device), torch.randn(B, N, 3, generator=g).to(device)
        d1, d2 = torch.empty(B, N, device=device), torch.empty(B, N, device=device)
        i1 = torch.empty(B, N, dtype=torch.int32, device=device)
        i2 = torch.empty(B, N, dtype=torch.int32, device=device)

        def fwd():
            L.check(lib.sv3d_chamfer_forward(vp(a), vp(b), B, N, N, vp(d1), vp(d2), vp(i1), vp(i2), stream), "chamfer")

        # warm-up long enough for the card's clocks to come back up (see the headline's pre-roll)
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) < 0.05:
            for _ in range(16):
                fwd()
            torch.cuda.synchronize()
        iters = 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fwd()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        tf = workmodel.chamfer_flops(B, N, N) / us / 1e6
        rows.append({"clouds": B, "points": N, "forward_us": us, "tflops": tf,
                     "frac_of_fp32_valu_peak": tf / workmodel.FP32_VALU_PEAK_TFLOPS})
    return rows


def _timed_train(cmr, args, device, world, extra_flags, steps_wanted, dist_on):
    """A trainer of examples/cmr_synthetic.py built from `extra_flags`, 6 warm-up steps, then
    `steps_wanted` timed steps; returns (trainer, options, steps, seconds: max over ranks)."""
    import tempfile
    tmp = tempfile.mkdtemp(prefix="sv3d_bench_")
    o = cmr.parse(["--print_scalars", "0", "--checkpoint_dir", tmp, "--name", "bench", "--dataset_size", "1024",
                   "--save_latest_freq", "1000000"] + extra_flags)
    o.epoch_fraction = 1.0
    from nnutils import train_utils
    env = train_utils.init_distributed(gpu_id=device.index, force=dist_on)
    t = cmr.ShapeTrainer(o, env)
    t.init_training()
    note("train_step %s: trainer built (%d parameters%s)"
         % (" ".join(extra_flags), sum(p.numel() for p in t.model.parameters()),
            ", DDP over %d rank(s), backend %s" % (world, env.backend) if dist_on else ""))
    o.num_iter = 6                                   # warm-up: library start-up, autotuning, allocator
    t.train()
    note("train_step: 6 warm-up steps done")
    o.num_iter = steps_wanted
    if dist_on:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = t.train()
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    return t, o, steps, dt


def train_step_leg(args, device, world, rank):
    """examples/cmr_synthetic.py's step (encoder, camera hypotheses, shared render passes, part maps,
    Chamfer, regularisers, distance-map loss, autocast + clip + fused Adam x2) at batch 4 per rank."""
    if os.environ.get("SV3D_BENCH_FAIL_RANK") == str(rank):
        raise RuntimeError("SV3D_BENCH_FAIL_RANK: this rank was asked to fail in the train_step leg")
    cmr = _load_cmr_module()
    dist_on = world > 1 or args.force_dist
    t, o, steps, dt = _timed_train(cmr, args, device, world, ["--batch_size", "4", "--seg_train", "1"]
                                   + args.train_args.split(), args.train_steps, dist_on)
    per_rank = steps * o.batch_size / dt
    graphed = None
    variants = None
    if not dist_on:
        # the same step replayed from a HIP graph (one process only: the eager figure above is the
        # one that is comparable across rank counts).  Mode 1 holds forward + loss + backward, mode 2
        # the optimizer steps as well.
        def replayed(mode):
            o.cuda_graph = mode
            o.num_iter = 4                           # capture, then a few replays
            t.train()
            o.num_iter = args.train_steps
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gsteps = t.train()
            torch.cuda.synchronize()
            return gsteps, time.perf_counter() - t0
        s1, d1 = replayed(1)
        gsteps, gdt = replayed(2)
        graphed = {"ms_per_step": 1e3 * gdt / gsteps, "dataset_images_per_s": gsteps * o.batch_size / gdt,
                   "what": "--cuda_graph 2: forward + loss + backward + unscale + clipping + the two fused Adam "
                           "steps + scaler update captured once in a HIP graph and replayed",
                   "forward_backward_only_ms_per_step": 1e3 * d1 / s1}
    if world == 1 and not args.train_args:
        del t
        gc.collect()
        torch.cuda.empty_cache()
        variants = {}
        for name, flags, what in (
                ("hd_mesh", ["--batch_size", "4", "--seg_train", "1", "--subdivide", "4"],
                 "the same step on the 2562-vertex / 5120-face mesh (subdivide 4, T = 6): 64 rendered images per pass"),
                ("batch64_one_hypothesis", ["--batch_size", "64", "--seg_train", "0", "--num_hypo_cams", "1",
                                            "--flip_pair", "0", "--num_parts", "0"],
                 "64 dataset images per step, one camera hypothesis, no mirror pair: one RGB + silhouette pass over "
                 "64 rendered images, no part maps")):
            tv, ov, sv, dv = _timed_train(cmr, args, device, world, flags, max(4, args.train_steps // 2), dist_on)
            variants[name] = {"what": what, "steps": sv, "ms_per_step": 1e3 * dv / sv,
                              "dataset_images_per_s": sv * ov.batch_size / dv}
            del tv
            gc.collect()
            torch.cuda.empty_cache()
    return {"what": "synthetic CMR-style training step (examples/cmr_synthetic.py): ResNet-18-shaped encoder under "
                    "autocast, 8 camera hypotheses x (4 + 4 mirrored) images = 64 rendered images per pass, RGB + "
                    "silhouette + 8 part maps from 2 shared passes, Chamfer, Laplacian + flatten, distance-map "
                    "loss, clip + fused Adam x2; LPIPS replaced by L1 (weights absent)",
            "batch_per_rank": o.batch_size, "rendered_images_per_pass": 16 * o.batch_size, "steps": steps,
            "ms_per_step": 1e3 * dt / steps, "dataset_images_per_s_per_rank": per_rank,
            "dataset_images_per_s": per_rank * world, "ranks": world, "hip_graph": graphed, "variants": variants,
            "gradient_all_reduce": ("%s via DistributedDataParallel over %d ranks"
                                    % ("RCCL" if torch.distributed.get_backend() == "nccl" else
                                       torch.distributed.get_backend(), torch.distributed.get_world_size()))
            if dist_on else "none (one rank)"}


_T0 = time.perf_counter()


def note(msg):
    """Progress line on stderr (rank, seconds since start): what a run that ends in a time limit was
    doing, and where its time went."""
    sys.stderr.write("[bench rank %s +%.1fs] %s\n" % (os.environ.get("RANK", "0"), time.perf_counter() - _T0, msg))
    sys.stderr.flush()


FAILED_KEY = "sv3d_bench_extras_failed"


def _group_store():
    """The process group's key-value store (None outside a process group)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.distributed_c10d._get_default_store()
    except Exception:                            # noqa: BLE001 - diagnostics only
        pass
    return None


def run_guarded(run, extras, out, rank, limit_s, poll_s=1.0):
    """run() on every rank under a time limit; returns when it finished.

    A rank whose run() raises writes the exception and its traceback to stderr at once and posts it
    in the process group's store; every rank's watcher thread polls that key, so rank 0 prints `out`
    with the legs finished so far and `extras_error` = "rank R: Error: text" within a second, and
    all ranks end there with status 0 (a rank waiting in a collective for the one that failed cannot
    be interrupted any other way).  If the limit passes with nothing posted, every rank first dumps
    the Python stack of each of its threads to stderr -- which collective or call it is stuck in --
    and rank 0 prints the line with a "did not finish" note."""
    import faulthandler
    import threading
    import traceback
    lock = threading.Lock()
    state = {"done": False}
    store = _group_store()
    t_end = time.monotonic() + limit_s

    def leave(note_text):
        if rank == 0:
            o = dict(out)
            o.update({k: v for k, v in extras.items() if not k.startswith("_")})
            o["extras_error"] = note_text
            print(json.dumps(o), flush=True)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)

    def posted():
        if store is None:
            return None
        try:
            if store.check([FAILED_KEY]):
                return store.get(FAILED_KEY).decode("utf-8", "replace")
        except Exception:                        # noqa: BLE001 - the store went away with its rank
            return None
        return None

    def watch():
        while True:
            time.sleep(min(poll_s, max(t_end - time.monotonic(), 0.0)) or 0.01)
            with lock:
                if state["done"]:
                    return
                msg = posted()
                if msg is not None:
                    note("leaving the extra legs: " + msg)
                    leave(msg)
                if time.monotonic() >= t_end:
                    note("the extra legs did not finish within %d s (last: %s); stacks of all threads follow"
                         % (limit_s, extras.get("_leg", "?")))
                    faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                    leave(extras.get("extras_error", "the extra legs did not finish within %d s (rank %d was in: %s)"
                                     % (limit_s, rank, extras.get("_leg", "?"))))

    watcher = threading.Thread(target=watch, daemon=True)
    watcher.start()
    try:
        run()
    except Exception as e:                       # noqa: BLE001 - reported in the line, the headline stands
        text = ("rank %d: %s: %s" % (rank, type(e).__name__, e))[:600]
        note("rank %d: an extra leg raised (%s):\n%s" % (rank, extras.get("_leg", "?"), traceback.format_exc()))
        extras["extras_error"] = text
        if store is not None:
            try:
                store.set(FAILED_KEY, text)
            except Exception:                    # noqa: BLE001
                pass
        if rank == 0:
            with lock:
                extras.pop("_leg", None)
                leave(text)
        threading.Event().wait()                 # until the watcher ends this process
    with lock:
        state["done"] = True
    extras.pop("_leg", None)


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
    args.alpha_only = args.mode == "none"      # the silhouette caller keeps the alpha plane alone (softras.py:64)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # diagnostic: every thread's Python stack on stderr if the run is still going after N seconds
    # (several ranks: armed by default at 0.8 of the extra legs' limit, counted from the start)
    watchdog = os.environ.get("SV3D_BENCH_WATCHDOG") or (str(0.8 * args.extras_limit_s) if world > 1 else "")
    if watchdog and float(watchdog) > 0:
        import faulthandler
        faulthandler.dump_traceback_later(float(watchdog), repeat=False)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        # SV3D_DIST_BACKEND=gloo+cuda rehearses the multi-rank run on one GPU (ranks share it,
        # collectives through gloo); the default is RCCL, one GPU per rank
        rehearsal = os.environ.get("SV3D_DIST_BACKEND") == "gloo+cuda"
        if rehearsal:
            local_rank %= max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    import sv3d  # noqa: F401  (loads the libraries; fails loudly when they are missing)

    case = RasterCase(device, args.batch, args.image_size, args.subdivide, args.mode, args.sigma, args.gamma,
                      args.tex_size, args.anti_aliasing, alpha_only=args.alpha_only, seed=1000 + rank)
    B, F, S = case.B, case.F, case.S
    aa = 2 if args.anti_aliasing else 1
    ev = HipEvents()
    K, W = args.steps, args.warmup
    step = case.step

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    extras = {}

    def run_extras():                           # own warm-up and timing each
        for name, leg in (("api", lambda: api_leg(args, device, rank)), ("chamfer", lambda: chamfer_leg(device)),
                          ("configs", lambda: configs_leg(device, ev)),
                          ("train_step", lambda: train_step_leg(args, device, world, rank))):
            extras["_leg"] = name
            note("leg %s starts" % name)
            extras[name] = leg()
            note("leg %s done" % name)
        extras.pop("_leg", None)

    if not args.no_extras and world == 1:
        # one rank: ahead of the headline, so that they also leave the GPU busy before it starts
        run_extras()
        gc.collect()
        torch.cuda.empty_cache()

    # no pause of Python's cyclic collector inside the timed region: a full collection of the
    # process's objects takes longer than the GPU's launch queue can cover.  It runs before the
    # warm-up, not after it: the GPU would sit idle through it and start the timed steps cold.
    gc.collect()
    gc.disable()
    # The card raises its clocks only after tens of milliseconds of continuous work: the same
    # kernels run 7 % slower in the first milliseconds after an idle spell (23.8 against 22.3 us
    # for the forward), which a 20-step run of one millisecond would otherwise be measured in.
    # The step is therefore repeated, untimed, for `--preroll-ms` before the W warm-up steps.
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.preroll_ms:
        for _ in range(64):
            step()
        torch.cuda.synchronize()
    for _ in range(W):
        step()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Kernel durations: a pass of its own, after the timed region (which carries no events), the GPU
    # still warm from it.
    for _ in range(16):
        step()
    fwd_ms, bwd_ms, launches = case.kernel_times(ev, launches=args.kernel_launches)
    name = "raster_backward_kernel" if bwd_ms >= fwd_ms else case.forward_kernel_name()
    roofline = case.roofline(fwd_ms, bwd_ms, launches, traffic=pmc_traffic(args, name))

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
                   "parallelism": "image-sharded x%d" % world,
                   "image_planes": "alpha only (the silhouette caller's view, nnutils/softras.py:64)"
                   if case.p.alpha_only else "all",
                   "clock_preroll_ms": args.preroll_ms},
        "roofline": roofline,
    }
    if not args.no_extras and world > 1:
        # Several ranks: the headline is measured first and the extra legs, whose training step
        # all-reduces gradients between the ranks, run under a time limit.  A rank that fails or
        # waits for one that failed must not cost the run its headline: rank 0 then prints the line
        # with the legs it has and the ranks leave without the group's shutdown handshake.
        run_guarded(run_extras, extras, out, rank, args.extras_limit_s)
    if rank != 0:
        return
    out.update({k: v for k, v in extras.items() if not k.startswith("_")})
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, case.fv[0], args.cpu_seconds)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
