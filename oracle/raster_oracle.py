"""CPU oracle for the soft rasterizer -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file; the product path never does.

PARITY UNPINNED.  The arithmetic of this path lives in the un-vendored
third-party package `soft_renderer` (ShichenLiu/SoftRas, cloned at HEAD per the
reference's README.md:25-31, no pinned version) which is absent from
/root/reference and from this image.  The reference holds no test or golden
vector for it.  This file restates the published SoftRas algorithm
(`soft_rasterize_cuda_kernel.cu`: face setup with the inverse barycentric
matrix / `sym` matrix / obtuse flags, the barycentric `t - w` point-to-triangle
distance, sigmoid fragments, product alpha, online z-softmax colour) as a dense
all-pixels x all-faces torch evaluation, anchored on the reference's own call
site nnutils/softras.py:31-77 (constants sigma 1e-5, gamma 1e-4, dist_eps
1e-10, anti-aliasing, look_at eye z -2.732, orthographic, ambient 0.8).

It deliberately follows the *upstream formulation* (barycentric algebra), not
the pixel-relative segment formulation the HIP kernels use, so that agreement
between the two is a real check.  Gradients are what autograd returns for this
forward with every discrete decision (cull, inside/outside, region/edge choice,
clamps, texel pick, running max) frozen.
"""
import math

import torch

DEFAULTS = dict(
    background=(0.0, 0.0, 0.0),
    near=1.0,
    far=100.0,
    eps=1e-3,
    sigma=1e-5,
    dist_eps=1e-10,
    gamma=1e-4,
    double_side=True,
    texture_mode="surface",   # "none" | "surface" | "vertex"
    pixels=None,              # optional LongTensor of flat pixel indices (row * S + col) to evaluate
)


def pixel_centres(S, dtype=torch.float64):
    """Row-major pixel centre coordinates; row 0 is the TOP of the image (y up)."""
    rows = torch.arange(S, dtype=dtype)
    cols = torch.arange(S, dtype=dtype)
    yi = (S - 1) - rows
    yp = (2.0 * yi + 1.0 - S) / S
    xp = (2.0 * cols + 1.0 - S) / S
    Y, X = torch.meshgrid(yp, xp, indexing="ij")
    return X.reshape(-1), Y.reshape(-1)


def _face_setup(fv):
    """fv [F,3,3] -> inverse barycentric matrix, sym matrix, obtuse flags."""
    x = fv[:, :, 0]
    y = fv[:, :, 1]
    x0, x1, x2 = x[:, 0], x[:, 1], x[:, 2]
    y0, y1, y2 = y[:, 0], y[:, 1], y[:, 2]
    det = x2 * (y0 - y1) + x0 * (y1 - y2) + x1 * (y2 - y0)
    det = torch.where(det > 0, det.clamp(min=1e-10), det.clamp(max=-1e-10))
    inv = torch.stack([
        torch.stack([y1 - y2, x2 - x1, x1 * y2 - x2 * y1], -1),
        torch.stack([y2 - y0, x0 - x2, x2 * y0 - x0 * y2], -1),
        torch.stack([y0 - y1, x1 - x0, x0 * y1 - x1 * y0], -1),
    ], 1) / det[:, None, None]                                  # [F,3,3]
    sym = x[:, :, None] * x[:, None, :] + y[:, :, None] * y[:, None, :] + 1.0   # [F,3,3]
    p = fv[:, :, :2].detach()
    obt = torch.zeros(fv.shape[0], 3, dtype=torch.bool)
    found = torch.zeros(fv.shape[0], dtype=torch.bool)
    for k in range(3):
        a = p[:, (k + 1) % 3] - p[:, k]
        b = p[:, (k + 2) % 3] - p[:, k]
        neg = ((a * b).sum(-1) < 0) & ~found
        obt[:, k] = neg
        found = found | neg
    return inv, sym, obt


def _edge_t(w, sym, v0):
    """Barycentric coordinates t [P,F,3] of the foot of the perpendicular from the
    pixel onto the line of edge (v0, v0+1); v0 is a python int."""
    v1 = (v0 + 1) % 3
    v2 = (v0 + 2) % 3
    a = sym[:, v0, :] - sym[:, v1, :]                           # [F,3]
    num = (w * a[None]).sum(-1) - a[None, :, v1]
    den = (a[:, v0] - a[:, v1])[None]                           # == |p_v1 - p_v0|^2
    # a zero-length projected edge makes upstream divide 0/0; here the foot is the point itself
    t0 = num / torch.where(den == 0, torch.ones_like(den), den)
    t = [None, None, None]
    t[v0] = t0
    t[v1] = 1.0 - t0
    t[v2] = torch.zeros_like(t0)
    return torch.stack(t, -1)


def rasterize_image(fv, tex, S, **kw):
    """Dense soft rasterization of ONE image.

    fv  [F,3,3] face vertices (x, y, z) as seen by the kernel (already transformed)
    tex None | [F,Ts,3] (surface) | [F,3,3] (vertex colours per face corner)
    Returns dict(rgba [4,S,S], aggr [2,S,S] (softmax sum, max), n_active, unstable [S,S] bool);
    with `pixels` given the spatial axes are replaced by one axis over those pixels.
    """
    P = dict(DEFAULTS)
    P.update(kw)
    dt = fv.dtype
    F = fv.shape[0]
    mode = P["texture_mode"] if tex is not None else "none"
    near, far, eps, sigma, gamma = P["near"], P["far"], P["eps"], P["sigma"], P["gamma"]
    threshold = math.log(1.0 / P["dist_eps"] - 1.0) * sigma
    margin = math.sqrt(threshold)
    tol = 64 * torch.finfo(torch.float32).eps

    xp, yp = pixel_centres(S, dt)                               # [P]
    if P["pixels"] is not None:
        xp, yp = xp[P["pixels"]], yp[P["pixels"]]
    npx = xp.shape[0]
    inv, sym, obt = _face_setup(fv)
    px = fv[:, :, 0]
    py = fv[:, :, 1]
    pz = fv[:, :, 2]
    with torch.no_grad():
        xmin, xmax = px.min(1).values, px.max(1).values
        ymin, ymax = py.min(1).values, py.max(1).values
        keep = ~((xp[:, None] > (xmax + margin)[None]) | (xp[:, None] < (xmin - margin)[None]) |
                 (yp[:, None] > (ymax + margin)[None]) | (yp[:, None] < (ymin - margin)[None]))
    n_active = int(keep.sum())

    pix = torch.stack([xp, yp, torch.ones_like(xp)], -1)        # [P,3]
    w = torch.einsum("fkc,pc->pfk", inv, pix)                   # [P,F,3]
    wd = w.detach()
    inside = ((wd > 0) & (wd < 1)).all(-1)
    unstable = torch.zeros(npx, dtype=torch.bool)

    # ---- signed squared euclidean distance -------------------------------------------
    cand = [_edge_t(w, sym, k) for k in range(3)]               # unclamped, [P,F,3] each
    pxy = fv[:, :, :2]                                          # [F,3,2]

    def delta(t):
        return torch.einsum("pfk,fkc->pfc", t - w, pxy)         # [P,F,2]

    d_in = torch.stack([delta(t) for t in cand], 0)             # [3,P,F,2]
    d2_in = (d_in ** 2).sum(-1)                                 # [3,P,F]
    best = d2_in.detach().argmin(0)                             # first wins ties
    d2_inside = torch.gather(d2_in, 0, best[None])[0]

    w0n, w1n, w2n = wd[..., 0] <= 0, wd[..., 1] <= 0, wd[..., 2] <= 0
    pd = torch.stack([xp, yp], -1)[:, None, None, :] - pxy.detach()[None]       # [P,F,3,2] pixel - p_k
    def dotk(k, j):
        return (pd[:, :, k] * (pxy.detach()[:, j] - pxy.detach()[:, k])[None]).sum(-1)
    v0 = torch.full((npx, F), -1, dtype=torch.long)
    sel = torch.zeros((npx, F), dtype=torch.bool)
    def assign(cond, val):
        nonlocal v0, sel
        c = cond & ~sel
        v0 = torch.where(c, val, v0)
        sel = sel | c
    c0 = w1n & w2n
    assign(c0, torch.where(obt[None, :, 0] & (dotk(0, 2) > 0), 2, 0).long().expand(npx, F))
    c1 = w2n & w0n
    assign(c1, torch.where(obt[None, :, 1] & (dotk(1, 0) > 0), 0, 1).long().expand(npx, F))
    c2 = w0n & w1n
    assign(c2, torch.where(obt[None, :, 2] & (dotk(2, 1) > 0), 1, 2).long().expand(npx, F))
    assign(w0n, torch.full((npx, F), 1, dtype=torch.long))
    assign(w1n, torch.full((npx, F), 2, dtype=torch.long))
    assign(w2n, torch.full((npx, F), 0, dtype=torch.long))
    v0 = v0.clamp(min=0)                                        # unassigned == inside, value unused
    d_out = torch.stack([delta(t.clamp(0.0, 1.0)) for t in cand], 0)
    d2_out_all = (d_out ** 2).sum(-1)
    d2_outside = torch.gather(d2_out_all, 0, v0[None])[0]

    d2 = torch.where(inside, d2_inside, d2_outside)
    sign = torch.where(inside, 1.0, -1.0).to(dt)
    active = keep & (inside | (d2.detach() < threshold))
    D = torch.sigmoid(sign * d2 / sigma)
    D = torch.where(active, D, torch.zeros_like(D))

    # ---- alpha: product ---------------------------------------------------------------
    alpha = 1.0 - torch.prod(1.0 - D, dim=1)                    # [P]

    bg = torch.tensor(P["background"], dtype=dt)
    if mode == "vertex8":
        # eight scalar per-corner channels in one pass; channel c is channel 0 of the reference's
        # render of part map c (main.py:422-426), so every channel takes background[0]
        bg = bg[:1].repeat(tex.shape[2])
    if mode == "none":
        rgb = bg[:, None].expand(3, npx).clone()
        aggr = torch.zeros(2, npx, dtype=dt)
    else:
        # ---- depth ---------------------------------------------------------------------
        wc = w.clamp(0.0, 1.0)
        wc = wc / wc.sum(-1, keepdim=True).clamp(min=1e-5)
        den = (wc / pz[None]).sum(-1)
        # pairs that take no part keep a finite depth (a face without area has w == 0 and would
        # put 1/0 -- and through it NaN gradients -- into branches that are masked out below)
        zp = 1.0 / torch.where(active, den, torch.ones_like(den))   # [P,F]
        zok = (zp.detach() >= near) & (zp.detach() <= far)
        front = ((py[:, 2] - py[:, 0]) * (px[:, 1] - px[:, 0]) <
                 (py[:, 1] - py[:, 0]) * (px[:, 2] - px[:, 0])).detach()
        contrib = active & zok
        if not P["double_side"]:
            contrib = contrib & front[None]
        zn = (far - zp) / (far - near)
        zn_masked = torch.where(contrib, zn.detach(), torch.full_like(zn, -float("inf")))
        m = torch.maximum(zn_masked.max(1).values, torch.tensor(eps, dtype=dt))   # [P]
        arg = torch.where(contrib, (zn - m[:, None]) / gamma, torch.zeros_like(zn))   # mask BEFORE exp
        E = torch.where(contrib, torch.exp(arg), torch.zeros_like(zn))
        bgw = math.exp(eps / gamma) * torch.exp((eps - m) / gamma)                 # [P]
        # ---- colour ---------------------------------------------------------------------
        if mode == "surface":
            Ts = tex.shape[1]
            R = int(round(math.sqrt(Ts)))
            wcd = wc.detach()
            fx = wcd[..., 0] * R
            fy = wcd[..., 1] * R
            wx = fx.floor().clamp(max=R - 1).long()
            wy = fy.floor().clamp(max=R - 1).long()
            # SoftRas writes (w0 + w1) * R - wx - wy <= 1; with w0 + w1 = 1 - w2 this is the same
            # test, and it is exact on the clamped edge w2 == 0 where the sum form is decided
            # by rounding noise (in fp64 as much as in fp32).
            diag = (1.0 - wcd[..., 2]) * R - wx - wy
            lower = diag <= 1
            tid = torch.where(lower, wy * R + wx, (R - 1 - wy) * R + (R - 1 - wx))
            # fp32 barycentrics of a face carry rounding of the order eps * |e|^2 / |det| (the edge
            # functions cancel against a small determinant on edge-on faces), so the band around a
            # texel-cell boundary inside which the pick is undecidable in single precision scales
            # with that conditioning number
            with torch.no_grad():
                e = torch.stack([px[:, (k + 1) % 3] - px[:, k] for k in range(3)], -1) ** 2 + \
                    torch.stack([py[:, (k + 1) % 3] - py[:, k] for k in range(3)], -1) ** 2
                det = ((px[:, 1] - px[:, 0]) * (py[:, 2] - py[:, 0]) -
                       (py[:, 1] - py[:, 0]) * (px[:, 2] - px[:, 0])).abs().clamp(min=1e-10)
                tolf = (tol * (e.max(-1).values / det).clamp(min=1.0))[None]     # [1,F]
            free = (wd > tolf[..., None]) & (wd < 1 - tolf[..., None])   # coordinate not pinned by the clamp
            near_edge = (((fx - fx.round()).abs() < tolf * R) & free[..., 0]) | \
                        (((fy - fy.round()).abs() < tolf * R) & free[..., 1]) | \
                        (((diag - 1).abs() < tolf * R) & free[..., 2])
            # a pick only matters through the fragment's share of the pixel's colour: faces hidden
            # behind nearer ones (share below 1e-7) cannot move the pixel whichever texel they take
            with torch.no_grad():
                share = (D * E) / (bgw + (D * E).sum(1))[:, None]
            unstable = unstable | (near_edge & contrib & (share > 1e-7)).any(-1)
            fidx = torch.arange(F)[None].expand(npx, F)
            col = tex[fidx, tid]                                # [P,F,3]
        else:
            col = torch.einsum("pfk,fkc->pfc", wc, tex)         # [P,F,3]
        wgt = D * E
        ssum = bgw + wgt.sum(1)
        C = bgw[:, None] * bg[None] + (wgt[..., None] * col).sum(1)
        rgb = (C / ssum[:, None]).t()
        aggr = torch.stack([ssum, m], 0)

    rgba = torch.cat([rgb, alpha[None]], 0)
    if P["pixels"] is None:
        rgba, aggr, unstable = rgba.reshape(-1, S, S), aggr.reshape(2, S, S), unstable.reshape(S, S)
    return dict(rgba=rgba, aggr=aggr, n_active=n_active, unstable=unstable)


def soft_rasterize(face_vertices, textures, S, **kw):
    """Batched wrapper: face_vertices [B,F,3,3], textures None|[B,F,Ts,3]|[B,F,3,3] -> rgba [B,4,S,S]."""
    outs = [rasterize_image(face_vertices[b], None if textures is None else textures[b], S, **kw)
            for b in range(face_vertices.shape[0])]
    return (torch.stack([o["rgba"] for o in outs]),
            dict(aggr=torch.stack([o["aggr"] for o in outs]),
                 n_active=[o["n_active"] for o in outs],
                 unstable=torch.stack([o["unstable"] for o in outs])))


def count_active_pairs(face_vertices, S, sigma, dist_eps, chunk=64):
    """Number of (pixel, face) pairs that pass the formula's own bounding-box cull,
    per image -- the unit of algorithmic work for the roofline.  fp64, memory bounded."""
    fv = face_vertices.detach().double()
    margin = math.sqrt(math.log(1.0 / dist_eps - 1.0) * sigma)
    xs = (2.0 * torch.arange(S, dtype=torch.float64) + 1.0 - S) / S
    out = []
    for b in range(fv.shape[0]):
        x, y = fv[b, :, :, 0], fv[b, :, :, 1]
        xmin, xmax = x.min(1).values - margin, x.max(1).values + margin
        ymin, ymax = y.min(1).values - margin, y.max(1).values + margin
        nx = ((xs[None] >= xmin[:, None]) & (xs[None] <= xmax[:, None])).sum(1)
        ny = ((xs[None] >= ymin[:, None]) & (xs[None] <= ymax[:, None])).sum(1)
        out.append(int((nx * ny).sum()))
    return out
