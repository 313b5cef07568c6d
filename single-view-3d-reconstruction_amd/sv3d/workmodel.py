"""Algorithmic work of the hot path, for roofline accounting.

Unit of raster work: one ACTIVE (pixel, face) pair = a pair that passes the formula's own
bounding-box cull (pixel centre inside the face's box widened by sqrt(ln(1/dist_eps-1)*sigma)).
FLOP per pair are counted on the evaluation the kernels perform (pixel-relative
point-to-segment form), with add/sub/mul = 1, fma = 2, rcp/exp = 1 and compares, selects,
min/max and clamps = 0.  Culled pairs, binning, setup and the per-pixel epilogue count
as zero, so `achieved` is a lower bound on executed arithmetic.

forward, silhouette (57):
   6  vertex - pixel (3 corners x 2)
  12  3 edges x [dot (mul+fma = 3) + scale by 1/|e|^2 (1)]          (clamp to [0,1] free)
  12  3 edges x foot - pixel (2 fma)
   9  3 edges x squared length (mul + fma)
   9  3 edges x edge function (mul + fma)
   3  barycentrics (x 1/det)
   5  exp argument (1), exp (1), 1+e (1), rcp (1), (1-D) = e*D (1)
   1  alpha product
forward, colour extra (48): interpolated 1/z numerator and denominator (5+5), Newton
   reciprocal (5), offset depth and zp (2), barycentric normalisation (2+1+3), shifted z (1),
   two softmax exponentials (3+3), weight (1), running sum (2), three colour sums (9),
   texel address arithmetic (6).
backward, silhouette (74): the 56 forward FLOP that rebuild the fragment, chain factor (1),
   distance chain 2*gx/sigma, (1-t), t (5), six accumulations (12).
backward, colour extra (106): depth / barycentric recompute (30), exponent (3), colour dot
   (6), fragment and depth chain factors (8), three z accumulations (15), barycentric
   gradient through normalisation and clamp (17), projection on the two axes (12), six
   accumulations (12), texel / vertex-colour accumulation (3).

Chamfer forward: 8 FLOP per (query, candidate) pair per direction (3 sub, 1 mul, 2 fma),
both directions computed: 16 * B * N * M.
"""
import math

import torch

FWD_SILHOUETTE = 57
FWD_COLOUR_EXTRA = 48
BWD_SILHOUETTE = 74
BWD_COLOUR_EXTRA = 106
CHAMFER_PER_PAIR = 8
FP32_VALU_PEAK_TFLOPS = 157.3      # 256 CU x 4 SIMD x 32 lanes x 2 FLOP x 2.4 GHz


def cull_margin(sigma, dist_eps):
    return math.sqrt(math.log(1.0 / dist_eps - 1.0) * sigma)


def count_active_pairs(face_vertices, raster_size, sigma, dist_eps):
    """Active (pixel, face) pairs per image, exact: a box test is separable, so the count is
    (#pixel columns inside the x range) * (#pixel rows inside the y range) summed over faces.
    face_vertices [B,F,3,3] on any device; evaluated in float64."""
    fv = face_vertices.detach().double()
    S = int(raster_size)
    m = cull_margin(sigma, dist_eps)
    centres = (2.0 * torch.arange(S, dtype=torch.float64, device=fv.device) + 1.0 - S) / S
    x, y = fv[..., 0], fv[..., 1]                                       # [B,F,3]
    lo_x, hi_x = x.min(-1).values - m, x.max(-1).values + m
    lo_y, hi_y = y.min(-1).values - m, y.max(-1).values + m
    nx = torch.searchsorted(centres, hi_x.contiguous(), right=True) - torch.searchsorted(centres, lo_x.contiguous())
    ny = torch.searchsorted(centres, hi_y.contiguous(), right=True) - torch.searchsorted(centres, lo_y.contiguous())
    return (nx.clamp(min=0) * ny.clamp(min=0)).sum(-1).tolist()


def raster_flops(n_pairs, colour):
    """(forward FLOP, backward FLOP) for `n_pairs` active pairs."""
    f = FWD_SILHOUETTE + (FWD_COLOUR_EXTRA if colour else 0)
    b = BWD_SILHOUETTE + (BWD_COLOUR_EXTRA if colour else 0)
    return n_pairs * f, n_pairs * b


def chamfer_flops(B, N, M):
    return 2 * CHAMFER_PER_PAIR * B * N * M
