"""Texture path on the device: entry points of the operators `sv3d::texture_sample` (with
`sv3d::texture_sample_backward`) and `sv3d::distance_transform` (csrc/torch_ops.cpp over the
functions of the same names in include/sv3d_hip.h).  GPU tensors only.
"""
from .ops import _ops

MAX_CHANNELS = 4


def texture_sample(images, flow):
    """Bilinear samples of `images` [B,C,H,W] (C <= 4) at `flow` [B,N,2] in [-1,1] (x, y; the
    corners of the image are -1 and +1, outside reads zero) -> [B,N,C], differentiable in `flow`."""
    if images.requires_grad:
        raise RuntimeError("sv3d.texture_sample has no gradient for the images")
    return _ops.texture_sample(images, flow)


def dt_barrier(masks, k=50.0):
    """[B,H,W] masks -> sigmoid(k (EDT(1 - mask) - EDT(mask)) / max(H,W)); no gradient."""
    return _ops.distance_transform(masks.detach(), True, float(k))


def dt(masks):
    """[B,H,W] masks -> EDT(1 - mask) / max(H,W); no gradient."""
    return _ops.distance_transform(masks.detach(), False, 0.0)
