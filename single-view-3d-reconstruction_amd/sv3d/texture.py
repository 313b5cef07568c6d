"""Texture path on the device: torch custom ops over `sv3d_texture_sample(_backward)` and
`sv3d_distance_transform` (include/sv3d_hip.h).  GPU tensors only.
"""
import ctypes

import torch
from torch import Tensor

from . import _lib as L
from .ops import _f32c, _on, _plain, _ptr, _stream

MAX_CHANNELS = 4


def _check_sample_args(images, flow):
    if not (images.is_cuda and flow.is_cuda) or images.device != flow.device:
        raise RuntimeError("sv3d.texture_sample runs on the GPU only, images and flow on one device")
    if images.dim() != 4 or not 1 <= images.shape[1] <= MAX_CHANNELS or images.shape[2] < 1 or images.shape[3] < 1:
        raise ValueError("images must be [B,C,H,W] with C <= %d, got %s" % (MAX_CHANNELS, tuple(images.shape)))
    if flow.dim() != 3 or flow.shape[2] != 2 or flow.shape[0] != images.shape[0] or flow.shape[1] == 0:
        raise ValueError("flow must be [B,N,2] with the images' batch size, got %s" % (tuple(flow.shape),))


def _sample_launch(images, flow):
    _check_sample_args(images, flow)
    img, fl = _f32c(images), _f32c(flow)
    B, C, H, W = img.shape
    N = fl.shape[1]
    with _on(img.device):
        out = torch.empty((B, N, C), dtype=torch.float32, device=img.device)
        L.check(L.lib().sv3d_texture_sample(_ptr(img), _ptr(fl), B, C, H, W, N, _ptr(out), _stream(img)),
                "sv3d_texture_sample")
    return out


def _sample_backward_launch(images, flow, grad_out):
    img, fl, g = _f32c(images), _f32c(flow), _f32c(grad_out)
    B, C, H, W = img.shape
    N = fl.shape[1]
    with _on(img.device):
        grad_flow = torch.empty_like(fl)
        L.check(L.lib().sv3d_texture_sample_backward(_ptr(img), _ptr(fl), _ptr(g), B, C, H, W, N, _ptr(grad_flow),
                                                     _stream(img)), "sv3d_texture_sample_backward")
    return grad_flow


@torch.library.custom_op("sv3d::texture_sample", mutates_args=(), device_types="cuda")
def texture_sample_op(images: Tensor, flow: Tensor) -> Tensor:
    return _sample_launch(images, flow)


@texture_sample_op.register_fake
def _(images, flow):
    return images.new_empty((flow.shape[0], flow.shape[1], images.shape[1]), dtype=torch.float32)


@torch.library.custom_op("sv3d::texture_sample_backward", mutates_args=(), device_types="cuda")
def texture_sample_backward_op(images: Tensor, flow: Tensor, grad_out: Tensor) -> Tensor:
    return _sample_backward_launch(images, flow, grad_out)


@texture_sample_backward_op.register_fake
def _(images, flow, grad_out):
    return flow.new_empty(flow.shape, dtype=torch.float32)


def _sample_setup_context(ctx, inputs, output):
    images, flow = inputs
    if images.requires_grad:
        raise RuntimeError("sv3d.texture_sample has no gradient for the images")
    ctx.save_for_backward(images, flow)
    ctx.dtype = flow.dtype


def _sample_backward(ctx, grad_out):
    images, flow = ctx.saved_tensors
    return None, texture_sample_backward_op(images, flow, grad_out).to(ctx.dtype)


texture_sample_op.register_autograd(_sample_backward, setup_context=_sample_setup_context)


class _SampleFn(torch.autograd.Function):
    """Eager path without the custom-op dispatcher."""

    @staticmethod
    def forward(ctx, images, flow):
        out = _sample_launch(images, flow)
        ctx.save_for_backward(images, flow)
        ctx.dtype = flow.dtype
        return out

    @staticmethod
    def backward(ctx, grad_out):
        images, flow = ctx.saved_tensors
        return None, _sample_backward_launch(images, flow, grad_out).to(ctx.dtype)


def texture_sample(images, flow):
    """Bilinear samples of `images` [B,C,H,W] (C <= 4) at `flow` [B,N,2] in [-1,1] (x, y; the
    corners of the image are -1 and +1, outside reads zero) -> [B,N,C], differentiable in `flow`."""
    if images.requires_grad:
        raise RuntimeError("sv3d.texture_sample has no gradient for the images")
    if _plain(images, flow):
        return _SampleFn.apply(images, flow)
    return texture_sample_op(images, flow)


@torch.library.custom_op("sv3d::distance_transform", mutates_args=(), device_types="cuda")
def distance_transform_op(masks: Tensor, barrier: bool, k: float) -> Tensor:
    if not masks.is_cuda:
        raise RuntimeError("sv3d.distance_transform runs on the GPU only")
    if masks.dim() != 3 or masks.numel() == 0:
        raise ValueError("masks must be a non-empty [B,H,W] tensor, got %s" % (tuple(masks.shape),))
    m = _f32c(masks)
    B, H, W = m.shape
    with _on(m.device):
        out = torch.empty_like(m)
        nbytes = ctypes.c_size_t()
        L.check(L.lib().sv3d_distance_transform_workspace(B, H, W, ctypes.byref(nbytes)),
                "sv3d_distance_transform_workspace")
        ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=m.device)
        L.check(L.lib().sv3d_distance_transform(_ptr(m), B, H, W, 1 if barrier else 0, float(k), _ptr(out),
                                                _ptr(ws), nbytes.value, _stream(m)), "sv3d_distance_transform")
    return out


@distance_transform_op.register_fake
def _(masks, barrier, k):
    return masks.new_empty(masks.shape, dtype=torch.float32)


def dt_barrier(masks, k=50.0):
    """[B,H,W] masks -> sigmoid(k (EDT(1 - mask) - EDT(mask)) / max(H,W)); no gradient."""
    return distance_transform_op(masks.detach(), True, float(k))


def dt(masks):
    """[B,H,W] masks -> EDT(1 - mask) / max(H,W); no gradient."""
    return distance_transform_op(masks.detach(), False, 0.0)
