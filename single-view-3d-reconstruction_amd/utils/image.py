"""Distance transforms of the ground-truth masks with the names of the reference's
utils/image.py:94-113, computed on the GPU for a whole batch at once.

The reference calls these per sample on the host (scipy) for every batch, twice -- the second
time on the mirrored mask (main.py:259-260); the transform of a mirrored mask is the mirrored
transform, so one batched call and a `torch.flip` give both.
"""
import numpy as np
import torch

import sv3d.texture


def _run(mask, fn):
    as_numpy = not torch.is_tensor(mask)
    m = torch.as_tensor(np.asarray(mask, dtype=np.float32)) if as_numpy else mask
    single = m.dim() == 2
    if m.dim() not in (2, 3):
        raise ValueError("mask must be [H,W] or [B,H,W], got %s" % (tuple(m.shape),))
    src = m.device
    if not m.is_cuda:
        m = m.cuda()                      # raises without a GPU: there is no host implementation
    out = fn(m[None] if single else m)
    out = out[0] if single else out
    if as_numpy:
        return out.cpu().numpy()
    return out.to(src)


def compute_dt(mask):
    """utils/image.py:94-100: Euclidean distance to the mask, over the larger image side.
    `mask` [H,W] or [B,H,W], tensor (result on the same device) or array (result an array)."""
    return _run(mask, sv3d.texture.dt)


def compute_dt_barrier(mask, k=50):
    """utils/image.py:102-113: sigmoid(k * signed distance to the mask boundary / image side)."""
    return _run(mask, lambda m: sv3d.texture.dt_barrier(m, k))
