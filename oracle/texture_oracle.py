"""CPU oracle of the texture path -- TEST INFRASTRUCTURE ONLY (imported by tests/, never by the
product path).

Both functions of the reference on this path are thin calls into libraries that ARE installed
here, so the oracle calls those same libraries on the CPU and parity is pinned by them:

* `sample_textures`: geom_utils.py:11-26 = `torch.nn.functional.grid_sample(images, flow,
  align_corners=True)` + permute (run here in float64);
* `compute_dt`, `compute_dt_barrier`: utils/image.py:94-113 = `scipy.ndimage.distance_transform_edt`
  of (1 - mask) and of mask, scaled by max(shape), through 1 / (1 + exp(-k d)).
"""
import numpy as np
import torch


def sample_textures(texture_flow, images):
    """texture_flow [B,F,T,T,2], images [B,C,N,N] (torch, any float dtype) -> [B,F,T*T,C]."""
    T = texture_flow.size(-2)
    F = texture_flow.size(1)
    grid = texture_flow.reshape(-1, F, T * T, 2)
    return torch.nn.functional.grid_sample(images, grid, align_corners=True).permute(0, 2, 3, 1)


def compute_dt(mask):
    from scipy.ndimage import distance_transform_edt
    mask = np.asarray(mask)
    return distance_transform_edt(1 - mask) / max(mask.shape)


def compute_dt_barrier(mask, k=50):
    from scipy.ndimage import distance_transform_edt
    mask = np.asarray(mask)
    diff = (distance_transform_edt(1 - mask) - distance_transform_edt(mask)) / max(mask.shape)
    return 1.0 / (1 + np.exp(k * -diff))
