"""`soft_renderer` drop-in (import soft_renderer as sr): the subset of SoftRas the reference
uses, over hand-written gfx950 kernels (see core.py)."""
from . import functional
from .core import (AmbientLighting, DirectionalLighting, Lighting, Mesh, SoftRasterizer,
                   SoftRenderer, Transform)
from .functional import load_obj, save_obj
from .losses import FlattenLoss, LaplacianLoss

__all__ = ["AmbientLighting", "DirectionalLighting", "Lighting", "Mesh", "SoftRasterizer",
           "SoftRenderer", "Transform", "functional", "load_obj", "save_obj", "FlattenLoss",
           "LaplacianLoss"]
