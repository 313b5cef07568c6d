"""sv3d: MI355X-native hot path (soft rasterizer + Chamfer NN, and the mesh / texture stages either
side of them) behind a C ABI.

`sv3d._lib` binds libsv3d_hip.so; `sv3d.ops`, `sv3d.meshreg` and `sv3d.texture` expose it as
PyTorch-ROCm custom ops (namespace `sv3d::`), all registered by importing this package.
"""
from . import _lib
from .ops import chamfer_nn, face_gather, face_gather_n, project_gather, render, soft_rasterize
from . import meshreg, texture
from .meshreg import flatten_loss, laplacian_loss, mesh_spmv
from .texture import dt, dt_barrier, texture_sample

__all__ = ["_lib", "chamfer_nn", "face_gather", "face_gather_n", "project_gather", "render", "soft_rasterize",
           "meshreg", "texture", "flatten_loss", "laplacian_loss", "mesh_spmv", "dt", "dt_barrier",
           "texture_sample"]
