"""sv3d: MI355X-native hot path (soft rasterizer + Chamfer NN) behind a C ABI.

`sv3d._lib` binds libsv3d_hip.so, `sv3d.ops` exposes it as PyTorch-ROCm custom ops.
"""
from . import _lib
from .ops import chamfer_nn, face_gather, face_gather_n, project_gather, render, soft_rasterize

__all__ = ["_lib", "chamfer_nn", "face_gather", "face_gather_n", "project_gather", "render", "soft_rasterize"]
