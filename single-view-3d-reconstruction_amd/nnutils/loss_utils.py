"""Losses of the reference's nnutils/loss_utils.py that reach native code: `chamfer_loss`
(loss_utils.py:75-88) and the cotangent `LaplacianLoss` (loss_utils.py:565-583); the
reference's other losses are plain torch elementwise / matmul code that runs unchanged on
PyTorch-ROCm.  Importing this module has no device side effects (the reference's module
moves a constant to the GPU at import time, loss_utils.py:14-15).
"""
import torch

from chamferdist.chamferdist import ChamferDistance


def chamfer_loss(verts_org, verts_trans):
    """Chamfer loss between two vertex sets; loss_utils.py:75-88.

    Args:
        verts_org:   B x Nv x 3
        verts_trans: B x Nv x 3
    Returns:
        0-dim tensor: the mean over B x Nv of the squared distance from each point of
        `verts_trans` to its nearest point of `verts_org` (the reference keeps dist2 only).
    """
    _, dist2, _, _ = ChamferDistance()(verts_org, verts_trans)
    return dist2.mean()


class LaplacianLoss(object):
    """Encourages minimal mean curvature shapes; loss_utils.py:565-583.

    `faces` is B x F x 3; the call returns the mean over all B N vertices of |L x|_2 with the
    cotangent Laplacian L of nnutils/laplacian.py (built from the first vertices it is given).
    """

    def __init__(self, faces):
        from nnutils.laplacian import LaplacianModule
        self.laplacian = LaplacianModule(faces)
        self.Lx = None

    def __call__(self, verts):
        self.Lx = self.laplacian(verts)
        Lx = self.Lx.view(-1, self.Lx.size(2))
        return torch.norm(Lx, p=2, dim=1).mean()
