"""Losses of the reference's nnutils/loss_utils.py that reach native code: `chamfer_loss`
(loss_utils.py:75-88), the cotangent `LaplacianLoss` (loss_utils.py:565-583) and the
texture-flow distance-transform loss (loss_utils.py:242-301); the
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


def texture_dt_loss_per_instance(texture_flow, dist_transf, vis_rend=None, cams=None, verts=None, tex_pred=None):
    """loss_utils.py:242-283: mean over a sample's texels of the distance map at the texel's flow.

    texture_flow: B x F x T x T x 2 in [-1, 1]; dist_transf: B x 1 x N x N -> B.
    The reference's interactive matplotlib branch (`vis_rend`) is not provided.
    """
    if vis_rend is not None:
        raise NotImplementedError("the interactive visualisation branch is not provided")
    from nnutils import geom_utils
    return geom_utils.sample_textures(texture_flow, dist_transf).mean([1, 2, 3])


def texture_dt_loss(texture_flow, dist_transf, vis_rend=None, cams=None, verts=None, tex_pred=None, probs=None,
                    num_cams=1):
    """loss_utils.py:286-301: the per-instance loss weighted by the camera probabilities
    `probs` (Nc x B; uniform 1/Nc when omitted), summed over cameras, averaged over the batch."""
    loss = texture_dt_loss_per_instance(texture_flow, dist_transf, vis_rend, cams, verts, tex_pred).view(num_cams, -1)
    if probs is None:
        probs = torch.ones_like(loss) / num_cams
    return (loss * probs).sum(0).mean()
