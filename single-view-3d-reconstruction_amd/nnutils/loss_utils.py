"""Hot-path loss of the reference's nnutils/loss_utils.py: `chamfer_loss`.

Only the function that reaches native code is provided here (loss_utils.py:75-88); the
reference's other losses are plain torch elementwise / matmul code that runs unchanged on
PyTorch-ROCm.  Importing this module has no device side effects (the reference's module
moves a constant to the GPU at import time, loss_utils.py:14-15).
"""
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
