"""ChamferDistance module with the call contract used at nnutils/loss_utils.py:85-86:
`dist1, dist2, idx1, idx2 = ChamferDistance()(xyz1, xyz2)`.

Backed by the `sv3d::chamfer_nn` op (hand-written gfx950 kernel behind the C ABI);
CUDA(HIP) tensors only -- there is no CPU path.
"""
import torch

import sv3d


class ChamferDistance(torch.nn.Module):
    """forward(xyz1 [B,N,3], xyz2 [B,M,3]) -> (dist1 [B,N], dist2 [B,M], idx1 [B,N], idx2 [B,M]).

    dist* are SQUARED nearest-neighbour distances (differentiable w.r.t. both clouds with
    the indices frozen), idx* are int32 indices into the other cloud, lowest index on ties.
    """

    def forward(self, xyz1, xyz2):
        return sv3d.chamfer_nn(xyz1, xyz2)
