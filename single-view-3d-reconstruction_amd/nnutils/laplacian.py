"""Cotangent-weight mesh Laplacian with the interface of the reference's nnutils/laplacian.py
(`LaplacianModule(faces)(V) -> L V`, `cotangent(V, F)`), kept on the device.

The reference assembles L with scipy on the first call and then multiplies on the host on every
call, forward and backward (laplacian.py:79-84,96-101: device -> numpy -> scipy -> device).  Here
the weights are computed once on the device and every product is one launch of the CSR kernel
behind `sv3d.meshreg.mesh_spmv`.
"""
import numpy as np
import torch

from sv3d import meshreg


def cotangent(V, F):
    """laplacian.py:106-137.  V: B x N x 3, F: B x F x 3 -> B x F x 3: per face, for the edges
    opposite its corners 1, 2, 3 (the reference's columns 23, 31, 12), the sum of the other two
    squared edge lengths minus its own, over 2 * (Heron area) * 4 -- half the cotangent of the angle
    opposite that edge."""
    batch = torch.arange(V.shape[0], device=V.device)[:, None]
    corners = [V[batch, F[:, :, k].long()] for k in range(3)]                       # three B x F x 3
    # edge k is opposite corner k
    length = torch.stack([(corners[(k + 1) % 3] - corners[(k + 2) % 3]).pow(2).sum(-1).sqrt() for k in range(3)], -1)
    half = length.sum(-1) * 0.5
    heron = half * (half - length[..., 0]) * (half - length[..., 1]) * (half - length[..., 2])
    sq = length.pow(2)
    return (sq.sum(-1, keepdim=True) - 2 * sq) / (2 * heron.sqrt())[..., None] / 4


class LaplacianModule(torch.nn.Module):
    """laplacian.py:26-60: L is built from the first batch of vertices it sees (one block per batch
    element, stacked to a B N x B N operator) and reused for every later call, which must
    therefore have the same batch size."""

    def __init__(self, faces):
        super(LaplacianModule, self).__init__()
        if faces.dim() != 3 or faces.shape[2] != 3:
            raise ValueError("faces must be B x F x 3, got %s" % (tuple(faces.shape),))
        self.SF = faces.detach()
        self.F_np = faces.detach().cpu().numpy()
        self.L = None

    def _build(self, V):
        B, N = V.shape[0], V.shape[1]
        if self.F_np.shape[0] != B:
            raise ValueError("faces hold %d meshes, vertices %d" % (self.F_np.shape[0], B))
        F = self.F_np.astype(np.int64) + (np.arange(B).reshape(-1, 1, 1) * N)
        F = F.reshape(-1, 3)
        if F.min() < 0 or self.F_np.max() >= N:
            raise ValueError("face index outside [0, %d)" % N)
        rows = F[:, [1, 2, 0]].reshape(-1)
        cols = F[:, [2, 0, 1]].reshape(-1)
        diag = np.arange(B * N)
        # L = C + C^T - diag(row sums of C + C^T): every weight lands on (r,c), (c,r), -(r,r), -(c,c)
        r = np.concatenate([rows, cols, rows, cols, diag])
        c = np.concatenate([cols, rows, rows, cols, diag])
        start, col, slot = meshreg.csr_from_coo(r, c, B * N)
        n = rows.size
        dev = V.device
        # summed on the host (once): a sequential index_add keeps the weights repeatable
        C = cotangent(V.detach().float(), self.SF.to(dev)).reshape(-1).cpu()
        values = torch.zeros(col.size, dtype=torch.float32)
        values.index_add_(0, torch.from_numpy(slot[:4 * n]), torch.cat([C, C, -C, -C]))
        self.L = (values[None].to(dev), torch.from_numpy(start).to(dev), torch.from_numpy(col).to(dev))
        self._shape = (B, N)

    def forward(self, V):
        if self.L is None:
            self._build(V)
        if tuple(V.shape[:2]) != self._shape or V.shape[2] != 3:
            raise ValueError("the Laplacian was built for vertices %s x 3, got %s"
                             % (self._shape, tuple(V.shape)))
        B, N = self._shape
        return meshreg.mesh_spmv(V.reshape(1, B * N, 3), self.L).reshape(B, N, 3)
