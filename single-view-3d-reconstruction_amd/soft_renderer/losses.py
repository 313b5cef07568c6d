"""Mesh regularisers with the constructor / call contract of `sr.LaplacianLoss(vertex, faces)`
and `sr.FlattenLoss(faces)` as the reference builds them at main.py:202-203 and calls them at
main.py:515-516.

Each call is one HIP launch that returns the loss and its vertex gradient (sv3d/meshreg.py): the
uniform Laplacian as a sparse neighbour stencil instead of the upstream dense V x V matmul, the
flatten loss without its ~40 elementwise launches.  GPU tensors only; CPU tensors raise.
"""
import numpy as np
import torch
import torch.nn as nn

from sv3d import meshreg

# the fused Laplacian kernel keeps 4 V floats in the LDS (150 KB at most)
_FUSED_MAX_VERTICES = 150 * 1024 // 16


def _np(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def _reduce(out, batch, average):
    return out.sum() / batch if average else out


class LaplacianLoss(nn.Module):
    """Uniform (graph) Laplacian, rows normalised by the vertex degree:
    loss[b] = sum_v |v - mean(neighbours of v)|^2."""

    def __init__(self, vertex, faces, average=False):
        super(LaplacianLoss, self).__init__()
        self.nv = int(vertex.shape[0])
        self.nf = int(faces.shape[0])
        self.average = average
        row_start, nbr = meshreg.neighbour_csr(_np(faces), self.nv)
        self.register_buffer('row_start', torch.from_numpy(row_start))
        self.register_buffer('nbr', torch.from_numpy(nbr))
        if self.nv > _FUSED_MAX_VERTICES:
            # y = L x through the CSR product, L = I - D^-1 A and its transpose
            deg = np.diff(row_start).astype(np.int64)
            rows = np.repeat(np.arange(self.nv), deg)
            live = np.flatnonzero(deg > 0)
            r = np.concatenate([rows, live])
            c = np.concatenate([nbr.astype(np.int64), live])
            w = np.concatenate([-1.0 / deg[rows], np.ones(live.size)]).astype(np.float32)
            for name, (rr, cc) in (('fwd', (r, c)), ('bwd', (c, r))):
                start, col, slot = meshreg.csr_from_coo(rr, cc, self.nv)
                val = np.zeros(col.size, np.float32)
                val[slot] = w
                self.register_buffer(name + '_val', torch.from_numpy(val)[None])
                self.register_buffer(name + '_start', torch.from_numpy(start))
                self.register_buffer(name + '_col', torch.from_numpy(col))

    def forward(self, x):
        if x.dim() != 3 or x.shape[1] != self.nv:
            raise ValueError("expected vertices [B,%d,3], got %s" % (self.nv, tuple(x.shape)))
        if self.nv > _FUSED_MAX_VERTICES:
            y = meshreg.mesh_spmv(x, (self.fwd_val, self.fwd_start, self.fwd_col),
                                  (self.bwd_val, self.bwd_start, self.bwd_col))
            out = y.pow(2).sum((1, 2))
        else:
            out = meshreg.laplacian_loss(x, self.row_start, self.nbr)
        return _reduce(out, x.size(0), self.average)


class FlattenLoss(nn.Module):
    """Penalises the dihedral angle across the edges of `meshreg.edge_quads` (the upstream
    module's list: edges found at corner positions (0,1) or (1,2) of some face): (cos + 1)^2
    summed over them, cos between the two in-face altitudes onto the shared edge."""

    def __init__(self, faces, average=False):
        super(FlattenLoss, self).__init__()
        self.nf = int(faces.shape[0])
        self.average = average
        f = _np(faces)
        self.nv = int(f.max()) + 1
        q = meshreg.edge_quads(f)
        for i, name in enumerate(('v0s', 'v1s', 'v2s', 'v3s')):
            self.register_buffer(name, torch.from_numpy(q[:, i].copy()))
        inc_start, inc = meshreg.incidence_csr(q, self.nv)
        self.register_buffer('quads', torch.from_numpy(q.astype(np.int32)))
        self.register_buffer('inc_start', torch.from_numpy(inc_start))
        self.register_buffer('inc', torch.from_numpy(inc))

    def forward(self, vertices, eps=1e-6):
        nv = vertices.shape[1]
        if vertices.dim() != 3 or nv < self.nv:
            raise ValueError("expected vertices [B,>=%d,3], got %s" % (self.nv, tuple(vertices.shape)))
        if nv != self.nv:                          # vertices beyond the last one a face uses
            pad = self.inc_start[-1:].expand(nv - self.nv)
            inc_start = torch.cat([self.inc_start, pad])
        else:
            inc_start = self.inc_start
        out = meshreg.flatten_loss(vertices, self.quads, inc_start, self.inc, eps)
        return _reduce(out, vertices.size(0), self.average)
