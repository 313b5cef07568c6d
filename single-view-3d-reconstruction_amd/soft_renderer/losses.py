"""Mesh regularisers with the constructor / call contract of `sr.LaplacianLoss(vertex, faces)`
and `sr.FlattenLoss(faces)` as the reference builds them at main.py:202-203 and calls them at
main.py:515-516.  Plain torch: these are O(V) / O(E) terms next to the rasterizer."""
import numpy as np
import torch
import torch.nn as nn


def _np(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


class LaplacianLoss(nn.Module):
    """Uniform (graph) Laplacian, rows normalised by the vertex degree:
    loss[b] = sum_v |v - mean(neighbours of v)|^2."""

    def __init__(self, vertex, faces, average=False):
        super(LaplacianLoss, self).__init__()
        self.nv = int(vertex.shape[0])
        self.nf = int(faces.shape[0])
        self.average = average
        f = _np(faces).astype(np.int64)
        lap = np.zeros((self.nv, self.nv), dtype=np.float32)
        for a, b in ((0, 1), (1, 0), (1, 2), (2, 1), (2, 0), (0, 2)):
            lap[f[:, a], f[:, b]] = -1
        idx = np.arange(self.nv)
        lap[idx, idx] = -lap.sum(1)
        lap /= np.maximum(lap[idx, idx], 1e-12)[:, None]
        self.register_buffer('laplacian', torch.from_numpy(lap))

    def forward(self, x):
        y = torch.matmul(self.laplacian, x)
        out = y.pow(2).sum(tuple(range(1, y.dim())))
        return out.sum() / x.size(0) if self.average else out


class FlattenLoss(nn.Module):
    """Penalises the dihedral angle across every interior edge: (cos + 1)^2 summed over
    edges, cos between the two in-face altitudes onto the shared edge."""

    def __init__(self, faces, average=False):
        super(FlattenLoss, self).__init__()
        self.nf = int(faces.shape[0])
        self.average = average
        f = _np(faces).astype(np.int64)
        opposite = {}
        for tri in f:
            for k in range(3):
                a, b, c = int(tri[k]), int(tri[(k + 1) % 3]), int(tri[(k + 2) % 3])
                opposite.setdefault((min(a, b), max(a, b)), []).append(c)
        quads = [(e[0], e[1], o[0], o[1]) for e, o in sorted(opposite.items()) if len(o) == 2]
        q = np.asarray(quads, dtype=np.int64).reshape(-1, 4)
        for i, name in enumerate(('v0s', 'v1s', 'v2s', 'v3s')):
            self.register_buffer(name, torch.from_numpy(q[:, i].copy()))

    def forward(self, vertices, eps=1e-6):
        v0, v1 = vertices[:, self.v0s], vertices[:, self.v1s]
        a = v1 - v0
        al2 = a.pow(2).sum(-1)

        def altitude(vk):
            b = vk - v0
            bl1 = (b.pow(2).sum(-1) + eps).sqrt()
            ab = (a * b).sum(-1)
            cos = ab / ((al2 + eps).sqrt() * bl1 + eps)
            sin = (1 - cos.pow(2) + eps).sqrt()
            return b - a * (ab / (al2 + eps))[..., None], bl1 * sin

        c1, l1 = altitude(vertices[:, self.v2s])
        c2, l2 = altitude(vertices[:, self.v3s])
        cos = (c1 * c2).sum(-1) / (l1 * l2 + eps)
        out = (cos + 1).pow(2).sum(tuple(range(1, cos.dim())))
        return out.sum() / vertices.size(0) if self.average else out
