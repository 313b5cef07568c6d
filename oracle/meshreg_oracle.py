"""CPU restatement of the mesh regularisers -- TEST INFRASTRUCTURE ONLY (imported by tests/, never
by the product path).  PARITY UNPINNED for every function here:

* `laplacian_loss`, `flatten_loss`: the uniform-Laplacian and flatten losses the reference takes
  from the third-party SoftRas package (`sr.LaplacianLoss`, `sr.FlattenLoss`; built at
  main.py:202-203, called at main.py:515-516).  That package is not part of the reference tree
  and is not installed: these two are restated from its published definitions.
* `cotangent`, `cotangent_laplacian`, `laplacian_apply`, `cotangent_laplacian_loss`: the in-tree
  cotangent Laplacian, nnutils/laplacian.py:26-137 and loss_utils.py:565-583, restated with dense
  numpy from a reading of that source.  The reference holds no test or stored output for it, and
  no vectors produced by the reference module itself are available.

Everything runs in the dtype of its input (pass float64 arrays for the truth, float32 for the
rounding floor).
"""
import numpy as np
import torch


def uniform_laplacian_matrix(num_vertices, faces):
    """Dense V x V operator: -1 for every pair of vertices sharing a face edge, the degree on the
    diagonal, each row divided by its diagonal (rows of isolated vertices stay zero)."""
    f = np.asarray(faces, np.int64)
    lap = np.zeros((num_vertices, num_vertices), np.float64)
    for a, b in ((0, 1), (1, 0), (1, 2), (2, 1), (2, 0), (0, 2)):
        lap[f[:, a], f[:, b]] = -1
    idx = np.arange(num_vertices)
    lap[idx, idx] = 0
    lap[idx, idx] = -lap.sum(1)
    diag = lap[idx, idx].copy()
    lap[diag > 0] /= diag[diag > 0, None]
    return lap


def laplacian_loss(vertices, faces):
    """vertices: torch [B,V,3] -> loss [B] = sum over vertices and axes of (L x)^2."""
    lap = torch.from_numpy(uniform_laplacian_matrix(vertices.shape[1], faces)).to(vertices.dtype)
    y = torch.matmul(lap, vertices)
    return y.pow(2).sum((1, 2))


def flatten_quads(faces):
    """The upstream module's edge table, built its way: the set of sorted vertex pairs taken from
    the face columns 0:2 and 1:3 only (`np.concatenate((faces[:, 0:2], faces[:, 1:3]))`, so an
    edge lying at corner positions (2,0) in both of its faces never enters); for each pair, going
    through the faces in order, the remaining vertex of the first face holding both ends is v2 and
    that of the second face is v3.  Returned in ascending (v0, v1) order (upstream iterates a
    python set; the loss is a sum, so the order is immaterial).  Pairs held by fewer than two
    faces are dropped (upstream's v2 / v3 lists would fall out of step on such a mesh)."""
    f = np.asarray(faces, np.int64)
    pairs = sorted(set(tuple(sorted(p)) for p in np.concatenate((f[:, 0:2], f[:, 1:3]), 0).tolist()))
    quads = []
    for v0, v1 in pairs:
        rest = [int([v for v in tri if v != v0 and v != v1][0]) for tri in f.tolist() if v0 in tri and v1 in tri]
        if len(rest) >= 2:
            quads.append((v0, v1, rest[0], rest[1]))
    return np.asarray(quads, np.int64).reshape(-1, 4)


def flatten_loss(vertices, faces, eps=1e-6):
    """vertices: torch [B,V,3] -> loss [B] = sum over the edges of `flatten_quads` of (cos + 1)^2, cos between
    the in-face altitudes onto the shared edge, eps placed as in the upstream module."""
    q = torch.from_numpy(flatten_quads(faces))
    v0, v1 = vertices[:, q[:, 0]], vertices[:, q[:, 1]]
    a = v1 - v0
    al2 = a.pow(2).sum(-1)

    def altitude(vk):
        b = vk - v0
        bl1 = (b.pow(2).sum(-1) + eps).sqrt()
        ab = (a * b).sum(-1)
        cos = ab / ((al2 + eps).sqrt() * bl1 + eps)
        sin = (1 - cos.pow(2) + eps).sqrt()
        return b - a * (ab / (al2 + eps))[..., None], bl1 * sin

    c1, l1 = altitude(vertices[:, q[:, 2]])
    c2, l2 = altitude(vertices[:, q[:, 3]])
    cos = (c1 * c2).sum(-1) / (l1 * l2 + eps)
    return (cos + 1).pow(2).sum(1)


def cotangent(V, F):
    """nnutils/laplacian.py:106-137.  V [B,N,3], F [B,Fn,3] (numpy) -> [B,Fn,3]: per face, for the
    edges 23, 31, 12, (sum of the other two squared edge lengths - its own) / (2 * Heron area) / 4,
    i.e. half the cotangent of the angle opposite that edge."""
    B = V.shape[0]
    bi = np.arange(B)[:, None]
    v1, v2, v3 = V[bi, F[:, :, 0]], V[bi, F[:, :, 1]], V[bi, F[:, :, 2]]
    l1 = np.sqrt(((v2 - v3) ** 2).sum(2))
    l2 = np.sqrt(((v3 - v1) ** 2).sum(2))
    l3 = np.sqrt(((v1 - v2) ** 2).sum(2))
    sp = (l1 + l2 + l3) * 0.5
    A = 2 * np.sqrt(sp * (sp - l1) * (sp - l2) * (sp - l3))
    C = np.stack([l2 ** 2 + l3 ** 2 - l1 ** 2, l1 ** 2 + l3 ** 2 - l2 ** 2, l1 ** 2 + l2 ** 2 - l3 ** 2], 2)
    return C / A[:, :, None] / 4


def cotangent_laplacian(V, F):
    """nnutils/laplacian.py:33-58 as dense per-mesh matrices [B,N,N]: entry (F[:,1],F[:,2]) gets
    the weight of edge 23, (F[:,2],F[:,0]) of 31, (F[:,0],F[:,1]) of 12; symmetrised; minus the
    diagonal of row sums."""
    B, N = V.shape[0], V.shape[1]
    C = cotangent(V, F)
    Lm = np.zeros((B, N, N), V.dtype)
    for b in range(B):
        rows = F[b][:, [1, 2, 0]].reshape(-1)
        cols = F[b][:, [2, 0, 1]].reshape(-1)
        np.add.at(Lm[b], (rows, cols), C[b].reshape(-1))
        Lm[b] = Lm[b] + Lm[b].T
        Lm[b] -= np.diag(Lm[b].sum(1))
    return Lm


def laplacian_apply(Lm, X):
    """Laplacian.forward / .backward, nnutils/laplacian.py:67-103: L x per mesh (L symmetric, so
    the backward is the same product applied to the incoming gradient)."""
    return np.einsum("bij,bjc->bic", Lm, X)


def cotangent_laplacian_loss(Lm, X):
    """loss_utils.py:578-583: mean over all B N vertices of |L x|_2."""
    return np.sqrt((laplacian_apply(Lm, X) ** 2).sum(-1)).mean()
