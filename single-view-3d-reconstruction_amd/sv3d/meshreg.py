"""Mesh regularisers on the device: entry points of the operators `sv3d::laplacian_loss`,
`sv3d::flatten_loss` and `sv3d::mesh_spmv` (csrc/torch_ops.cpp over `sv3d_laplacian_loss`,
`sv3d_flatten_loss`, `sv3d_mesh_spmv` of include/sv3d_hip.h), and the host-side index tables they
read (built once per topology with numpy).

The two loss operators return `(loss [B], dloss/dvertices [B,V,3])` from one launch; autograd only
scales the second output.  There is no CPU implementation: CPU tensors fail in the dispatcher.
"""
import numpy as np
import torch

from .ops import _ops


# ----------------------------------------------------------------------------------------------
# index tables
def _faces_np(faces):
    f = faces.detach().cpu().numpy() if torch.is_tensor(faces) else np.asarray(faces)
    f = f.astype(np.int64)
    if f.ndim != 2 or f.shape[1] != 3:
        raise ValueError("faces must be [F,3], got %s" % (f.shape,))
    return f


def neighbour_csr(faces, num_vertices):
    """(row_start [V+1], nbr [nnz]) int32: for every vertex the distinct vertices it shares a
    face edge with, ascending."""
    f = _faces_np(faces)
    if f.size and (f.min() < 0 or f.max() >= num_vertices):
        raise ValueError("face index outside [0, %d)" % num_vertices)
    a = np.concatenate([f[:, [0, 1]], f[:, [1, 0]], f[:, [1, 2]], f[:, [2, 1]], f[:, [2, 0]], f[:, [0, 2]]])
    a = a[a[:, 0] != a[:, 1]]
    key = np.unique(a[:, 0] * num_vertices + a[:, 1])
    rows, cols = key // num_vertices, key % num_vertices
    row_start = np.zeros(num_vertices + 1, np.int64)
    np.add.at(row_start, rows + 1, 1)
    return np.cumsum(row_start).astype(np.int32), cols.astype(np.int32)


def edge_quads(faces):
    """[E,4] int64 (v0, v1, v2, v3), the edge list of the upstream `FlattenLoss` (built from the
    faces at main.py:203): the distinct sorted pairs (v0 < v1) found at corner positions (0,1) and
    (1,2) of the faces -- an edge that sits at positions (2,0) in both of its faces is not in the
    list (about one edge in nine on an icosphere) -- ordered by (v0, v1), with v2 / v3 the
    opposite vertex of the first / second face, in face order, that holds the edge.  Edges with
    one face only are left out."""
    f = _faces_np(faces)
    nv = int(f.max()) + 1 if f.size else 0
    a = np.concatenate([f[:, [0, 1, 2]], f[:, [1, 2, 0]], f[:, [2, 0, 1]]])      # edge (a,b), opposite c
    face_id = np.tile(np.arange(f.shape[0]), 3)
    lo, hi = np.minimum(a[:, 0], a[:, 1]), np.maximum(a[:, 0], a[:, 1])
    order = np.lexsort((face_id, hi, lo))
    key = (lo * nv + hi)[order]
    opp = a[order, 2]
    uniq, first, count = np.unique(key, return_index=True, return_counts=True)
    listed = np.unique((lo * nv + hi)[: 2 * f.shape[0]])      # positions (0,1) and (1,2) only
    take = (count == 2) & np.isin(uniq, listed)
    sel = first[take]
    return np.stack([uniq[take] // nv, uniq[take] % nv, opp[sel], opp[sel + 1]], 1).reshape(-1, 4)


def incidence_csr(quads, num_vertices):
    """(inc_start [V+1], inc [4E]) int32: per vertex the entries 4 e + k with quads[e,k] == v,
    ascending."""
    q = np.asarray(quads, np.int64).reshape(-1, 4)
    verts = q.reshape(-1)
    order = np.argsort(verts, kind="stable")
    inc_start = np.zeros(num_vertices + 1, np.int64)
    np.add.at(inc_start, verts + 1, 1)
    return np.cumsum(inc_start).astype(np.int32), order.astype(np.int32)


def csr_from_coo(rows, cols, num_rows):
    """Structure of a sparse matrix from (possibly repeated) coordinates: (row_start [R+1],
    col [nnz], slot [len(rows)]) where slot[i] is the CSR entry coordinate i adds into; entries
    are ordered by (row, col)."""
    rows, cols = np.asarray(rows, np.int64), np.asarray(cols, np.int64)
    ncols = int(cols.max()) + 1 if cols.size else 1
    key, slot = np.unique(rows * ncols + cols, return_inverse=True)
    row_start = np.zeros(num_rows + 1, np.int64)
    np.add.at(row_start, key // ncols + 1, 1)
    return np.cumsum(row_start).astype(np.int32), (key % ncols).astype(np.int32), slot.astype(np.int64)


# ----------------------------------------------------------------------------------------------
# operators
def laplacian_loss(vertices, row_start, nbr):
    """loss [B] = sum_v |x_v - mean(neighbours)|^2 for neighbour lists from `neighbour_csr`."""
    return _ops.laplacian_loss(vertices, row_start, nbr)[0]


def flatten_loss(vertices, quads, inc_start, inc, eps=1e-6):
    """loss [B] = sum over the edges of `edge_quads` of (cos + 1)^2."""
    return _ops.flatten_loss(vertices, quads, inc_start, inc, float(eps))[0]


def mesh_spmv(x, csr, csr_t=None):
    """y = A x for A = (values [B or 1, nnz], row_start, col); `csr_t` is A's transpose in the same
    form (used by the backward), A itself when omitted (symmetric operators)."""
    csr_t = csr if csr_t is None else csr_t
    return _ops.mesh_spmv(x, csr[0], csr[1], csr[2], csr_t[0], csr_t[1], csr_t[2])
