"""Mesh regularisers on the device: torch custom ops over `sv3d_laplacian_loss`,
`sv3d_flatten_loss` and `sv3d_mesh_spmv` (include/sv3d_hip.h), and the host-side index tables
they read (built once per topology with numpy).

The two loss ops return `(loss [B], dloss/dvertices [B,V,3])` from one launch; autograd only
scales the second output.  There is no CPU implementation: CPU tensors raise.
"""
import numpy as np
import torch
from torch import Tensor

from . import _lib as L
from .ops import _f32c, _on, _plain, _ptr, _stream


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
# ops
def _check_vertices(x):
    if not x.is_cuda:
        raise RuntimeError("sv3d mesh regularisers run on the GPU only (got a %s tensor)" % x.device.type)
    if x.dim() != 3 or x.shape[2] != 3 or x.shape[0] == 0 or x.shape[1] == 0:
        raise ValueError("vertices must be a non-empty [B,V,3] tensor, got %s" % (tuple(x.shape),))


def _check_i32(t, name, like=None):
    if t.dtype != torch.int32 or not t.is_contiguous():
        raise ValueError("%s must be a contiguous int32 tensor" % name)
    if like is not None and t.device != like.device:
        # a host pointer handed to a kernel is a GPU memory fault, not an exception: refuse it here
        raise RuntimeError("%s is on %s but the vertices are on %s (move the loss module to the GPU)"
                           % (name, t.device, like.device))


def _laplacian_launch(vertices, row_start, nbr):
    _check_vertices(vertices)
    _check_i32(row_start, "row_start", vertices)
    _check_i32(nbr, "nbr", vertices)
    x = _f32c(vertices)
    B, V = x.shape[0], x.shape[1]
    if row_start.numel() != V + 1:
        raise ValueError("row_start must hold V + 1 = %d entries, got %d" % (V + 1, row_start.numel()))
    with _on(x.device):
        loss = torch.empty((B,), dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x)
        L.check(L.lib().sv3d_laplacian_loss(_ptr(x), _ptr(row_start), _ptr(nbr), B, V, _ptr(loss), _ptr(grad),
                                            _stream(x)), "sv3d_laplacian_loss")
    return loss, grad


@torch.library.custom_op("sv3d::laplacian_loss", mutates_args=(), device_types="cuda")
def laplacian_loss_op(vertices: Tensor, row_start: Tensor, nbr: Tensor) -> tuple[Tensor, Tensor]:
    return _laplacian_launch(vertices, row_start, nbr)


@laplacian_loss_op.register_fake
def _(vertices, row_start, nbr):
    return (vertices.new_empty((vertices.shape[0],), dtype=torch.float32),
            vertices.new_empty(vertices.shape, dtype=torch.float32))


def _flatten_launch(vertices, quads, inc_start, inc, eps):
    _check_vertices(vertices)
    for t, name in ((quads, "quads"), (inc_start, "inc_start"), (inc, "inc")):
        _check_i32(t, name, vertices)
    x = _f32c(vertices)
    B, V, E = x.shape[0], x.shape[1], quads.shape[0]
    if quads.dim() != 2 or quads.shape[1] != 4 or E == 0:
        raise ValueError("quads must be a non-empty [E,4] tensor, got %s" % (tuple(quads.shape),))
    if inc_start.numel() != V + 1 or inc.numel() != 4 * E:
        raise ValueError("incidence tables must hold V + 1 and 4 E entries")
    with _on(x.device):
        loss = torch.empty((B,), dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x)
        L.check(L.lib().sv3d_flatten_loss(_ptr(x), _ptr(quads), _ptr(inc_start), _ptr(inc), B, V, E, float(eps),
                                          _ptr(loss), _ptr(grad), _stream(x)), "sv3d_flatten_loss")
    return loss, grad


@torch.library.custom_op("sv3d::flatten_loss", mutates_args=(), device_types="cuda")
def flatten_loss_op(vertices: Tensor, quads: Tensor, inc_start: Tensor, inc: Tensor,
                    eps: float) -> tuple[Tensor, Tensor]:
    return _flatten_launch(vertices, quads, inc_start, inc, eps)


@flatten_loss_op.register_fake
def _(vertices, quads, inc_start, inc, eps):
    return (vertices.new_empty((vertices.shape[0],), dtype=torch.float32),
            vertices.new_empty(vertices.shape, dtype=torch.float32))


def _loss_setup_context(ctx, inputs, output):
    ctx.save_for_backward(output[1])
    ctx.dtype = inputs[0].dtype
    ctx.set_materialize_grads(False)


def _laplacian_backward(ctx, grad_loss, _grad_of_grad):
    (g,) = ctx.saved_tensors
    if grad_loss is None:
        return None, None, None
    return (grad_loss.to(torch.float32)[:, None, None] * g).to(ctx.dtype), None, None


def _flatten_backward(ctx, grad_loss, _grad_of_grad):
    (g,) = ctx.saved_tensors
    if grad_loss is None:
        return None, None, None, None, None
    return (grad_loss.to(torch.float32)[:, None, None] * g).to(ctx.dtype), None, None, None, None


laplacian_loss_op.register_autograd(_laplacian_backward, setup_context=_loss_setup_context)
flatten_loss_op.register_autograd(_flatten_backward, setup_context=_loss_setup_context)


@torch.library.custom_op("sv3d::mesh_spmv", mutates_args=(), device_types="cuda")
def mesh_spmv_op(x: Tensor, values: Tensor, row_start: Tensor, col: Tensor,
                 values_t: Tensor, row_start_t: Tensor, col_t: Tensor) -> Tensor:
    _check_vertices(x)
    _check_i32(row_start, "row_start", x)
    _check_i32(col, "col", x)
    if values.device != x.device:
        raise RuntimeError("values are on %s but x is on %s" % (values.device, x.device))
    xs, vals = _f32c(x), _f32c(values)
    B, V, nnz = xs.shape[0], xs.shape[1], col.numel()
    if vals.dim() != 2 or vals.shape[1] != nnz or vals.shape[0] not in (1, B):
        raise ValueError("values must be [B or 1, nnz=%d], got %s" % (nnz, tuple(vals.shape)))
    if row_start.numel() != V + 1:
        raise ValueError("row_start must hold V + 1 = %d entries, got %d" % (V + 1, row_start.numel()))
    with _on(xs.device):
        y = torch.empty_like(xs)
        L.check(L.lib().sv3d_mesh_spmv(_ptr(vals), _ptr(row_start), _ptr(col), _ptr(xs), B, V, nnz, vals.shape[0],
                                       _ptr(y), _stream(xs)), "sv3d_mesh_spmv")
    return y


@mesh_spmv_op.register_fake
def _(x, values, row_start, col, values_t, row_start_t, col_t):
    return x.new_empty(x.shape, dtype=torch.float32)


def _spmv_setup_context(ctx, inputs, output):
    x, values, row_start, col, values_t, row_start_t, col_t = inputs
    ctx.save_for_backward(values, row_start, col, values_t, row_start_t, col_t)
    ctx.dtype = x.dtype


def _spmv_backward(ctx, grad_y):
    values, row_start, col, values_t, row_start_t, col_t = ctx.saved_tensors
    gx = mesh_spmv_op(grad_y, values_t, row_start_t, col_t, values, row_start, col)
    return gx.to(ctx.dtype), None, None, None, None, None, None


mesh_spmv_op.register_autograd(_spmv_backward, setup_context=_spmv_setup_context)


class _FusedLossFn(torch.autograd.Function):
    """Eager path of the two loss ops without the custom-op dispatcher (which costs more host time
    than the kernels take): `op` is the op's own python body."""

    @staticmethod
    def forward(ctx, vertices, op, *tables):
        loss, grad = op(vertices, *tables)
        ctx.save_for_backward(grad)
        ctx.dtype, ctx.n = vertices.dtype, len(tables)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        (g,) = ctx.saved_tensors
        return ((grad_loss.to(torch.float32)[:, None, None] * g).to(ctx.dtype), None) + (None,) * ctx.n


def laplacian_loss(vertices, row_start, nbr):
    """loss [B] = sum_v |x_v - mean(neighbours)|^2 for neighbour lists from `neighbour_csr`."""
    if _plain(vertices, row_start, nbr):
        return _FusedLossFn.apply(vertices, _laplacian_launch, row_start, nbr)
    return laplacian_loss_op(vertices, row_start, nbr)[0]


def flatten_loss(vertices, quads, inc_start, inc, eps=1e-6):
    """loss [B] = sum over the edges of `edge_quads` of (cos + 1)^2."""
    if _plain(vertices, quads, inc_start, inc):
        return _FusedLossFn.apply(vertices, _flatten_launch, quads, inc_start, inc, eps)
    return flatten_loss_op(vertices, quads, inc_start, inc, eps)[0]


def mesh_spmv(x, csr, csr_t=None):
    """y = A x for A = (values [B or 1, nnz], row_start, col); `csr_t` is A's transpose in the same
    form (used by the backward), A itself when omitted (symmetric operators)."""
    csr_t = csr if csr_t is None else csr_t
    return mesh_spmv_op(x, csr[0], csr[1], csr[2], csr_t[0], csr_t[1], csr_t[2])
