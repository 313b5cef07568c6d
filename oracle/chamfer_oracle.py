"""CPU oracle for the Chamfer nearest-neighbour loss -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  PARITY UNPINNED: see the header of chamfer_oracle.c (the algorithm
lives in the un-vendored `chamferdist` package; the reference's call site is
nnutils/loss_utils.py:75-88, which keeps dist2.mean() only).

Two restatements of the same contract:
  * `forward_c` / `backward_c`: the plain-C port (real fmaf), the bit-exact anchor;
  * `forward_np`: numpy, with fmaf emulated through float64 (a*b is exact in
    double, the sum then rounds twice; a disagreement with the C port needs a
    result within 2^-29 of an fp32 midpoint and has not been observed).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libchamfer_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/libchamfer_oracle.so missing: run `make -C oracle`")
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def forward_c(xyz1, xyz2):
    xyz1 = np.ascontiguousarray(xyz1, dtype=np.float32)
    xyz2 = np.ascontiguousarray(xyz2, dtype=np.float32)
    B, N, _ = xyz1.shape
    M = xyz2.shape[1]
    d1 = np.empty((B, N), np.float32)
    d2 = np.empty((B, M), np.float32)
    i1 = np.empty((B, N), np.int32)
    i2 = np.empty((B, M), np.int32)
    _lib().chamfer_oracle_forward(_p(xyz1, ctypes.c_float), _p(xyz2, ctypes.c_float),
                                  ctypes.c_int(B), ctypes.c_int(N), ctypes.c_int(M),
                                  _p(d1, ctypes.c_float), _p(d2, ctypes.c_float),
                                  _p(i1, ctypes.c_int32), _p(i2, ctypes.c_int32))
    return d1, d2, i1, i2


def backward_c(xyz1, xyz2, gd1, gd2, idx1, idx2):
    xyz1 = np.ascontiguousarray(xyz1, dtype=np.float32)
    xyz2 = np.ascontiguousarray(xyz2, dtype=np.float32)
    gd1 = np.ascontiguousarray(gd1, dtype=np.float32)
    gd2 = np.ascontiguousarray(gd2, dtype=np.float32)
    idx1 = np.ascontiguousarray(idx1, dtype=np.int32)
    idx2 = np.ascontiguousarray(idx2, dtype=np.int32)
    B, N, _ = xyz1.shape
    M = xyz2.shape[1]
    g1 = np.empty((B, N, 3), np.float64)
    g2 = np.empty((B, M, 3), np.float64)
    _lib().chamfer_oracle_backward(_p(xyz1, ctypes.c_float), _p(xyz2, ctypes.c_float),
                                   ctypes.c_int(B), ctypes.c_int(N), ctypes.c_int(M),
                                   _p(gd1, ctypes.c_float), _p(gd2, ctypes.c_float),
                                   _p(idx1, ctypes.c_int32), _p(idx2, ctypes.c_int32),
                                   _p(g1, ctypes.c_double), _p(g2, ctypes.c_double))
    return g1, g2


def _fmaf(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def _nn_np(a, b, chunk=256):
    n = a.shape[0]
    dist = np.empty(n, np.float32)
    idx = np.empty(n, np.int32)
    for s in range(0, n, chunk):
        q = a[s:s + chunk, None, :]
        dx = (b[None, :, 0] - q[..., 0]).astype(np.float32)
        dy = (b[None, :, 1] - q[..., 1]).astype(np.float32)
        dz = (b[None, :, 2] - q[..., 2]).astype(np.float32)
        d = _fmaf(dz, dz, _fmaf(dy, dy, (dx * dx).astype(np.float32)))
        j = d.argmin(1)                      # numpy argmin returns the FIRST minimum
        idx[s:s + chunk] = j
        dist[s:s + chunk] = d[np.arange(d.shape[0]), j]
    return dist, idx


def forward_np(xyz1, xyz2):
    xyz1 = np.asarray(xyz1, dtype=np.float32)
    xyz2 = np.asarray(xyz2, dtype=np.float32)
    out = [[], [], [], []]
    for b in range(xyz1.shape[0]):
        d1, i1 = _nn_np(xyz1[b], xyz2[b])
        d2, i2 = _nn_np(xyz2[b], xyz1[b])
        for o, v in zip(out, (d1, d2, i1, i2)):
            o.append(v)
    return tuple(np.stack(o) for o in out)


def chamfer_loss(verts_org, verts_trans):
    """nnutils/loss_utils.py:75-88: mean of dist2 (trans -> org direction) only."""
    _, d2, _, _ = forward_c(verts_org, verts_trans)
    return float(d2.astype(np.float64).mean())
