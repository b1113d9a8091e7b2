"""TEST INFRASTRUCTURE ONLY — numpy-facing ctypes binding of the C oracle."""
import ctypes
import os

import numpy as np

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.build(verbose=False)
        _lib = ctypes.CDLL(path)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def chamfer_nn(q, c):
    """q (B,Nq,3), c (B,Nc,3) float32 -> (min sq dist (B,Nq) f32, argmin (B,Nq) i64)."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    c = np.ascontiguousarray(c, dtype=np.float32)
    B, Nq, _ = q.shape
    Nc = c.shape[1]
    mind = np.empty((B, Nq), np.float32)
    arg = np.empty((B, Nq), np.int64)
    lib().pno_chamfer_nn(_p(q), _p(c), B, Nq, Nc, _p(mind), _p(arg))
    return mind, arg


def knn(x, k, mode=0):
    """x (B,C,N) float32 channel-first -> idx (B,N,k) int64, best first (see pno_knn)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, C, N = x.shape
    idx = np.empty((B, N, k), np.int64)
    lib().pno_knn(_p(x), B, C, N, k, mode, _p(idx))
    return idx


def knn_row_values(xb, i, mode=0):
    """One row of the reference's (negated) pairwise-distance matrix for a single (C,N) item."""
    xb = np.ascontiguousarray(xb, dtype=np.float32)
    C, N = xb.shape
    c1 = C if mode == 0 else 3
    xx = np.zeros(N, np.float32)
    for c in range(c1):
        xx = (xb[c].astype(np.float64) * xb[c].astype(np.float64) + xx.astype(np.float64)).astype(np.float32)
    v = np.empty(N, np.float32)
    lib().pno_knn_values(_p(xb), C, N, mode, int(i), _p(xx), _p(v))
    return v


def dot_argmax(centers, x):
    """centers (Nc,D), x (Nq,D) float32 -> (Nq,) int64 index of the centre with the largest dot."""
    centers = np.ascontiguousarray(centers, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape[0], np.int64)
    lib().pno_dot_argmax(_p(centers), centers.shape[0], _p(x), x.shape[0], x.shape[1], _p(out))
    return out


def kth_largest_dot(x, K):
    """x (N,D) -> (N,) K-th largest entry of every row of x x^T (fmaf-chain dots)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape[0], np.float32)
    lib().pno_kth_largest_dot(_p(x), x.shape[0], x.shape[1], int(K), _p(out))
    return out
