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
