"""Tensor-level wrappers over the C ABI.  Each function validates layout, allocates
outputs/workspace through torch's caching allocator and launches on torch's current
HIP stream.  No arithmetic happens here."""
import torch

from . import _lib
from ._lib import check, current_stream, ptr, require_cuda


def _f32c(t, name):
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    return t.contiguous()


def chamfer_nn(a, b, side_a=True, side_b=True):
    """Nearest neighbours between two batched clouds.

    a: (B, Na, 3), b: (B, Nb, 3) fp32 on the GPU.
    Returns (minA, argA, minB, argB): for every a_i the squared distance to and index of
    its nearest b_j (ties -> smallest j), and vice versa.  Entries of a side that was not
    requested are None.
    """
    require_cuda(a, b)
    a = _f32c(a, "a")
    b = _f32c(b, "b")
    if a.dim() != 3 or b.dim() != 3 or a.shape[2] != 3 or b.shape[2] != 3 or a.shape[0] != b.shape[0]:
        raise ValueError("chamfer_nn expects (B,Na,3) and (B,Nb,3), got %s and %s"
                         % (tuple(a.shape), tuple(b.shape)))
    B, Na, _ = a.shape
    Nb = b.shape[1]
    if Na == 0 or Nb == 0 or B == 0:
        raise ValueError("chamfer_nn: empty cloud")
    lib = _lib.load()
    dev = a.device
    minA = argA = minB = argB = None
    if side_a:
        minA = torch.empty((B, Na), dtype=torch.float32, device=dev)
        argA = torch.empty((B, Na), dtype=torch.int64, device=dev)
    if side_b:
        minB = torch.empty((B, Nb), dtype=torch.float32, device=dev)
        argB = torch.empty((B, Nb), dtype=torch.int64, device=dev)
    wsz = lib.pn_chamfer_nn_workspace(B, Na, Nb)
    ws = torch.empty(wsz, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.pn_chamfer_nn_f32(ptr(a), ptr(b), B, Na, Nb, ptr(minA), ptr(argA), ptr(minB),
                                   ptr(argB), ptr(ws), wsz, current_stream(dev))
    check(rc, "pn_chamfer_nn_f32")
    return minA, argA, minB, argB


def knn(x, k, metric="feature"):
    """k nearest neighbours in feature space, self included, best first.

    x: (B, C, N) fp32 channel-first (the layout the reference's encoders use).
    metric: "feature" (src/model.py:9-22, src/PointNet.py:9-26) or "points_normals"
    (src/PointNet.py:29-69, C must be 6).  Returns idx (B, N, k) int64.
    """
    require_cuda(x)
    x = _f32c(x, "x")
    if x.dim() != 3:
        raise ValueError("knn expects (B,C,N), got %s" % (tuple(x.shape),))
    B, C, N = x.shape
    lib = _lib.load()
    dev = x.device
    idx = torch.empty((B, N, k), dtype=torch.int64, device=dev)
    wsz = lib.pn_knn_workspace(B, C, N, k)
    ws = torch.empty(wsz, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        if metric == "feature":
            rc = lib.pn_knn_f32(ptr(x), B, C, N, k, ptr(idx), ptr(ws), wsz, current_stream(dev))
        elif metric == "points_normals":
            if C != 6:
                raise ValueError("points_normals metric needs 6 channels, got %d" % C)
            rc = lib.pn_knn_pn_f32(ptr(x), B, N, k, ptr(idx), ptr(ws), wsz, current_stream(dev))
        else:
            raise ValueError("unknown metric %r" % (metric,))
    check(rc, "pn_knn")
    return idx
