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
    with _lib.on_device(dev):
        rc = lib.pn_chamfer_nn_f32(ptr(a), ptr(b), B, Na, Nb, ptr(minA), ptr(argA), ptr(minB),
                                   ptr(argB), ptr(ws), wsz, current_stream(dev))
    check(rc, "pn_chamfer_nn_f32")
    return minA, argA, minB, argB


def knn(x, k, metric="feature", int32=False):
    """k nearest neighbours in feature space, self included, best first.

    x: (B, C, N) fp32 channel-first (the layout the reference's encoders use).
    metric: "feature" (src/model.py:9-22, src/PointNet.py:9-26) or "points_normals"
    (src/PointNet.py:29-69, C must be 6).  Returns idx (B, N, k) int64 — the dtype of the reference's
    ``topk`` indices — or, with ``int32``, the same values as int32: the form the library's edge-conv
    kernels take (half the index bytes), which the encoders use between their own layers.
    """
    require_cuda(x)
    x = _f32c(x, "x")
    if x.dim() != 3:
        raise ValueError("knn expects (B,C,N), got %s" % (tuple(x.shape),))
    if metric not in ("feature", "points_normals"):
        raise ValueError("unknown metric %r" % (metric,))
    B, C, N = x.shape
    if metric == "points_normals" and C != 6:
        raise ValueError("points_normals metric needs 6 channels, got %d" % C)
    lib = _lib.load()
    dev = x.device
    idx = torch.empty((B, N, k), dtype=torch.int32 if int32 else torch.int64, device=dev)
    wsz = lib.pn_knn_workspace(B, C, N, k)
    ws = torch.empty(wsz, dtype=torch.uint8, device=dev)
    with _lib.on_device(dev):
        if int32:
            rc = lib.pn_knn_graph_i32(ptr(x), B, C, N, k, int(metric == "points_normals"), ptr(idx), ptr(ws), wsz,
                                      current_stream(dev))
        elif metric == "feature":
            rc = lib.pn_knn_f32(ptr(x), B, C, N, k, ptr(idx), ptr(ws), wsz, current_stream(dev))
        else:
            rc = lib.pn_knn_pn_f32(ptr(x), B, N, k, ptr(idx), ptr(ws), wsz, current_stream(dev))
    check(rc, "pn_knn")
    return idx


def knn3_ragged(pts, off, max_n, k, f64=False, want_dist=False):
    """Neighbours of 3-D points by coordinate differences inside the segments of a ragged batch
    (csrc/knn3.hip; src/fitting_utils.py:150-164, 704-710).  pts (total,3) fp32, off (S+1) int32 device
    offsets, max_n >= the largest segment (a host integer).  Returns idx (total,k) int32 LOCAL to the
    segment — nearest first, the point itself first — and, with ``want_dist``, their distances
    (float64 when ``f64``)."""
    require_cuda(pts, off)
    pts = _f32c(pts, "pts")
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise ValueError("knn3_ragged expects (total,3) points, got %s" % (tuple(pts.shape),))
    if off.dtype != torch.int32:
        raise TypeError("off must be int32")
    S = off.numel() - 1
    total = pts.shape[0]
    idx = torch.empty((total, k), dtype=torch.int32, device=pts.device)
    dist = torch.empty((total, k), dtype=torch.float64 if f64 else torch.float32, device=pts.device) if want_dist else None
    if total == 0 or S <= 0:
        return (idx, dist) if want_dist else idx
    with _lib.on_device(pts.device):
        rc = _lib.load().pn_knn3_ragged(ptr(pts), ptr(off.contiguous()), S, int(max_n), int(k), int(bool(f64)), ptr(idx),
                                        ptr(dist), current_stream(pts.device))
    check(rc, "pn_knn3_ragged")
    return (idx, dist) if want_dist else idx


def _i64c(t, name):
    if t.dtype != torch.int64:
        raise TypeError("%s must be int64, got %s" % (name, t.dtype))
    return t.contiguous()


def _graph_c(t, name):
    """A kNN graph: int64 (the API dtype) or int32 (the library's own, kernels.knn(..., int32=True))."""
    if t.dtype not in (torch.int64, torch.int32):
        raise TypeError("%s must be int64 or int32, got %s" % (name, t.dtype))
    return t.contiguous()


def transpose12(x):
    """(B,R,C) -> contiguous (B,C,R) through the LDS-tiled HIP transpose."""
    require_cuda(x)
    x = _f32c(x, "x")
    B, R, C = x.shape
    out = torch.empty((B, C, R), dtype=torch.float32, device=x.device)
    with _lib.on_device(x.device):
        rc = _lib.load().pn_transpose_f32(ptr(x), ptr(out), B, R, C, current_stream(x.device))
    check(rc, "pn_transpose_f32")
    return out


def edge_feature_fwd(xt, idx):
    """xt (B,N,C) point-major, idx (B,N,k) -> feat (B,N,k,2C) = cat(x_j - x_i, x_i)."""
    require_cuda(xt, idx)
    xt = _f32c(xt, "xt")
    idx = _i64c(idx, "idx")
    B, N, C = xt.shape
    k = idx.shape[2]
    feat = torch.empty((B, N, k, 2 * C), dtype=torch.float32, device=xt.device)
    with _lib.on_device(xt.device):
        rc = _lib.load().pn_edge_feature_fwd_f32(ptr(xt), ptr(idx), B, N, k, C, ptr(feat),
                                                 current_stream(xt.device))
    check(rc, "pn_edge_feature_fwd_f32")
    return feat


def edge_feature_bwd(gfeat, idx):
    """gfeat (B,N,k,2C) -> gradient w.r.t. xt (B,N,C)."""
    require_cuda(gfeat, idx)
    gfeat = _f32c(gfeat, "gfeat")
    idx = _i64c(idx, "idx")
    B, N, k, C2 = gfeat.shape
    C = C2 // 2
    gxt = torch.empty((B, N, C), dtype=torch.float32, device=gfeat.device)
    lib = _lib.load()
    wsz = lib.pn_edgeconv_bwd_workspace(B, N, k)
    ws = torch.empty(wsz, dtype=torch.uint8, device=gfeat.device)
    with _lib.on_device(gfeat.device):
        rc = lib.pn_edge_feature_bwd_f32(ptr(gfeat), ptr(idx), B, N, k, C, ptr(gxt), ptr(ws), wsz,
                                         current_stream(gfeat.device))
    check(rc, "pn_edge_feature_bwd_f32")
    return gxt


def edgeconv_reduce_fwd(PQ, idx, gamma, groups, per_sample):
    """PQ (B,N,2*Cout) = [P | Q], idx (B,N,k).  Returns yext, argk (uint8), s1 (all (B,N,Cout))
    and the fp64 group moments stats ((B or 1), groups, 2) of y = P[j] + Q[i] over all edges."""
    require_cuda(PQ, idx, gamma)
    PQ = _f32c(PQ, "PQ")
    idx = _graph_c(idx, "idx")
    gamma = _f32c(gamma, "gamma")
    B, N, C2 = PQ.shape
    Cout = C2 // 2
    k = idx.shape[2]
    dev = PQ.device
    yext = torch.empty((B, N, Cout), dtype=torch.float32, device=dev)
    s1 = torch.empty((B, N, Cout), dtype=torch.float32, device=dev)
    argk = torch.empty((B, N, Cout), dtype=torch.uint8, device=dev)
    stats = torch.empty((B if per_sample else 1, groups, 2), dtype=torch.float64, device=dev)
    lib = _lib.load()
    wsz = lib.pn_edgeconv_reduce_workspace(B, N, Cout, groups)
    ws = torch.empty(wsz, dtype=torch.uint8, device=dev)
    fn = lib.pn_edgeconv_reduce_fwd_i32 if idx.dtype == torch.int32 else lib.pn_edgeconv_reduce_fwd_f32
    with _lib.on_device(dev):
        rc = fn(ptr(PQ), ptr(idx), ptr(gamma), B, N, k, Cout, groups, int(per_sample), ptr(yext), ptr(argk),
                ptr(s1), ptr(stats), ptr(ws), wsz, current_stream(dev))
    check(rc, "pn_edgeconv_reduce_fwd")
    return yext, argk, s1, stats


def moments(stats, count, eps):
    """fp64 (sum, sum of squares) -> fp32 (mean, rstd) per group."""
    require_cuda(stats)
    n = stats.numel() // 2
    mean = torch.empty(stats.shape[:-1], dtype=torch.float32, device=stats.device)
    rstd = torch.empty_like(mean)
    with _lib.on_device(stats.device):
        rc = _lib.load().pn_moments_f32(ptr(stats), n, float(count), float(eps), ptr(mean), ptr(rstd),
                                        current_stream(stats.device))
    check(rc, "pn_moments_f32")
    return mean, rstd


def edgeconv_finalize_fwd(yext, mean, rstd, gamma, beta, groups, per_sample, slope):
    """out (B,Cout,N) = LeakyReLU(gamma * (yext - mean) * rstd + beta), channel-first."""
    require_cuda(yext)
    B, N, Cout = yext.shape
    out = torch.empty((B, Cout, N), dtype=torch.float32, device=yext.device)
    with _lib.on_device(yext.device):
        rc = _lib.load().pn_edgeconv_finalize_fwd_f32(ptr(yext), ptr(_f32c(mean, "mean")),
                                                      ptr(_f32c(rstd, "rstd")), ptr(_f32c(gamma, "gamma")),
                                                      ptr(_f32c(beta, "beta")), B, N, Cout, groups,
                                                      int(per_sample), float(slope), ptr(out),
                                                      current_stream(yext.device))
    check(rc, "pn_edgeconv_finalize_fwd_f32")
    return out


def edgeconv_bwd_prep(gout, yext, mean, rstd, gamma, beta, groups, per_sample, slope):
    """gout (B,Cout,N) -> gz, yhat as (B,N,Cout)."""
    require_cuda(gout, yext)
    gout = _f32c(gout, "gout")
    B, N, Cout = yext.shape
    gz = torch.empty_like(yext)
    yhat = torch.empty_like(yext)
    with _lib.on_device(yext.device):
        rc = _lib.load().pn_edgeconv_bwd_prep_f32(ptr(gout), ptr(yext), ptr(mean), ptr(rstd),
                                                  ptr(_f32c(gamma, "gamma")), ptr(_f32c(beta, "beta")), B, N,
                                                  Cout, groups, int(per_sample), float(slope), ptr(gz),
                                                  ptr(yhat), current_stream(yext.device))
    check(rc, "pn_edgeconv_bwd_prep_f32")
    return gz, yhat


def edgeconv_csr_build(idx):
    """The transposed kNN graph of idx (B,N,k) for edgeconv_bwd(..., csr=...): a workspace tensor.  Depends on idx
    alone — graph.py builds it during the forward pass on a side stream."""
    require_cuda(idx)
    idx = _graph_c(idx, "idx")
    B, N, k = idx.shape
    lib = _lib.load()
    wsz = lib.pn_edgeconv_bwd_workspace(B, N, k)
    ws = torch.empty(wsz, dtype=torch.uint8, device=idx.device)
    with _lib.on_device(idx.device):
        rc = lib.pn_edgeconv_csr_build(ptr(idx), int(idx.dtype == torch.int32), B, N, k, ptr(ws), wsz,
                                       current_stream(idx.device))
    check(rc, "pn_edgeconv_csr_build")
    return ws


def edgeconv_bwd(PQ, idx, t, s1, argk, mean, rstd, c1c2, groups, per_sample, dense, csr=None):
    """Edge-level normalisation gradient -> dPQ (B,N,2*Cout); see csrc/edge.hip.  ``csr``: the workspace of
    edgeconv_csr_build(idx) (the transposed graph, already built), else it is built here."""
    require_cuda(PQ, idx, t)
    B, N, C2 = PQ.shape
    Cout = C2 // 2
    k = idx.shape[2]
    dPQ = torch.empty_like(PQ)
    lib = _lib.load()
    wsz = lib.pn_edgeconv_bwd_workspace(B, N, k)
    idx = _graph_c(idx, "idx")
    if csr is not None:
        if csr.numel() < wsz:
            raise ValueError("edgeconv_bwd: the prebuilt graph does not belong to this idx")
        with _lib.on_device(PQ.device):
            rc = lib.pn_edgeconv_bwd_prebuilt(ptr(PQ), ptr(idx), int(idx.dtype == torch.int32), ptr(_f32c(t, "t")), ptr(s1),
                                              ptr(argk), ptr(mean), ptr(rstd), ptr(_f32c(c1c2, "c1c2")), B, N, k, Cout,
                                              groups, int(per_sample), int(dense), ptr(dPQ), ptr(csr), csr.numel(),
                                              current_stream(PQ.device))
        check(rc, "pn_edgeconv_bwd_prebuilt")
        return dPQ
    ws = torch.empty(wsz, dtype=torch.uint8, device=PQ.device)
    fn = lib.pn_edgeconv_bwd_i32 if idx.dtype == torch.int32 else lib.pn_edgeconv_bwd_f32
    with _lib.on_device(PQ.device):
        rc = fn(ptr(PQ), ptr(idx), ptr(_f32c(t, "t")), ptr(s1), ptr(argk), ptr(mean), ptr(rstd),
                ptr(_f32c(c1c2, "c1c2")), B, N, k, Cout, groups, int(per_sample), int(dense), ptr(dPQ), ptr(ws), wsz,
                current_stream(PQ.device))
    check(rc, "pn_edgeconv_bwd")
    return dPQ


def dot_select(q, c, k, want_value):
    """q (B,Nq,C), c (B,Nc,C) point-major.  Returns (result, flags) where result is either the
    (B,Nq,k) int64 indices of the k largest dot products (best first) or the (B,Nq) k-th largest
    dot product, and flags (B,Nq) int32 marks rows the caller must recompute.  Returns None when
    the shape is outside the kernel's fast path."""
    require_cuda(q, c)
    q = _f32c(q, "q")
    c = _f32c(c, "c")
    B, Nq, C = q.shape
    Nc = c.shape[1]
    lib = _lib.load()
    wsz = lib.pn_dot_select_workspace(B, C, Nq, Nc, int(k), int(bool(want_value)))
    if wsz == 0:
        return None
    dev = q.device
    ws = torch.empty(wsz, dtype=torch.uint8, device=dev)
    flags = torch.empty((B, Nq), dtype=torch.int32, device=dev)
    if want_value:
        out = torch.empty((B, Nq), dtype=torch.float32, device=dev)
        args = (None, ptr(out))
    else:
        out = torch.empty((B, Nq, k), dtype=torch.int64, device=dev)
        args = (ptr(out), None)
    with _lib.on_device(dev):
        rc = lib.pn_dot_select_f32(ptr(q), Nq, ptr(c), Nc, B, C, int(k), args[0], args[1], ptr(flags),
                                   ptr(ws), wsz, current_stream(dev))
    check(rc, "pn_dot_select_f32")
    return out, flags


def dot_kth_x3(q, c, k):
    """K-th largest dot product between the rows of q (B,Nq,C) and c (B,Nc,C) with both distance
    passes in bf16 x 3 arithmetic (fp32-grade, |error| ~1e-7 on unit rows): (values (B,Nq), flags),
    or None outside that path (C <= 128, Nc >= 2048, the fast-path shape rules)."""
    require_cuda(q, c)
    q, c = _f32c(q, "q"), _f32c(c, "c")
    B, Nq, C = q.shape
    Nc = c.shape[1]
    lib = _lib.load()
    wsz = lib.pn_dot_select_workspace(B, C, Nq, Nc, int(k), 1)
    if wsz == 0:
        return None
    dev = q.device
    ws = torch.empty(wsz, dtype=torch.uint8, device=dev)
    flags = torch.empty((B, Nq), dtype=torch.int32, device=dev)
    out = torch.empty((B, Nq), dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        rc = lib.pn_dot_kth_x3_f32(ptr(q), Nq, ptr(c), Nc, B, C, int(k), ptr(out), ptr(flags), ptr(ws), wsz,
                                   current_stream(dev))
    if rc == -4:      # PN_ERR_UNSUPPORTED: shape outside the split passes
        return None
    check(rc, "pn_dot_kth_x3_f32")
    return out, flags


def dot_kth_unit(q, c_image, Nc, k):
    """K-th largest dot product between the unit rows of q (B,Nq,128) and the candidates whose
    fp16 x 2 tile images are ``c_image`` (meanshift_h2_split): (values (B,Nq), flags) with the
    distance passes on the fp16 matrix cores (|error| ~1e-7), or None outside the fast path."""
    require_cuda(q)
    q = _f32c(q, "q")
    B, Nq, C = q.shape
    lib = _lib.load()
    wsz = lib.pn_dot_select_workspace(B, C, Nq, int(Nc), int(k), 1)
    if wsz == 0 or C != 128:
        return None
    dev = q.device
    ws = torch.empty(wsz, dtype=torch.uint8, device=dev)
    flags = torch.empty((B, Nq), dtype=torch.int32, device=dev)
    out = torch.empty((B, Nq), dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        rc = lib.pn_dot_kth_unit_h2_f32(ptr(q), Nq, ptr(c_image), int(Nc), B, C, int(k), ptr(out), ptr(flags),
                                        ptr(ws), wsz, current_stream(dev))
    check(rc, "pn_dot_kth_unit_h2_f32")
    return out, flags


class MeanShiftWorkspace:
    """Scratch buffers of the mean-shift kernels for one (B,N,D) problem, allocated once per
    mean_shift call and reused by every iteration."""

    def __init__(self, B, N, D, device, backward=False, exact_f32=True):
        lib = _lib.load()
        self.B, self.N, self.D = B, N, D
        self.Np = (N + 63) // 64 * 64
        self.S = lib.pn_meanshift_slices(B, N)
        f = dict(dtype=torch.float32, device=device)
        self.opart = torch.empty((B, self.S, N, D), **f)
        self.rpart = torch.empty((B, self.S, N), **f)
        if backward:
            self.gu = torch.empty((B, N, D), **f)
            # row scalars (+ the per-tile maxima of the fp16 path)
            self.cs = torch.empty(3 * B * N + B * (self.Np // 32), **f)
            self.opart_x = torch.empty((B, self.S, N, D), **f)
            if exact_f32:   # the bf16 x 3 path needs neither go nor the channel-first copies
                self.go = torch.empty((B, N, D), **f)
                self.qt = torch.empty((B, D, self.Np), **f)
                self.gut = torch.empty((B, D, self.Np), **f)


def meanshift_pack(x):
    require_cuda(x)
    x = _f32c(x, "x")
    B, N, D = x.shape
    Np = (N + 63) // 64 * 64
    xt = torch.empty((B, D, Np), dtype=torch.float32, device=x.device)
    with _lib.on_device(x.device):
        rc = _lib.load().pn_meanshift_pack_f32(ptr(x), B, N, D, ptr(xt), current_stream(x.device))
    check(rc, "pn_meanshift_pack_f32")
    return xt


def meanshift_x3_split(x):
    """x (B,N,128) -> pre-split bf16 x 3 tile images for the matrix-core path."""
    require_cuda(x)
    x = _f32c(x, "x")
    B, N, D = x.shape
    lib = _lib.load()
    img = torch.empty(lib.pn_meanshift_x3_image_bytes(B, N), dtype=torch.uint8, device=x.device)
    with _lib.on_device(x.device):
        rc = lib.pn_meanshift_x3_split_f32(ptr(x), B, N, D, ptr(img), current_stream(x.device))
    check(rc, "pn_meanshift_x3_split_f32")
    return img


def meanshift_x3_tileinfo(z):
    """z (B,N,128) unit rows -> (centres (B,T,2,128), angular radii (B,T,2), row counts (B,T,2)) of
    the two bounding caps of every 32-row tile, T = align_up(N, 64) / 32 (radius < 0: empty cap): the
    geometry the block-sparse plan is derived from."""
    require_cuda(z)
    z = _f32c(z, "z")
    B, N, D = z.shape
    T = (N + 63) // 64 * 2
    cen = torch.empty((B, T, 2, D), dtype=torch.float32, device=z.device)
    rho = torch.empty((B, T, 2), dtype=torch.float32, device=z.device)
    cnt = torch.empty((B, T, 2), dtype=torch.float32, device=z.device)
    with _lib.on_device(z.device):
        rc = _lib.load().pn_meanshift_x3_tileinfo_f32(ptr(z), B, N, D, ptr(cen), ptr(rho), ptr(cnt),
                                                      current_stream(z.device))
    check(rc, "pn_meanshift_x3_tileinfo_f32")
    return cen, rho, cnt


def kmeans_assign(x, cen):
    """x (B,N,128), cen (B,K,128) -> (B,N) int32: the centre with the largest dot product (ties -> smaller index)."""
    require_cuda(x, cen)
    x, cen = _f32c(x, "x"), _f32c(cen, "cen")
    B, N, D = x.shape
    lab = torch.empty((B, N), dtype=torch.int32, device=x.device)
    with _lib.on_device(x.device):
        rc = _lib.load().pn_kmeans_assign_f32(ptr(x), ptr(cen), B, N, D, cen.shape[1], ptr(lab), current_stream(x.device))
    check(rc, "pn_kmeans_assign_f32")
    return lab


def kmeans_centres(x, lab, old):
    """Normalised sums of the points of every cell (lab (B,N) int32 from kmeans_assign); an empty cell keeps
    ``old`` (B,K,128)."""
    require_cuda(x, lab, old)
    x, old = _f32c(x, "x"), _f32c(old, "old")
    if lab.dtype != torch.int32 or not lab.is_contiguous():
        raise ValueError("kmeans_centres: lab must be a contiguous int32 tensor")
    B, N, D = x.shape
    cen = torch.empty_like(old)
    with _lib.on_device(x.device):
        rc = _lib.load().pn_kmeans_centres_f32(ptr(x), ptr(lab), ptr(old), B, N, D, old.shape[1], ptr(cen),
                                               current_stream(x.device))
    check(rc, "pn_kmeans_centres_f32")
    return cen


CELL_ORDER_MAX_CELLS = 768


def cell_order(rank, home, fine):
    """Stable argsort of rank[home[fine]] * F + fine as ONE counting-sort launch: rank (B,P), home (B,F), fine (B,N),
    all int32 -> (B,N) int64.  F <= CELL_ORDER_MAX_CELLS."""
    require_cuda(rank, home, fine)
    for t, nm in ((rank, "rank"), (home, "home"), (fine, "fine")):
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise ValueError("cell_order: %s must be a contiguous int32 tensor" % nm)
    B, N = fine.shape
    perm = torch.empty((B, N), dtype=torch.int64, device=fine.device)
    with _lib.on_device(fine.device):
        rc = _lib.load().pn_cell_order_i32(ptr(rank), ptr(home), ptr(fine), B, N, rank.shape[1], home.shape[1],
                                           ptr(perm), current_stream(fine.device))
    check(rc, "pn_cell_order_i32")
    return perm


def meanshift_chain_order(sim, as_long=True):
    """sim (B,128,128) similarities of cell centres -> (B,128) int64 (int32 with ``as_long=False``) position of
    every cell in the greedy nearest-neighbour chain that starts at cell 0."""
    require_cuda(sim)
    sim = _f32c(sim, "sim")
    B, P, _ = sim.shape
    rank = torch.empty((B, P), dtype=torch.int32, device=sim.device)
    with _lib.on_device(sim.device):
        rc = _lib.load().pn_meanshift_chain_order_f32(ptr(sim), B, P, ptr(rank), current_stream(sim.device))
    check(rc, "pn_meanshift_chain_order_f32")
    return rank.long() if as_long else rank


def meanshift_x3_plan_bytes(B, N):
    return int(_lib.load().pn_meanshift_x3_plan_bytes(B, N))


def meanshift_x3_plan_buffer(B, N, iterations, device):
    """One buffer for the plans of ``iterations`` launches: plan t lives at t * core bytes, the scratch of the plan
    call (dead once it returns) of plan t overlaps the plans still to be written, ONE scratch region at the end.
    Returns (buffer, [views of plan_bytes each], core bytes)."""
    lib = _lib.load()
    full, core = int(lib.pn_meanshift_x3_plan_bytes(B, N)), int(lib.pn_meanshift_x3_plan_core_bytes(B, N))
    buf = torch.empty(iterations * core + (full - core), dtype=torch.uint8, device=device)
    return buf, [buf[t * core:t * core + full] for t in range(iterations)], core


def meanshift_x3_plan(q_info, x_info, bsq, N, rel_eps=1e-9, out=None):
    """Block-sparse plan of one iteration (which tile pairs can contribute more than ``rel_eps`` of
    the smallest row sum): an opaque byte tensor for meanshift_x3_iter_fwd / _bwd.  The product passes
    mean_shift.PLAN_REL_EPS (1e-6) for the forward-only training path and PLAN_REL_EPS_DENSE_BWD (1e-9)
    where a dense backward reuses the plan; the default here is the tighter one.  ``out``: a contiguous uint8
    tensor of meanshift_x3_plan_bytes(B, N) to write into (a row of the buffer that holds all plans of a call)."""
    cq, rq = q_info[0], q_info[1]
    cx, rx = x_info[0], x_info[1]
    nx = x_info[2] if len(x_info) > 2 else None      # rows of the data caps (None: conservative bounds)
    B = cq.shape[0]
    lib = _lib.load()
    plan = out if out is not None else torch.empty(lib.pn_meanshift_x3_plan_bytes(B, N), dtype=torch.uint8,
                                                   device=cq.device)
    with _lib.on_device(cq.device):
        rc = lib.pn_meanshift_x3_plan_f32(ptr(cq), ptr(rq), ptr(cx), ptr(rx), ptr(nx), ptr(bsq), B, N, float(rel_eps),
                                          ptr(plan), current_stream(cq.device))
    check(rc, "pn_meanshift_x3_plan_f32")
    return plan


def meanshift_x3_nearest(xq, xc, q_info, c_info, perm=None):
    """arg-max_j xq_i . xc_j for unit rows xq, xc (B,N,128) given in one common locality order, with
    their tile caps (meanshift_x3_tileinfo); perm (B,N) int64 position -> original index.  Returns
    (B,N) int64 indexed by / naming ORIGINAL indices: exactly dot_select(xq, xc, 1) of the unpermuted
    tensors (fp32 fma chains, ties to the smaller index), evaluated on the tile pairs that can hold
    a maximum only."""
    require_cuda(xq, xc)
    xq, xc = _f32c(xq, "xq"), _f32c(xc, "xc")
    B, N, D = xq.shape
    lib = _lib.load()
    out = torch.empty((B, N), dtype=torch.int64, device=xq.device)
    wsz = lib.pn_meanshift_x3_plan_bytes(B, N)
    ws = torch.empty(wsz, dtype=torch.uint8, device=xq.device)
    if perm is not None:
        perm = _i64c(perm, "perm")
    with _lib.on_device(xq.device):
        rc = lib.pn_meanshift_x3_nearest_f32(ptr(xq), ptr(xc), ptr(q_info[0]), ptr(q_info[1]), ptr(c_info[0]),
                                             ptr(c_info[1]), ptr(perm), B, N, D, ptr(out), ptr(ws), wsz,
                                             current_stream(xq.device))
    check(rc, "pn_meanshift_x3_nearest_f32")
    return out


def meanshift_x3_plan_stats(plan, B, N):
    """(active fraction of tile pairs, of pass-0 / pass-1 / pass-2 block x tile pairs) — diagnostics."""
    T = (N + 63) // 64 * 2
    nb0, nb1, nb2 = -(-N // 256), -(-N // 128), -(-N // 256)
    oc = (B * T * T + 255) // 256 * 256
    pairs = plan[:B * T * T].float().mean().item()
    counts = plan[oc:oc + B * (nb0 + nb1 + nb2) * 4].view(torch.int32).reshape(B, -1).float()
    return (pairs, counts[:, :nb0].mean().item() / T, counts[:, nb0:nb0 + nb1].mean().item() / T,
            counts[:, nb0 + nb1:].mean().item() / T)


def meanshift_x3_plan_visited(plans, B, N):
    """Device scalar: the share of the N^2 tile pairs the plans keep, averaged over the plans — what
    the planned launches execute relative to dense ones (their duration follows it: measured 0.73 of
    the dense launch at 0.71 of the pairs, 0.83 at 0.85).  No synchronisation (the caller downloads
    it with something it waits for anyway)."""
    T = (N + 63) // 64 * 2
    n = B * T * T
    if isinstance(plans, tuple):
        # all plans of the call in one buffer (meanshift_x3_plan_buffer), ``core`` bytes apart: ONE reduction (the
        # flags are 0 / 1 bytes, the fp32 sum of <= 2^24 of them is exact in any order)
        buf, count, core = plans
        return buf.as_strided((count, n), (core, 1)).sum(dtype=torch.float32) / float(n * count)
    return torch.stack([p[:n].sum(dtype=torch.float32) for p in plans]).mean() / float(n)


def meanshift_x3_iter_fwd(q, x_image, bsq, ws, plan=None, out=None, want_info=False):
    """``out`` = (y (B,N,D), rsum (B,N), unorm (B,N)) contiguous fp32 tensors to write into (slices of
    the buffers that keep all iterates of a call together), or None: allocated here.  ``want_info``: also
    return meanshift_x3_tileinfo(y) — the caps come out of the launch that combines the partial results."""
    B, N, D = q.shape
    if out is not None:
        y, rsum, unorm = out
        for t, shp in ((y, (B, N, D)), (rsum, (B, N)), (unorm, (B, N))):
            if tuple(t.shape) != shp or t.dtype != torch.float32 or not t.is_contiguous() or t.device != q.device:
                raise ValueError("meanshift_x3_iter_fwd: out tensors must be contiguous fp32 of shapes (B,N,D), (B,N), (B,N)")
    else:
        y = torch.empty_like(q)
        rsum = torch.empty((B, N), dtype=torch.float32, device=q.device)
        unorm = torch.empty((B, N), dtype=torch.float32, device=q.device)
    info = None
    if want_info:
        T = (N + 63) // 64 * 2
        info = (torch.empty((B, T, 2, D), dtype=torch.float32, device=q.device),
                torch.empty((B, T, 2), dtype=torch.float32, device=q.device),
                torch.empty((B, T, 2), dtype=torch.float32, device=q.device))
    with _lib.on_device(q.device):
        rc = _lib.load().pn_meanshift_x3_iter_fwd_info_f32(ptr(q), ptr(x_image), ptr(bsq), B, N, D, ptr(ws.opart),
                                                           ptr(ws.rpart), ptr(y), ptr(rsum), ptr(unorm),
                                                           ptr(plan), ptr(info[0]) if info else None,
                                                           ptr(info[1]) if info else None,
                                                           ptr(info[2]) if info else None, current_stream(q.device))
    check(rc, "pn_meanshift_x3_iter_fwd_info_f32")
    return (y, rsum, unorm, info) if want_info else (y, rsum, unorm)


def meanshift_x3_iter_bwd(gy, y, q, x, x_image, rsum, unorm, bsq, ws, gx, plan=None):
    """bf16 x 3 counterpart of meanshift_iter_bwd: returns dL/dq, adds into ``gx``."""
    B, N, D = x.shape
    gy = _f32c(gy, "gy")
    gq = torch.empty_like(x)
    lib = _lib.load()
    if getattr(ws, "x3_imgs", None) is None:
        nbytes = lib.pn_meanshift_x3_image_bytes(B, N)
        ws.x3_imgs = [torch.empty(nbytes, dtype=torch.uint8, device=x.device) for _ in range(2)]
    im = ws.x3_imgs
    with _lib.on_device(x.device):
        rc = lib.pn_meanshift_x3_iter_bwd_plan_f32(ptr(gy), ptr(y), ptr(q), ptr(x), ptr(x_image), ptr(rsum),
                                                   ptr(unorm), ptr(bsq), B, N, D, ptr(ws.gu), ptr(ws.cs),
                                                   ptr(im[0]), ptr(im[1]), ptr(ws.opart), ptr(ws.opart_x), ptr(gq),
                                                   ptr(gx), ptr(plan), current_stream(x.device))
    check(rc, "pn_meanshift_x3_iter_bwd_plan_f32")
    return gq


def gemm_x3_weight_image(w, transposed=False):
    """Pre-split bf16 x 3 image of a weight w (M,K) for gemm_x3 (transposed: of w^T, for the gradient
    w.r.t. the activations).  Returns a uint8 tensor."""
    require_cuda(w)
    w = _f32c(w, "w")
    if w.dim() != 2:
        raise ValueError("gemm_x3_weight_image expects (M,K), got %s" % (tuple(w.shape),))
    M, Kd = w.shape
    lib = _lib.load()
    nbytes = lib.pn_gemm_x3_weight_image_bytes(Kd, M) if transposed else lib.pn_gemm_x3_weight_image_bytes(M, Kd)
    img = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    with _lib.on_device(w.device):
        rc = lib.pn_gemm_x3_weight_image_f32(ptr(w), M, Kd, 1 if transposed else 0, ptr(img), current_stream(w.device))
    check(rc, "pn_gemm_x3_weight_image_f32")
    return img


def gemm_x3(img_a, M, x, bias=None):
    """out (B,M,N) = A x (+ bias): img_a = gemm_x3_weight_image of the (M,K) operand A, x (B,K,N) fp32
    channel-first.  fp32-grade products on the bf16 matrix cores (csrc/gemm_x3.hip)."""
    require_cuda(img_a, x)
    x = _f32c(x, "x")
    if x.dim() != 3:
        raise ValueError("gemm_x3 expects x (B,K,N), got %s" % (tuple(x.shape),))
    B, Kd, N = x.shape
    lib = _lib.load()
    if img_a.numel() != lib.pn_gemm_x3_weight_image_bytes(M, Kd):
        raise ValueError("gemm_x3: the weight image does not belong to a (%d,%d) operand" % (M, Kd))
    if bias is not None:
        bias = _f32c(bias, "bias")
        if bias.numel() != M:
            raise ValueError("gemm_x3: bias must have %d elements" % M)
    out = torch.empty((B, M, N), dtype=torch.float32, device=x.device)
    wsz = lib.pn_gemm_x3_points_image_bytes(B, Kd, N)
    ws = torch.empty(wsz, dtype=torch.uint8, device=x.device)
    with _lib.on_device(x.device):
        rc = lib.pn_gemm_x3_f32(ptr(img_a), ptr(x), ptr(bias), B, M, Kd, N, ptr(out), ptr(ws), wsz,
                                current_stream(x.device))
    check(rc, "pn_gemm_x3_f32")
    return out


def standardize_select(w, kf):
    """The confident points of S segments (src/fitting_utils.py:515-523): w (S,n) -> byte mask (S,n): w > 0.8, or
    the kf largest memberships of a row with fewer than 400 such points (ties to the smaller index)."""
    require_cuda(w)
    w = _f32c(w, "w")
    S, n = w.shape
    sel = torch.empty((S, n), dtype=torch.uint8, device=w.device)
    with _lib.on_device(w.device):
        rc = _lib.load().pn_standardize_select_f32(ptr(w), S, n, int(kf), ptr(sel), current_stream(w.device))
    check(rc, "pn_standardize_select_f32")
    return sel


def standardize_scale(Pr, w, sel, eps):
    """Pr (S,n,3) rotated centred points -> (pts (S,n,3) = Pr / (std + eps), std (S,3) = | max - min | of the
    weighted selected points per axis): src/fitting_utils.py:545-552."""
    Pr, w = _f32c(Pr, "Pr"), _f32c(w, "w")
    S, n, _ = Pr.shape
    pts = torch.empty_like(Pr)
    std = torch.empty((S, 3), dtype=torch.float32, device=Pr.device)
    with _lib.on_device(Pr.device):
        rc = _lib.load().pn_standardize_scale_f32(ptr(Pr), ptr(w), ptr(sel), S, n, float(eps), ptr(pts), ptr(std),
                                                  current_stream(Pr.device))
    check(rc, "pn_standardize_scale_f32")
    return pts, std


def gemm_x3_cat(img_a, M, xs, bias=None):
    """gemm_x3 applied to the concatenation of xs (a list of 1 to 4 tensors (B,C_s,N), every C_s a multiple of 8)
    along the channels — without writing the concatenation (src/model.py:150 builds it with torch.cat)."""
    import ctypes
    require_cuda(img_a, *xs)
    xs = [_f32c(t, "x") for t in xs]
    if not 1 <= len(xs) <= 4 or any(t.dim() != 3 or t.shape[0] != xs[0].shape[0] or t.shape[2] != xs[0].shape[2]
                                    or t.shape[1] % 8 for t in xs):
        raise ValueError("gemm_x3_cat expects 1 to 4 tensors (B,C_s,N) with C_s a multiple of 8")
    B, _, N = xs[0].shape
    Kd = sum(t.shape[1] for t in xs)
    lib = _lib.load()
    if img_a.numel() != lib.pn_gemm_x3_weight_image_bytes(M, Kd):
        raise ValueError("gemm_x3_cat: the weight image does not belong to a (%d,%d) operand" % (M, Kd))
    if bias is not None:
        bias = _f32c(bias, "bias")
    out = torch.empty((B, M, N), dtype=torch.float32, device=xs[0].device)
    wsz = lib.pn_gemm_x3_points_image_bytes(B, Kd, N)
    ws = torch.empty(wsz, dtype=torch.uint8, device=out.device)
    ptrs = (ctypes.c_void_p * len(xs))(*[t.data_ptr() for t in xs])
    chans = (ctypes.c_int * len(xs))(*[t.shape[1] for t in xs])
    with _lib.on_device(out.device):
        rc = lib.pn_gemm_x3_cat_f32(ptr(img_a), ptrs, chans, len(xs), ptr(bias), B, M, N, ptr(out), ptr(ws), wsz,
                                    current_stream(out.device))
    check(rc, "pn_gemm_x3_cat_f32")
    return out


def gemm_x3_wgrad(gy, x, want_bias=False):
    """Weight gradient of y[b] = W x[b] (+ bias): gw (M,K) = sum_b gy[b] x[b]^T over the points, bf16 x 3 on the
    matrix cores with a fixed-order split over the points (csrc/gemm_x3.hip).  gy (B,M,N), x (B,K,N) fp32
    channel-first.  Returns gw, or (gw, gb (M)) with ``want_bias``."""
    require_cuda(gy, x)
    gy, x = _f32c(gy, "gy"), _f32c(x, "x")
    if gy.dim() != 3 or x.dim() != 3 or gy.shape[0] != x.shape[0] or gy.shape[2] != x.shape[2]:
        raise ValueError("gemm_x3_wgrad expects gy (B,M,N) and x (B,K,N), got %s and %s"
                         % (tuple(gy.shape), tuple(x.shape)))
    B, M, N = gy.shape
    Kd = x.shape[1]
    lib = _lib.load()
    gw = torch.empty((M, Kd), dtype=torch.float32, device=x.device)
    gb = torch.empty((M,), dtype=torch.float32, device=x.device) if want_bias else None
    wsz = lib.pn_gemm_x3_wgrad_workspace(B, M, Kd, N)
    ws = torch.empty(wsz, dtype=torch.uint8, device=x.device)
    with _lib.on_device(x.device):
        rc = lib.pn_gemm_x3_wgrad_f32(ptr(gy), ptr(x), B, M, Kd, N, ptr(gw), ptr(gb), ptr(ws), wsz,
                                      current_stream(x.device))
    check(rc, "pn_gemm_x3_wgrad_f32")
    return (gw, gb) if want_bias else gw


def meanshift_rows_bwd(gy, y, q, rsum, unorm, x, bsq, gx, ws=None):
    """One step of the mean-shift backward restricted to R <= 64 rows per batch item (csrc/meanshift_rows.hip;
    src/mean_shift.py:45-79 maps every row on its own): gy, y, q (B,R,128), rsum, unorm (B,R), x (B,N,128),
    bsq (B).  Returns gq (B,R,128); ADDS the step's gradient w.r.t. the data into gx (B,N,128)."""
    require_cuda(gy, y, q, rsum, unorm, x, bsq, gx)
    gy, y, q, x = _f32c(gy, "gy"), _f32c(y, "y"), _f32c(q, "q"), _f32c(x, "x")
    rsum, unorm, bsq = _f32c(rsum, "rsum"), _f32c(unorm, "unorm"), _f32c(bsq, "bsq")
    B, N, D = x.shape
    R = q.shape[1]
    if tuple(gy.shape) != (B, R, D) or tuple(y.shape) != (B, R, D) or tuple(q.shape) != (B, R, D):
        raise ValueError("meanshift_rows_bwd: gy, y, q must be (B,R,D)")
    if tuple(rsum.shape) != (B, R) or tuple(unorm.shape) != (B, R) or tuple(gx.shape) != (B, N, D):
        raise ValueError("meanshift_rows_bwd: rsum, unorm (B,R), gx (B,N,D)")
    if gx.dtype != torch.float32 or not gx.is_contiguous():
        raise TypeError("meanshift_rows_bwd: gx must be contiguous fp32")
    lib = _lib.load()
    wsz = lib.pn_meanshift_rows_bwd_workspace(B, N)
    if ws is None or ws.numel() < wsz:
        ws = torch.empty(wsz, dtype=torch.uint8, device=x.device)
    gq = torch.empty_like(q)
    with _lib.on_device(x.device):
        rc = lib.pn_meanshift_rows_bwd_f32(ptr(gy), ptr(y), ptr(q), ptr(rsum), ptr(unorm), ptr(x), ptr(bsq), B, N, D, R,
                                           ptr(gq), ptr(gx), ptr(ws), ws.numel(), current_stream(x.device))
    check(rc, "pn_meanshift_rows_bwd_f32")
    return gq


def meanshift_rows_workspace(B, N, device):
    return torch.empty(_lib.load().pn_meanshift_rows_bwd_workspace(B, N), dtype=torch.uint8, device=device)


def meanshift_rows_scatter_add(gx, rows, g):
    """gx[b, rows[b,r], :] += g[b, r, :], r ascending, in place (rows (B,R) int64, repeats allowed)."""
    require_cuda(gx, rows, g)
    g = _f32c(g, "g")
    rows = _i64c(rows, "rows")
    B, N, D = gx.shape
    R = rows.shape[1]
    if tuple(g.shape) != (B, R, D) or gx.dtype != torch.float32 or not gx.is_contiguous():
        raise ValueError("meanshift_rows_scatter_add: g (B,R,D), gx contiguous fp32 (B,N,D)")
    with _lib.on_device(gx.device):
        rc = _lib.load().pn_meanshift_rows_scatter_add_f32(ptr(g), ptr(rows), B, N, D, R, ptr(gx),
                                                           current_stream(gx.device))
    check(rc, "pn_meanshift_rows_scatter_add_f32")
    return gx


def meanshift_h2_split(x):
    """x (B,N,128), unit rows -> pre-split fp16 x 2 tile images for the matrix-core path."""
    require_cuda(x)
    x = _f32c(x, "x")
    B, N, D = x.shape
    lib = _lib.load()
    img = torch.empty(lib.pn_meanshift_h2_image_bytes(B, N), dtype=torch.uint8, device=x.device)
    with _lib.on_device(x.device):
        rc = lib.pn_meanshift_h2_split_f32(ptr(x), B, N, D, ptr(img), current_stream(x.device))
    check(rc, "pn_meanshift_h2_split_f32")
    return img


def meanshift_h2_iter_fwd(q, x_image, bsq, ws):
    B, N, D = q.shape
    y = torch.empty_like(q)
    rsum = torch.empty((B, N), dtype=torch.float32, device=q.device)
    unorm = torch.empty((B, N), dtype=torch.float32, device=q.device)
    with _lib.on_device(q.device):
        rc = _lib.load().pn_meanshift_h2_iter_fwd_f32(ptr(q), ptr(x_image), ptr(bsq), B, N, D, ptr(ws.opart),
                                                      ptr(ws.rpart), ptr(y), ptr(rsum), ptr(unorm),
                                                      current_stream(q.device))
    check(rc, "pn_meanshift_h2_iter_fwd_f32")
    return y, rsum, unorm


def meanshift_h2_iter_bwd(gy, y, q, x, x_image, rsum, unorm, bsq, ws, gx):
    """fp16 x 2 counterpart of meanshift_iter_bwd: returns dL/dq, adds into ``gx``."""
    B, N, D = x.shape
    gy = _f32c(gy, "gy")
    gq = torch.empty_like(x)
    lib = _lib.load()
    if getattr(ws, "h2_imgs", None) is None:
        nbytes = lib.pn_meanshift_h2_image_bytes(B, N)
        ws.h2_imgs = [torch.empty(nbytes, dtype=torch.uint8, device=x.device) for _ in range(2)]
    im = ws.h2_imgs
    with _lib.on_device(x.device):
        rc = lib.pn_meanshift_h2_iter_bwd_f32(ptr(gy), ptr(y), ptr(q), ptr(x), ptr(x_image), ptr(rsum),
                                              ptr(unorm), ptr(bsq), B, N, D, ptr(ws.gu), ptr(ws.cs), ptr(im[0]),
                                              ptr(im[1]), ptr(ws.opart), ptr(ws.opart_x), ptr(gq), ptr(gx),
                                              current_stream(x.device))
    check(rc, "pn_meanshift_h2_iter_bwd_f32")
    return gq


def meanshift_iter_fwd(q, x, xt, bsq, ws):
    B, N, D = x.shape
    y = torch.empty_like(x)
    rsum = torch.empty((B, N), dtype=torch.float32, device=x.device)
    unorm = torch.empty((B, N), dtype=torch.float32, device=x.device)
    with _lib.on_device(x.device):
        rc = _lib.load().pn_meanshift_iter_fwd_f32(ptr(q), ptr(x), ptr(xt), ptr(bsq), B, N, D, ptr(ws.opart),
                                                   ptr(ws.rpart), ptr(y), ptr(rsum), ptr(unorm),
                                                   current_stream(x.device))
    check(rc, "pn_meanshift_iter_fwd_f32")
    return y, rsum, unorm


def meanshift_iter_bwd(gy, y, q, x, xt, rsum, unorm, bsq, ws, gx):
    """Returns dL/dq (B,N,D) and adds this iteration's contribution to dL/dx into ``gx``."""
    B, N, D = x.shape
    gy = _f32c(gy, "gy")
    gq = torch.empty_like(x)
    with _lib.on_device(x.device):
        rc = _lib.load().pn_meanshift_iter_bwd_f32(ptr(gy), ptr(y), ptr(q), ptr(x), ptr(xt), ptr(rsum),
                                                   ptr(unorm), ptr(bsq), B, N, D, ptr(ws.gu), ptr(ws.go),
                                                   ptr(ws.cs), ptr(ws.qt), ptr(ws.gut), ptr(ws.opart),
                                                   ptr(ws.opart_x), ptr(gq), ptr(gx),
                                                   current_stream(x.device))
    check(rc, "pn_meanshift_iter_bwd_f32")
    return gq


def sym3_eig(G):
    """G (M,3,3) float64 symmetric -> (evals (M,3) descending, evecs (M,3,3), columns)."""
    require_cuda(G)
    if G.dtype != torch.float64:
        raise TypeError("sym3_eig expects float64")
    G = G.contiguous()
    M = G.shape[0]
    evals = torch.empty((M, 3), dtype=torch.float64, device=G.device)
    evecs = torch.empty((M, 3, 3), dtype=torch.float64, device=G.device)
    with _lib.on_device(G.device):
        rc = _lib.load().pn_sym3_eig_f64(ptr(G), M, ptr(evals), ptr(evecs), current_stream(G.device))
    check(rc, "pn_sym3_eig_f64")
    return evals, evecs


def _i32c(t, name):
    if t.dtype != torch.int32:
        raise TypeError("%s must be int32, got %s" % (name, t.dtype))
    return t.contiguous()


def chamfer_nn_ragged(a, off_a, max_a, b, off_b, max_b, side_a=True, side_b=True):
    """Ragged batch of nearest-neighbour searches: item i is a[off_a[i]:off_a[i+1]] against
    b[off_b[i]:off_b[i+1]] (a (TA,3), b (TB,3) fp32; offsets (B+1,) int32 on the GPU; max_* the
    largest item sizes, known to the host).  Returns (minA (TA,), argA (TA,), minB (TB,), argB (TB,))
    with indices local to the item; sides not requested are None."""
    require_cuda(a, b, off_a, off_b)
    a = _f32c(a, "a")
    b = _f32c(b, "b")
    off_a, off_b = _i32c(off_a, "off_a"), _i32c(off_b, "off_b")
    TA, TB = a.shape[0], b.shape[0]
    B = off_a.shape[0] - 1
    if B < 1 or off_b.shape[0] != B + 1 or TA == 0 or TB == 0:
        raise ValueError("chamfer_nn_ragged: empty batch")
    lib = _lib.load()
    dev = a.device
    minA = argA = minB = argB = None
    if side_a:
        minA = torch.empty(TA, dtype=torch.float32, device=dev)
        argA = torch.empty(TA, dtype=torch.int64, device=dev)
    if side_b:
        minB = torch.empty(TB, dtype=torch.float32, device=dev)
        argB = torch.empty(TB, dtype=torch.int64, device=dev)
    wsz = lib.pn_chamfer_nn_ragged_workspace(TA, TB)
    ws = torch.empty(wsz, dtype=torch.uint8, device=dev)
    with _lib.on_device(dev):
        rc = lib.pn_chamfer_nn_ragged_f32(ptr(a), ptr(off_a), TA, int(max_a), ptr(b), ptr(off_b), TB, int(max_b),
                                          B, ptr(minA), ptr(argA), ptr(minB), ptr(argB), ptr(ws), wsz,
                                          current_stream(dev))
    check(rc, "pn_chamfer_nn_ragged_f32")
    return minA, argA, minB, argB


def chamfer_ragged_reduce(minA, off_a, minB, off_b):
    """(mean minA + mean minB) / 2 per item of the ragged batch, fixed summation order -> (S,)."""
    require_cuda(minA, minB)
    S = off_a.shape[0] - 1
    out = torch.empty(S, dtype=torch.float32, device=minA.device)
    with _lib.on_device(minA.device):
        rc = _lib.load().pn_chamfer_ragged_reduce_f32(ptr(_f32c(minA, "minA")), ptr(_i32c(off_a, "off_a")),
                                                      ptr(_f32c(minB, "minB")), ptr(_i32c(off_b, "off_b")), S,
                                                      ptr(out), current_stream(minA.device))
    check(rc, "pn_chamfer_ragged_reduce_f32")
    return out


def chamfer_ragged_bwd(pred, off_a, max_a, gt, off_b, argA, argB, g):
    """Gradient of chamfer_ragged_reduce with respect to the predictions (TA,3); g (S,)."""
    require_cuda(pred, gt, g)
    pred, gt, g = _f32c(pred, "pred"), _f32c(gt, "gt"), _f32c(g, "g")
    S = off_a.shape[0] - 1
    gpred = torch.empty_like(pred)
    with _lib.on_device(pred.device):
        rc = _lib.load().pn_chamfer_ragged_bwd_f32(ptr(pred), ptr(_i32c(off_a, "off_a")), int(max_a), ptr(gt),
                                                   ptr(_i32c(off_b, "off_b")), ptr(_i64c(argA, "argA")),
                                                   ptr(_i64c(argB, "argB")), ptr(g), S, ptr(gpred),
                                                   current_stream(pred.device))
    check(rc, "pn_chamfer_ragged_bwd_f32")
    return gpred


def gather_rows3_bwd(g, idx, N):
    """g (B,M,3), idx (B,M) int64 -> (B,N,3): rows of g added into the rows idx names, ascending m."""
    require_cuda(g, idx)
    g, idx = _f32c(g, "g"), _i64c(idx, "idx")
    B, M, _ = g.shape
    out = torch.empty((B, N, 3), dtype=torch.float32, device=g.device)
    with _lib.on_device(g.device):
        rc = _lib.load().pn_gather_rows3_bwd_f32(ptr(g), ptr(idx), B, M, N, ptr(out), current_stream(g.device))
    check(rc, "pn_gather_rows3_bwd_f32")
    return out


# ---- batched primitive fits (csrc/fitbatch.hip) ---------------------------------------------
PRIM_PLANE, PRIM_SPHERE, PRIM_CYLINDER, PRIM_CONE = 0, 1, 2, 3
FIT_NPAR = 16


def weighted_moments(P, Nrm, W, seg_shape, seg_row, stride, eps):
    """P, Nrm (B,N,3), W (B,Cp,N) fp32; seg_* (S,) int32 -> partial moment sums (S, chunks, 64) fp64."""
    require_cuda(P, Nrm, W, seg_shape, seg_row)
    P, Nrm, W = _f32c(P, "P"), _f32c(Nrm, "Nrm"), _f32c(W, "W")
    B, N, _ = P.shape
    Cp = W.shape[1]
    S = seg_shape.shape[0]
    lib = _lib.load()
    partial = torch.empty((S, lib.pn_weighted_moments_chunks(), lib.pn_weighted_moments_count()),
                          dtype=torch.float64, device=P.device)
    with _lib.on_device(P.device):
        rc = lib.pn_weighted_moments_f64(ptr(P), ptr(Nrm), ptr(W), B, N, Cp, int(stride), float(eps),
                                         ptr(_i32c(seg_shape, "seg_shape")), ptr(_i32c(seg_row, "seg_row")), S,
                                         ptr(partial), current_stream(P.device))
    check(rc, "pn_weighted_moments_f64")
    return partial


def primitive_fit(partial, seg_type, seg_rows):
    """partial (S,chunks,64) fp64 -> params (S,16) fp64, jac (S,16,64) fp64, status (S,) int32."""
    require_cuda(partial, seg_type, seg_rows)
    S = partial.shape[0]
    dev = partial.device
    params = torch.empty((S, FIT_NPAR), dtype=torch.float64, device=dev)
    jac = torch.empty((S, FIT_NPAR, partial.shape[2]), dtype=torch.float64, device=dev)
    status = torch.empty(S, dtype=torch.int32, device=dev)
    with _lib.on_device(dev):
        rc = _lib.load().pn_primitive_fit_f64(ptr(partial.contiguous()), ptr(_i32c(seg_type, "seg_type")),
                                              ptr(_i32c(seg_rows, "seg_rows")), S, ptr(params), ptr(jac),
                                              ptr(status), current_stream(dev))
    check(rc, "pn_primitive_fit_f64")
    return params, jac, status


def cone_angle(P, W, seg_shape, seg_row, seg_type, status, params, jac, stride, eps):
    """Second pass of the cone fit; updates params[:,6] and jac[:,6,:] in place, returns cone_direct (S,)."""
    require_cuda(P, W, params, jac)
    P, W = _f32c(P, "P"), _f32c(W, "W")
    B, N, _ = P.shape
    S = seg_shape.shape[0]
    cone_direct = torch.empty(S, dtype=torch.float64, device=P.device)
    with _lib.on_device(P.device):
        rc = _lib.load().pn_cone_angle_f64(ptr(P), ptr(W), B, N, W.shape[1], int(stride), float(eps),
                                           ptr(seg_shape), ptr(seg_row), ptr(seg_type), ptr(status), S,
                                           ptr(params), ptr(jac), ptr(cone_direct), current_stream(P.device))
    check(rc, "pn_cone_angle_f64")
    return cone_direct


def primitive_residual(P, seg_shape, seg_type, gt_off, gt_idx, params, status, sqrt_flag=False):
    """Mean residual of the ground-truth points of every segment: dist (S,) fp32, dparam (S,16) fp64."""
    require_cuda(P, gt_off, gt_idx, params)
    B, N, _ = P.shape
    S = seg_shape.shape[0]
    dist = torch.empty(S, dtype=torch.float32, device=P.device)
    dparam = torch.zeros((S, FIT_NPAR), dtype=torch.float64, device=P.device)
    with _lib.on_device(P.device):
        rc = _lib.load().pn_primitive_residual_f32(ptr(_f32c(P, "P")), B, N, ptr(seg_shape), ptr(seg_type),
                                                   ptr(_i32c(gt_off, "gt_off")), ptr(_i32c(gt_idx, "gt_idx")), S,
                                                   ptr(params), int(bool(sqrt_flag)), ptr(dist), ptr(dparam),
                                                   ptr(status), current_stream(P.device))
    check(rc, "pn_primitive_residual_f32")
    return dist, dparam


def weighted_moments_bwd(P, Nrm, W, seg_shape, seg_row, seg_type, g_dist, dparam, jac, params, cone_direct,
                         stride, eps):
    """d loss / d W (B,Cp,N) fp32 of the batched fits given g_dist (S,) = d loss / d dist."""
    require_cuda(P, Nrm, W, g_dist)
    P, Nrm, W = _f32c(P, "P"), _f32c(Nrm, "Nrm"), _f32c(W, "W")
    B, N, _ = P.shape
    S = seg_shape.shape[0]
    gW = torch.zeros_like(W)
    with _lib.on_device(P.device):
        rc = _lib.load().pn_weighted_moments_bwd_f32(ptr(P), ptr(Nrm), ptr(W), B, N, W.shape[1], int(stride),
                                                     float(eps), ptr(seg_shape), ptr(seg_row), ptr(seg_type), S,
                                                     ptr(_f32c(g_dist, "g_dist")), ptr(dparam), ptr(jac),
                                                     ptr(params), ptr(cone_direct), ptr(gW),
                                                     current_stream(P.device))
    check(rc, "pn_weighted_moments_bwd_f32")
    return gW


def bspline_eval(nu, nv, ctrl, affine=None, wrap=False):
    """ctrl (S,cu,cv,3), nu (gu,cu), nv (gv,cv) fp32 -> (S,(gu+wrap)*gv,3); affine (S,3,4) optional."""
    require_cuda(nu, nv, ctrl, affine)
    nu, nv, ctrl = _f32c(nu, "nu"), _f32c(nv, "nv"), _f32c(ctrl, "ctrl")
    S, cu, cv, _ = ctrl.shape
    gu, gv = nu.shape[0], nv.shape[0]
    if affine is not None:
        affine = _f32c(affine, "affine")
    out = torch.empty((S, (gu + int(wrap)) * gv, 3), dtype=torch.float32, device=ctrl.device)
    with _lib.on_device(ctrl.device):
        rc = _lib.load().pn_bspline_eval_f32(ptr(nu), ptr(nv), ptr(ctrl), ptr(affine), S, gu, gv, cu, cv, int(wrap),
                                             ptr(out), current_stream(ctrl.device))
    check(rc, "pn_bspline_eval_f32")
    return out


def bspline_eval_bwd(nu, nv, gout, affine, cu, cv, wrap=False):
    require_cuda(nu, nv, gout, affine)
    nu, nv, gout = _f32c(nu, "nu"), _f32c(nv, "nv"), _f32c(gout, "gout")
    S = gout.shape[0]
    gu, gv = nu.shape[0], nv.shape[0]
    if affine is not None:
        affine = _f32c(affine, "affine")
    gctrl = torch.empty((S, cu, cv, 3), dtype=torch.float32, device=gout.device)
    with _lib.on_device(gout.device):
        rc = _lib.load().pn_bspline_eval_bwd_f32(ptr(nu), ptr(nv), ptr(gout), ptr(affine), S, gu, gv, cu, cv,
                                                 int(wrap), ptr(gctrl), current_stream(gout.device))
    check(rc, "pn_bspline_eval_bwd_f32")
    return gctrl


# ---- round-3 fusions (csrc/fused.hip) ----------------------------------------------------------
def edgeconv_bwd_stats(gout, yext, mean, rstd, gamma, beta, groups, per_sample, dense, slope, k):
    """Step A of the fused edge-conv backward with its reductions: returns (t (B,N,Cout), dgamma,
    dbeta (Cout), c1c2 ((B or 1), groups, 2))."""
    require_cuda(gout, yext)
    gout = _f32c(gout, "gout")
    B, N, Cout = yext.shape
    dev = yext.device
    t = torch.empty_like(yext)
    dgamma = torch.empty(Cout, dtype=torch.float32, device=dev)
    dbeta = torch.empty(Cout, dtype=torch.float32, device=dev)
    c1c2 = torch.empty((B if per_sample else 1, groups, 2), dtype=torch.float32, device=dev)
    lib = _lib.load()
    wsz = lib.pn_edgeconv_bwd_stats_workspace(B, N, Cout)
    ws = torch.empty(wsz, dtype=torch.uint8, device=dev)
    with _lib.on_device(dev):
        rc = lib.pn_edgeconv_bwd_stats_f32(ptr(gout), ptr(yext), ptr(mean), ptr(rstd), ptr(_f32c(gamma, "gamma")),
                                           ptr(_f32c(beta, "beta")), B, N, int(k), Cout, int(groups),
                                           int(per_sample), int(dense), float(slope), ptr(t), ptr(dgamma),
                                           ptr(dbeta), ptr(c1c2), ptr(ws), wsz, current_stream(dev))
    check(rc, "pn_edgeconv_bwd_stats_f32")
    return t, dgamma, dbeta, c1c2


def triplet_fwd(E, ia, ib, w, margin):
    """E (rows,128) fp32; ia, ib (P,num) int64 row indices; w (P,) fp32 -> (loss (1,), item_scale (P,))."""
    require_cuda(E, ia, ib, w)
    E = _f32c(E, "E")
    ia, ib = _i64c(ia, "ia"), _i64c(ib, "ib")
    P, num = ia.shape
    dev = E.device
    item_loss = torch.empty(P, dtype=torch.float32, device=dev)
    item_scale = torch.empty(P, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        rc = _lib.load().pn_triplet_fwd_f32(ptr(E), E.shape[0], E.shape[1], ptr(ia), ptr(ib), ptr(_f32c(w, "w")), P,
                                            num, float(margin), ptr(item_loss), ptr(item_scale), ptr(loss),
                                            current_stream(dev))
    check(rc, "pn_triplet_fwd_f32")
    return loss, item_scale


def triplet_bwd(E, ia, ib, item_scale, gout, margin):
    """d loss / d E (rows,128) scaled by gout (1,)."""
    require_cuda(E, gout)
    P, num = ia.shape
    gE = torch.zeros_like(E)
    lib = _lib.load()
    wsz = lib.pn_triplet_bwd_workspace(P, num, E.shape[1])
    ws = torch.empty(wsz, dtype=torch.uint8, device=E.device)
    with _lib.on_device(E.device):
        rc = lib.pn_triplet_bwd_f32(ptr(E), E.shape[0], E.shape[1], ptr(ia), ptr(ib), ptr(item_scale),
                                    ptr(_f32c(gout.reshape(1), "gout")), P, num, float(margin), ptr(gE),
                                    ptr(ws), wsz, current_stream(E.device))
    check(rc, "pn_triplet_bwd_f32")
    return gE


def membership_fwd(cen, emb, bw, ncl, eps, want_labels=False):
    """cen (B,CP,128), emb (B,N,128), bw (B,), ncl (B,) int64 -> Wraw, prob, Wn (B,CP,N), rowstat (B,CP,4),
    labels (B,N) int64 or None."""
    require_cuda(cen, emb, bw, ncl)
    cen, emb, bw = _f32c(cen, "cen"), _f32c(emb, "emb"), _f32c(bw, "bw")
    ncl = _i64c(ncl, "ncl")
    B, CP, D = cen.shape
    N = emb.shape[1]
    dev = emb.device
    Wraw = torch.empty((B, CP, N), dtype=torch.float32, device=dev)
    prob = torch.empty_like(Wraw)
    Wn = torch.empty_like(Wraw)
    rowstat = torch.empty((B, CP, 4), dtype=torch.float32, device=dev)
    labels = torch.empty((B, N), dtype=torch.int64, device=dev) if want_labels else None
    with _lib.on_device(dev):
        rc = _lib.load().pn_membership_fwd_f32(ptr(cen), ptr(emb), ptr(bw), ptr(ncl), B, CP, N, D, float(eps),
                                               ptr(Wraw), ptr(prob), ptr(Wn), ptr(rowstat), ptr(labels),
                                               current_stream(dev))
    check(rc, "pn_membership_fwd_f32")
    return Wraw, prob, Wn, rowstat, labels


def membership_bwd(gWn, Wraw, prob, rowstat, bw, ncl):
    require_cuda(gWn, Wraw)
    gWn = _f32c(gWn, "gWn")
    B, CP, N = Wraw.shape
    rowgrad = torch.empty((B, CP, 2), dtype=torch.float32, device=Wraw.device)
    gWraw = torch.empty_like(Wraw)
    with _lib.on_device(Wraw.device):
        rc = _lib.load().pn_membership_bwd_f32(ptr(gWn), ptr(Wraw), ptr(prob), ptr(rowstat), ptr(bw), ptr(ncl), B, CP,
                                               N, ptr(rowgrad), ptr(gWraw), current_stream(Wraw.device))
    check(rc, "pn_membership_bwd_f32")
    return gWraw


def affine_act_fwd(x, scale, shift, act, slope=0.0):
    """act(x * scale[c] + shift[c]) on (B,C,N); act 0 none, 1 ReLU, 2 LeakyReLU(slope)."""
    require_cuda(x, scale, shift)
    x = _f32c(x, "x")
    B, C, N = x.shape
    y = torch.empty_like(x)
    with _lib.on_device(x.device):
        rc = _lib.load().pn_affine_act_fwd_f32(ptr(x), ptr(_f32c(scale, "scale")), ptr(_f32c(shift, "shift")), B, C, N,
                                               int(act), float(slope), ptr(y), current_stream(x.device))
    check(rc, "pn_affine_act_fwd_f32")
    return y


def affine_act_bwd(gy, y, scale, act, slope=0.0):
    gy = _f32c(gy, "gy")
    B, C, N = y.shape
    gx = torch.empty_like(y)
    with _lib.on_device(y.device):
        rc = _lib.load().pn_affine_act_bwd_f32(ptr(gy), ptr(y), ptr(_f32c(scale, "scale")), B, C, N, int(act),
                                               float(slope), ptr(gx), current_stream(y.device))
    check(rc, "pn_affine_act_bwd_f32")
    return gx


def nms_occupied(membership, U):
    """membership (B,N) int64 -> (counts (B,N) int32, uq (B,U) int64 ascending occupied centres, nocc (B,) int64)."""
    require_cuda(membership)
    membership = _i64c(membership, "membership")
    B, N = membership.shape
    dev = membership.device
    counts = torch.empty((B, N), dtype=torch.int32, device=dev)
    uq = torch.empty((B, U), dtype=torch.int64, device=dev)
    nocc = torch.empty(B, dtype=torch.int64, device=dev)
    with _lib.on_device(dev):
        rc = _lib.load().pn_nms_occupied_f32(ptr(membership), B, N, int(U), ptr(counts), ptr(uq), ptr(nocc),
                                             current_stream(dev))
    check(rc, "pn_nms_occupied_f32")
    return counts, uq, nocc


def nms_vote(G, uq, nocc, counts, bw, cmax):
    """G (B,U,U) Gram matrix of the occupied centres -> (cid (B,cmax) int64 ascending voted centres, ncl (B,))."""
    require_cuda(G, uq, nocc, counts, bw)
    G = _f32c(G, "G")
    B, U, _ = G.shape
    N = counts.shape[1]
    dev = G.device
    hits = torch.empty((B, N), dtype=torch.int32, device=dev)
    cid = torch.empty((B, cmax), dtype=torch.int64, device=dev)
    ncl = torch.empty(B, dtype=torch.int64, device=dev)
    with _lib.on_device(dev):
        rc = _lib.load().pn_nms_vote_f32(ptr(G), ptr(uq), ptr(nocc), ptr(counts), ptr(_f32c(bw, "bw")), B, N, U,
                                         int(cmax), ptr(hits), ptr(cid), ptr(ncl), current_stream(dev))
    check(rc, "pn_nms_vote_f32")
    return cid, ncl


def weighted_max_fwd(x, scale, shift, w, act, slope=0.0):
    """max over n of act(x * scale[c] + shift[c]) * w[s, n]: x (S,C,N), w (S,N) -> (out (S,C), idx int32, val)."""
    require_cuda(x, scale, shift, w)
    x, w = _f32c(x, "x"), _f32c(w, "w")
    S, C, N = x.shape
    dev = x.device
    out = torch.empty((S, C), dtype=torch.float32, device=dev)
    idx = torch.empty((S, C), dtype=torch.int32, device=dev)
    val = torch.empty((S, C), dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        rc = _lib.load().pn_weighted_max_fwd_f32(ptr(x), ptr(_f32c(scale, "scale")), ptr(_f32c(shift, "shift")), ptr(w),
                                                 S, C, N, int(act), float(slope), ptr(out), ptr(idx), ptr(val),
                                                 current_stream(dev))
    check(rc, "pn_weighted_max_fwd_f32")
    return out, idx, val


def weighted_max_bwd(g, idx, val, N):
    g = _f32c(g, "g")
    S, C = g.shape
    gw = torch.empty((S, N), dtype=torch.float32, device=g.device)
    with _lib.on_device(g.device):
        rc = _lib.load().pn_weighted_max_bwd_f32(ptr(g), ptr(idx), ptr(val), S, C, int(N), ptr(gw),
                                                 current_stream(g.device))
    check(rc, "pn_weighted_max_bwd_f32")
    return gw
