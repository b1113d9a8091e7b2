"""The training-mode fitting stage of a whole batch of shapes in a few dozen launches.

Same arithmetic and decisions as the per-segment functions of ``fitting.py`` — which follow
src/residual_utils.py:86-208 (Evaluation.fitting_loss / residual_train_mode / separate_losses),
src/primitive_forward.py:925-1047 (fit_one_shape_torch), :708-843 (Fit.fit_*_torch), :34-102 /
:347-415 (SplineNet forward wrappers), src/fitting_utils.py:306-325 (weights_normalize), :493-553
(standardize_point_torch), src/mean_shift.py:19-179 and src/primitives.py:18-206 — but organised
by STAGE over all shapes and segments instead of by shape and segment:

  clustering   bandwidths, mean-shift iterations and non-maximum suppression of all shapes as
               batched launches (occupied centres / cluster centres are padded lists instead of
               torch.unique / nonzero results): one small download sizes the neighbour matrix,
               ONE download brings the cluster ids of all shapes to the host;
  host         Hungarian matching and the segment table of every shape (what the reference also
               does on the host), ONE packed upload;
  primitives   csrc/fitbatch.hip: weighted moments -> fits -> cone pass -> residuals, 4 launches
               forward and 1 backward for every plane / sphere / cylinder / cone of the step;
  splines      standardisation of all spline segments as batched tensor expressions (ONE
               download of the 3x3 covariances: the minor axis is taken from LAPACK geev on the
               host like the reference, whose sign convention fixes the canonical frame), ONE
               SplineNet forward per net over its segments, B-spline evaluation with the
               de-standardisation folded in, ONE ragged Chamfer launch per direction;
  loss         per-shape means, ONE download (distances, fit status, predicted types).

The reference's serial path costs ~250 launches and ~10 host synchronisations per segment; this
one ~60 launches per shape and 4 synchronisations per STEP (one per shape at batch 4)."""
import os

import numpy as np
import torch
from torch.profiler import record_function

from . import kernels as K
from . import mean_shift as MSM
from ._lib import _PinnedRing, h2d, pinned_like, wait_event
from .dp import FitStatusError
from .losses import normalized_rows

EPS = float(np.finfo(np.float32).eps)
SPLINE_TYPES = (0, 2, 6, 7, 9, 8)
CLOSED_TYPES = (0, 9, 6, 7)
OPEN_TYPES = (2, 8)
PRIM_CODE = {1: K.PRIM_PLANE, 5: K.PRIM_SPHERE, 4: K.PRIM_CYLINDER, 3: K.PRIM_CONE}
PRIM_NAME = {K.PRIM_PLANE: "plane", K.PRIM_SPHERE: "sphere", K.PRIM_CYLINDER: "cylinder", K.PRIM_CONE: "cone"}
CMAX = 64          # padded list of cluster centres (the guard retries above 49 anyway)
# mean-shift backward through the centre rows only (mean_shift._CentreRows; 0: dense passes over all rows)
ROWS_BWD = os.environ.get("PARSENET_MS_ROWS_BWD", "1") != "0"
# the open and the closed SplineNet of a fitting stage on two streams (0: one after the other)
SPLINE_STREAMS = os.environ.get("PARSENET_SPLINE_STREAMS", "1") != "0"
_SIDE = {}


def _side_stream(dev):
    key = (dev.type, dev.index)
    st = _SIDE.get(key)
    if st is None:
        st = _SIDE[key] = torch.cuda.Stream(device=dev)
    return st


# -------------------------------------------------------------------------------------------
# clustering
# -------------------------------------------------------------------------------------------
_CONSTS = {}


def _device_const(key, make):
    """Small constant tensors uploaded once: an upload from pageable memory makes the host wait for
    everything queued on the stream (see _lib.h2d), once per step is once too often."""
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = make()
    return t


def _fitter_bases(fitter, dev):
    """Device copies of the fitter's B-spline bases, kept ON the fitter (they die with it; a
    module-level cache keyed by id() could hand a recycled id another fitter's grid)."""
    cache = fitter.__dict__.setdefault("_bases_dev", {})
    hit = cache.get(dev)
    if hit is None or hit[0] is not fitter.nu or hit[1] is not fitter.nv:
        hit = cache[dev] = (fitter.nu, fitter.nv, fitter.nu.to(dev), fitter.nv.to(dev))
    return hit[2], hit[3]


def bandwidth_batch(X, quantile, num_samples=10000):
    """MeanShift.compute_bandwidth (src/mean_shift.py:115-137) for every shape of X (B,N,128):
    (bw (B,) clamped at 0.003, flagged rows per shape (B,)) or None outside the selection
    kernel's fast path.  Needs num_samples >= N (every row is used, the shuffle is immaterial;
    the caller still draws it from numpy's RNG)."""
    B, N, D = X.shape
    Kq = int(quantile * num_samples)
    if N > num_samples or not (1 <= Kq <= N):
        return None
    Xc = X.detach().contiguous()
    res = None
    if MSM.ARITH == "fp16x2" and D == 128:
        res = K.dot_kth_unit(Xc, K.meanshift_h2_split(Xc), N, Kq)
    elif MSM.ARITH == "bf16x3":
        # the statistic in the arithmetic of the iterations it parametrises: fp32-grade dot products
        # from the error-free bf16 x 3 split (1e-7 on unit rows; the bandwidth is a mean over N of them)
        res = K.dot_kth_x3(Xc, Xc, Kq)
    if res is None:
        res = K.dot_select(Xc, Xc, Kq, want_value=True)
    if res is None:
        return None
    kth_dot, flags = res
    kth = 2.0 - 2.0 * kth_dot
    bw = torch.mean(torch.sqrt(torch.clamp(kth, min=1e-6)), 1)
    return torch.clamp(bw, min=0.003), (flags != 0).sum(1)


def nms_batch(new_X, X, bw, width=None, labels=True, nearest=None):
    """MeanShift.nms (src/mean_shift.py:139-179) for all shapes at once.  new_X, X (B,N,128)
    detached, bw (B,).  Returns a dict: labels (B,N) int64 (None with ``labels=False``: the caller
    takes them from the membership kernel it runs anyway), cid (B,CMAX) int64 ascending centre ids
    (padded), ncl, nocc, nflag (B,) and the padded ``width`` of the neighbour matrix.  None outside
    the kernel's fast path.

    The number of occupied centres sizes the neighbour matrix.  Any width >= max(nocc) gives the
    same result (padding is masked), so ``width`` may be a guess — the caller downloads nocc
    together with the cluster ids and calls again with ``width=None`` if the guess was too small;
    without a guess nocc is downloaded here (one more synchronisation).  nflag counts the rows the
    selection kernel flagged (massive ties).

    The reference scores every occupied centre u against ALL N shifted points j with
    [dist(u,j) < b] * members(j); unoccupied j score 0 and the row maximum is at least members(u)
    > 0, so only occupied columns can win: the neighbour matrix is (occupied x occupied), in
    ascending centre order — the first-index tie rule is unchanged.  Counting, the ordered lists of
    occupied / voted centres and the vote itself are three small kernels (csrc/fused.hip:
    pn_nms_occupied_f32, pn_nms_vote_f32) around one GEMM."""
    B, N, D = X.shape
    if nearest is not None:
        # (B,N) from the planned iterations (kernels.meanshift_x3_nearest): the same indices as the
        # selection engine's, evaluated on the tile pairs that can hold a maximum only; nothing to flag
        member = nearest
        nflag = torch.zeros(B, dtype=torch.int64, device=X.device)
        if os.environ.get("PARSENET_CHECK_NEAREST") == "1":       # developer check against the selection engine
            ref, fl = K.dot_select(X, new_X, 1, want_value=False)
            bad = (ref[:, :, 0] != nearest) & (fl == 0)
            if bool(bad.any()):
                raise AssertionError("pruned nearest differs from dot_select in %d rows (B=%d N=%d)" % (int(bad.sum()), B, N))
    else:
        res = K.dot_select(X, new_X, 1, want_value=False)
        if res is None:
            return None
        idx, flags = res
        member = idx[:, :, 0]
        nflag = (flags != 0).sum(1)
    cap = N if width is None else min(int(width), N)
    counts, uq, nocc = K.nms_occupied(member, cap)
    U = cap
    if width is None:
        U = max(int(nocc.max().item()), 1)                                    # sync: sizes the next launches
        uq = uq[:, :U].contiguous()
    Cu = torch.gather(new_X, 1, uq.unsqueeze(2).expand(-1, -1, D))
    cid, ncl = K.nms_vote(torch.bmm(Cu, Cu.transpose(1, 2)), uq, nocc, counts, bw, CMAX)
    out = {"labels": None, "cid": cid, "ncl": ncl, "nocc": nocc, "nflag": nflag, "width": U}
    if labels:
        out["labels"] = centre_labels(new_X, X, cid, ncl, bw)
    return out


def centre_labels(new_X, X, cid, ncl, bw):
    """labels = first arg-max over the pruned centres of centre . x (src/mean_shift.py:176-178)."""
    B, N, D = X.shape
    Csel = torch.gather(new_X, 1, cid.unsqueeze(2).expand(-1, -1, D))
    if D == 128 and cid.shape[1] in (16, 32, 64):
        return K.membership_fwd(Csel, X, bw, torch.clamp(ncl, max=cid.shape[1]), EPS, want_labels=True)[4]
    cvalid = torch.arange(cid.shape[1], device=X.device).unsqueeze(0) < ncl.unsqueeze(1)
    sc = torch.bmm(Csel, X.transpose(1, 2))                                     # (B,CMAX,N)
    sc = torch.where(cvalid.unsqueeze(2), sc, torch.full_like(sc, float("-inf")))
    return MSM._first_argmax(sc, 1)


def nms_width_guess(ev, B, N):
    """Width for the next nms_batch call of this problem size: a quarter above what the previous
    one needed, in steps of 256 (None before the first call).  The memory lives on the owning
    Evaluation object, not in the module."""
    return ev.__dict__.setdefault("_nms_width", {}).get((B, N))


def nms_width_update(ev, B, N, nocc_max):
    ev.__dict__.setdefault("_nms_width", {})[(B, N)] = min(N, (int(nocc_max * 1.25) // 256 + 1) * 256)


# -------------------------------------------------------------------------------------------
# memberships
# -------------------------------------------------------------------------------------------
def weights_normalize_batch(Wraw, bw, ncl):
    """fitting_utils.weights_normalize (src/fitting_utils.py:306-325) on padded (B,Cp,N) centre /
    point dot products: rows >= ncl[b] are padding and come out as zeros."""
    B, Cp, N = Wraw.shape
    rowmask = (torch.arange(Cp, device=Wraw.device).unsqueeze(0) < ncl.unsqueeze(1)).unsqueeze(2)
    x = Wraw / (bw.reshape(B, 1, 1) ** 2) / 2
    prob = torch.exp(torch.clamp(x, max=75, min=-75))
    prob = torch.where(rowmask, prob, torch.zeros_like(prob))
    prob = prob / torch.sum(prob, 1, keepdim=True)
    shifted = prob - torch.min(prob, 2, keepdim=True)[0]
    shifted = shifted / (torch.max(shifted, 2, keepdim=True)[0] + EPS)
    return torch.where((ncl > 1).reshape(B, 1, 1), shifted, prob)


class _Membership(torch.autograd.Function):
    """centres (B,CP,128) padded, embedding (B,N,128), bandwidths (B,), cluster counts (B,) ->
    (Wn, Wraw) (B,CP,N): src/residual_utils.py:120 + fitting_utils.weights_normalize in two
    launches (csrc/fused.hip); backward: two launches + the two GEMMs onto centres and embedding."""

    @staticmethod
    def forward(ctx, cen, emb, bw, ncl):
        cen, emb = cen.contiguous(), emb.contiguous()
        Wraw, prob, Wn, rowstat, labels = K.membership_fwd(cen, emb, bw, ncl, EPS, want_labels=True)
        ctx.save_for_backward(cen, emb, bw, ncl, Wraw, prob, rowstat)
        ctx.mark_non_differentiable(Wraw, labels)
        return Wn, Wraw, labels

    @staticmethod
    def backward(ctx, gWn, _gWraw, _glabels):
        cen, emb, bw, ncl, Wraw, prob, rowstat = ctx.saved_tensors
        gWraw = K.membership_bwd(gWn, Wraw, prob, rowstat, bw, ncl)
        return torch.bmm(gWraw, emb), torch.bmm(gWraw.transpose(1, 2), cen), None, None


def memberships(cen, emb, bw, ncl):
    """(Wn, Wraw) for padded centre rows; the fused kernels for 128-d embeddings and at most 64
    centres (the guard retries above 49 anyway), tensor expressions otherwise."""
    B, Cp, D = cen.shape
    if D == 128 and Cp <= 64:
        CP = 16 if Cp <= 16 else 32 if Cp <= 32 else 64
        cen = torch.nn.functional.pad(cen, (0, 0, 0, CP - Cp))
        return _Membership.apply(cen, emb, bw.contiguous(), ncl)[:2]
    Wraw = torch.bmm(cen, emb.transpose(1, 2))
    return weights_normalize_batch(Wraw, bw, ncl), Wraw


# -------------------------------------------------------------------------------------------
# analytic primitives
# -------------------------------------------------------------------------------------------
class _PrimitiveFitLoss(torch.autograd.Function):
    """Wn (B,Cp,N) normalised memberships -> mean residual distance of every analytic segment.
    Forward: moments, fits, cone pass, residuals (4 launches); backward: one launch."""

    @staticmethod
    def forward(ctx, Wn, P, Nrm, tab, stride, sqrt_flag):
        Wn, P, Nrm = Wn.contiguous(), P.contiguous(), Nrm.contiguous()
        partial = K.weighted_moments(P, Nrm, Wn, tab["shape"], tab["row"], stride, EPS)
        params, jac, status = K.primitive_fit(partial, tab["type"], tab["rows"])
        cone_direct = K.cone_angle(P, Wn, tab["shape"], tab["row"], tab["type"], status, params, jac, stride, EPS)
        dist, dparam = K.primitive_residual(P, tab["shape"], tab["type"], tab["gt_off"], tab["gt_idx"], params,
                                            status, sqrt_flag)
        ctx.save_for_backward(Wn, P, Nrm, params, jac, dparam, cone_direct)
        ctx.tab, ctx.stride = tab, stride
        ctx.mark_non_differentiable(params, status)
        return dist, params, status

    @staticmethod
    def backward(ctx, g_dist, _gp, _gs):
        Wn, P, Nrm, params, jac, dparam, cone_direct = ctx.saved_tensors
        tab = ctx.tab
        gW = K.weighted_moments_bwd(P, Nrm, Wn, tab["shape"], tab["row"], tab["type"], g_dist.contiguous(), dparam,
                                    jac, params, cone_direct, ctx.stride, EPS)
        return gW, None, None, None, None, None


# -------------------------------------------------------------------------------------------
# splines
# -------------------------------------------------------------------------------------------
class _BSplineEval(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ctrl, nu, nv, affine, wrap):
        ctx.save_for_backward(nu, nv, affine)
        ctx.wrap, ctx.cu, ctx.cv = wrap, ctrl.shape[1], ctrl.shape[2]
        return K.bspline_eval(nu, nv, ctrl, affine, wrap)

    @staticmethod
    def backward(ctx, gout):
        nu, nv, affine = ctx.saved_tensors
        return K.bspline_eval_bwd(nu, nv, gout.contiguous(), affine, ctx.cu, ctx.cv, ctx.wrap), None, None, None, None


class _RaggedChamfer(torch.autograd.Function):
    """chamfer_distance_single_shape (src/utils.py:326-358, two-sided, squared, reduced) of S
    (prediction, ground truth) pairs of different sizes: pred (TA,3) concatenated predictions
    (differentiable), gt (TB,3) concatenated targets -> (S,) values
    (mean_i min_j + mean_j min_i) / 2.  The kernel's minima are the reference's values bit for
    bit; the backward routes each minimum to its pair of points."""

    @staticmethod
    def forward(ctx, pred, gt, off_a, off_b, max_a, max_b):
        minA, argA, minB, argB = K.chamfer_nn_ragged(pred.detach(), off_a, max_a, gt, off_b, max_b)
        ctx.save_for_backward(pred, gt, argA, argB, off_a, off_b)
        ctx.max_a = max_a
        # per-item means in a fixed order (index_add_ would use fp32 atomics: not reproducible)
        return K.chamfer_ragged_reduce(minA, off_a, minB, off_b)

    @staticmethod
    def backward(ctx, g):
        pred, gt, argA, argB, off_a, off_b = ctx.saved_tensors
        # d/d minA_i = g_s / (2 nA), the 2 of the square cancels; targets whose nearest prediction is
        # row i are gathered in ascending order (csrc/chamfer.hip)
        gpred = K.chamfer_ragged_bwd(pred.detach(), off_a, ctx.max_a, gt, off_b, argA, argB, g.contiguous())
        return gpred, None, None, None, None, None


def _host_minor_axis_rotation(cov):
    """standardize_point_torch's host step (src/fitting_utils.py:532-540): eigenvectors of the 3x3
    covariance from LAPACK geev (its sign convention decides the canonical frame), minor axis
    rotated to +x.  cov (3,3) float32 CPU tensor -> R (3,3) float32 numpy."""
    from .fitting import rotation_matrix_a_to_b
    w, v = torch.linalg.eig(cov)
    U = v.real
    smallest_ev = U[:, torch.min(w.real, 0)[1]].numpy()
    return rotation_matrix_a_to_b(smallest_ev, np.array([1, 0, 0])).astype(np.float32)


def host_minor_axis_rotations(cov):
    """_host_minor_axis_rotation for a stack cov (S,3,3) (float32 CPU tensor) -> (S,3,3) float32
    numpy, bit for bit the per-matrix function's result but ~6x cheaper on the host (the device
    idles meanwhile): ONE batched geev call (LAPACK is still entered once per matrix, with the same
    input), and rotation_matrix_a_to_b specialised to B = (1, 0, 0) with the two np.cross calls and
    the dots written out in the float64 scalar operations numpy performs for them; norm, inv and
    the two matrix products stay numpy's (tests/test_host_logic.py pins the equality)."""
    w, v = torch.linalg.eig(cov)
    k = torch.min(w.real, 1)[1].numpy()
    U = v.real.numpy()
    out = np.empty((cov.shape[0], 3, 3), dtype=np.float32)
    for s in range(cov.shape[0]):
        A = U[s][:, k[s]]
        a0, a1, a2 = float(A[0]), float(A[1]), float(A[2])
        cos = a0 * 1.0 + a1 * 0.0 + a2 * 0.0                      # np.dot(A, B)
        wv = np.array([0.0 * a2 - 0.0 * a1, 0.0 * a0 - 1.0 * a2, 1.0 * a1 - 0.0 * a0])    # np.cross(B, A)
        sin = np.linalg.norm(wv)
        vv = np.array([1.0 - cos * a0, 0.0 - cos * a1, 0.0 - cos * a2])                  # B - dot * A
        Fm = np.stack([A.astype(np.float64), vv / (np.linalg.norm(vv) + EPS), wv / (np.linalg.norm(wv) + EPS)], 1)
        G = np.array([[cos, -sin, 0], [sin, cos, 0], [0, 0, 1]])
        try:
            out[s] = (Fm @ G @ np.linalg.inv(Fm)).astype(np.float32)
        except np.linalg.LinAlgError:
            out[s] = np.eye(3, dtype=np.float32)
    return out


STD_FUSED = os.environ.get("PARSENET_STD_FUSED", "1") != "0"


def standardize_segments(P2, w):
    """standardize_point_torch (src/fitting_utils.py:512-553) for S segments at once.
    P2 (S,n,3) sub-sampled points of each segment's shape, w (S,n) memberships (+EPS).
    Returns (points (S,n,3), std (S,3), mean (S,3), R (S,3,3)); one download (covariances)."""
    S, n, _ = P2.shape
    with torch.no_grad():
        # the selection (w > 0.8, or the top quarter / half of the memberships when fewer than 400 qualify) and, below,
        # extents + scaling as one launch each (csrc/fused.hip: pn_standardize_*; round 5: a topk over half of every row
        # and 31 launches in all).  Mean, covariance and rotation keep round 5's expressions — and bits: the sign of
        # LAPACK's eigenvector follows the last bits of the covariance (profiles/r06_std_sign_probe.txt)
        kf = n // 4 if n >= 7500 else n // 2
        if STD_FUSED:
            selb = K.standardize_select(w, max(kf, 1))
            self_ = selb.float()
        else:               # PARSENET_STD_FUSED=0 (developer A/B): round 5's tensor expressions
            hi = w > 0.8
            cnt = hi.sum(1, keepdim=True)
            top = torch.topk(w, kf, dim=1)[1]
            fb = torch.zeros_like(hi).scatter_(1, top, torch.ones_like(top, dtype=torch.bool))
            selb = torch.where(cnt < 400, fb, hi)
            self_ = selb.float()
        wsel = w * self_
        mean = (P2 * wsel.unsqueeze(2)).sum(1) / (wsel.sum(1, keepdim=True) + EPS)
        Pc = P2 - mean.unsqueeze(1)
        cov = torch.bmm((Pc * self_.unsqueeze(2)).transpose(1, 2), Pc)
        cov_h, slot = pinned_like(cov.shape, cov.dtype, hold=True)   # host step of the reference: download ...
        try:
            cov_h.copy_(cov, non_blocking=True)
            if slot is not None:
                _PinnedRing.arm(slot)
                wait_event(slot["event"])
            else:
                torch.cuda.current_stream(cov.device).synchronize()
            rot = host_minor_axis_rotations(cov_h)                 # ... batched geev ...
        finally:
            _PinnedRing.release(slot)
        R = h2d(rot, P2.device)                                    # ... upload
        Pr = torch.bmm(Pc, R.transpose(1, 2))
        if STD_FUSED:
            pts, std = K.standardize_scale(Pr, w, selb, EPS)
        else:
            wp = Pr * w.unsqueeze(2)
            big = torch.full_like(wp, float("inf"))
            selx = selb.unsqueeze(2)
            std = torch.abs(torch.where(selx, wp, -big).max(1)[0] - torch.where(selx, wp, big).min(1)[0])
            pts = Pr / (std.unsqueeze(1) + EPS)
    return pts, std, mean, R


# -------------------------------------------------------------------------------------------
# host side: matching and the segment table
# -------------------------------------------------------------------------------------------
def precompute_ground_truth(labels, primitives):
    """The part of build_segment_table that depends on the ground truth alone — label array, modal
    primitive type of every gt segment, the point lists of the gt segments — so that the host can
    prepare it while the device is still clustering and it has nothing else to do."""
    g = np.asarray(labels).astype(np.int64)
    if g.size and (g.min() < 0 or g.max() >= 50):
        raise ValueError("labels must lie in [0, 50) (one-hot width of the reference)")
    modal = np.bincount(g * 10 + np.asarray(primitives).astype(np.int64), minlength=500).reshape(50, 10).argmax(1)
    lists = {int(c): np.flatnonzero(g == c) for c in np.flatnonzero(np.bincount(g, minlength=50))}
    return g, modal, lists


def build_segment_table(labels, primitives, cluster_ids, N, pre=None):
    """What residual_train_mode + fit_one_shape_torch decide on the host for ONE shape
    (src/residual_utils.py:154-200, src/primitive_forward.py:938-1020): Hungarian matching, the
    ground-truth segment of every predicted cluster, its modal primitive type, the sub-sampling
    and minimum-size rules, at most 4 splines.  Returns (segments, match) where a segment is a dict
    {row, key, type, kind ('prim' | 'open' | 'closed'), gt (indices)} in the order the reference
    fits them, and match = (rids, cids, unique_pred, gt counts, confusion matrix, first index of
    every gt label) for the metrics.  Everything comes from two histograms of the shape: the
    50 x 50 confusion matrix and the 50 x 10 (gt label, primitive type) table."""
    from .fitting import solve_dense
    p = np.asarray(cluster_ids).astype(np.int64)
    g, modal, gt_lists = pre if pre is not None else precompute_ground_truth(labels, primitives)
    if p.size and (p.min() < 0 or p.max() >= 50):
        raise ValueError("labels must lie in [0, 50) (one-hot width of the reference)")
    conf = np.bincount(p * 50 + g, minlength=2500).reshape(50, 50)
    dots = conf.astype(np.float32)
    # relaxed_iou_fast on the one-hot encodings (segment_utils.py:356-374): integer counts, the
    # same fp32 expression -> the reference's matrix bit for bit
    iou = dots / (dots.sum(1, keepdims=True, dtype=np.float32) + dots.sum(0, keepdims=True, dtype=np.float32)
                  - dots + np.float32(1e-7))
    rids, cids = solve_dense(1.0 - iou)
    pcount, gcount = conf.sum(1), conf.sum(0)
    unique_pred = np.flatnonzero(pcount)                                     # = np.unique(cluster_ids)
    n2 = (N + 1) // 2
    n4 = (n2 + 1) // 2
    segs, spline_count = [], 0
    for index, i in enumerate(unique_pred):
        c = cids[i]
        if gcount[c] == 0:
            continue
        seg_type = int(modal[c])             # most frequent type of the gt segment, smallest on ties
        if seg_type in SPLINE_TYPES:
            spline_count += 1
            if spline_count > 4 or n2 < 100:
                continue
            kind = "closed" if seg_type in CLOSED_TYPES else "open"
        elif seg_type in PRIM_CODE:
            if n4 < 20:
                continue
            kind = "prim"
        else:
            raise ValueError("unknown primitive type %r" % (seg_type,))
        segs.append({"row": index, "key": int(i), "type": seg_type, "kind": kind, "gt": gt_lists[int(c)]})
    return segs, (rids, cids, unique_pred, gcount, conf, g)


def siou_matched_segments_fast(match, prim_pred_per_cluster, primitives):
    """segment_utils.SIOU_matched_segments (src/segment_utils.py:139-187) from the confusion matrix
    of build_segment_table instead of one boolean mask pair per match: identical values."""
    from .fitting import _merge_types
    rids, cids, _, ng, conf, g = match
    np_ = conf.sum(1)
    prim = None
    ious, ok, pairs = [], [], []
    for r, c in zip(rids, cids):
        if ng[c] == 0 or np_[r] == 0 or ng[c] < 100:
            continue
        inter = conf[r, c]
        ious.append(inter / ((np_[r] + ng[c] - inter) + 1e-8))
        if prim is None:
            prim = _merge_types(primitives)
        gt_type = prim[np.argmax(g == c)]          # primitives[gt segment][0]: its first point
        ok.append(gt_type == prim_pred_per_cluster[r])
        pairs.append([gt_type, prim_pred_per_cluster[r]])
    return (np.mean(ious) if ious else float("nan"), np.mean(ok) if ok else float("nan"), [[rids, cids]], pairs)


# -------------------------------------------------------------------------------------------
# the stage
# -------------------------------------------------------------------------------------------
def fitting_losses_train(ev, embedding, points, normals, labels, primitives, primitives_log_prob, quantile,
                         iterations, lamb, defer_metrics=False):
    """Training-mode Evaluation.fitting_loss for every shape of the batch.  Returns a list (one
    entry per shape) of ([Loss, geometric mean, spline mean, s_iou, p_iou], [parameters, cluster
    ids, weights]) exactly as the reference's call with that single shape would.
    ``ev``: the owning fitting.Evaluation (SplineNets, mean-shift object, slow-path helpers)."""
    stage = _fitting_stage(ev, embedding, points, normals, labels, primitives, primitives_log_prob, quantile,
                           iterations, lamb, defer_metrics)
    next(stage)                      # clustering queued, the cluster-id download under way
    try:
        next(stage)                  # host matching, fits, losses
    except StopIteration as done:
        return done.value
    raise RuntimeError("fitting stage: unexpected second suspension")


def fitting_losses_train_pipelined(ev, embedding, points, normals, labels, primitives, primitives_log_prob, quantile,
                                   iterations, lamb, chunks=2):
    """The same stage for a batch cut into ``chunks`` groups of shapes, software-pipelined: the
    clustering of EVERY group is queued first (bandwidth, ten mean-shift iterations, NMS,
    memberships, and the stream-ordered copy of its cluster ids into pinned memory), only then
    does the host turn to group 0 — it waits for that group's copy alone, and while it runs the
    Hungarian matching and builds the segment tables the device is busy with the next group's
    iterations instead of idling (one group: the device waits for the host at that point, ~2-5 ms
    per step, more on a busy host).  The reference processes shapes strictly one after the other
    (train_parsenet_e2e.py:190-241); shapes are independent, so the grouping changes no result
    (numpy's RNG is consumed in shape order: all draws happen in the host part).
    Returns (losses (B,) on the device, finish) like ``fitting_losses_train(defer_metrics=True)``."""
    B = embedding.shape[0]
    chunks = max(1, min(int(chunks), B))
    bounds = [(B * c // chunks, B * (c + 1) // chunks) for c in range(chunks)]
    labels, primitives = np.asarray(labels), np.asarray(primitives)
    stages = [_fitting_stage(ev, embedding[lo:hi], points[lo:hi], normals[lo:hi], labels[lo:hi], primitives[lo:hi],
                             primitives_log_prob[lo:hi], quantile, iterations, lamb, True) for lo, hi in bounds]
    for st in stages:
        next(st)
    parts = []
    for st in stages:
        try:
            next(st)
        except StopIteration as done:
            parts.append(done.value)
            continue
        raise RuntimeError("fitting stage: unexpected second suspension")
    loss_b = torch.cat([p[0] for p in parts])

    def finish():
        out = []
        for _, fin in parts:
            out += fin()
        return out
    return loss_b, finish


def _fitting_stage(ev, embedding, points, normals, labels, primitives, primitives_log_prob, quantile, iterations,
                   lamb, defer_metrics):
    """Generator behind the two functions above: runs up to the point where the host needs the
    cluster ids (their copy into pinned memory is queued, an event recorded), yields ONCE, and on
    resumption waits for that copy and does the rest.  Returns (via StopIteration) what
    fitting_losses_train returns."""
    B, N, D = embedding.shape
    dev = embedding.device
    labels, primitives = np.asarray(labels), np.asarray(primitives)
    points, normals = points.contiguous(), normals.contiguous()
    # (contiguous ONCE: the callers hand over a permuted view of the network's (B,128,N) output and the bandwidth, the
    # iterations and the memberships each made their own (B,N,128) copy of it)
    emb = normalized_rows(embedding).contiguous()        # (shared with the embedding loss of the same step)
    fitter = ev.fitter

    # ---- clustering, all shapes ---------------------------------------------------------
    state = None
    with torch.no_grad(), record_function("fit:bandwidth"):
        bwres = bandwidth_batch(emb, quantile)
    if bwres is not None and D == 128:
        bw, bwflag = bwres
        with record_function("fit:meanshift_fwd"):
            # (a planned call also returns every point's nearest shifted point, the first step of the NMS:
            # the iterations hold both sides in the locality order its exact pruning needs)
            MSM.WANT_NEAREST = os.environ.get("PARSENET_MS_NEAREST", "1") != "0"
            try:
                # (no autograd graph through the N rows: the gradient enters at the centre rows only —
                # MSM.centre_rows below — and the backward then runs those rows alone; PARSENET_MS_ROWS_BWD=0:
                # the dense backward passes over all rows, same gradient up to the summation order)
                if ROWS_BWD:
                    new_X, ms_state = MSM.mean_shift_iterations_state(emb, bw, iterations)
                else:
                    new_X, ms_state = MSM.mean_shift_iterations(emb, bw, iterations), None
            finally:
                MSM.WANT_NEAREST = False
            nearest, MSM.LAST_NEAREST = MSM.LAST_NEAREST, None

        def cluster(width):
            """NMS down to the padded list of centres, then — still on the device, before anything is
            downloaded — the memberships of ALL CMAX padded centre rows (rows >= ncl are masked in
            the kernel), whose first output are the cluster labels the host is waiting for."""
            with torch.no_grad(), record_function("fit:nms"):
                st = nms_batch(new_X.detach(), emb.detach(), bw, width, labels=False, nearest=nearest)
            if st is not None:
                with record_function("fit:memberships"):
                    if ms_state is not None:
                        cen_all = MSM.centre_rows(emb, ms_state, st["cid"])                            # (B,CMAX,D)
                    else:
                        cen_all = torch.gather(new_X, 1, st["cid"].unsqueeze(2).expand(-1, -1, D))
                    st["cen"] = cen_all
                    st["Wn"], st["Wraw"], st["labels"] = _Membership.apply(cen_all, emb, bw, torch.clamp(st["ncl"], max=CMAX))
            return st
        state = cluster(nms_width_guess(ev, B, N))
    with torch.no_grad():
        # SIOU_matched_segments merges the predicted types before the per-cluster vote
        # (src/segment_utils.py:152-161: 0, 6, 7 -> 9; 8 -> 2)
        lut = _device_const(("merge_lut", dev), lambda: h2d(np.asarray([9, 1, 2, 3, 4, 5, 9, 9, 2, 9], dtype=np.int64), dev))
        prim_pred = lut[torch.max(primitives_log_prob, 1)[1]]
    if state is not None:
        # auto mode of the block-sparse iterations: the share of list entries the plans of this call
        # kept (a device scalar, in 1e-6 units) rides in front of the pack
        auto_stat, MSM.AUTO_STAT = MSM.AUTO_STAT, None
        head = (torch.zeros(1, device=dev) if auto_stat is None else auto_stat.reshape(1) * 1e6 + 1.0).long()

        def start_download(st):
            """Stream-ordered copy of the cluster ids (and the few counters riding along) into pinned
            memory + an event: the host blocks on THIS copy only, whatever else is queued behind it."""
            dev_pack = torch.cat([head, st["labels"].reshape(-1), st["cid"].reshape(-1), st["ncl"], bwflag,
                                  st["nocc"], st["nflag"]]).to(torch.int32)
            # staging ring: no pinned allocation per step; HELD until the host has copied the ids out
            host, slot = pinned_like(dev_pack.shape, torch.int32, hold=True)
            host.copy_(dev_pack, non_blocking=True)
            if slot is not None:
                _PinnedRing.arm(slot)
                return host, slot["event"], slot
            done = torch.cuda.Event()
            done.record()
            return host, done, None

        def collect(pend):
            """Wait for a started download and copy it out of its ring slot (the slot goes back into
            the rotation: the uploads of the host stage below — or of another group — may take it)."""
            try:
                wait_event(pend[1])
                return pend[0].numpy().copy()
            finally:
                _PinnedRing.release(pend[2])

        def download(st):
            return collect(start_download(st))
        pending = start_download(state)
        yield                                                    # (a pipelined caller queues the next group here)
        # numpy RNG: one shuffle per mean_shift call of the reference (src/mean_shift.py:121-122), i.e. one
        # per shape on the fast path.  They are drawn HERE, while the device is still clustering and the
        # host would only wait, instead of after the download when the device waits for the host; should
        # a shape turn out to need the synchronous path the state is rewound and the draws are redone
        # in the reference's order below.
        rng_state = np.random.get_state()
        for _ in range(B):
            np.random.shuffle(np.arange(N))
        # ... and so is everything the matching needs from the ground truth alone
        gt_pre = [precompute_ground_truth(labels[b], primitives[b]) for b in range(B)]
        pack = collect(pending)                                  # download: cluster ids
        if pack[0] > 0:
            MSM.auto_report(B, N, (float(pack[0]) - 1.0) * 1e-6)
        if int(pack[-2 * B:-B].max()) > state["width"]:
            # the guessed width of the neighbour matrix was too small (the clustering changed a lot
            # since the last step): once more with the exact one
            state = cluster(None)
            pack = download(state)
        nms_width_update(ev, B, N, int(pack[-2 * B:-B].max()))
        o = 1
        lab_h = pack[o:o + B * N].reshape(B, N); o += B * N
        cid_h = pack[o:o + B * CMAX].reshape(B, CMAX); o += B * CMAX
        ncl_h, bwflag_h = pack[o:o + B], pack[o + B:o + 2 * B]
        nflag_h = pack[-B:]
        predrawn = all(nflag_h[b] == 0 and bwflag_h[b] == 0 and ncl_h[b] <= 49 for b in range(B))
        if not predrawn:
            np.random.set_state(rng_state)
    else:
        predrawn = False
        gt_pre = [None] * B
        yield
    centers, bws, cluster_ids = [], [], []
    all_fast = state is not None
    for b in range(B):
        fast = state is not None and nflag_h[b] == 0 and bwflag_h[b] == 0 and ncl_h[b] <= CMAX
        if fast and not predrawn:
            np.random.shuffle(np.arange(N))
        if fast and ncl_h[b] <= 49:
            centers.append(state["cen"][b, :int(ncl_h[b])])
            bws.append(bw[b])
            cluster_ids.append(lab_h[b].astype(np.int64))
        else:
            # tie-flagged selection rows or the guard's retry above 49 clusters: this shape alone on
            # the synchronous path
            all_fast = False
            q = quantile * 1.2 if (fast and ncl_h[b] > 49) else quantile
            c, bwb, ids = ev.guard_mean_shift(emb[b], q, iterations, kernel_type="gaussian")
            centers.append(c)
            bws.append(bwb.reshape(()))
            cluster_ids.append(ids.cpu().numpy().astype(np.int64))
    ncl_list = [int(c.shape[0]) for c in centers]
    if all_fast:
        # the memberships the device computed before the download are the ones to use
        Wn, Wraw, bwt, Cp = state["Wn"], state["Wraw"], bw, CMAX
    else:
        Cp = max(ncl_list)
        cen = torch.stack([torch.nn.functional.pad(c, (0, 0, 0, Cp - c.shape[0])) for c in centers], 0)   # (B,Cp,D)
        bwt = torch.stack(bws).detach()
        ncl_t = h2d(np.asarray(ncl_list, dtype=np.int64), dev)
        with record_function("fit:memberships"):
            Wn, Wraw = memberships(cen, emb, bwt, ncl_t)                         # (B,Cp,N), Cp padded to 16/32/64
            Cp = Wn.shape[1]

    # ---- host: matching + segment tables -------------------------------------------------
    tables, matches = [], []
    for b in range(B):
        segs, m = build_segment_table(labels[b], primitives[b], cluster_ids[b], N, gt_pre[b])
        tables.append(segs)
        matches.append(m)
    prim_segs = [(b, s) for b in range(B) for s in tables[b] if s["kind"] == "prim"]
    spl_segs = [(b, s) for b in range(B) for s in tables[b] if s["kind"] != "prim"]
    spl_segs.sort(key=lambda t: t[1]["kind"] != "open")       # open first, then closed (stable)
    n_open = sum(1 for _, s in spl_segs if s["kind"] == "open")
    all_segs = prim_segs + spl_segs
    S_p, S_s = len(prim_segs), len(spl_segs)

    # ONE packed upload of every table (float tables travel as their bit patterns)
    parts, where, floats = [], {}, set()

    def add(name, arr, dtype=np.int32):
        a = np.ascontiguousarray(arr, dtype=dtype).reshape(-1)
        if dtype == np.float32:
            floats.add(name)
            a = a.view(np.int32)
        where[name] = (sum(p.size for p in parts), a.size)
        parts.append(a)
    gt_lists = [s["gt"] for _, s in all_segs]
    gt_off = np.concatenate([[0], np.cumsum([g.size for g in gt_lists])]) if all_segs else np.zeros(1)
    add("seg_shape", [b for b, _ in all_segs])
    add("seg_row", [s["row"] for _, s in all_segs])
    add("seg_type", [PRIM_CODE.get(s["type"], 0) for _, s in all_segs])
    add("seg_rows", [((N + 1) // 2 + 1) // 2] * len(all_segs))
    add("gt_off", gt_off)
    add("gt_idx", np.concatenate(gt_lists) if all_segs else np.zeros(0))
    add("gt_flat", np.concatenate([g + b * N for (b, _), g in zip(all_segs, gt_lists)]) if all_segs else np.zeros(0))
    if S_s:
        na = [900 if s["kind"] == "open" else 930 for _, s in spl_segs]
        nb = [g.size for g in gt_lists[S_p:]]
        add("off_a", np.concatenate([[0], np.cumsum(na)]))
        add("off_b", np.concatenate([[0], np.cumsum(nb)]))
    add("scale", [lamb if s_["kind"] != "prim" else 1.0 for _, s_ in all_segs], np.float32)
    add("cnt_shape", np.maximum(np.bincount(np.asarray([b for b, _ in all_segs], dtype=np.int64), minlength=B), 1),
        np.float32)
    # per-shape sums as a masked row sum (fixed order; index_add_ would use fp32 atomics)
    add("seg_of_shape", (np.arange(B)[:, None] == np.asarray([b for b, _ in all_segs], dtype=np.int64)[None, :]),
        np.float32)
    dev_tab = h2d(np.concatenate(parts) if parts else np.zeros(1, np.int32), dev)
    T = {k: (dev_tab[o:o + n].view(torch.float32) if k in floats else dev_tab[o:o + n]) for k, (o, n) in where.items()}

    dists = []
    params_p = status = None
    # ---- analytic primitives ------------------------------------------------------------
    if S_p:
        tab = {"shape": T["seg_shape"][:S_p], "row": T["seg_row"][:S_p], "type": T["seg_type"][:S_p],
               "rows": T["seg_rows"][:S_p], "gt_off": T["gt_off"][:S_p + 1], "gt_idx": T["gt_idx"]}
        with record_function("fit:primitives"):
            d_p, params_p, status = _PrimitiveFitLoss.apply(Wn, points, normals, tab, 4, False)
        dists.append(d_p)
    # ---- splines ------------------------------------------------------------------------
    recs = []
    if S_s:
        sb = T["seg_shape"][S_p:].long()
        sr = T["seg_row"][S_p:].long()
        with record_function("fit:standardize"):
            P2 = points[:, 0::2].detach()[sb]                                # (S_s,n2,3)
            w2 = Wn[:, :, 0::2][sb, sr] + EPS                                 # (S_s,n2), differentiable
            pts_std, std, mean, R = standardize_segments(P2, w2.detach())      # sync 2
            affine = torch.cat([torch.linalg.inv(R) * std.unsqueeze(1), mean.unsqueeze(2)], 2).contiguous()
        nu, nv = _fitter_bases(fitter, dev)

        def spline_group(lo, hi, net, wrap):
            with record_function("fit:splinenet"):
                # (S,n,3) -> (S,3,n) through the LDS-tiled transpose: the tensor library's strided copy takes 0.19 ms
                # for these 90 000 floats (profiles/r06_copy_sites.txt), three times per step
                ctrl = net(K.transpose12(pts_std[lo:hi]), w2[lo:hi])
            return _BSplineEval.apply(ctrl.reshape(hi - lo, 20, 20, 3), nu, nv, affine[lo:hi], wrap)
        groups = [g for g in ((0, n_open, fitter.open_control_decoder, False),
                              (n_open, S_s, fitter.closed_control_decoder, True)) if g[1] > g[0]]
        # The two SplineNets work on a handful of segments each: most of their kernels fill a fraction of the 256
        # CUs.  With both kinds of segments present the closed net runs on a side stream next to the open one
        # (same kernels, same inputs, same results; autograd runs each node's backward on its forward stream).
        two_streams = SPLINE_STREAMS and len(groups) == 2 and dev.type == "cuda"
        if two_streams:
            main, side = torch.cuda.current_stream(dev), _side_stream(dev)
            side.wait_stream(main)                     # pts_std, w2, affine, the bases: produced on main
            with torch.cuda.stream(side):
                rec_side = spline_group(*groups[1])
            rec_main = spline_group(*groups[0])
            main.wait_stream(side)
            rec_side.record_stream(main)               # allocated on the side stream, consumed on main
            rec_list = [rec_main, rec_side]
        else:
            rec_list = [spline_group(*g) for g in groups]
        pieces = []
        for (lo, hi, _, _), rec in zip(groups, rec_list):
            pieces.append(rec.reshape(-1, 3))
            recs += [rec[k:k + 1] for k in range(hi - lo)]
        pred = torch.cat(pieces, 0)
        gt_cloud = points.reshape(B * N, 3)[T["gt_flat"][int(gt_off[S_p]):].long()]
        with record_function("fit:chamfer"):
            d_s = _RaggedChamfer.apply(pred, gt_cloud, T["off_a"], T["off_b"], max(na), max(nb))
        dists.append(d_s)

    # ---- losses, metrics, ONE download ----------------------------------------------------
    S_all = S_p + S_s
    with torch.no_grad():
        hot = torch.nn.functional.one_hot(prim_pred, 10).to(Wraw.dtype).transpose(1, 2)      # (B,10,N)
        ptype = torch.max(torch.bmm(hot, Wraw.detach().transpose(1, 2)), 1)[1]               # (B,Cp)
    if S_all:
        d_all = torch.cat(dists)
        d_used = torch.where(d_all.detach() > 1, torch.full_like(d_all, 0.1), d_all)   # degenerate case -> constant
        loss_b = (T["seg_of_shape"].reshape(B, S_all) * (d_used * T["scale"]).unsqueeze(0)).sum(1)
        loss_b = loss_b / T["cnt_shape"]
        tail = [d_all.detach().double()]
        if S_p:
            tail.append(status.double())
        tail_dev = torch.cat(tail + [ptype.reshape(-1).double()])
    else:
        loss_b = torch.zeros(B, dtype=torch.float32, device=dev)
        tail_dev = ptype.reshape(-1).double()
    pf = params_p.float() if S_p else None
    tail_host = tail_event = tail_slot = None
    if defer_metrics and tail_dev.is_cuda:
        # stream-ordered copy into pinned memory right here, behind the forward kernels: finish()
        # then waits for THIS copy only — not for the backward pass and the optimizer step the
        # caller queues in between — and the host is free to queue the next step meanwhile
        tail_host, slot = pinned_like(tail_dev.shape, tail_dev.dtype, hold=True)     # held until finish() has read it
        tail_host.copy_(tail_dev, non_blocking=True)
        tail_slot = slot
        if slot is not None:
            _PinnedRing.arm(slot)
            tail_event = slot["event"]
        else:
            tail_event = torch.cuda.Event()
            tail_event.record()

    def finish():
        """Host side of the results: ONE download (distances, fit status, voted types), then the
        per-shape records.  With ``defer_metrics`` the caller runs this after it has queued the
        backward pass, so the device never waits for the host between the two."""
        if tail_event is not None:
            try:
                wait_event(tail_event)
                host = tail_host.numpy().copy()
            finally:
                _PinnedRing.release(tail_slot)
        else:
            host = tail_dev.cpu().numpy()                                                      # sync 3
        d_h = host[:S_all]
        st_h = host[S_all:S_all + S_p].astype(np.int64) if S_p else np.zeros(0, np.int64)
        ptype_h = host[S_all + S_p:].astype(np.int64).reshape(B, Cp)
        if not np.isfinite(d_h).all():
            bad = int(np.nonzero(~np.isfinite(d_h))[0][0])
            raise FitStatusError("fitting: non-finite residual distance in segment %d of shape %d"
                                 % (all_segs[bad][1]["key"], all_segs[bad][0]))
        if (st_h & 5).any():
            bad = int(np.nonzero(st_h & 5)[0][0])
            raise FitStatusError("fitting: %s in segment %d of shape %d" % (
                "non-finite design matrix / no full-rank ridge system (lstsq)" if st_h[bad] & 1 else
                "NaN residual distance", prim_segs[bad][1]["key"], prim_segs[bad][0]))
        out = []
        for b in range(B):
            parameters, geo, spl = {}, [], []
            for k, (bb, s) in enumerate(all_segs):
                if bb != b:
                    continue
                dv = 0.1 if d_h[k] > 1 else d_h[k]
                if s["kind"] == "prim":
                    code, p = PRIM_CODE[s["type"]], pf[k]
                    if code == K.PRIM_PLANE:
                        parameters[s["key"]] = ["plane", p[0:3].reshape(3, 1), p[3]]
                    elif code == K.PRIM_SPHERE:
                        parameters[s["key"]] = ["sphere", p[0:3].reshape(1, 3), p[3]]
                    elif code == K.PRIM_CYLINDER:
                        parameters[s["key"]] = ["cylinder", p[0:3].reshape(3, 1), p[3:6].reshape(1, 3), p[6]]
                    else:
                        parameters[s["key"]] = ["cone", p[0:3].reshape(1, 3), p[3:6].reshape(3, 1), p[6:7]]
                    geo.append(float(dv))
                else:
                    parameters[s["key"]] = ["open-spline" if s["kind"] == "open" else "closed-spline", recs[k - S_p]]
                    spl.append(float(dv))
            fitted = {s["key"] for bb, s in all_segs if bb == b}
            for i in matches[b][2]:              # skipped segments are recorded as None like the reference
                if int(i) not in fitted and matches[b][3][matches[b][1][i]] > 0:
                    parameters[int(i)] = None
            nseg = sum(1 for bb, _ in all_segs if bb == b)
            Loss = loss_b[b] if nseg else torch.zeros(1, device=dev)
            s_iou, p_iou, _, _ = siou_matched_segments_fast(matches[b], ptype_h[b], primitives[b])
            ev.stats["shapes"] += 1
            ev.stats["clusters"] += ncl_list[b]
            ev.stats["fitted"] += nseg
            out.append(([Loss, float(np.mean(geo)) if geo else None, float(np.mean(spl)) if spl else None, s_iou,
                         p_iou], [parameters, cluster_ids[b], Wraw[b, :ncl_list[b]]]))
        return out
    if defer_metrics:
        return loss_b, finish
    return finish()
