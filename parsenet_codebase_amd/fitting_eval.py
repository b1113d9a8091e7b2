"""The EVALUATION-mode fitting stage, stage by stage over all shapes and segments of a batch.

What it computes is ``Evaluation.fitting_loss(eval=True)`` of the reference —
src/residual_utils.py:210-331 (residual_eval_mode: hard memberships, the modal PREDICTED primitive type
of every predicted segment, fits on the segment's own points, sqrt residuals), src/primitive_forward.py:
925-1047 (fit_one_shape_torch with eval=True), src/fitting_utils.py:704-710 (remove_outliers: open3d's
statistical outlier removal), :202-219 (up_sample_points_in_range: kNN-centroid up-sampling into the range
the SplineNets were trained on), src/primitive_forward.py:153-296 (the optional LS refit) — organised like
the training stage of fitting_batch.py instead of one segment after the other:

  clustering   bandwidths, mean-shift iterations, NMS and labels of all shapes as batched launches, ONE
               download of the cluster ids (fitting_batch's kernels; no autograd graph);
  host         Hungarian matching of every shape, the modal predicted type and the member lists of its
               segments, ONE packed upload;
  primitives   the four launches of csrc/fitbatch.hip for every analytic segment of the batch, with HARD
               one-hot membership rows (weight 1 + eps on the members, exactly 0 elsewhere, eps = 0 in the
               kernel: the moments of the members alone) and sqrt residuals;
  splines      outlier statistic of ALL spline segments from one ragged float64 neighbour search
               (csrc/knn3.hip, k = 20), ONE download of the surviving counts (they size numpy's draws);
               the up-sampling rounds as ragged float32 searches (k = 5) over the segments that still need
               one; ONE SplineNet forward per net over its re-sampled segments (all of one size), B-spline
               evaluation, the optional refit with its solves batched, ONE ragged Chamfer call.

numpy's generator is consumed in the reference's order — per shape: the shuffle of its mean-shift call(s),
then, segment by segment, the re-sampling draw and the draws of the refit — because every draw depends on
counts only, which the host has after the two downloads.  The per-segment functions of fitting.py
(``Evaluation.batched = False``) compute the same numbers one segment at a time; tests hold the two equal."""
import os

import numpy as np
import torch
from torch.profiler import record_function

from . import _lsa_worker
from . import kernels as K
from . import mean_shift as MSM
from ._lib import h2d
from .fitting_batch import (CMAX, EPS, PRIM_CODE, _BSplineEval, _fitter_bases, bandwidth_batch, nms_batch,
                            standardize_segments)

_CLOSED_TYPES, _OPEN_TYPES = (0, 9, 6, 7), (2, 8)
_RESAMPLE = {"closed": (1400, 1800), "open": (1000, 1500)}      # src/primitive_forward.py:989-996, 1030-1036
_REFIT = {"open": dict(size_u=20, size_v=20, up=(1600, 2000), sub=1600, degree=2, bgrid=20),   # :229-296
          "closed": dict(size_u=21, size_v=20, up=(2000, 2100), sub=None, degree=3, bgrid=30)}  # :153-226


# ---------------------------------------------------------------------------------------------
# clustering of all shapes (no gradient)
# ---------------------------------------------------------------------------------------------
def cluster_shapes(ev, emb, quantile, iterations):
    """guard_mean_shift of every shape (src/residual_utils.py:69-84) -> (clusters, calls): clusters[b] =
    (centres (C,128), bandwidth (0-dim tensor), cluster ids (N,) int64 numpy), calls[b] = the number of
    mean-shift calls the guard needed for shape b.  The fast path is fitting_batch's: one batched bandwidth
    selection, the iterations of all shapes per launch, NMS and labels on the device, one download; a shape
    with flagged selection rows or more than 49 clusters goes through the per-shape API (the guard's retry
    with quantile x 1.2).

    numpy's generator is left where it was: the reference shuffles once per mean-shift call
    (src/mean_shift.py:121-122; every row is a sample, so the shuffle does not change the result) and the
    caller replays calls[b] shuffles at shape b's place in the stream."""
    B, N, D = emb.shape
    dev = emb.device
    fast = None
    with torch.no_grad():
        bwres = bandwidth_batch(emb, quantile) if D == 128 else None
        if bwres is not None:
            bw, bwflag = bwres
            MSM.WANT_NEAREST = True
            try:
                new_X, _ = MSM.mean_shift_iterations_state(emb, bw, iterations)
            finally:
                MSM.WANT_NEAREST = False
            nearest, MSM.LAST_NEAREST = MSM.LAST_NEAREST, None
            st = nms_batch(new_X, emb, bw, None, labels=True, nearest=nearest)
            if st is not None:
                pack = torch.cat([st["labels"].reshape(-1), st["cid"].reshape(-1), st["ncl"], bwflag,
                                  st["nflag"]]).to(torch.int32).cpu().numpy()              # download 1
                fast = (new_X, bw, pack[:B * N].reshape(B, N), pack[B * N:B * N + B * CMAX].reshape(B, CMAX),
                        pack[-3 * B:-2 * B], pack[-2 * B:-B], pack[-B:])
    clusters, calls = [], []
    state = np.random.get_state()
    for b in range(B):
        ok = fast is not None and fast[6][b] == 0 and fast[5][b] == 0 and fast[4][b] <= CMAX
        if ok and fast[4][b] <= 49:
            new_X, bw, lab_h, cid_h, ncl_h = fast[:5]
            cid = h2d(cid_h[b, :int(ncl_h[b])].astype(np.int64), dev)
            clusters.append((new_X[b][cid], bw[b], lab_h[b].astype(np.int64)))
            calls.append(1)
            continue
        # the guard's loop on the per-shape API; a fast attempt that found more than 49 clusters was its first call
        q, n = (quantile * 1.2, 1) if ok else (quantile, 0)
        while True:
            _, center, bandwidth, ids = ev.ms.mean_shift(emb[b], 10000, q, iterations, kernel_type="gaussian")
            n += 1
            if center.shape[0] > 49 and torch.unique(ids).shape[0] > 49:
                q *= 1.2
            else:
                break
        clusters.append((center, bandwidth.reshape(()), ids.cpu().numpy().astype(np.int64)))
        calls.append(n)
    np.random.set_state(state)
    return clusters, calls


# ---------------------------------------------------------------------------------------------
# host: matching and the segment list of one shape
# ---------------------------------------------------------------------------------------------
def eval_segments(labels, cluster_ids, pred_primitives):
    """residual_eval_mode's loop (src/residual_utils.py:233-262) + the dispatch rules of
    fit_one_shape_torch(eval=True): every matched predicted segment with its member indices, the indices of
    its ground-truth segment, the modal predicted type and the value of its hard membership column.
    Returns (segments in the reference's order, match (rows, cols, unique_target, unique_pred))."""
    from .fitting import match
    labels = np.asarray(labels)
    pred_primitives = np.asarray(pred_primitives)
    rows, cols, unique_target, unique_pred = match(labels, cluster_ids)
    width = np.unique(cluster_ids).shape[0]                 # columns of the one-hot membership matrix
    segs = []
    for index, i in enumerate(unique_pred):
        gt_idx = np.flatnonzero(labels == cols[index])
        pred_idx = np.flatnonzero(cluster_ids == i)
        if gt_idx.size == 0 or pred_idx.size == 0:
            continue
        seg_type = int(np.bincount(pred_primitives[pred_idx].astype(np.int64)).argmax())
        kind = "closed" if seg_type in _CLOSED_TYPES else "open" if seg_type in _OPEN_TYPES else "prim"
        if kind == "prim" and seg_type not in PRIM_CODE:
            raise ValueError("unknown primitive type %r" % (seg_type,))
        # the segment reads column ``index`` of the one-hot of the cluster ids (src/primitive_forward.py:940):
        # 1 where the id equals that column
        wv = (1.0 if (int(i) == index and index < width) else 0.0) + EPS
        segs.append({"index": index, "key": int(i), "type": seg_type, "kind": kind, "pred": pred_idx, "gt": gt_idx,
                     "wv": np.float32(wv), "fit": pred_idx.size >= (20 if kind == "prim" else 100)})
    return segs, (rows, cols, unique_target, unique_pred)


# ---------------------------------------------------------------------------------------------
# ragged helpers
# ---------------------------------------------------------------------------------------------
def _ragged(counts, dev):
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return off, h2d(off, dev)


def outlier_keep_mask(pts, off_h, off_d):
    """remove_outliers (src/fitting_utils.py:704-710 -> open3d 0.9 remove_statistical_outlier(20, 0.5)) for a
    ragged batch of segments: pts (total,3) fp32, offsets on host and device.  Mean distance of every point to
    its 20 nearest neighbours (itself included) in float64, threshold = mean + 0.5 std (Bessel) over the
    segment, keep 0 < mean distance < threshold.  Returns the keep mask (total,) bool; the statistics of a
    segment are sums over its padded row in index order (no atomics)."""
    S = off_h.shape[0] - 1
    counts = np.diff(off_h)
    nmax = int(counts.max())
    k = int(min(20, counts.min()))
    if counts.min() < 20:       # a segment shorter than the neighbourhood: the search returns what there is (n)
        raise ValueError("outlier removal of a segment with fewer than 20 points")
    if nmax <= KNN3_MAX_POINTS:
        _, dist = K.knn3_ragged(pts, off_d, nmax, k, f64=True, want_dist=True)
    else:
        # a segment beyond the kernel's size (shapes of more than 10 240 points): float64 differences block-wise
        # and topk, segment by segment — what fitting._knn_points_by_differences and the reference fall back on
        dist = torch.cat([_mean_dist_broadcast(pts[off_h[s]:off_h[s + 1]].double(), k) for s in range(S)], 0)
    avg = dist.mean(1)                                                         # (total,) fp64
    dev = pts.device
    seg = torch.repeat_interleave(torch.arange(S, device=dev), h2d(counts.astype(np.int64), dev),
                                  output_size=int(off_h[-1]))
    pos = torch.arange(int(off_h[-1]), device=dev) - off_d[:-1].long()[seg]
    pad = torch.zeros((S, nmax), dtype=torch.float64, device=dev)
    valid = avg > 0
    n_t = h2d(counts.astype(np.float64), dev)
    pad[seg, pos] = torch.where(valid, avg, torch.zeros_like(avg))
    cloud_mean = pad.sum(1) / n_t
    dev2 = torch.zeros((S, nmax), dtype=torch.float64, device=dev)
    dev2[seg, pos] = torch.where(valid, (avg - cloud_mean[seg]) ** 2, torch.zeros_like(avg))
    std = torch.sqrt(dev2.sum(1) / torch.clamp(n_t - 1, min=1))
    std = torch.where(n_t > 1, std, torch.zeros_like(std))
    return valid & (avg < (cloud_mean + 0.5 * std)[seg])


KNN3_MAX_POINTS = 10240          # csrc/knn3.hip: 160 values per lane


def _mean_dist_broadcast(p, k):
    """(n,k) float64 distances of every point to its k nearest (itself included), 2 048 queries at a time."""
    out = []
    for s0 in range(0, p.shape[0], 2048):
        q = p[s0:s0 + 2048]
        dx, dy, dz = (q[:, None, c] - p[None, :, c] for c in range(3))
        d = (dx * dx + dy * dy) + dz * dz
        out.append(torch.sqrt(torch.topk(d, k, 1, largest=False)[0]))
    return torch.cat(out, 0)


def upsample_rounds(pts, counts, rounds):
    """up_sample_points_torch (src/fitting_utils.py:150-164) ``rounds[s]`` times on segment s of a ragged batch:
    each round appends the centroid of the 4 nearest neighbours of every point (csrc/knn3.hip, k = 5, fp32
    differences).  pts (total,3), counts / rounds host arrays.  Returns (pts, counts) after all rounds; a
    round runs over the segments that still need one."""
    dev = pts.device
    counts = np.asarray(counts, dtype=np.int64).copy()
    rounds = np.asarray(rounds, dtype=np.int64).copy()
    while rounds.max(initial=0) > 0:
        S = counts.shape[0]
        off = np.concatenate([[0], np.cumsum(counts)])
        act = np.flatnonzero(rounds > 0)
        # the active segments as a ragged batch of their own
        a_counts = counts[act]
        a_off_h, a_off_d = _ragged(a_counts, dev)
        src = np.concatenate([np.arange(off[s], off[s + 1]) for s in act])
        src_d = h2d(src, dev)
        sub = pts[src_d]
        idx = K.knn3_ragged(sub, a_off_d, int(a_counts.max()), 5)                 # local indices, self first
        seg = torch.repeat_interleave(torch.arange(act.shape[0], device=dev), h2d(a_counts, dev),
                                      output_size=int(a_off_h[-1]))
        glob = idx[:, 1:].long() + a_off_d[:-1].long()[seg].unsqueeze(1)
        centers = torch.mean(sub[glob], 1)
        # new layout: every active segment doubles to [points, centres]
        new_counts = counts.copy()
        new_counts[act] *= 2
        new_off = np.concatenate([[0], np.cumsum(new_counts)])
        dst_old = np.concatenate([np.arange(new_off[s], new_off[s] + counts[s]) for s in range(S)])
        dst_new = np.concatenate([np.arange(new_off[s] + counts[s], new_off[s + 1]) for s in act])
        out = torch.empty((int(new_off[-1]), 3), dtype=pts.dtype, device=dev)
        out[h2d(dst_old, dev)] = pts
        out[h2d(dst_new, dev)] = centers
        pts, counts = out, new_counts
        rounds[act] -= 1
    return pts, counts


def _rounds_to_reach(n, a_max):
    """up_sample_points(_torch)_in_range: at least one doubling, until n >= a_max."""
    if n <= 0:
        raise ValueError("up-sampling of an empty segment")
    r = 1
    while n * (1 << r) < a_max:
        r += 1
    return r


# ---------------------------------------------------------------------------------------------
# the stage
# ---------------------------------------------------------------------------------------------
def fitting_losses_eval(ev, embedding, points, normals, labels, primitives, primitives_log_prob, quantile, iterations,
                        lamb, if_optimize=False):
    """Evaluation-mode Evaluation.fitting_loss for every shape of the batch: a list (one entry per shape) of
    ([Loss, geometric mean, spline mean, s_iou, p_iou], [parameters, cluster ids, weights]) as the reference's
    call with that single shape returns them."""
    from .fitting import SIOU_matched_segments, to_one_hot
    B, N, D = embedding.shape
    dev = embedding.device
    labels, primitives = np.asarray(labels), np.asarray(primitives)
    points, normals = points.contiguous().float(), normals.contiguous().float()
    fitter = ev.fitter
    with torch.no_grad():
        emb = torch.nn.functional.normalize(embedding.detach(), p=2, dim=2)
        prim_pred = torch.max(primitives_log_prob, 1)[1].data.cpu().numpy()
        with record_function("eval:clustering"):
            clusters, ms_calls = cluster_shapes(ev, emb, quantile, iterations)
        seglists = [eval_segments(labels[b], clusters[b][2], prim_pred[b])[0] for b in range(B)]

        prim_segs = [(b, s) for b in range(B) for s in seglists[b] if s["kind"] == "prim" and s["fit"]]
        spl_segs = [(b, s) for b in range(B) for s in seglists[b] if s["kind"] != "prim" and s["fit"]]
        ids_dev = h2d(np.stack([c[2] for c in clusters]).astype(np.int64), dev)           # (B,N)

        # ---- analytic primitives: hard membership rows, the members' moments --------------------------
        params_h = status_h = dist_p = None
        if prim_segs:
            Cp = max(sum(1 for bb, _ in prim_segs if bb == b) for b in range(B))
            row_of = {}
            seg_id = np.full((B, Cp), -1, np.int64)
            seg_wv = np.zeros((B, Cp), np.float32)
            nxt = [0] * B
            for b, s in prim_segs:
                row_of[(b, s["key"])] = nxt[b]
                seg_id[b, nxt[b]] = s["key"]
                seg_wv[b, nxt[b]] = s["wv"]
                nxt[b] += 1
            W = (ids_dev.unsqueeze(1) == h2d(seg_id, dev).unsqueeze(2)).float() * h2d(seg_wv, dev).unsqueeze(2)
            gt_lists = [s["gt"] for _, s in prim_segs]
            tab = {"shape": h2d(np.asarray([b for b, _ in prim_segs], np.int32), dev),
                   "row": h2d(np.asarray([row_of[(b, s["key"])] for b, s in prim_segs], np.int32), dev),
                   "type": h2d(np.asarray([PRIM_CODE[s["type"]] for _, s in prim_segs], np.int32), dev),
                   "rows": h2d(np.asarray([s["pred"].size for _, s in prim_segs], np.int32), dev),
                   "gt_off": h2d(np.concatenate([[0], np.cumsum([g.size for g in gt_lists])]).astype(np.int32), dev),
                   "gt_idx": h2d(np.concatenate(gt_lists).astype(np.int32), dev)}
            with record_function("eval:primitives"):
                partial = K.weighted_moments(points, normals, W, tab["shape"], tab["row"], 1, 0.0)
                params, jac, status = K.primitive_fit(partial, tab["type"], tab["rows"])
                K.cone_angle(points, W, tab["shape"], tab["row"], tab["type"], status, params, jac, 1, 0.0)
                dist_p, _ = K.primitive_residual(points, tab["shape"], tab["type"], tab["gt_off"], tab["gt_idx"],
                                                 params, status, True)

        # ---- splines -----------------------------------------------------------------------------------
        recs = {}
        dist_s = None
        if spl_segs:
            flat_pts = points.reshape(B * N, 3)
            members = np.concatenate([s["pred"] + b * N for b, s in spl_segs])
            counts0 = np.asarray([s["pred"].size for _, s in spl_segs], np.int64)
            seg_pts = flat_pts[h2d(members, dev)]
            off_h, off_d = _ragged(counts0, dev)
            with record_function("eval:outliers"):
                keep = outlier_keep_mask(seg_pts, off_h, off_d)
                csum = torch.cumsum(keep.long(), 0)
                ends = off_d[1:].long() - 1
                tot = csum[ends]
                kept = torch.cat([tot[:1], tot[1:] - tot[:-1]])
                kept_h = kept.cpu().numpy().astype(np.int64)                              # download 2
                if (kept_h < 5).any():
                    # every mean distance equal (std 0: nothing is below the mean) or coincident points: the
                    # reference's up-sampling raises in topk on such a cloud (src/fitting_utils.py:155-158) and
                    # its caller skips the shape — never pad a segment with self-neighbours
                    j = int(np.flatnonzero(kept_h < 5)[0])
                    raise RuntimeError("fitting: outlier removal left %d point(s) of spline segment %d of shape %d "
                                       "(up-sampling needs 5)" % (kept_h[j], spl_segs[j][1]["key"], spl_segs[j][0]))
                seg_pts = seg_pts[keep]
            # numpy's draws in the reference's order: per shape the shuffle(s) of its mean-shift call(s), then,
            # segment by segment, the re-sampling draw and the draws of the refit (every draw depends on counts
            # alone, which the host now has)
            draws = {}
            for b in range(B):
                for _ in range(ms_calls[b]):
                    np.random.shuffle(np.arange(N))
                for j, (bb, s) in enumerate(spl_segs):           # (spl_segs keeps the segments of a shape in order)
                    if bb != b:
                        continue
                    a_max = _RESAMPLE[s["kind"]][1]
                    n = int(kept_h[j])
                    d = {}
                    if n > a_max:
                        d["rounds"] = 0
                        d["L"] = np.random.choice(np.arange(n), a_max, replace=False)
                    else:
                        d["rounds"] = _rounds_to_reach(n, a_max)
                        d["L"] = np.random.choice(np.arange(n << d["rounds"]), a_max, replace=False)
                    if if_optimize and (s["kind"] == "open" or s["pred"].size > 200):
                        d["refit"] = _refit_draws(s["kind"], a_max)
                    draws[j] = d
            with record_function("eval:upsample"):
                up_pts, up_counts = upsample_rounds(seg_pts, kept_h, [draws[j]["rounds"] for j in range(len(spl_segs))])
            up_off = np.concatenate([[0], np.cumsum(up_counts)])
            nu, nv = _fitter_bases(fitter, dev)
            groups = {"open": [j for j, (_, s) in enumerate(spl_segs) if s["kind"] == "open"],
                      "closed": [j for j, (_, s) in enumerate(spl_segs) if s["kind"] == "closed"]}
            pending_groups = []
            for kind, js in groups.items():
                if not js:
                    continue
                a_max = _RESAMPLE[kind][1]
                sel = np.concatenate([up_off[j] + draws[j]["L"] for j in js])
                P = up_pts[h2d(sel, dev)].reshape(len(js), a_max, 3)
                w = h2d(np.asarray([spl_segs[j][1]["wv"] for j in js], np.float32), dev).reshape(-1, 1).expand(-1, a_max)
                w = w.contiguous()
                with record_function("eval:splinenet"):
                    pts_std, std, mean, R = standardize_segments(P, w)
                    affine = torch.cat([torch.linalg.inv(R) * std.unsqueeze(1), mean.unsqueeze(2)], 2).contiguous()
                    net = fitter.open_control_decoder if kind == "open" else fitter.closed_control_decoder
                    ctrl = net(K.transpose12(pts_std), w).reshape(len(js), 20, 20, 3)
                    rec = _BSplineEval.apply(ctrl, nu, nv, affine, kind == "closed")     # (S,900|930,3)
                # if_optimize: the refits of BOTH kinds are submitted (their matchings run on the assignment pool) before
                # either is finished
                pending_groups.append((js, _refit_submit(kind, js, spl_segs, draws, P, ctrl, affine, rec)
                                       if if_optimize else (lambda rec=rec: rec)))
            for js, fin in pending_groups:
                rec = fin()
                for t, j in enumerate(js):
                    recs[j] = rec[t:t + 1]
            # two-sided Chamfer with guard_sqrt on every nearest-neighbour distance (src/utils.py:326-358)
            order = list(range(len(spl_segs)))
            pred = torch.cat([recs[j].reshape(-1, 3) for j in order], 0)
            na = [int(recs[j].shape[1]) for j in order]
            nb = [spl_segs[j][1]["gt"].size for j in order]
            gt_cloud = flat_pts[h2d(np.concatenate([spl_segs[j][1]["gt"] + spl_segs[j][0] * N for j in order]), dev)]
            off_a = h2d(np.concatenate([[0], np.cumsum(na)]).astype(np.int32), dev)
            off_b = h2d(np.concatenate([[0], np.cumsum(nb)]).astype(np.int32), dev)
            with record_function("eval:chamfer"):
                minA, _, minB, _ = K.chamfer_nn_ragged(pred, off_a, max(na), gt_cloud, off_b, max(nb))
                gs = lambda x: torch.sqrt(torch.clamp(x, min=1e-5))                       # noqa: E731
                dist_s = K.chamfer_ragged_reduce(gs(minA), off_a, gs(minB), off_b)

        else:
            for b in range(B):
                for _ in range(ms_calls[b]):
                    np.random.shuffle(np.arange(N))

        # ---- ONE download of the distances (and the fit status), then the per-shape records -------------
        tail = []
        if prim_segs:
            tail += [dist_p.double(), status.double(), params.reshape(-1)]
        if spl_segs:
            tail.append(dist_s.double())
        host = torch.cat(tail).cpu().numpy() if tail else np.zeros(0)                      # download 3
        o = 0
        Sp, Ss = len(prim_segs), len(spl_segs)
        if prim_segs:
            dp_h, st_h = host[:Sp], host[Sp:2 * Sp].astype(np.int64)
            pf_h = host[2 * Sp:2 * Sp + Sp * params.shape[1]].reshape(Sp, -1)
            o = 2 * Sp + pf_h.size
            if (st_h & 5).any():
                bad = int(np.nonzero(st_h & 5)[0][0])
                raise RuntimeError("fitting: %s in segment %d of shape %d" % (
                    "non-finite design matrix / no full-rank ridge system (lstsq)" if st_h[bad] & 1 else
                    "NaN residual distance", prim_segs[bad][1]["key"], prim_segs[bad][0]))
            pf = params.float()
        ds_h = host[o:o + Ss]
        out = []
        for b in range(B):
            parameters, geo, spl, loss_terms = {}, [], [], []
            fitted = {}
            for k, (bb, s) in enumerate(prim_segs):
                if bb == b:
                    fitted[s["key"]] = ("prim", k)
            for j, (bb, s) in enumerate(spl_segs):
                if bb == b:
                    fitted[s["key"]] = ("spline", j)
            for s in seglists[b]:
                if s["key"] not in fitted:
                    parameters[s["key"]] = None
                    continue
                what, k = fitted[s["key"]]
                if what == "prim":
                    code, p = PRIM_CODE[s["type"]], pf[k]
                    if code == K.PRIM_PLANE:
                        parameters[s["key"]] = ["plane", p[0:3].reshape(3, 1), p[3]]
                    elif code == K.PRIM_SPHERE:
                        parameters[s["key"]] = ["sphere", p[0:3].reshape(1, 3), p[3]]
                    elif code == K.PRIM_CYLINDER:
                        parameters[s["key"]] = ["cylinder", p[0:3].reshape(3, 1), p[3:6].reshape(1, 3), p[6]]
                    else:
                        parameters[s["key"]] = ["cone", p[0:3].reshape(1, 3), p[3:6].reshape(3, 1), p[6:7]]
                    d = float(dp_h[k])
                else:
                    parameters[s["key"]] = ["open-spline" if s["kind"] == "open" else "closed-spline", recs[k]]
                    d = float(ds_h[k])
                if not np.isfinite(d):
                    raise RuntimeError("fitting: non-finite residual distance in segment %d of shape %d" % (s["key"], b))
            # separate_losses (src/residual_utils.py:333-378): in ascending key order
            for key in sorted(k_ for k_, v in parameters.items() if v is not None):
                what, k = fitted[key]
                d = float(dp_h[k]) if what == "prim" else float(ds_h[k])
                dt = dist_p[k] if what == "prim" else dist_s[k]
                if d > 1:                               # most probably a degenerate case
                    d, dt = 0.1, torch.ones((), device=dev) * 0.1
                if what == "prim":
                    geo.append(float(d))
                    loss_terms.append(dt)
                else:
                    spl.append(float(d))
                    loss_terms.append(dt * lamb)
            Loss = torch.mean(torch.stack(loss_terms)) if loss_terms else torch.zeros(1, device=dev)
            ids_b = clusters[b][2]
            weights = to_one_hot(ids_b, np.unique(ids_b).shape[0], device_id=dev.index).T
            s_iou, p_iou, _, _ = SIOU_matched_segments(labels[b], ids_b, prim_pred[b], primitives[b], weights.T)
            out.append(([Loss, np.mean(geo) if geo else None, np.mean(spl) if spl else None, s_iou, p_iou],
                        [parameters, ids_b, weights]))
    return out


def _refit_draws(kind, a_max):
    """numpy's draws of one optimize_*_spline_kronecker call (src/primitive_forward.py:153-296), in its order:
    the uniform (u, v) parameters, the re-sampling of the up-sampled input points, the sub-sample."""
    from .fitting import boundary_parameterization
    cfg = _REFIT[kind]
    nbound = boundary_parameterization(cfg["bgrid"]).shape[0]
    d = {"uv": np.random.random((1600 - nbound, 2))}
    lo, hi = cfg["up"]
    n = a_max
    if n > hi:
        d["rounds"] = 0
        d["L"] = np.random.choice(np.arange(n), hi, replace=False)
    else:
        d["rounds"] = _rounds_to_reach(n, hi)
        d["L"] = np.random.choice(np.arange(n << d["rounds"]), hi, replace=False)
    if cfg["sub"] is not None:
        d["sub"] = np.random.choice(np.arange(hi), cfg["sub"], replace=False)
    return d


# ---------------------------------------------------------------------------------------------
# assignment pool: the Hungarian matchings of a batch's refits side by side
# ---------------------------------------------------------------------------------------------
_LSA_POOL = None
REFIT_POOL = os.environ.get("PARSENET_REFIT_POOL", "1") != "0"


def assignment_pool():
    """A persistent pool of worker PROCESSES (spawned: they import numpy / scipy only, never torch) for the
    linear_sum_assignment calls of the LS refit — scipy holds the GIL, threads would run them one after the other.
    Sized by the CPUs the job may use (dp.usable_cpus() - 1, at most 16).  None when PARSENET_REFIT_POOL=0 or a single CPU:
    the caller then solves in place."""
    global _LSA_POOL
    if not REFIT_POOL:
        return None
    if _LSA_POOL is None:
        import atexit
        import multiprocessing
        from concurrent.futures import ProcessPoolExecutor
        from .dp import usable_cpus
        n = min(16, usable_cpus() - 1)      # (a batch of 4 shapes has up to 16 spline segments: one matching per worker)
        if n < 2:
            return None
        _LSA_POOL = ProcessPoolExecutor(max_workers=n, mp_context=multiprocessing.get_context("spawn"))
        atexit.register(_LSA_POOL.shutdown, wait=False, cancel_futures=True)
    return _LSA_POOL


def _refit_batch(kind, js, spl_segs, draws, P, ctrl, affine, rec):
    """The LS refit of one kind in one go (submit, then finish)."""
    return _refit_submit(kind, js, spl_segs, draws, P, ctrl, affine, rec)()


def _refit_submit(kind, js, spl_segs, draws, P, ctrl, affine, rec):
    """The LS refit (src/primitive_forward.py:153-296) of the segments ``js`` of one kind: samples of the
    predicted surface at the drawn parameters, the input points up-sampled (ragged, together) and re-sampled,
    the Hungarian matching per segment on the host (scipy), the 100 x 100 normal equations of all segments
    solved together, the refitted surfaces sampled on the regular grid.  Segments the reference does not refit
    (closed, 200 members or fewer) keep their network prediction.

    Two phases (round 6): this function queues everything up to the distance matrices and hands them to the
    assignment pool; the returned ``finish()`` collects the matchings and runs the solves.  The caller submits BOTH
    kinds of a batch before it finishes either, so all matchings of the batch run side by side on the host's cores
    (14.1 s -> see profiles/r06_named_kernels_kbench.txt per batch of 4 shapes, all of it scipy's assignment)."""
    from .approximation import fit_bezier_surface_fit_kronecker
    from .bspline import basis_matrix, uniform_knots
    from .fitting import boundary_parameterization, regular_parameterization, solve_dense
    cfg = _REFIT[kind]
    dev = P.device
    todo = [t for t, j in enumerate(js) if "refit" in draws[j]]
    if not todo:
        return lambda: rec
    S = len(todo)
    a_max = P.shape[1]
    su, sv = cfg["size_u"], cfg["size_v"]
    # control grid in the input frame; closed: the first row appended again (21 x 20)
    grid = ctrl[todo]
    if kind == "closed":
        grid = torch.cat([grid, grid[:, 0:1]], 1)
    aff = affine[todo]
    grid = grid.reshape(S, su * sv, 3) @ aff[:, :, :3].transpose(1, 2) + aff[:, :, 3].unsqueeze(1)
    bound = boundary_parameterization(cfg["bgrid"])
    ku, kv = uniform_knots(su, 3), uniform_knots(sv, 3)
    ku2 = uniform_knots(10, cfg["degree"])
    reg = regular_parameterization(30, 30)
    RU = torch.from_numpy(basis_matrix(reg[:, 0], 10, cfg["degree"], ku2)).to(dev)
    RV = torch.from_numpy(basis_matrix(reg[:, 1], 10, cfg["degree"], ku2)).to(dev)
    # input points of all segments up-sampled together
    rounds = [draws[js[t]]["refit"]["rounds"] for t in todo]
    up, cnt = upsample_rounds(P[todo].reshape(S * a_max, 3), [a_max] * S, rounds)
    off = np.concatenate([[0], np.cumsum(cnt)])
    out = rec.clone()
    pool = assignment_pool()
    pending = []
    for q, t in enumerate(todo):
        d = draws[js[t]]["refit"]
        parameters = np.concatenate([d["uv"], bound], 0)
        bu = torch.from_numpy(basis_matrix(parameters[:, 0], su, 3, ku)).to(dev)
        bv = torch.from_numpy(basis_matrix(parameters[:, 1], sv, 3, kv)).to(dev)
        samples = torch.einsum("ni,nj,ijc->nc", bu, bv, grid[q].reshape(su, sv, 3).double())      # (1600,3) fp64
        inp = up[h2d(off[q] + d["L"], dev)]
        if "sub" in d:
            inp = inp[h2d(d["sub"], dev)]
        inp = inp.double()
        dist = torch.cdist(samples, inp, compute_mode="donot_use_mm_for_euclid_dist")
        cost = dist.cpu().numpy()
        job = pool.submit(_lsa_worker.solve, cost) if pool is not None else None
        pending.append((t, parameters, inp, cost, job))

    def finish():
        for t, parameters, inp, cost, job in pending:
            cids = job.result() if job is not None else solve_dense(cost)[1]
            matched = inp[h2d(np.asarray(cids), dev)]
            NU = torch.from_numpy(basis_matrix(parameters[:, 0], 10, cfg["degree"], ku2)).to(dev)
            NV = torch.from_numpy(basis_matrix(parameters[:, 1], 10, cfg["degree"], ku2)).to(dev)
            new_ctrl = fit_bezier_surface_fit_kronecker(matched, NU, NV)
            pts = torch.einsum("ni,nj,ijc->nc", RU, RV, new_ctrl).float()
            if kind == "closed":
                pts = pts.reshape(30, 30, 3)
                pts = torch.cat([pts, pts[0:1]], 0).reshape(930, 3)
            out[t] = pts
        return out
    return finish
