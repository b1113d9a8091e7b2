"""kNN graph, edge features and the fused edge convolution, with autograd.

Public names mirror the reference (src/model.py:9-53, src/PointNet.py:9-140); everything
numerically heavy is a HIP kernel reached through ``kernels`` (C ABI).  Only tiny
per-channel reductions and the point-level GEMM are left to torch (rocBLAS).
"""
import contextlib
import os
import threading
import weakref

import numpy as np
import torch

from . import kernels as K

# Folded constants of FROZEN layers (the SplineNets of the fitting stage: W -> [Wa | Wb - Wa],
# rsqrt(running_var + eps), the affine map of an evaluation-mode BatchNorm1d), keyed by the layer
# object and validated by (data_ptr, _version) of the tensors they were built from.  They live HERE,
# not in the module's __dict__: torch.save(module) / copy.deepcopy do not carry them along.  An
# in-place edit through ``.data`` (``p.data.copy_(...)``) does not bump ``_version`` — call
# invalidate_frozen_caches(module) after such an edit; load_state_dict / optimizer steps / ``copy_``
# on the parameter itself are seen without help.
_FROZEN = weakref.WeakKeyDictionary()


def frozen_cache(layer):
    """The cache dictionary of one layer object (created on first use)."""
    d = _FROZEN.get(layer)
    if d is None:
        d = _FROZEN[layer] = {}
    return d


def invalidate_frozen_caches(module=None):
    """Forget the folded constants of ``module`` and its sub-modules (all layers if None)."""
    from . import encoders            # (the split images of frozen weights, encoders._weight_image)
    if module is None:
        _FROZEN.clear()
        encoders._W_IMAGES.clear()
        return
    for m in module.modules():
        _FROZEN.pop(m, None)
    for prm in module.parameters():
        encoders._W_IMAGES.pop(id(prm), None)


# --------------------------------------------------------------------------------------
# kNN — no gradient flows through the indices (the reference wraps them in no_grad)
# --------------------------------------------------------------------------------------
def _as_bcn(x):
    if x.dim() != 3:
        raise ValueError("expected a (B,C,N) tensor, got %s" % (tuple(x.shape),))
    return x.detach()


# The public functions return the reference's int64 indices (torch.topk's dtype).  Between the layers of the
# encoders the graph never leaves the library: inside ``library_graphs()`` the same functions return the same
# indices as int32, the form the edge-conv kernels read with half the bytes (gather and transposed-graph build).
_LOCAL = threading.local()


@contextlib.contextmanager
def library_graphs():
    prev = getattr(_LOCAL, "int32", False)
    _LOCAL.int32 = True
    try:
        yield
    finally:
        _LOCAL.int32 = prev


def _int32():
    return getattr(_LOCAL, "int32", False)


# TEST HOOK (None in production): a callable (x (B,C,N), k, metric) -> (B,N,k) indices or None.  Parity tests that
# compare whole networks against the CPU oracle pin BOTH sides to one graph per layer: the oracle's and the
# product's features of a layer differ by fp32 rounding, and a feature-space kNN graph has near-ties that such a
# difference flips (one flipped neighbour moves a SplineNet output by percents) — with the hook the product takes
# the graph the oracle used, so that everything BEHIND the graphs is compared at the arithmetic's own accuracy.
# The graph kernels themselves are pinned bit for bit elsewhere (tests/test_knn_gpu.py, test_fullsize_gpu.py).
GRAPH_HOOK = None


def _graph(x, k, metric):
    x = _as_bcn(x)
    if GRAPH_HOOK is not None:
        idx = GRAPH_HOOK(x, int(k), metric)
        if idx is not None:
            return idx.to(device=x.device, dtype=torch.int32 if _int32() else torch.int64).contiguous()
    return K.knn(x, int(k), metric, int32=_int32())


def _dilate(idx, k1, k2):
    if k1 != k2:
        cols = torch.as_tensor(np.arange(0, k2, k2 // k1), device=idx.device)
        idx = idx.index_select(2, cols).contiguous()
    return idx


def knn(x, k):
    """src/model.py:9-22.  x (B,C,N) -> idx (B,N,k) int64, nearest first, self included."""
    with torch.no_grad():
        return _graph(x, k, "feature")


def knn_dilated(x, k1, k2):
    """src/PointNet.py:9-26: top-k2, keeping columns arange(0, k2, k2 // k1)."""
    with torch.no_grad():
        return _dilate(_graph(x, k2, "feature"), k1, k2)


def knn_points_normals(x, k1, k2):
    """src/PointNet.py:29-69: rows 0:3 xyz, 3:6 unit normals; metric |dp|^2 (1 + (2 - 2 ni.nj))."""
    with torch.no_grad():
        return _dilate(_graph(x, k2, "points_normals"), k1, k2)


# --------------------------------------------------------------------------------------
# get_graph_feature, API form
# --------------------------------------------------------------------------------------
class _EdgeFeature(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx):
        xt = K.transpose12(x)  # (B,N,C)
        feat = K.edge_feature_fwd(xt, idx)
        ctx.save_for_backward(idx)
        return feat

    @staticmethod
    def backward(ctx, gfeat):
        (idx,) = ctx.saved_tensors
        gxt = K.edge_feature_bwd(gfeat.contiguous(), idx)
        return K.transpose12(gxt), None


def graph_feature(x, idx):
    """x (B,C,N), idx (B,N,k) per-item indices -> (B,2C,N,k) view of (B,N,k,2C) memory,
    exactly the tensor src/model.py:49-51 returns."""
    x = x.contiguous()
    feat = _EdgeFeature.apply(x, idx)
    return feat.permute(0, 3, 1, 2)


# --------------------------------------------------------------------------------------
# fused edge convolution: conv1x1 -> norm -> LeakyReLU -> max_k
# --------------------------------------------------------------------------------------
# The transposed graph the backward gathers through depends on idx alone, so it CAN be built during the forward pass
# on a side stream (five small latency-bound launches, 0.13 ms per layer at cfg4's size) instead of at the head of
# the layer's backward (round-5 verdict, item 3c).  Built and measured in round 6 (tools/jobs/r6e.sh, alternating on
# one box): cfg5 22.93 / 22.86 ms per step with the prefetch against 23.01 / 22.92 without (inside the spread), cfg4
# 9.44 / 9.48 against 9.21 / 9.19 — the side stream's launches run between the waves of the distance passes that
# fill the chip and slow THEM down by more than the backward saves.  Same kernels, same lists, bit-identical
# results (tests/test_edgeconv_gpu.py); opt-in: PARSENET_CSR_PREFETCH=1.
CSR_PREFETCH = os.environ.get("PARSENET_CSR_PREFETCH", "0") == "1"
_CSR_STREAMS = {}


def _csr_stream(dev):
    s = _CSR_STREAMS.get(dev)
    if s is None:
        s = _CSR_STREAMS[dev] = torch.cuda.Stream(device=dev)
    return s


class _EdgeConvNormMax(torch.autograd.Function):
    """PQ (B,N,2*Cout) point-level products, idx (B,N,k) -> out (B,Cout,N).

    ``per_sample`` selects GroupNorm statistics (per item and group) or BatchNorm statistics
    (per channel over the batch).  ``fixed`` = (mean, rstd) replaces the batch statistics by
    constants (eval-mode BatchNorm).  Also returns the fp64 moments so that BatchNorm can
    update its running estimates.
    """

    @staticmethod
    def forward(ctx, PQ, idx, gamma, beta, groups, per_sample, eps, slope, fixed):
        PQ = PQ.contiguous()
        B, N, C2 = PQ.shape
        Cout = C2 // 2
        k = idx.shape[2]
        yext, argk, s1, stats = K.edgeconv_reduce_fwd(PQ, idx, gamma.detach(), groups, per_sample)
        if fixed is None:
            count = (Cout // groups) * N * k * (1 if per_sample else B)
            mean, rstd = K.moments(stats, count, eps)
            dense = True
        else:
            mean, rstd = fixed
            mean = mean.reshape(1, groups).contiguous().float()
            rstd = rstd.reshape(1, groups).contiguous().float()
            dense = False
        out = K.edgeconv_finalize_fwd(yext, mean, rstd, gamma.detach(), beta.detach(), groups, per_sample,
                                      slope)
        ctx.save_for_backward(PQ, idx, gamma, beta, yext, argk, s1, mean, rstd)
        ctx.cfg = (groups, per_sample, slope, dense, k)
        ctx.csr = None
        if CSR_PREFETCH and PQ.is_cuda and ctx.needs_input_grad[0]:
            main, side = torch.cuda.current_stream(PQ.device), _csr_stream(PQ.device)
            side.wait_stream(main)                       # idx was produced on main
            with torch.cuda.stream(side):
                csr = K.edgeconv_csr_build(idx)
                ready = torch.cuda.Event()
                ready.record(side)
            idx.record_stream(side)
            ctx.csr = (csr, ready)
        ctx.mark_non_differentiable(stats)
        return out, stats

    @staticmethod
    def backward(ctx, gout, _gstats):
        PQ, idx, gamma, beta, yext, argk, s1, mean, rstd = ctx.saved_tensors
        groups, per_sample, slope, dense, k = ctx.cfg
        B, N, Cout = yext.shape
        # one fused launch group: t = gamma * gout * LeakyReLU'(z), d gamma, d beta and the group means
        # c1, c2 of the normalisation gradient (fixed-order partial sums, fp64 combination)
        t, dgamma, dbeta, c1c2 = K.edgeconv_bwd_stats(gout, yext, mean, rstd, gamma.detach(), beta.detach(), groups,
                                                      per_sample, dense, slope, k)
        csr = None
        if ctx.csr is not None:
            csr, ready = ctx.csr
            ctx.csr = None
            main = torch.cuda.current_stream(PQ.device)
            main.wait_event(ready)
            csr.record_stream(main)                      # allocated on the side stream, consumed here
        dPQ = K.edgeconv_bwd(PQ, idx, t, s1, argk, mean, rstd, c1c2, groups, per_sample, dense, csr=csr)
        return dPQ, None, dgamma, dbeta, None, None, None, None, None


class _EdgeWeight(torch.autograd.Function):
    """w (Cout, 2C) = [Wa | Wb] -> (C, 2Cout) = [Wa ; Wb - Wa]^T, the operand of the one GEMM on points.  The
    backward pass forms dWa = Ga - Gb (autograd's Ga + (-Gb): the same fp32 value) and dWb = Gb in two launches;
    slicing, subtracting and concatenating through autograd took ten per layer and step (two zero fills, two
    copies, a negation and two additions among them)."""

    @staticmethod
    def forward(ctx, w, C):
        wa, wb = w[:, :C], w[:, C:]
        return torch.cat([wa, wb - wa], 0).t()

    @staticmethod
    def backward(ctx, g):
        gt = g.t()                                          # (2Cout, C)
        co = gt.shape[0] // 2
        ga, gb = gt[:co], gt[co:]
        return torch.cat([ga - gb, gb], 1), None


def edge_conv_norm_max(x, idx, weight, norm, slope=0.2):
    """One DGCNN edge-conv layer on the kNN graph ``idx``:
    max_k LeakyReLU(norm(conv1x1(cat(x_j - x_i, x_i)))) -> (B,Cout,N).

    x (B,C,N); weight (Cout, 2C, 1, 1) or (Cout, 2C) without bias; ``norm`` is the layer's
    torch.nn.GroupNorm / BatchNorm2d module (its parameters, statistics mode and running
    buffers are honoured, including the running-statistics update in training mode).
    """
    B, C, N = x.shape
    w = weight.reshape(weight.shape[0], -1)
    Cout = w.shape[0]
    if w.shape[1] != 2 * C:
        raise ValueError("edge conv weight expects %d input channels, got %d" % (2 * C, w.shape[1]))
    # W [xj - xi ; xi] = Wa xj + (Wb - Wa) xi: one GEMM on points
    frozen = not (weight.requires_grad and torch.is_grad_enabled())
    hit = frozen_cache(norm).get("wcat") if frozen else None
    key = (weight.data_ptr(), weight._version, C)
    if hit is not None and hit[0] == key:
        wcat_t = hit[1]                                    # frozen layer (the SplineNets of the fitting stage)
    else:
        wcat_t = _EdgeWeight.apply(w, C)                    # (C, 2Cout)
        if frozen:
            wcat_t = wcat_t.detach().contiguous()
            frozen_cache(norm)["wcat"] = (key, wcat_t)
    # batched with stride 0 on the weight: no transposing copy of x (see encoders.weight_bmm)
    PQ = torch.bmm(x.transpose(1, 2), wcat_t.unsqueeze(0).expand(B, -1, -1))   # (B,N,2Cout)
    if isinstance(norm, torch.nn.GroupNorm):
        out, _ = _EdgeConvNormMax.apply(PQ, idx, norm.weight, norm.bias, norm.num_groups, True, norm.eps,
                                        slope, None)
        return out
    if isinstance(norm, torch.nn.modules.batchnorm._BatchNorm):
        gamma = norm.weight if norm.weight is not None else torch.ones(Cout, device=x.device)
        beta = norm.bias if norm.bias is not None else torch.zeros(Cout, device=x.device)
        use_batch = norm.training or norm.running_mean is None
        if use_batch:
            out, stats = _EdgeConvNormMax.apply(PQ, idx, gamma, beta, Cout, False, norm.eps, slope, None)
            if norm.training and norm.track_running_stats:
                with torch.no_grad():
                    k = idx.shape[2]
                    M = float(B * N * k)
                    mean = stats[0, :, 0] / M
                    var = (stats[0, :, 1] / M - mean * mean).clamp_min(0.0)
                    norm.num_batches_tracked += 1
                    mom = norm.momentum
                    if mom is None:
                        mom = 1.0 / float(norm.num_batches_tracked)
                    norm.running_mean.mul_(1 - mom).add_(mean.float(), alpha=mom)
                    norm.running_var.mul_(1 - mom).add_((var * (M / max(M - 1.0, 1.0))).float(), alpha=mom)
            return out
        rkey = (norm.running_var.data_ptr(), norm.running_var._version, norm.eps)
        rhit = frozen_cache(norm).get("rstd")
        if rhit is None or rhit[0] != rkey:
            rhit = (rkey, torch.rsqrt(norm.running_var + norm.eps))
            frozen_cache(norm)["rstd"] = rhit
        out, _ = _EdgeConvNormMax.apply(PQ, idx, gamma, beta, Cout, False, norm.eps, slope,
                                        (norm.running_mean, rhit[1]))
        return out
    raise TypeError("unsupported norm layer %r" % (norm,))
