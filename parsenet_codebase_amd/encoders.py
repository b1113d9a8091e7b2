"""DGCNN encoders of the ParSeNet hot path on the HIP kernels.

Parameter names and shapes are those of the reference modules, so its checkpoints load
unchanged (with or without DataParallel's ``module.`` prefix):
  DGCNNControlPoints          src/model.py:56-180     (SplineNet, BatchNorm)
  DGCNNEncoderGn              src/PointNet.py:143-220 (GroupNorm)
  PrimitivesEmbeddingDGCNGn   src/PointNet.py:223-289
Edge-conv layers run through graph.edge_conv_norm_max (kNN + fused gather-reduce kernels);
the per-point heads are dense GEMMs left to rocBLAS via torch.
"""
import os
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import graph
from .norms import group_norm_relu, group_norm_relu_max
from ._lib import require_cuda


# fp32-grade GEMMs on the bf16 matrix cores (csrc/gemm_x3.hip) for the large per-point layers.  As accurate as the
# rocBLAS fp32 product (tests/test_gemm_gpu.py), 1.2-2.6 x faster per forward product.  PARSENET_GEMM_X3:
#   "frozen" (default) products with a FROZEN weight only (requires_grad False: the SplineNets inside an end-to-end
#            step, whose 1152 -> 1024 layer on S x 5 000 points is the step's largest product); their weight images are
#            cached, nothing that is trained changes its rounding;
#   "1"      every product above the thresholds below — also the trained layers: forward, the gradient w.r.t. the
#            activations and (round 6) the weight + bias gradient (pn_gemm_x3_wgrad_f32: split over the points, fixed-
#            order sum).  Measured in round 6 on one box, alternating (tools/jobs/r6c.sh, profiles/r06_gemm_x3_ab.txt):
#            cfg4 504 against 518 shapes/s, cfg5 inside the spread, cfg3 +1 %, cfg2 +14 %.  The weight gradient is
#            where it loses: the split images of BOTH operands (gy and x over 40 000 points: 0.03-0.09 ms) are written
#            for ONE use, and rocBLAS already runs these skinny products at 90-100 TFLOP/s — 0.275 against 0.234 ms
#            for the 1024 x 256 layer, 0.107 against 0.056 ms for 128 x 256; it wins from ~1M weight elements on
#            (cfg3's 1024 x 1152 conv5: 0.432 against 0.492 ms).  And a change of the trained products' rounding trains
#            ANOTHER network over hundreds of steps: the three whole-step parity tests pre-train their own network and
#            their pinned partitions flip (re-pin search in tools/jobs/r6c.sh: every candidate recipe moves some shape
#            across a merge) — so the trained layers stay on rocBLAS by default;
#   "0"      rocBLAS everywhere.
# Below GEMM_X3_MIN_FLOP / GEMM_X3_MIN_ROWS the split images do not pay.
GEMM_X3_MODE = os.environ.get("PARSENET_GEMM_X3", "frozen")
GEMM_X3 = GEMM_X3_MODE in ("1", "frozen")
GEMM_X3_MIN_FLOP = float(os.environ.get("PARSENET_GEMM_X3_MIN_GFLOP", "2")) * 1e9
GEMM_X3_MIN_ROWS = int(os.environ.get("PARSENET_GEMM_X3_MIN_ROWS", "512"))
_W_IMAGES = {}          # id(frozen parameter) -> {view: ((version, data_ptr), image)}; entries die with the parameter


def _weight_image(w, transposed):
    """Split image of w (or of w^T).  A TRAINABLE weight gets a fresh image on every call (two small launches;
    nothing to go stale when the optimizer moves it).  The image of a frozen one (requires_grad False: the
    SplineNets inside an e2e step) is kept with the parameter object — not with its address, which the allocator
    hands to the next module — and rebuilt when the parameter's version counter or storage changes."""
    from . import kernels as K
    base = w._base if w._base is not None else w
    if base.requires_grad:
        return K.gemm_x3_weight_image(w.detach(), transposed)
    ent = _W_IMAGES.get(id(base))
    if ent is None:
        ent = _W_IMAGES[id(base)] = {}
        weakref.finalize(base, _W_IMAGES.pop, id(base), None)
    key = (transposed, w.storage_offset(), tuple(w.shape), tuple(w.stride()))
    state = (base._version, base.data_ptr())
    hit = ent.get(key)
    if hit is None or hit[0] != state:
        hit = ent[key] = (state, K.gemm_x3_weight_image(w.detach(), transposed))
    return hit[1]


class _WeightGemmX3(torch.autograd.Function):
    """w (Co,Ci) applied to x (B,Ci,N) (+ bias) on the bf16 matrix cores; the gradient w.r.t. x the same way
    with the image of w^T, the gradient w.r.t. w (and the bias) by pn_gemm_x3_wgrad_f32."""

    @staticmethod
    def forward(ctx, w, x, bias):
        from . import kernels as K
        ctx.save_for_backward(w, x)
        ctx.has_bias = bias is not None
        return K.gemm_x3(_weight_image(w, False), w.shape[0], x, bias)

    @staticmethod
    def backward(ctx, gy):
        from . import kernels as K
        w, x = ctx.saved_tensors
        gy = gy.contiguous()
        gw = gx = gb = None
        if ctx.needs_input_grad[1]:
            if _gemm_x3_rows_pay(w.shape[1]):
                gx = K.gemm_x3(_weight_image(w, True), w.shape[1], gy, None)
            else:
                gx = torch.bmm(w.t().unsqueeze(0).expand(gy.shape[0], -1, -1), gy)
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[0]:
            # over the B N points: split-K on the matrix cores, fixed-order sum; the bias gradient from the same call
            res = K.gemm_x3_wgrad(gy, x, want_bias=want_b)
            gw, gb = res if want_b else (res, None)
        elif want_b:
            gb = gy.sum((0, 2))
        return gw, gx, gb


def _gemm_x3_rows_pay(M):
    """The split image of the activations costs ~10 bytes per element of x whatever M is, the matrix-core GEMM
    saves ~40 % of 2 M K N / 100 TFLOP/s: measured (tools/kbench.py gemm) even at M = 256, 1.2-1.3 x at
    M >= 512."""
    return M >= GEMM_X3_MIN_ROWS


def _gemm_x3_pays(w, x):
    if GEMM_X3_MODE == "frozen" and (w._base if w._base is not None else w).requires_grad:
        return False
    return (GEMM_X3 and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and
            _gemm_x3_rows_pay(w.shape[0]) and w.shape[1] >= 64 and
            2.0 * w.shape[0] * w.shape[1] * x.shape[0] * x.shape[2] >= GEMM_X3_MIN_FLOP)


def weight_bmm(w, x, bias=None):
    """w (Co,Ci) applied to x (B,Ci,N) -> (B,Co,N) (+ bias (Co)).  Large products: bf16 x 3 on the matrix
    cores (_WeightGemmX3).  Otherwise a strided-batched rocBLAS GEMM with batch stride 0 on the weight:
    torch.matmul would fold the batch into the rows instead, which costs a transposing copy of the
    activations on the way in and on the way out (13 % of a cfg4 step)."""
    if _gemm_x3_pays(w, x):
        return _WeightGemmX3.apply(w, x, bias)
    y = torch.bmm(w.unsqueeze(0).expand(x.shape[0], -1, -1), x)
    if bias is not None:
        y = y + bias.view(1, -1, 1)
    return y


def conv1x1(x, conv):
    """nn.Conv1d(kernel_size=1) applied as a plain GEMM (rocBLAS) instead of a MIOpen convolution:
    same arithmetic, no per-shape algorithm search (the fitting stage feeds a new point count for
    every segment).  The weight goes in as a VIEW (flatten), not as ``weight[:, :, 0]``: a select's backward node
    fills a zero tensor and copies the gradient into it, two launches per weight and step for the same bits."""
    return weight_bmm(conv.weight.flatten(1), x, conv.bias)


def batch_norm_1d(x, bn):
    """nn.BatchNorm1d applied to (B,C,N).  Evaluation mode is written out as the per-channel
    affine map it is: torch routes F.batch_norm to MIOpen, which searches / compiles a kernel for
    every new tensor shape — and the fitting stage feeds a different point count for every
    segment (tens of milliseconds per first-seen size).  Training mode (fixed batch shapes) keeps
    the library call, including the running-statistics update."""
    if bn.training or bn.running_mean is None:
        return bn(x)
    scale = torch.rsqrt(bn.running_var + bn.eps)
    shift = -bn.running_mean * scale
    if bn.weight is not None:
        scale = scale * bn.weight
        shift = shift * bn.weight + bn.bias
    return x * scale.view(1, -1, 1) + shift.view(1, -1, 1)


class _AffineAct(torch.autograd.Function):
    """act(x * scale[c] + shift[c]) on (B,C,N) in one launch (csrc/fused.hip); scale / shift are
    constants of a frozen evaluation-mode BatchNorm (no gradient flows to them)."""

    @staticmethod
    def forward(ctx, x, scale, shift, act, slope):
        from . import kernels as K
        y = K.affine_act_fwd(x, scale, shift, act, slope)
        ctx.save_for_backward(y, scale)
        ctx.cfg = (act, slope)
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import kernels as K
        y, scale = ctx.saved_tensors
        return K.affine_act_bwd(gy, y, scale, *ctx.cfg), None, None, None, None


_ACT = {"none": 0, "relu": 1, "leaky": 2}
_WMAX_BWD_MAX_POINTS = 13600     # pn_weighted_max_bwd_f32 keeps 12 bytes per point in LDS (160 KiB - 256 - 256)


def _frozen_affine(conv, bn, x):
    """(scale, shift) of ``bn(conv(.) )`` as a per-channel affine map after the bias-free GEMM, or
    None unless ``bn`` is a frozen evaluation-mode BatchNorm (running statistics, no tensor of it
    or the convolution's bias wants a gradient) on the GPU.  Cached (graph.frozen_cache: outside the
    module, validated by data_ptr / _version of its tensors; after an edit through ``.data`` call
    graph.invalidate_frozen_caches) until one of its tensors changes."""
    frozen = not (bn.training or bn.running_mean is None) and x.is_cuda and not (
        torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (bn.weight, bn.bias, conv.bias)))
    if not frozen:
        return None
    src = (bn.running_mean, bn.running_var, bn.weight, bn.bias, conv.bias)
    key = tuple((t.data_ptr(), t._version) if t is not None else None for t in src) + (bn.eps,)
    hit = graph.frozen_cache(bn).get("affine")
    if hit is None or hit[0] != key:
        with torch.no_grad():
            scale = torch.rsqrt(bn.running_var + bn.eps)
            shift = -bn.running_mean * scale
            if bn.weight is not None:
                scale = scale * bn.weight
                shift = shift * bn.weight + bn.bias
            if conv.bias is not None:
                shift = shift + conv.bias * scale
        hit = (key, scale.contiguous(), shift.contiguous())
        graph.frozen_cache(bn)["affine"] = hit
    return hit[1], hit[2]


def conv_bn_act(x, conv, bn, act, slope=0.0):
    """act(bn(conv1x1(x))) for x (B,Ci,N).  A FROZEN evaluation-mode BatchNorm1d (the SplineNets of
    the fitting stage: src/model.py:160-176 under eval(), parameters without gradient) is the
    per-channel affine map it is, folded with the convolution's bias and the activation into ONE
    launch after the GEMM.  Training mode, or parameters that want a gradient: the generic
    expressions."""
    aff = _frozen_affine(conv, bn, x)
    if aff is None:
        y = batch_norm_1d(conv1x1(x, conv), bn)
        return F.relu(y) if act == "relu" else F.leaky_relu(y, slope) if act == "leaky" else y
    return _AffineAct.apply(weight_bmm(conv.weight.flatten(1), x), aff[0], aff[1], _ACT[act], slope)


class _WeightedMax(torch.autograd.Function):
    """max_n act(y * scale + shift) * w for y (S,C,N), w (S,N) -> (S,C): the frozen SplineNet's
    conv5 / bn5 / LeakyReLU, membership weighting and max pool in one pass (csrc/fused.hip); the
    only gradient is the one to the memberships w."""

    @staticmethod
    def forward(ctx, y, scale, shift, w, act, slope):
        from . import kernels as K
        out, idx, val = K.weighted_max_fwd(y, scale, shift, w, act, slope)
        ctx.save_for_backward(idx, val)
        ctx.n = y.shape[2]
        return out

    @staticmethod
    def backward(ctx, g):
        from . import kernels as K
        idx, val = ctx.saved_tensors
        return None, None, None, K.weighted_max_bwd(g, idx, val, ctx.n), None, None


def _edge_layer(cin2, cout, norm):
    # Sequential only to reproduce the reference's parameter names ("convN.0.weight"); the
    # forward pass feeds the weight to the fused kernel instead of calling it.
    return nn.Sequential(nn.Conv2d(cin2, cout, kernel_size=1, bias=False), norm,
                         nn.LeakyReLU(negative_slope=0.2))


class DGCNNControlPoints(nn.Module):
    """Control-point regression network: (B,3,N) points -> (B, G*G, 3) control grid.

    mode 0: open SplineNet, mode 1: closed SplineNet.  ``num_points`` is the number of
    neighbours (the reference's name).  ``weights`` (only with B == 1): per-point membership
    multiplying the 1024-d point features before the global max pool (src/model.py:165-167).
    """

    _WIDTHS = {0: (64, 64, 128, 256), 1: (128, 256, 256, 512)}

    def __init__(self, num_control_points, num_points=40, mode=0):
        super().__init__()
        if mode not in self._WIDTHS:
            raise ValueError("mode must be 0 or 1")
        self.k = num_points
        self.mode = mode
        self.drop = 0.0
        self.controlpoints = num_control_points
        w = self._WIDTHS[mode]
        self.bn1 = nn.BatchNorm2d(w[0])
        self.bn2 = nn.BatchNorm2d(w[1])
        self.bn3 = nn.BatchNorm2d(w[2])
        self.bn4 = nn.BatchNorm2d(w[3])
        self.bn5 = nn.BatchNorm1d(1024)
        self.conv1 = _edge_layer(6, w[0], self.bn1)
        self.conv2 = _edge_layer(2 * w[0], w[1], self.bn2)
        self.conv3 = _edge_layer(2 * w[1], w[2], self.bn3)
        self.conv4 = _edge_layer(2 * w[2], w[3], self.bn4)
        self.conv5 = nn.Sequential(nn.Conv1d(sum(w), 1024, kernel_size=1, bias=False), self.bn5,
                                   nn.LeakyReLU(negative_slope=0.2))
        self.conv6 = nn.Conv1d(1024, 1024, 1)
        self.conv7 = nn.Conv1d(1024, 1024, 1)
        self.conv8 = nn.Conv1d(1024, 3 * (num_control_points ** 2), 1)
        self.bn6 = nn.BatchNorm1d(1024)
        self.bn7 = nn.BatchNorm1d(1024)
        self.tanh = nn.Tanh()

    def forward(self, x, weights=None):
        require_cuda(x)
        batch_size = x.size(0)
        feats = []
        for conv, bn in ((self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3),
                         (self.conv4, self.bn4)):
            with graph.library_graphs():
                idx = graph.knn(x, self.k)
            x = graph.edge_conv_norm_max(x, idx, conv[0].weight, bn, slope=0.2)
            feats.append(x)
        w5 = self.conv5[0].weight.flatten(1)
        aff = _frozen_affine(self.conv5[0], self.bn5, feats[0])
        npts = feats[0].shape[2]
        grad5 = torch.is_grad_enabled() and (any(f.requires_grad for f in feats) or self.conv5[0].weight.requires_grad)

        def conv5_product():
            # frozen network (the fitting stage's SplineNets): W applied to the four layers' outputs where they lie —
            # the concatenation of src/model.py:150 (S x 512 / 1152 x 5000 floats, 0.37 ms of copies per cfg5 step)
            # is never written
            if not grad5 and _gemm_x3_pays(w5, feats[0]) and all(f.shape[1] % 8 == 0 for f in feats):
                from . import kernels as K
                return K.gemm_x3_cat(_weight_image(w5, False), w5.shape[0], feats)
            return weight_bmm(w5, torch.cat(feats, dim=1))
        if (aff is not None and isinstance(weights, torch.Tensor) and weights.numel() == batch_size * npts
                and npts <= _WMAX_BWD_MAX_POINTS and not grad5):
            # frozen network inside the fitting stage: activation, membership weighting and the max over
            # the points in ONE pass over the 1024-channel features (they are never written out)
            x = _WeightedMax.apply(conv5_product(), aff[0], aff[1],
                                   weights.reshape(batch_size, -1), _ACT["leaky"], 0.2).unsqueeze(2)
            x = conv_bn_act(x, self.conv6, self.bn6, "relu")
            x = conv_bn_act(x, self.conv7, self.bn7, "relu")
            x = self.tanh(conv1x1(x, self.conv8)[:, :, 0])
            return x.view(batch_size, self.controlpoints * self.controlpoints, 3)
        if aff is None:
            x = conv_bn_act(torch.cat(feats, dim=1), self.conv5[0], self.bn5, "leaky", 0.2)
        else:
            x = _AffineAct.apply(conv5_product(), aff[0], aff[1], _ACT["leaky"], 0.2)
        if isinstance(weights, torch.Tensor):
            # the reference reshapes to (1,1,-1) (one segment per call); a (B,n) matrix weights
            # every item of a batch of segments with its own memberships
            x = x * weights.reshape((batch_size if weights.numel() == batch_size * x.shape[2] else 1, 1, -1))
        # max over the points (F.adaptive_max_pool1d(x, 1) in the reference; torch's pooling kernel
        # takes 0.5 ms on a 1024 x 5000 input, the reduction 20 us)
        x = x.max(dim=2, keepdim=True)[0]
        x = conv_bn_act(x, self.conv6, self.bn6, "relu")
        x = conv_bn_act(x, self.conv7, self.bn7, "relu")
        x = self.tanh(conv1x1(x, self.conv8)[:, :, 0])
        return x.view(batch_size, self.controlpoints * self.controlpoints, 3)


class DGCNNEncoderGn(nn.Module):
    """Three edge-conv layers (k = nn_nb) + global 1024-d feature.  mode 5 uses the
    points+normals metric for the first graph (6 input channels)."""

    def __init__(self, mode=0, input_channels=3, nn_nb=80):
        super().__init__()
        self.k = nn_nb
        self.dilation_factor = 1
        self.mode = mode
        self.drop = 0.0
        if mode not in (0, 1, 5):
            raise ValueError("DGCNNEncoderGn: unsupported mode %r" % (mode,))
        self.bn1 = nn.GroupNorm(2, 64)
        self.bn2 = nn.GroupNorm(2, 64)
        self.bn3 = nn.GroupNorm(2, 128)
        # bn4 / bn5 are never used by the reference's forward either; they exist so that
        # its checkpoints load (src/PointNet.py:154-155)
        self.bn4 = nn.GroupNorm(4, 256)
        self.bn5 = nn.GroupNorm(8, 1024)
        self.conv1 = _edge_layer(input_channels * 2, 64, self.bn1)
        self.conv2 = _edge_layer(64 * 2, 64, self.bn2)
        self.conv3 = _edge_layer(64 * 2, 128, self.bn3)
        self.mlp1 = nn.Conv1d(256, 1024, 1)
        self.bnmlp1 = nn.GroupNorm(8, 1024)

    def forward(self, x):
        require_cuda(x)
        k = self.k
        with graph.library_graphs():          # the graphs stay inside the library: int32
            if self.mode == 5:
                idx = graph.knn_points_normals(x, k, k)
            else:
                idx = graph.knn_dilated(x, k, k)
            x1 = graph.edge_conv_norm_max(x, idx, self.conv1[0].weight, self.bn1, slope=0.2)
            x2 = graph.edge_conv_norm_max(x1, graph.knn_dilated(x1, k, k), self.conv2[0].weight, self.bn2, 0.2)
            x3 = graph.edge_conv_norm_max(x2, graph.knn_dilated(x2, k, k), self.conv3[0].weight, self.bn3, 0.2)
        x_features = torch.cat((x1, x2, x3), dim=1)
        # GroupNorm + ReLU + max over the points in one pass (norm and ReLU are monotone per channel); the
        # convolution's bias is added inside the norm's kernels (round 6: no pass of its own over (B,1024,N))
        x4 = group_norm_relu_max(weight_bmm(self.mlp1.weight.flatten(1), x_features), self.bnmlp1,
                                 rowbias=self.mlp1.bias)
        return x4, x_features


class PrimitivesEmbeddingDGCNGn(nn.Module):
    """Segmentation network: per-point embedding (emb_size) and primitive-type log
    probabilities.  The embedding loss is evaluated inside forward, as in the reference, so
    that every data-parallel rank computes it on its own shard."""

    def __init__(self, emb_size=50, num_primitives=8, primitives=False, embedding=False, mode=0,
                 num_channels=3, loss_function=None, nn_nb=80):
        super().__init__()
        self.mode = mode
        self.encoder = DGCNNEncoderGn(mode=mode, input_channels=num_channels, nn_nb=nn_nb)
        self.drop = 0.0
        self.loss_function = loss_function
        if mode in (0, 3, 4, 5, 6):
            self.conv1 = nn.Conv1d(1024 + 256, 512, 1)
        elif mode in (1, 2):
            self.conv1 = nn.Conv1d(1024 + 512, 512, 1)
        else:
            raise ValueError("unsupported mode %r" % (mode,))
        self.bn1 = nn.GroupNorm(8, 512)
        self.conv2 = nn.Conv1d(512, 256, 1)
        self.bn2 = nn.GroupNorm(4, 256)
        self.softmax = nn.Softmax(dim=1)
        self.logsoftmax = nn.LogSoftmax(dim=1)
        self.tanh = nn.Tanh()
        self.emb_size = emb_size
        self.primitives = primitives
        self.embedding = embedding
        if self.embedding:
            self.mlp_seg_prob1 = nn.Conv1d(256, 256, 1)
            self.mlp_seg_prob2 = nn.Conv1d(256, self.emb_size, 1)
            self.bn_seg_prob1 = nn.GroupNorm(4, 256)
        if primitives:
            self.mlp_prim_prob1 = nn.Conv1d(256, 256, 1)
            self.mlp_prim_prob2 = nn.Conv1d(256, num_primitives, 1)
            self.bn_prim_prob1 = nn.GroupNorm(4, 256)

    def forward(self, points, labels, compute_loss=True):
        require_cuda(points)
        x, first_layer_features = self.encoder(points)
        # conv1 on cat(global repeated over N, local): the global part is the same for every
        # point, so it is applied once per item and broadcast (same sum, 5x fewer FLOPs)
        ng = x.shape[1]
        w = self.conv1.weight.flatten(1)
        glob = torch.addmm(self.conv1.bias, x, w[:, :ng].t())            # (B,512)
        # the per-item global term and the biases of the layers that feed a GroupNorm are added inside the norm's
        # kernels (norms.py ``rowbias``: the same fp32 addition at load, no pass of its own over (B,C,N))
        x = group_norm_relu(weight_bmm(w[:, ng:], first_layer_features), self.bn1, rowbias=glob)
        x_all = group_norm_relu(weight_bmm(self.conv2.weight.flatten(1), x), self.bn2, rowbias=self.conv2.bias)
        embedding = None
        primitives_log_prob = None
        if self.embedding:
            x = group_norm_relu(weight_bmm(self.mlp_seg_prob1.weight.flatten(1), x_all), self.bn_seg_prob1,
                                rowbias=self.mlp_seg_prob1.bias)
            embedding = conv1x1(x, self.mlp_seg_prob2)
        if self.primitives:
            x = group_norm_relu(weight_bmm(self.mlp_prim_prob1.weight.flatten(1), x_all), self.bn_prim_prob1,
                                rowbias=self.mlp_prim_prob1.bias)
            primitives_log_prob = self.logsoftmax(conv1x1(x, self.mlp_prim_prob2))
        if compute_loss:
            lab = labels.data.cpu().numpy() if torch.is_tensor(labels) else labels
            embed_loss = self.loss_function(embedding, lab)
        else:
            embed_loss = torch.zeros(1, device=points.device)
        return embedding, primitives_log_prob, embed_loss


class PrimitivesEmbeddingDGCNGne2e(PrimitivesEmbeddingDGCNGn):
    """src/PointNet.py:292-380: the same network with the fitting loss evaluated inside
    ``forward`` (no script of the reference instantiates it).  ``self.evaluation`` must be set to
    an ``Evaluation`` object by the caller, like in the reference; the embedding loss is called
    as ``loss_function(embedding, points (B,N,C), labels)``.  Same parameters and state_dict keys
    as PrimitivesEmbeddingDGCNGn."""

    evaluation = None

    def forward(self, points, labels, primitives, quantile, debug, compute_loss=True):
        loss_function, self.loss_function = self.loss_function, None
        try:
            embedding, primitives_log_prob, _ = super().forward(points, labels, compute_loss=False)
        finally:
            self.loss_function = loss_function
        if compute_loss:
            lab = labels.data.cpu().numpy() if torch.is_tensor(labels) else labels
            embed_loss = self.loss_function(embedding, points.permute(0, 2, 1), lab)
        else:
            embed_loss = torch.zeros(1, device=points.device)
        if self.evaluation is None:
            raise RuntimeError("PrimitivesEmbeddingDGCNGne2e: set .evaluation (an Evaluation object) first")
        normals = points[:, 3:, :].permute(0, 2, 1)
        res_loss = self.evaluation.fitting_loss(embedding.permute(0, 2, 1), points.permute(0, 2, 1)[:, :, 0:3],
                                                normals, labels, primitives, primitives_log_prob, quantile=0.025,
                                                debug=False)
        return res_loss, embedding, primitives_log_prob, embed_loss
