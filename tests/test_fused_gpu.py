"""GPU: the round-3 fused kernels (csrc/fused.hip) against the tensor expressions they replace —
which are the oracle-checked restatements of the reference's lines (tests/test_golden_gpu.py,
test_fitting_batch_gpu.py, test_encoder_gpu.py run the same paths against the fixtures)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


@pytest.mark.parametrize("per_sample,dense,Cout,groups", [(True, True, 64, 2), (False, True, 128, 128),
                                                          (False, False, 64, 64), (True, True, 256, 4)])
def test_edgeconv_backward_stats(gpu, per_sample, dense, Cout, groups):
    """t, d gamma, d beta, c1 / c2 of graph._EdgeConvNormMax.backward: the fused launch group vs
    pn_edgeconv_bwd_prep_f32 + the fp64 tensor reductions it replaces."""
    from parsenet_codebase_amd import kernels as K
    torch.manual_seed(0)
    B, N, k, slope = 3, 1000, 20, 0.2
    yext = torch.randn(B, N, Cout, device=gpu)
    gout = torch.randn(B, Cout, N, device=gpu)
    gamma = torch.randn(Cout, device=gpu)
    beta = torch.randn(Cout, device=gpu)
    S = B if per_sample else 1
    mean = 0.1 * torch.randn(S, groups, device=gpu)
    rstd = 1.0 + 0.1 * torch.rand(S, groups, device=gpu)
    t, dgamma, dbeta, c1c2 = K.edgeconv_bwd_stats(gout, yext, mean, rstd, gamma, beta, groups, per_sample, dense,
                                                  slope, k)
    gz, yhat = K.edgeconv_bwd_prep(gout, yext, mean, rstd, gamma, beta, groups, per_sample, slope)
    assert torch.equal(t, gz * gamma)
    assert _rel(dbeta, gz.double().sum((0, 1))) < 1e-6 and _rel(dgamma, (gz.double() * yhat.double()).sum((0, 1))) < 1e-6
    Cg = Cout // groups
    tv = (gz * gamma).view(B, N, groups, Cg).double()
    yv = yhat.view(B, N, groups, Cg).double()
    if per_sample:
        ref = torch.stack([tv.sum((1, 3)), (tv * yv).sum((1, 3))], -1) / float(Cg * N * k)
    else:
        ref = torch.stack([tv.sum((0, 1, 3)), (tv * yv).sum((0, 1, 3))], -1).unsqueeze(0) / float(Cg * N * k * B)
    if not dense:
        ref = torch.zeros_like(ref)
    assert c1c2.shape == ref.shape and float((c1c2.double() - ref).abs().max()) <= 1e-6 * float(ref.abs().max() + 1e-30)


def test_edgeconv_layer_gradients_unchanged(gpu):
    """The whole layer through autograd (GroupNorm statistics) against a plain torch edge conv."""
    from parsenet_codebase_amd import graph
    torch.manual_seed(1)
    B, C, N, k, Cout = 2, 16, 600, 12, 64
    x = torch.randn(B, C, N, device=gpu, requires_grad=True)
    w = (0.2 * torch.randn(Cout, 2 * C, 1, 1, device=gpu)).requires_grad_(True)
    gn = torch.nn.GroupNorm(2, Cout).to(gpu)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(Cout))
        gn.bias.copy_(torch.randn(Cout))
    idx = graph.knn(x, k)
    out = graph.edge_conv_norm_max(x, idx, w, gn, 0.2)
    g = torch.randn_like(out)
    (out * g).sum().backward()
    got = [x.grad.clone(), w.grad.clone(), gn.weight.grad.clone(), gn.bias.grad.clone()]
    x.grad = w.grad = gn.weight.grad = gn.bias.grad = None
    xt = x.transpose(1, 2)
    nb = torch.gather(xt.unsqueeze(1).expand(-1, N, -1, -1), 2, idx.unsqueeze(3).expand(-1, -1, -1, C))
    feat = torch.cat([nb - xt.unsqueeze(2), xt.unsqueeze(2).expand(-1, -1, k, -1)], 3).permute(0, 3, 1, 2)
    ref = torch.nn.functional.leaky_relu(gn(torch.nn.functional.conv2d(feat, w)), 0.2).max(3)[0]
    assert _rel(out, ref) < 1e-5
    (ref * g).sum().backward()
    for a, b in zip(got, [x.grad, w.grad, gn.weight.grad, gn.bias.grad]):
        assert _rel(a, b) < 2e-4, _rel(a, b)


@pytest.mark.parametrize("num", [30, 7])
def test_triplet_loss_kernel(gpu, num):
    """EmbeddingLoss.triplet_loss (src/segment_loss.py:31-124): fused kernels vs the tensor-expression
    form, same numpy RNG draws; loss and gradient with respect to the network output."""
    from parsenet_codebase_amd import losses
    B, N, S = 3, 400, 6
    rng = np.random.RandomState(0)
    labels = rng.randint(0, S, (B, N))
    if num < 30:
        N = 40
        labels = labels[:, :N] % 8          # N // S + 1 < 30 -> fewer samples per segment
    out = torch.randn(B, 128, N, device=gpu)
    res = {}
    for fused in (True, False):
        losses.FUSED = fused
        try:
            o = out.clone().requires_grad_(True)
            np.random.seed(5)
            l = losses.EmbeddingLoss(margin=1.0).triplet_loss(o, labels)
            l.sum().backward()
            res[fused] = (l.detach().clone(), o.grad.clone(), np.random.get_state()[1][:4].tolist())
        finally:
            losses.FUSED = True
    assert res[True][2] == res[False][2]
    assert _rel(res[True][0], res[False][0]) < 2e-6
    assert _rel(res[True][1], res[False][1]) < 2e-5


@pytest.mark.parametrize("ncls", [[1, 5, 16], [33, 2, 49], [12, 12, 7]])
def test_membership_kernels(gpu, ncls):
    """Wraw, weights_normalize and its gradient onto centres and embedding: fused vs
    fitting_batch.weights_normalize_batch(bmm(...)) through autograd; labels = first arg-max."""
    from parsenet_codebase_amd import fitting_batch as FB, kernels as K
    torch.manual_seed(2)
    B, N, D = len(ncls), 3000, 128
    Cp = max(ncls)
    emb = torch.nn.functional.normalize(torch.randn(B, N, D, device=gpu), dim=2)
    pick = torch.randint(0, N, (B, Cp), device=gpu)
    cen = torch.gather(emb, 1, pick.unsqueeze(2).expand(-1, -1, D)) + 0.05 * torch.randn(B, Cp, D, device=gpu)
    ncl = torch.tensor(ncls, device=gpu)
    cen = cen * (torch.arange(Cp, device=gpu).unsqueeze(0) < ncl.unsqueeze(1)).unsqueeze(2)    # padded rows: zeros
    bw = torch.tensor([0.3, 0.11, 0.45][:B], device=gpu)
    g = torch.randn(B, Cp, N, device=gpu)
    c1, e1 = cen.clone().requires_grad_(True), emb.clone().requires_grad_(True)
    Wn1, Wraw1 = FB.memberships(c1, e1, bw, ncl)
    (Wn1[:, :Cp] * g).sum().backward()
    c2, e2 = cen.clone().requires_grad_(True), emb.clone().requires_grad_(True)
    Wraw2 = torch.bmm(c2, e2.transpose(1, 2))
    Wn2 = FB.weights_normalize_batch(Wraw2, bw, ncl)
    (Wn2 * g).sum().backward()
    assert Wn1.shape[1] in (16, 32, 64)
    if Wn1.shape[1] > Cp:
        assert float(Wn1[:, Cp:].abs().max()) == 0
    assert _rel(Wraw1[:, :Cp], Wraw2) < 2e-6
    assert float((Wn1[:, :Cp] - Wn2).abs().max()) < 2e-5
    assert _rel(c1.grad, c2.grad) < 2e-4 and _rel(e1.grad, e2.grad) < 2e-4
    # labels: first arg-max over the valid centre rows of the kernel's own Wraw
    CP = Wn1.shape[1]
    cpad = torch.nn.functional.pad(cen, (0, 0, 0, CP - Cp))
    Wraw, _, _, _, lab = K.membership_fwd(cpad, emb, bw, ncl, 1e-7, want_labels=True)
    valid = torch.arange(CP, device=gpu).view(1, CP, 1) < ncl.view(B, 1, 1)
    sc = torch.where(valid, Wraw, torch.full_like(Wraw, float("-inf")))
    from parsenet_codebase_amd.mean_shift import _first_argmax
    assert torch.equal(lab, _first_argmax(sc, 1))


@pytest.mark.parametrize("act", ["leaky", "relu", "none"])
def test_frozen_batchnorm_affine(gpu, act):
    """encoders.conv_bn_act with a frozen evaluation-mode BatchNorm1d vs conv -> bn -> activation."""
    from parsenet_codebase_amd.encoders import conv_bn_act
    torch.manual_seed(3)
    conv = torch.nn.Conv1d(96, 160, 1).to(gpu)
    bn = torch.nn.BatchNorm1d(160).to(gpu)
    with torch.no_grad():
        bn.running_mean.copy_(torch.randn(160))
        bn.running_var.copy_(torch.rand(160) + 0.5)
        bn.weight.copy_(torch.randn(160))
        bn.bias.copy_(torch.randn(160))
    for p in list(conv.parameters()) + list(bn.parameters()):
        p.requires_grad = False
    bn.eval()
    x = torch.randn(5, 96, 333, device=gpu, requires_grad=True)
    y = conv_bn_act(x, conv, bn, act, 0.2)
    g = torch.randn_like(y)
    (y * g).sum().backward()
    gx = x.grad.clone()
    x.grad = None
    r = bn(conv(x))
    r = torch.relu(r) if act == "relu" else torch.nn.functional.leaky_relu(r, 0.2) if act == "leaky" else r
    (r * g).sum().backward()
    assert _rel(y, r) < 2e-6 and _rel(gx, x.grad) < 2e-5
    # the cache follows in-place changes of the statistics
    with torch.no_grad():
        bn.running_mean.add_(1.0)
    y2 = conv_bn_act(x, conv, bn, act, 0.2)
    r2 = bn(conv(x))
    r2 = torch.relu(r2) if act == "relu" else torch.nn.functional.leaky_relu(r2, 0.2) if act == "leaky" else r2
    assert _rel(y2, r2) < 2e-6


def test_frozen_splinenet_head_weighted_max(gpu):
    """DGCNNControlPoints with frozen parameters in evaluation mode and per-segment memberships:
    the fused conv5 -> bn5 -> LeakyReLU -> x * weights -> max path vs the generic expressions
    (same module, gradient mode of one BatchNorm parameter switched on to force them): control
    points and the gradient with respect to the memberships."""
    from parsenet_codebase_amd.encoders import DGCNNControlPoints
    torch.manual_seed(4)
    net = DGCNNControlPoints(20, num_points=10, mode=0).to(gpu).eval()
    with torch.no_grad():
        for bn in (net.bn5, net.bn6, net.bn7):
            bn.running_mean.copy_(0.1 * torch.randn_like(bn.running_mean))
            bn.running_var.copy_(torch.rand_like(bn.running_var) + 0.5)
    for p in net.parameters():
        p.requires_grad = False
    pts = torch.randn(3, 3, 1500, device=gpu) * 0.3
    res = []
    w0 = torch.rand(3, 1500, device=gpu)
    for fused in (True, False):
        net.bn5.weight.requires_grad = not fused          # a parameter that wants a gradient: generic path
        w = w0.clone().requires_grad_(True)
        out = net(pts, w)
        g = torch.randn_like(out)
        (out * g).sum().backward() if fused else (out * res[0][2]).sum().backward()
        res.append((out.detach(), w.grad.clone(), g))
    net.bn5.weight.requires_grad = False
    assert _rel(res[0][0], res[1][0]) < 5e-6
    assert _rel(res[0][1], res[1][1]) < 5e-5


def test_weighted_max_backward_adds_shared_points_in_channel_order(gpu):
    """pn_weighted_max_bwd_f32: gw[s][n] = the terms g * val of the channels whose arg-max is n, added one
    after the other in channel order (what a serial loop does) — bit for bit, with most channels of a chunk
    of 64 naming one of a few points (the in-wave resolution), others a point of their own (the direct
    path), and a channel count that is not a multiple of 64."""
    from parsenet_codebase_amd import kernels as K
    rng = np.random.RandomState(2)
    S, C, N = 3, 1024 + 37, 700
    idx = rng.randint(0, N, (S, C)).astype(np.int32)
    hot = rng.rand(S, C) < 0.45
    idx[hot] = rng.choice([5, 6, 311, 699], int(hot.sum())).astype(np.int32)
    idx[1, 64:128] = 17                                  # a whole chunk on one point
    idx[2, 200:264] = np.arange(64)                      # a chunk without any shared point
    g = rng.standard_normal((S, C)).astype(np.float32)
    val = rng.standard_normal((S, C)).astype(np.float32)
    want = np.zeros((S, N), np.float32)
    for s in range(S):
        for c in range(C):
            want[s, idx[s, c]] = np.float32(want[s, idx[s, c]] + np.float32(g[s, c] * val[s, c]))
    got = K.weighted_max_bwd(torch.from_numpy(g).to(gpu), torch.from_numpy(idx).to(gpu),
                             torch.from_numpy(val).to(gpu), N).cpu().numpy()
    assert np.array_equal(got, want)


def test_flat_adam_follows_torch_adam_and_round_trips_its_state(gpu):
    """optim.FlatAdam (one launch on flat buffers) against torch.optim.Adam on the same parameters and gradients
    over several steps, a learning-rate change in between (ReduceLROnPlateau edits param_groups), and a
    state_dict / load_state_dict round trip in torch's format (checkpoints interchange with torch.optim.Adam:
    train_parsenet.py:96, :255-262)."""
    import copy
    from parsenet_codebase_amd.dp import FlatGradBucket
    from parsenet_codebase_amd.optim import FlatAdam
    torch.manual_seed(5)
    shapes = [(64, 6, 1, 1), (64,), (1024, 256, 1), (1024,), (7, 3), (1,)]
    ref_params = [torch.nn.Parameter(torch.randn(s, device=gpu)) for s in shapes]
    params = [torch.nn.Parameter(p.detach().clone()) for p in ref_params]
    bucket = FlatGradBucket(params)
    opt = FlatAdam(bucket, lr=1e-2)
    ref = torch.optim.Adam(ref_params, lr=1e-2)
    assert all(torch.equal(a, b) for a, b in zip(params, ref_params))      # moving into the flat buffer keeps the values

    def one_step(k):
        bucket.zero()
        for p, r in zip(params, ref_params):
            g = torch.randn(p.shape, device=gpu, generator=None) * (0.1 + k)
            p.grad.copy_(g)
            r.grad = g.clone()
        opt.step()
        ref.step()
    for k in range(4):
        one_step(k)
    for grp in opt.param_groups + ref.param_groups:
        grp["lr"] = 5e-3
    one_step(4)
    for p, r in zip(params, ref_params):
        assert float((p - r).abs().max()) <= 2e-6 * float(r.abs().max()) + 1e-7
    # torch's format: the state of one loads into the other
    sd = copy.deepcopy(opt.state_dict())
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 5.0
    ref2_params = [torch.nn.Parameter(p.detach().clone()) for p in params]
    ref2 = torch.optim.Adam(ref2_params, lr=5e-3)
    ref2.load_state_dict(copy.deepcopy(sd))
    params3 = [torch.nn.Parameter(p.detach().clone()) for p in params]
    bucket3 = FlatGradBucket(params3)
    opt3 = FlatAdam(bucket3, lr=5e-3)
    opt3.load_state_dict(copy.deepcopy(ref2.state_dict()))
    assert opt3.steps == 5
    g = [torch.randn(p.shape, device=gpu) for p in params]
    bucket.zero(); bucket3.zero()
    for p, p3, r, gg in zip(params, params3, ref2_params, g):
        p.grad.copy_(gg); p3.grad.copy_(gg); r.grad = gg.clone()
    opt.step(); opt3.step(); ref2.step()
    for p, p3, r in zip(params, params3, ref2_params):
        assert torch.equal(p, p3)                                   # a reloaded optimizer continues bit for bit
        assert float((p - r).abs().max()) <= 2e-6 * float(r.abs().max()) + 1e-7
    # a parameter that leaves the flat buffer is an error, not a silent no-op
    params[0].data = params[0].data.clone()
    with pytest.raises(RuntimeError):
        opt.step()


def test_bucket_gather_is_one_launch_and_equals_the_accumulated_bucket(gpu):
    """dp.FlatGradBucket.begin() / gather() on the GPU (pn_gather_flat_f32: the gradients autograd hands over go
    into the flat buffer in one launch) against zero() + accumulation into the views: the same bucket, also for
    more than 64 parameters (two launches), non-contiguous gradients and a parameter without a gradient."""
    from parsenet_codebase_amd.dp import FlatGradBucket
    torch.manual_seed(2)
    shapes = [(64, 6, 1, 1), (64,), (7, 3), (1,), (1024, 256, 1)] + [(5, k + 1) for k in range(70)]
    params = [torch.nn.Parameter(torch.randn(s, device=gpu)) for s in shapes]
    bucket = FlatGradBucket(params)
    x = torch.randn(3, device=gpu)

    def loss():
        # parameter 3 gets no gradient; parameter 2's gradient arrives as a transposed (non-contiguous) tensor
        tot = (params[2].t() * torch.arange(3, device=gpu).view(3, 1)).sum() * x.sum()
        for k, p in enumerate(params):
            if k not in (2, 3):
                tot = tot + (p * p).sum() * (k + 1)
        return tot
    bucket.zero()
    loss().backward()
    want = bucket.flat.clone()
    bucket.begin()
    assert all(p.grad is None for p in params)
    loss().backward()
    got = bucket.gather()
    assert torch.equal(got, want)
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(params, bucket.views))
    # a second step in which parameter 0 gets no gradient: its slot is cleared, not stale
    bucket.begin()
    (params[1] * 2).sum().backward()
    g2 = bucket.gather()
    assert float(g2[:params[0].numel()].abs().max()) == 0.0
    assert torch.equal(bucket.views[1], torch.full_like(params[1], 2.0))
