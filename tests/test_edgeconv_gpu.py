"""Edge features (API form) and the fused edge convolution against the torch-CPU oracle.
Forward and every gradient (input, conv weight, norm scale/shift) within 1e-5 relative —
the fused path evaluates W[xj-xi; xi] as P[j] + Q[i], so results differ from the oracle by
fp32 rounding only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _rel(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def _rand_graph(B, N, k, seed):
    g = torch.Generator().manual_seed(seed)
    idx = torch.stack([torch.stack([torch.randperm(N, generator=g)[:k] for _ in range(N)]) for _ in range(B)])
    idx[:, :, 0] = torch.arange(N)
    return idx


@pytest.mark.parametrize("B,C,N,k", [(2, 3, 50, 5), (1, 6, 300, 20), (2, 64, 200, 10), (1, 128, 97, 7),
                                     (1, 24, 64, 3)])
def test_graph_feature_fwd_bwd(gpu, B, C, N, k):
    from oracle import ref_torch as R
    from parsenet_codebase_amd import graph
    torch.manual_seed(C * N)
    x = torch.randn(B, C, N)
    idx = _rand_graph(B, N, k, 1)
    w = torch.randn(B, 2 * C, N, k)
    xr = x.clone().requires_grad_(True)
    fr = R.graph_feature(xr, idx)
    (fr * w).sum().backward()
    xg = x.to(gpu).requires_grad_(True)
    fg = graph.graph_feature(xg, idx.to(gpu))
    assert fg.shape == fr.shape
    assert torch.equal(fg.cpu(), fr)          # pure data movement + one subtraction: exact
    (fg * w.to(gpu)).sum().backward()
    assert _rel(xg.grad, xr.grad) < TOL


def _norm(kind, Cout, groups):
    if kind == "gn":
        return torch.nn.GroupNorm(groups, Cout)
    return torch.nn.BatchNorm2d(Cout)


@pytest.mark.parametrize("kind,B,C,Cout,N,k,groups,train", [
    ("gn", 2, 6, 64, 130, 8, 2, True),
    ("gn", 1, 64, 64, 333, 20, 2, True),
    ("gn", 2, 64, 128, 100, 80, 2, True),
    ("bn", 3, 3, 64, 70, 10, 0, True),
    ("bn", 2, 64, 128, 90, 10, 0, True),
    ("bn", 2, 128, 256, 64, 10, 0, True),
    ("bn", 1, 256, 512, 40, 10, 0, True),
    ("bn", 2, 64, 128, 90, 10, 0, False),
    ("gn", 1, 5, 40, 60, 6, 4, True),      # generic-width kernel
    ("gn", 2, 6, 64, 10000, 12, 2, True),  # BASELINE point count: transposed graph through the LDS histograms
    ("gn", 1, 3, 64, 16500, 4, 2, True),   # more points than LDS counters: global-atomic counting sort
])
def test_edge_conv_norm_max_fwd_bwd(gpu, kind, B, C, Cout, N, k, groups, train):
    from oracle import ref_torch as R
    from parsenet_codebase_amd import graph
    torch.manual_seed(N + Cout)
    x = torch.randn(B, C, N)
    idx = _rand_graph(B, N, k, 2)
    conv = torch.nn.Conv2d(2 * C, Cout, 1, bias=False)
    norm = _norm(kind, Cout, groups)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(Cout))      # both signs: exercises the min branch
        norm.bias.copy_(torch.randn(Cout) * 0.3)
        if kind == "bn":
            norm.running_mean.copy_(torch.randn(Cout) * 0.1)
            norm.running_var.copy_(torch.rand(Cout) + 0.5)
    import copy
    conv_g, norm_g = copy.deepcopy(conv).to(gpu), copy.deepcopy(norm).to(gpu)
    norm.train(train)
    norm_g.train(train)
    wout = torch.randn(B, Cout, N)

    xr = x.clone().requires_grad_(True)
    yr = R.edge_conv(xr, idx, conv, norm)
    (yr * wout).sum().backward()

    xg = x.to(gpu).requires_grad_(True)
    yg = graph.edge_conv_norm_max(xg, idx.to(gpu), conv_g.weight, norm_g)
    (yg * wout.to(gpu)).sum().backward()

    assert _rel(yg, yr) < TOL
    assert _rel(xg.grad, xr.grad) < 2e-5
    assert _rel(conv_g.weight.grad, conv.weight.grad) < 2e-5
    assert _rel(norm_g.weight.grad, norm.weight.grad) < 2e-5
    assert _rel(norm_g.bias.grad, norm.bias.grad) < 2e-5
    if kind == "bn" and train:
        assert _rel(norm_g.running_mean, norm.running_mean) < TOL
        assert _rel(norm_g.running_var, norm.running_var) < TOL
        assert int(norm_g.num_batches_tracked) == int(norm.num_batches_tracked)


@pytest.mark.parametrize("B,C,Cout,N,k,int32", [(2, 64, 64, 10000, 80, True), (3, 6, 64, 700, 10, False),
                                               (1, 3, 128, 16500, 4, False)])
def test_transposed_graph_prefetched_on_a_side_stream_equals_the_sequential_build(gpu, B, C, Cout, N, k, int32,
                                                                                 monkeypatch):
    """The transposed graph of an edge-conv layer is built during the forward pass on a side stream
    (graph.CSR_PREFETCH, round 6) instead of at the head of the backward: every gradient is bit-identical to the
    sequential form — int32 and int64 graphs, hub graphs, more points than LDS counters — and a second backward
    through a retained graph still works (the prefetched graph is consumed once, then rebuilt)."""
    from parsenet_codebase_amd import graph
    torch.manual_seed(N + k)
    x0 = torch.randn(B, C, N, device=gpu)
    idx = _rand_graph(B, N, k, 2).to(gpu)
    idx[:, : N // 3, 0] = 7                          # a hub: a third of the points name point 7
    if int32:
        idx = idx.int()
    conv = torch.nn.Conv2d(2 * C, Cout, 1, bias=False).to(gpu)
    norm = torch.nn.GroupNorm(2, Cout).to(gpu)
    wout = torch.randn(B, Cout, N, device=gpu)
    res = {}
    for on in (True, False):
        monkeypatch.setattr(graph, "CSR_PREFETCH", on)
        conv.zero_grad(); norm.zero_grad()
        x = x0.clone().requires_grad_(True)
        y = graph.edge_conv_norm_max(x, idx, conv.weight, norm)
        (y * wout).sum().backward(retain_graph=on)
        res[on] = [y.detach().clone(), x.grad.clone(), conv.weight.grad.clone(), norm.weight.grad.clone(),
                   norm.bias.grad.clone()]
        if on:
            x.grad = None
            (y * wout).sum().backward()              # the retained graph once more: the graph is rebuilt in place
            assert torch.equal(x.grad, res[on][1])
    torch.cuda.synchronize()
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)
