"""Fused GroupNorm(+ReLU)(+max over N) kernels against plain PyTorch fp32 ops on the same device."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("B,C,N,G,relu", [(2, 64, 1000, 4, True), (3, 512, 333, 8, True), (1, 256, 10000, 4, True),
                                          (2, 32, 77, 2, False)])
def test_group_norm_relu(gpu, B, C, N, G, relu):
    from parsenet_codebase_amd.norms import group_norm_relu
    torch.manual_seed(C + N)
    gn = torch.nn.GroupNorm(G, C).to(gpu)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(C))
        gn.bias.copy_(torch.randn(C) * 0.3)
    y = (torch.randn(B, C, N, device=gpu) * 2 + 0.5)
    w = torch.randn(B, C, N, device=gpu)
    y1 = y.clone().requires_grad_(True)
    o1 = gn(y1)
    o1 = F.relu(o1) if relu else o1
    (o1 * w).sum().backward()
    g1 = [y1.grad.clone(), gn.weight.grad.clone(), gn.bias.grad.clone()]
    gn.zero_grad()
    y2 = y.clone().requires_grad_(True)
    o2 = group_norm_relu(y2, gn, relu)
    (o2 * w).sum().backward()
    assert _rel(o2, o1) < 1e-5
    assert _rel(y2.grad, g1[0]) < 2e-5
    assert _rel(gn.weight.grad, g1[1]) < 2e-5 and _rel(gn.bias.grad, g1[2]) < 2e-5


@pytest.mark.parametrize("B,C,N,G", [(2, 1024, 2000, 8), (4, 64, 301, 2)])
def test_group_norm_relu_max(gpu, B, C, N, G):
    from parsenet_codebase_amd.norms import group_norm_relu_max
    torch.manual_seed(N)
    gn = torch.nn.GroupNorm(G, C).to(gpu)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(C))
        gn.bias.copy_(torch.randn(C) * 0.3)
    y = torch.randn(B, C, N, device=gpu)
    w = torch.randn(B, C, device=gpu)
    y1 = y.clone().requires_grad_(True)
    o1 = F.relu(gn(y1)).max(dim=2)[0]
    (o1 * w).sum().backward()
    g1 = [y1.grad.clone(), gn.weight.grad.clone(), gn.bias.grad.clone()]
    gn.zero_grad()
    y2 = y.clone().requires_grad_(True)
    o2 = group_norm_relu_max(y2, gn)
    (o2 * w).sum().backward()
    assert _rel(o2, o1) < 1e-5
    assert _rel(y2.grad, g1[0]) < 2e-5
    assert _rel(gn.weight.grad, g1[1]) < 2e-5 and _rel(gn.bias.grad, g1[2]) < 2e-5


@pytest.mark.parametrize("per_item", [False, True])
@pytest.mark.parametrize("B,C,N,G", [(4, 256, 10000, 4), (2, 64, 333, 2), (3, 1024, 2001, 8)])
def test_row_bias_inside_the_kernels_is_bit_identical_to_the_separate_addition(gpu, B, C, N, G, per_item):
    """``rowbias``: the convolution's bias (C,) / conv1's per-item global term (B,C) added at load inside the
    GroupNorm kernels (round 6) against adding it to the tensor first — the same fp32 addition: outputs, the
    gradient w.r.t. the input, gamma, beta AND the bias are bit-identical, for the plain and the max variant."""
    from parsenet_codebase_amd.norms import group_norm_relu, group_norm_relu_max
    torch.manual_seed(B * C + N)
    gn = torch.nn.GroupNorm(G, C).to(gpu)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(C))
        gn.bias.copy_(torch.randn(C) * 0.3)
    y = torch.randn(B, C, N, device=gpu) * 1.5
    bias0 = torch.randn((B, C) if per_item else (C,), device=gpu)
    for fn, wshape in ((lambda t, rb: group_norm_relu(t, gn, True, rowbias=rb), (B, C, N)),
                       (lambda t, rb: group_norm_relu_max(t, gn, rowbias=rb), (B, C))):
        w = torch.randn(wshape, device=gpu)
        res = []
        for fused in (False, True):
            gn.zero_grad()
            yy = y.clone().requires_grad_(True)
            bb = bias0.clone().requires_grad_(True)
            if fused:
                out = fn(yy, bb)
            else:
                out = fn(yy + (bb.unsqueeze(2) if per_item else bb.view(1, -1, 1)), None)
            (out * w).sum().backward()
            res.append([out.detach().clone(), yy.grad.clone(), bb.grad.clone(), gn.weight.grad.clone(),
                        gn.bias.grad.clone()])
        for a, b in zip(*res):
            assert torch.equal(a, b)


@pytest.mark.parametrize("B,C,N,G", [(4, 1024, 10000, 8), (3, 64, 301, 2)])
def test_max_variant_tail_kernels_are_bit_identical_to_the_tensor_expressions(gpu, B, C, N, G):
    """csrc/gn.hip pn_gn_max_finish_f32 / pn_gn_max_bwd_prep_f32 (round 6: the (B,C) tail of the max variant as one
    launch each way) against the ten + four tensor-library operations they replace (PARSENET_GN_MAX_TAIL=0): output
    and every gradient bit for bit, with gammas of both signs and a zero gamma."""
    from parsenet_codebase_amd import norms
    torch.manual_seed(C + N)
    gn = torch.nn.GroupNorm(G, C).to(gpu)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(C))
        gn.weight[5] = 0.0
        gn.bias.copy_(torch.randn(C) * 0.3)
    y = torch.randn(B, C, N, device=gpu) * 1.5
    w = torch.randn(B, C, device=gpu)
    res = []
    old = norms.MAX_TAIL_KERNELS
    try:
        for on in (False, True):
            norms.MAX_TAIL_KERNELS = on
            gn.zero_grad()
            yy = y.clone().requires_grad_(True)
            out = norms.group_norm_relu_max(yy, gn)
            (out * w).sum().backward()
            res.append([out.detach().clone(), yy.grad.clone(), gn.weight.grad.clone(), gn.bias.grad.clone()])
    finally:
        norms.MAX_TAIL_KERNELS = old
    for a, b in zip(*res):
        assert torch.equal(a, b)
