"""The per-point layers' GEMM on the bf16 matrix cores (csrc/gemm_x3.hip: operands split error-free into
three bf16 pieces, fp32 accumulation) against float64 products and against the rocBLAS path it replaces
(nn.Conv1d(kernel_size=1): src/model.py:56-180, src/PointNet.py:143-289)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,M,K,N", [(1, 64, 64, 4096), (3, 70, 100, 1000), (2, 128, 1152, 700), (4, 1024, 256, 2500),
                                     (1, 33, 17, 31), (2, 512, 520, 333)])
def test_gemm_x3_is_fp32_grade(gpu, B, M, K, N):
    """Every output as close to the float64 product as an fp32 dot product is, whatever its summation order:
    the error relative to sum |w||x| stays within 3 x that of the rocBLAS fp32 product of the same operands
    (measured: rms 1.6e-7 against 2.1e-7 at K = 256, tools/dbg/gx_diag.py) — ragged rows, contraction lengths
    and point counts, with and without bias."""
    from parsenet_codebase_amd import kernels as Kn
    g = torch.Generator().manual_seed(M * K + N)
    w = (torch.randn(M, K, generator=g) * torch.rand(M, 1, generator=g) * 3).to(gpu)
    x = (torch.randn(B, K, N, generator=g) * (0.1 + torch.rand(B, K, 1, generator=g) * 5)).to(gpu)
    bias = torch.randn(M, generator=g).to(gpu)
    img = Kn.gemm_x3_weight_image(w)
    for bs in (None, bias):
        got = Kn.gemm_x3(img, M, x, bs)
        want = torch.matmul(w.double(), x.double()) + (0 if bs is None else bs.double().view(1, -1, 1))
        scale = torch.matmul(w.abs().double(), x.abs().double()) + 1e-30
        err = ((got.double() - want).abs() / scale).max().item()
        blas = torch.matmul(w, x) + (0 if bs is None else bs.view(1, -1, 1))
        err_blas = ((blas.double() - want).abs() / scale).max().item()
        assert err < max(3 * err_blas, 4 * 2.0 ** -24) and err < 1e-5, (err, err_blas)
        assert torch.equal(got, Kn.gemm_x3(img, M, x, bs))            # bit-reproducible
    # the image of w^T: the gradient w.r.t. the activations
    gy = torch.randn(B, M, N, generator=g).to(gpu)
    imgT = Kn.gemm_x3_weight_image(w, transposed=True)
    got = Kn.gemm_x3(imgT, K, gy, None)
    want = torch.matmul(w.double().t(), gy.double())
    scale = torch.matmul(w.abs().double().t(), gy.abs().double()) + 1e-30
    err = ((got.double() - want).abs() / scale).max().item()
    err_blas = ((torch.matmul(w.t(), gy).double() - want).abs() / scale).max().item()
    assert err < max(3 * err_blas, 4 * 2.0 ** -24) and err < 1e-5, (err, err_blas)


@pytest.mark.parametrize("B,M,K,N", [(4, 1024, 256, 10000), (4, 256, 512, 10000), (32, 1024, 512, 700), (3, 70, 100, 1000),
                                     (1, 33, 17, 31), (2, 512, 520, 333), (4, 128, 256, 9999)])
def test_weight_gradient_is_fp32_grade_and_bit_reproducible(gpu, B, M, K, N):
    """pn_gemm_x3_wgrad_f32: gw = sum_b gy[b] x[b]^T over the points (split over the points, fixed-order sum of the
    partial results) and the bias gradient: as close to the float64 result as the rocBLAS product, relative to
    sum |gy||x|; two calls give the same bits; ragged rows / columns / point counts."""
    from parsenet_codebase_amd import kernels as Kn
    g = torch.Generator().manual_seed(M + 3 * K + N)
    gy = (torch.randn(B, M, N, generator=g) * torch.rand(B, M, 1, generator=g) * 2).to(gpu)
    x = (torch.randn(B, K, N, generator=g) * (0.1 + torch.rand(B, K, 1, generator=g) * 5)).to(gpu)
    gw, gb = Kn.gemm_x3_wgrad(gy, x, want_bias=True)
    want = torch.einsum("bmn,bkn->mk", gy.double(), x.double())
    scale = torch.einsum("bmn,bkn->mk", gy.abs().double(), x.abs().double()) + 1e-30
    err = ((gw.double() - want).abs() / scale).max().item()
    blas = torch.bmm(gy, x.transpose(1, 2)).sum(0)
    err_blas = ((blas.double() - want).abs() / scale).max().item()
    assert err < max(3 * err_blas, 4 * 2.0 ** -24) and err < 1e-5, (err, err_blas)
    wb = gy.double().sum((0, 2))
    assert float(((gb.double() - wb).abs() / (gy.abs().double().sum((0, 2)) + 1e-30)).max()) < 1e-6
    gw2, gb2 = Kn.gemm_x3_wgrad(gy, x, want_bias=True)
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)
    assert torch.equal(Kn.gemm_x3_wgrad(gy, x), gw)


@pytest.mark.parametrize("chans", [(64, 64, 128, 256), (128, 256, 256, 512), (8,), (16, 24)])
def test_product_with_the_concatenation_without_writing_it(gpu, chans):
    """pn_gemm_x3_cat_f32: W applied to channels that lie in up to four tensors (the SplineNets' conv5 on the four
    edge-conv outputs, src/model.py:150-157) — the same bits as the product with torch.cat of them."""
    from parsenet_codebase_amd import kernels as Kn
    g = torch.Generator().manual_seed(sum(chans))
    B, N, M = 3, 1500, 96
    xs = [torch.randn(B, c, N, generator=g).to(gpu) for c in chans]
    w = torch.randn(M, sum(chans), generator=g).to(gpu)
    bias = torch.randn(M, generator=g).to(gpu)
    img = Kn.gemm_x3_weight_image(w)
    want = Kn.gemm_x3(img, M, torch.cat(xs, 1).contiguous(), bias)
    got = Kn.gemm_x3_cat(img, M, xs, bias)
    assert torch.equal(got, want)
    with pytest.raises(ValueError):
        Kn.gemm_x3_cat(img, M, [torch.randn(B, 12, N, device=gpu)])          # not a multiple of 8


def test_conv1x1_forward_and_backward_on_both_paths(gpu, monkeypatch):
    """encoders.conv1x1 above the size threshold runs on the matrix-core path; outputs and all three gradients
    agree with the rocBLAS path to fp32 noise; a weight edited in place gets a new image."""
    from parsenet_codebase_amd import encoders as E
    torch.manual_seed(3)
    conv = torch.nn.Conv1d(256, 512, 1).to(gpu)
    x = torch.randn(4, 256, 5000, device=gpu)
    gy = torch.randn(4, 512, 5000, device=gpu)
    res = {}
    for on in (True, False):
        monkeypatch.setattr(E, "GEMM_X3", on)
        monkeypatch.setattr(E, "GEMM_X3_MODE", "1" if on else "0")
        conv.zero_grad()
        xi = x.clone().requires_grad_(True)
        y = E.conv1x1(xi, conv)
        assert (y.grad_fn.name().startswith("_WeightGemmX3")) == on
        (y * gy).sum().backward()
        res[on] = (y.detach(), xi.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone())
    for a, b in zip(res[True], res[False]):
        assert float((a - b).abs().max() / b.abs().max()) < 2e-6
    monkeypatch.setattr(E, "GEMM_X3", True)
    monkeypatch.setattr(E, "GEMM_X3_MODE", "1")
    with torch.no_grad():
        conv.weight.mul_(2.0)
        y2 = E.conv1x1(x, conv) - conv.bias.view(1, -1, 1)
        y1 = res[True][0] - conv.bias.view(1, -1, 1)
    assert float((y2 - 2 * y1).abs().max() / y2.abs().max()) < 1e-6


def test_frozen_weight_images_follow_the_parameter_not_its_address(gpu, monkeypatch):
    """The image of a frozen weight is cached with the parameter object: a new module whose weight lands on the
    address of a deleted one (same shape, same version) must not see the old image; an in-place edit of a frozen
    weight (version bump) rebuilds it."""
    import gc
    from parsenet_codebase_amd import encoders as E
    monkeypatch.setattr(E, "GEMM_X3", True)
    x = torch.randn(2, 128, 6000, device=gpu)
    outs = []
    for seed in (1, 2, 3):
        torch.manual_seed(seed)
        conv = torch.nn.Conv1d(128, 512, 1, bias=False).to(gpu).requires_grad_(False)
        y = E.conv1x1(x, conv)
        ref = torch.matmul(conv.weight[:, :, 0].double(), x.double())
        assert float((y.double() - ref).abs().max() / ref.abs().max()) < 1e-6
        with torch.no_grad():
            conv.weight.mul_(-0.5)
        y2 = E.conv1x1(x, conv)
        assert float((y2.double() + 0.5 * ref).abs().max() / ref.abs().max()) < 1e-6
        outs.append(y)
        del conv, y, y2
        gc.collect()
    assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[2])


def test_default_mode_takes_frozen_weights_and_mode_1_the_trained_ones_too(gpu, monkeypatch):
    """PARSENET_GEMM_X3 unset ("frozen"): a layer whose weight is trained keeps the rocBLAS product (its rounding is
    what the pre-trained states and the whole-step parity bars are pinned to), a frozen one — the SplineNets inside
    an end-to-end step — runs on the matrix cores.  Mode "1" (measured, opt-in: encoders.py) puts the trained layers
    there as well: forward, the gradient with respect to the input, and the weight and bias gradients
    (pn_gemm_x3_wgrad_f32); results agree to fp32 noise."""
    from parsenet_codebase_amd import encoders as E
    assert E.GEMM_X3_MODE == "frozen" and E.GEMM_X3
    torch.manual_seed(5)
    conv = torch.nn.Conv1d(1152, 1024, 1).to(gpu)
    x = torch.randn(2, 1152, 5000, device=gpu, requires_grad=True)
    gy = torch.randn(2, 1024, 5000, device=gpu)
    yb = E.conv1x1(x, conv)
    assert not yb.grad_fn.name().startswith("_WeightGemmX3")
    gx0, gw0, gb0 = torch.autograd.grad(yb, (x, conv.weight, conv.bias), gy)
    monkeypatch.setattr(E, "GEMM_X3_MODE", "1")
    y = E.conv1x1(x, conv)
    assert y.grad_fn.name().startswith("_WeightGemmX3")
    gx1, gw1, gb1 = torch.autograd.grad(y, (x, conv.weight, conv.bias), gy)
    for a, b in ((y, yb), (gx1, gx0), (gw1, gw0), (gb1, gb0)):
        assert float((a - b).abs().max() / b.abs().max()) < 1e-5       # (>= 1 024 terms per output, two roundings)
    monkeypatch.setattr(E, "GEMM_X3_MODE", "frozen")
    conv.requires_grad_(False)
    yf = E.conv1x1(x, conv)
    assert yf.grad_fn.name().startswith("_WeightGemmX3")               # frozen weights: on the matrix cores
    assert torch.equal(yf, y)
