"""Module-level parity: the HIP encoders against the torch-CPU oracle modules carrying the
same state_dict, on the same kNN graphs (the oracle's knn is hooked to the C oracle, whose
arithmetic order is the documented one the kernels follow)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _hook_knn():
    from oracle import cbind, ref_torch as R

    def impl(x, k, mode):
        return torch.from_numpy(cbind.knn(x.detach().numpy(), k, mode))
    R.KNN_IMPL = impl
    return R


def _rel(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def _cloud(B, N, seed, normals=True):
    rng = np.random.RandomState(seed)
    p = rng.uniform(-0.5, 0.5, (B, 3, N)).astype(np.float32)
    if not normals:
        return torch.from_numpy(p)
    n = rng.normal(size=(B, 3, N)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    return torch.from_numpy(np.concatenate([p, n], 1))


@pytest.mark.parametrize("mode,ch", [(5, 6), (0, 3)])
def test_parsenet_forward_backward(gpu, mode, ch):
    R = _hook_knn()
    try:
        from parsenet_codebase_amd.encoders import PrimitivesEmbeddingDGCNGn
        from parsenet_codebase_amd.losses import EmbeddingLoss, primitive_loss
        torch.manual_seed(0)
        B, N, k = 2, 700, 20
        ref = R.PrimitivesEmbeddingDGCNGn(embedding=True, emb_size=128, primitives=True, num_primitives=10,
                                          loss_function=R.EmbeddingLoss(1.0).triplet_loss, mode=mode,
                                          num_channels=ch, nn_nb=k)
        # perturb the norm scales so both signs / non-trivial shifts are exercised
        with torch.no_grad():
            for m in ref.modules():
                if isinstance(m, torch.nn.GroupNorm):
                    m.weight.copy_(torch.randn_like(m.weight))
                    m.bias.copy_(0.2 * torch.randn_like(m.bias))
        hip = PrimitivesEmbeddingDGCNGn(embedding=True, emb_size=128, primitives=True, num_primitives=10,
                                        loss_function=EmbeddingLoss(1.0).triplet_loss, mode=mode,
                                        num_channels=ch, nn_nb=k)
        missing = hip.load_state_dict(ref.state_dict(), strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        hip.to(gpu)
        x = _cloud(B, N, 3, normals=(ch == 6))
        rng = np.random.RandomState(0)
        labels = rng.randint(0, 6, (B, N))
        prim = torch.from_numpy(rng.randint(0, 10, (B, N)))

        np.random.seed(7)
        e_r, p_r, l_r = ref(x, labels, True)
        loss_r = l_r.mean() + R.primitive_loss(p_r, prim)
        loss_r.backward()

        np.random.seed(7)
        e_g, p_g, l_g = hip(x.to(gpu), torch.from_numpy(labels), True)
        loss_g = l_g.mean() + primitive_loss(p_g, prim.to(gpu))
        loss_g.backward()

        assert _rel(e_g, e_r) < 1e-4
        assert _rel(p_g, p_r) < 1e-4
        assert abs(loss_g.item() - loss_r.item()) / abs(loss_r.item()) < 1e-4
        gr = dict(ref.named_parameters())
        worst = 0.0
        for name, p in hip.named_parameters():
            if gr[name].grad is None:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0
                continue
            worst = max(worst, _rel(p.grad, gr[name].grad))
        # The max over the k neighbours is a discrete choice: a near tie (relative gap < 1e-6)
        # resolved differently by two fp32 evaluation orders re-routes one gradient entry and moves
        # the weight gradient of that layer by 3e-4 .. 2e-3.  The fp32 oracle shows the same jumps
        # against its own fp64 evaluation (measured: 5e-6 without a flip, 2.9e-4 with one), so
        # the whole-network bar sits above one flip; the layer-level tests
        # (test_edgeconv_gpu.py) hold the backward to 1e-5 on identical inputs.
        assert worst < 5e-3, worst
    finally:
        R.KNN_IMPL = None


@pytest.mark.parametrize("mode", [0, 1])
def test_splinenet_forward_backward(gpu, mode):
    R = _hook_knn()
    try:
        from parsenet_codebase_amd.encoders import DGCNNControlPoints
        torch.manual_seed(1)
        # B = 8: training-mode BatchNorm over a batch of 3 is ill-conditioned (it amplifies the
        # 1e-6 layer-level differences to 1e-3); from B = 8 on the outputs agree to ~2e-5
        B, N = 8, 300
        ref = R.DGCNNControlPoints(20, num_points=10, mode=mode)
        hip = DGCNNControlPoints(20, num_points=10, mode=mode)
        hip.load_state_dict(ref.state_dict(), strict=True)
        hip.to(gpu)
        x = _cloud(B, N, 5, normals=False)
        tgt = torch.randn(B, 400, 3)
        yr = ref(x)
        ((yr - tgt) ** 2).mean().backward()
        yg = hip(x.to(gpu))
        ((yg - tgt.to(gpu)) ** 2).mean().backward()
        assert yg.shape == (B, 400, 3)
        assert _rel(yg, yr) < 1e-4
        # Whole-network gradients are ill-conditioned here: a 2e-7 relative perturbation of one
        # weight changes the ORACLE's own gradients by 10-50 % (arg-max and neighbour flips in
        # four stacked kNN/max layers, BatchNorm over 8 samples).  The strict 2e-5 gradient
        # checks live at the operator level (test_edgeconv_gpu.py); here we assert direction.
        gr = dict(ref.named_parameters())
        gmax = max(float(p.grad.abs().max()) for p in ref.parameters())
        for n, p in hip.named_parameters():
            a, b = p.grad.detach().cpu().double().flatten(), gr[n].grad.double().flatten()
            if float(b.abs().max()) < 1e-6 * gmax:
                # e.g. a conv bias feeding a training-mode BatchNorm: the true gradient is 0
                assert float(a.abs().max()) < 1e-5 * gmax, n
                continue
            cos = float((a @ b) / (a.norm() * b.norm()))
            assert cos > 0.99, (n, cos)
        # running statistics of every BatchNorm moved identically
        br = dict(ref.named_buffers())
        for n, bgpu in hip.named_buffers():
            if bgpu.dtype.is_floating_point:
                assert _rel(bgpu, br[n]) < 1e-4, n
        # eval mode with membership weights (B must be 1): the e2e use of SplineNet
        ref.eval()
        hip.eval()
        w = torch.rand(1, N)
        y1 = ref(x[:1], w)
        y2 = hip(x[:1].to(gpu), w.to(gpu))
        assert _rel(y2, y1) < 1e-5
    finally:
        R.KNN_IMPL = None
