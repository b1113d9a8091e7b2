"""End-to-end fitting loss (cfg5 stage: embedding -> mean-shift -> matching -> weighted fits /
SplineNets -> residuals) against the torch-CPU oracle on a synthetic shape."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _shape_with_embedding(seed, n_points=3000):
    """A synthetic shape plus a well separated 128-d embedding (segment prototype + noise)."""
    from parsenet_codebase_amd import synthetic
    pts, nrm, lab, prim = synthetic.make_shape(seed, n_points, min_segments=4, max_segments=5)
    g = torch.Generator().manual_seed(seed)
    S = int(lab.max()) + 1
    proto = torch.nn.functional.normalize(torch.randn(S, 128, generator=g), dim=1)
    emb = proto[torch.from_numpy(lab)] + 0.15 * torch.randn(n_points, 128, generator=g) / np.sqrt(128)
    return (torch.from_numpy(pts), torch.from_numpy(nrm), lab, prim, emb)


def test_fitting_loss_forward_backward(gpu):
    from oracle import cbind, ref_fitting as RF, ref_torch as R
    from parsenet_codebase_amd.encoders import DGCNNControlPoints
    from parsenet_codebase_amd.fitting import Evaluation
    torch.cuda.set_device(gpu)
    pts, nrm, lab, prim, emb = _shape_with_embedding(7)
    torch.manual_seed(0)
    open_r, closed_r = R.DGCNNControlPoints(20, 10, 0), R.DGCNNControlPoints(20, 10, 1)
    open_g, closed_g = DGCNNControlPoints(20, 10, 0), DGCNNControlPoints(20, 10, 1)
    open_g.load_state_dict(open_r.state_dict())
    closed_g.load_state_dict(closed_r.state_dict())
    R.KNN_IMPL = lambda x, k, mode: torch.from_numpy(cbind.knn(x.detach().numpy(), k, mode))
    try:
        er = emb.clone().requires_grad_(True)
        np.random.seed(1)
        ev_r = RF.Evaluation(closed_r, open_r)
        loss_r, (params_r, ids_r, w_r) = ev_r.fitting_loss(er.unsqueeze(0), pts.unsqueeze(0), nrm.unsqueeze(0),
                                                           lab[None], prim[None], quantile=0.025,
                                                           iterations=10, lamb=0.1)
        loss_r[0].backward()
    finally:
        R.KNN_IMPL = None
    eg = emb.to(gpu).requires_grad_(True)
    np.random.seed(1)
    ev_g = Evaluation(closed_path=closed_g, open_path=open_g)
    logp = torch.log_softmax(torch.randn(1, 10, pts.shape[0], device=gpu), 1)
    loss_g, (params_g, ids_g, w_g) = ev_g.fitting_loss(eg.unsqueeze(0), pts.to(gpu).unsqueeze(0),
                                                       nrm.to(gpu).unsqueeze(0), lab[None], prim[None], logp,
                                                       quantile=0.025, iterations=10, lamb=0.1)
    loss_g[0].backward()
    assert len(loss_g) == 5
    # same segmentation (as a partition) and the same set of fitted primitive kinds
    def canon(l):
        _, first = np.unique(l, return_index=True)
        remap = {int(v): i for i, v in enumerate(l[np.sort(first)])}
        return np.array([remap[int(v)] for v in l])
    assert np.array_equal(canon(ids_g), canon(ids_r))
    kinds_r = sorted(v[0] for v in params_r.values() if v is not None)
    kinds_g = sorted(v[0] for v in params_g.values() if v is not None)
    assert kinds_g == kinds_r
    assert abs(loss_g[0].item() - loss_r[0].item()) / abs(loss_r[0].item()) < 1e-3
    for a, b in ((loss_g[1], loss_r[1]), (loss_g[2], loss_r[2])):
        assert (a is None) == (b is None)
        if a is not None:
            assert abs(a - b) / abs(b) < 1e-3
    ga, gb = eg.grad.cpu().double().flatten(), er.grad.double().flatten()
    assert float(gb.norm()) > 0
    cos = float(ga @ gb / (ga.norm() * gb.norm()))
    assert cos > 0.99, cos


def test_side_stream_prefetch_equals_sequential(gpu):
    """The end-to-end step with the clustering of shape b+1 prefetched on a side stream must give
    the loss and the gradients of the strictly sequential step (same kernels, same RNG order)."""
    from parsenet_codebase_amd.workloads import ParsenetE2EStep
    outs = []
    for overlap in (False, True):
        np.random.seed(3)
        torch.manual_seed(3)
        step = ParsenetE2EStep(gpu, batch=3, num_points=2500, seed=4)
        step.overlap = overlap
        np.random.seed(11)
        loss = step.step()
        torch.cuda.synchronize()
        outs.append((float(loss), step.bucket.flat.clone(), float(step.last_res)))
    (l0, g0, r0), (l1, g1, r1) = outs
    assert abs(l0 - l1) <= 1e-6 * abs(l0), (l0, l1)
    assert abs(r0 - r1) <= 1e-6 * max(abs(r0), 1e-12)
    assert float((g0 - g1).abs().max()) <= 1e-5 * float(g0.abs().max())


def test_e2e_network_variant(gpu):
    """PrimitivesEmbeddingDGCNGne2e (PointNet.py:292-380): same parameters as the segmentation
    network, fitting loss inside forward."""
    from parsenet_codebase_amd import synthetic
    from src.PointNet import PrimitivesEmbeddingDGCNGn, PrimitivesEmbeddingDGCNGne2e
    from src.model import DGCNNControlPoints
    from src.residual_utils import Evaluation
    torch.manual_seed(0)
    np.random.seed(0)
    kw = dict(embedding=True, emb_size=128, primitives=True, num_primitives=10, mode=5, num_channels=6, nn_nb=40)
    base = PrimitivesEmbeddingDGCNGn(**kw)
    net = PrimitivesEmbeddingDGCNGne2e(loss_function=lambda e, p, l: (e ** 2).mean().reshape(1), **kw)
    assert list(net.state_dict().keys()) == list(base.state_dict().keys())
    net.to(gpu).eval()
    net.evaluation = Evaluation(closed_path=DGCNNControlPoints(20, num_points=10, mode=1),
                                open_path=DGCNNControlPoints(20, num_points=10, mode=0))
    pts, nrm, lab, prim = synthetic.make_batch(0, 1, 2500)
    x = torch.from_numpy(np.concatenate([pts, nrm], 2).transpose(0, 2, 1).copy()).to(gpu)
    res, emb, logp, el = net(x, lab, prim, 0.025, False)
    assert tuple(emb.shape) == (1, 128, 2500) and tuple(logp.shape) == (1, 10, 2500)
    assert np.isfinite(float(res[0][0])) and np.isfinite(float(el))
    (res[0][0] + el.mean()).backward()
