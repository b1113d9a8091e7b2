"""cfg2 / cfg3 as parity cases: one full SplineNet training step of the HIP path
against the CPU oracle's restatement of train_open_splines.py:140-186 /
train_closed_control_points.py:141-176 on the same synthetic patches and the same weights."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("closed", [False, True])
def test_splinenet_training_step_matches_oracle(gpu, closed):
    import bench
    from oracle import ref_fitting as RF, ref_torch as R
    from parsenet_codebase_amd.workloads import SplineNetStep
    B = 8     # training-mode BatchNorm: see test_encoder_gpu.py for why not smaller
    step = SplineNetStep(gpu, closed=closed, batch=B, num_points=700, first_shape=3, seed=5)
    ref = R.DGCNNControlPoints(20, 10, 1 if closed else 0)
    ref.load_state_dict({k: v.cpu() for k, v in step.model.state_dict().items()}, strict=True)
    # kNN ties (a zero gap between the k-th and (k+1)-th neighbour does occur on these patches)
    # are ill-posed in the reference; the graph is pinned to the C oracle's fixed order, which
    # the HIP kernel reproduces bit for bit (case 1).  A single flipped
    # neighbour moves the output by 6 % here: BatchNorm over a batch of 8 pooled vectors.
    from oracle import cbind
    R.KNN_IMPL = lambda x, k, mode: torch.from_numpy(cbind.knn(x.detach().numpy(), k, mode))
    nu, nv = step.nu.cpu(), step.nv.cpu()
    nu_o, nv_o = RF.uniform_knot_bspline(20, 20, 3, 3, 30 if closed else 40)
    assert np.allclose(nu_o, nu.numpy(), atol=1e-6) and np.allclose(nv_o, nv.numpy(), atol=1e-6)
    loss_r, cd_r, reg_r, lap_r, out_r = bench.oracle_splinenet_step(ref, closed, step.points.cpu(),
                                                                   step.control_points.cpu(), nu, nv)
    loss_r.backward()
    R.KNN_IMPL = None

    step.bucket.zero()
    out_g = step.model(step.points)
    loss_g, cd_g, reg_g, lap_g = step.losses(out_g)
    loss_g.backward()

    # control points and Chamfer: the 1e-5 bar of BASELINE.json's metric
    assert _rel(out_g, out_r) < 1e-5 * 5, _rel(out_g, out_r)
    assert abs(cd_g.item() - cd_r.item()) / abs(cd_r.item()) < 1e-5 * 5
    assert abs(reg_g.item() - reg_r.item()) / abs(reg_r.item()) < 1e-5 * 5
    if not closed:
        assert abs(lap_g.item() - lap_r.item()) / abs(lap_r.item()) < 1e-4
    assert abs(loss_g.item() - loss_r.item()) / abs(loss_r.item()) < 1e-5 * 5
    # gradients of the head (the edge-conv layers' whole-network gradients are noise-amplified by
    # training-mode BatchNorm, see test_encoder_gpu.py): direction and size must agree
    gr = dict(ref.named_parameters())
    scale = max(float(g.grad.norm()) for g in gr.values() if g.grad is not None)
    for name, p in step.model.named_parameters():
        if p.grad is None or gr[name].grad is None:
            continue
        a, b = p.grad.detach().double().cpu().flatten(), gr[name].grad.double().flatten()
        if float(b.norm()) < 1e-3 * scale:
            # shifts in front of a training-mode BatchNorm cancel: the gradient is rounding noise
            assert float((a - b).norm()) < 1e-3 * scale, (name, float((a - b).norm()), scale)
            continue
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-300))
        assert cos > 0.999, (name, cos)
    # and the step itself runs (optimizer, bucket)
    l0 = float(step.step())
    assert np.isfinite(l0)
