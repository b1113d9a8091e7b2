"""Training loops: they run, accumulate, schedule, checkpoint and resume."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def test_train_parsenet_two_epochs(gpu, tmp_path):
    from parsenet_codebase_amd.trainer import TrainConfig, build_parsenet, train_parsenet
    np.random.seed(0)
    torch.manual_seed(0)
    cfg = TrainConfig(num_train=2, num_val=2, num_test=2, num_points=1500, epochs=2, batch_size=1, lr=1e-3,
                      out_dir=str(tmp_path), max_steps_per_epoch=2)
    hist = train_parsenet(cfg, device=gpu, log=lambda s: None, keep_points=1200)
    assert len(hist) == 2 and all(np.isfinite(h["train_loss"]) and np.isfinite(h["test_emb"]) for h in hist)
    assert hist[0]["saved"] is not None and os.path.exists(hist[0]["saved"])
    # the checkpoint loads into a fresh model under the reference's parameter names
    model = build_parsenet(cfg, gpu)
    model.load_state_dict(torch.load(hist[0]["saved"], map_location=gpu), strict=True)
    assert os.path.exists(hist[0]["saved"].replace(".pth", "_optimizer.pth"))


def test_train_parsenet_e2e_one_step(gpu, tmp_path):
    from parsenet_codebase_amd.trainer import TrainConfig, train_parsenet_e2e
    np.random.seed(1)
    torch.manual_seed(1)
    cfg = TrainConfig(num_train=5, num_val=2, num_test=2, num_points=2500, epochs=1, batch_size=1, lr=1e-4,
                      out_dir=str(tmp_path), max_steps_per_epoch=1)
    lines = []
    hist = train_parsenet_e2e(cfg, device=gpu, log=lines.append, keep_train=2000, keep_val=2000)
    h = hist[0]
    assert h["skipped_steps"] in (0, 1)
    if h["skipped_steps"] == 0:
        assert np.isfinite(h["train_loss"]) and np.isfinite(h["train_res"])
    assert np.isfinite(h["test_res"]), lines
    assert h["saved"] is not None


@pytest.mark.parametrize("closed", [False, True])
def test_train_splinenet(gpu, tmp_path, closed):
    from parsenet_codebase_amd.trainer import TrainConfig, train_splinenet
    np.random.seed(2)
    torch.manual_seed(2)
    cfg = TrainConfig(num_train=8, num_test=4, epochs=2, batch_size=4, lr=1e-3, loss_weight=0.9,
                      out_dir=str(tmp_path), max_steps_per_epoch=2, model_path="splinenet_{}")
    hist = train_splinenet(cfg, closed=closed, device=gpu, log=lambda s: None)
    assert len(hist) == 2 and all(np.isfinite(h["train_cd"]) and np.isfinite(h["test_cd"]) for h in hist)
    assert any(h["saved"] for h in hist)


@pytest.mark.parametrize("on_device", ["0", "1"])
def test_train_parsenet_from_files(gpu, tmp_path, on_device, monkeypatch):
    """TrainConfig.dataset: the reference's four-array schema read from {train,val}_data.npz through
    data.Dataset (shuffle, augmentation, normal noise, canonicalisation as the reference's script),
    on the host or — PARSENET_DATA_ON_DEVICE=1 — with the splits resident on the GPU."""
    from parsenet_codebase_amd import synthetic
    from parsenet_codebase_amd.trainer import TrainConfig, train_parsenet
    monkeypatch.setenv("PARSENET_DATA_ON_DEVICE", on_device)
    for split, first in (("train", 0), ("val", 50)):
        pts, nrm, lab, prim = synthetic.make_batch(first, 4, 1500)
        np.savez(tmp_path / (split + "_data.npz"), points=pts * 3.0 + 1.0, normals=nrm, labels=lab, prim=prim)
    np.random.seed(5)
    torch.manual_seed(5)
    cfg = TrainConfig(num_train=4, num_val=4, num_test=4, num_points=1500, epochs=1, batch_size=2, lr=1e-3,
                      out_dir=str(tmp_path / "out"), max_steps_per_epoch=1, dataset=str(tmp_path))
    hist = train_parsenet(cfg, device=gpu, log=lambda s: None, keep_points=1200)
    assert len(hist) == 1 and np.isfinite(hist[0]["train_loss"]) and np.isfinite(hist[0]["test_emb"])


def test_train_splinenet_from_files(gpu, tmp_path):
    """File-backed SplineNet training: data.DataSetControlPointsPoisson (anisotropic canonicalisation,
    augmentation) + the rescaling of outputs, points and control grids before the losses."""
    from parsenet_codebase_amd import synthetic
    from parsenet_codebase_amd.trainer import TrainConfig, train_splinenet
    pts, ctrl = synthetic.make_spline_patches(0, 24, 900, 20, closed=False)
    np.savez(tmp_path / "open_splines.npz", points=pts * np.array([1.0, 0.7, 0.4], np.float32) + 0.3,
             controlpoints=ctrl * np.array([1.0, 0.7, 0.4], np.float32) + 0.3)
    cfg = TrainConfig(num_train=12, num_val=8, num_test=4, epochs=1, batch_size=4, lr=1e-3, loss_weight=0.9,
                      out_dir=str(tmp_path / "out"), max_steps_per_epoch=2, model_path="splinenet_{}",
                      dataset=str(tmp_path))
    cfg.split_at = (12, 20)
    np.random.seed(3)
    torch.manual_seed(3)
    hist = train_splinenet(cfg, closed=False, device=gpu, log=lambda s: None)
    assert len(hist) == 1 and np.isfinite(hist[0]["train_cd"]) and np.isfinite(hist[0]["test_cd"])


def test_device_resident_dataset(gpu):
    """SURVEY 8(f) rank 4, "augmentation on GPU": the data layer with the split resident in HBM."""
    from tests.test_host_logic import _device_dataset_equals_host
    _device_dataset_equals_host(gpu, 3e-5)
