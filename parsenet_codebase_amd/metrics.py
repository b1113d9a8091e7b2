"""Evaluation metrics of the reference's test.py:39-47, 157-185 on the HIP Chamfer kernel:
surface coverage (s-k), point coverage (p-k), their thresholds at 1 % and 2 % of the unit box,
and the symmetric Chamfer figure the paper reports; plus segment / primitive-type IoU
(``SIOU_matched_segments``) and the per-point primitive mIoU (``evaluate_miou``) re-exported."""
import numpy as np
import torch

from .chamfer import chamfer_distance_single_shape
from .fitting import SIOU_matched_segments  # noqa: F401
from .losses import evaluate_miou  # noqa: F401


def continuous_labels(labels_):
    """test.py:39-47: relabel to 0..K-1 in order of np.unique."""
    labels_ = np.asarray(labels_)
    return np.unique(labels_, return_inverse=True)[1].reshape(labels_.shape).astype(labels_.dtype)


def coverage_metrics(pred_points, points):
    """pred_points (M,3): samples of the reconstructed surfaces; points (N,3): the input cloud.
    cd1[i] = distance of input point i to the nearest sample (how well the surfaces cover the
    shape, "s"), cd2[j] = distance of sample j to the nearest input point ("p"); both with
    guard_sqrt, like test.py:157-160.  Returns the dict test.py:172-180 builds (without the IoUs)."""
    pred_points = torch.as_tensor(pred_points)
    points = torch.as_tensor(points)
    cd1 = chamfer_distance_single_shape(pred_points, points, sqrt=True, one_side=True, reduce=False)
    cd2 = chamfer_distance_single_shape(points, pred_points, sqrt=True, one_side=True, reduce=False)
    sk, pk = torch.mean(cd1).item(), torch.mean(cd2).item()
    return {"sk_1": torch.mean((cd1 < 0.01).float()).item(), "sk_2": torch.mean((cd1 < 0.02).float()).item(),
            "sk": sk, "pk_1": torch.mean((cd2 < 0.01).float()).item(),
            "pk_2": torch.mean((cd2 < 0.02).float()).item(), "pk": pk, "cd": (sk + pk) / 2.0}
