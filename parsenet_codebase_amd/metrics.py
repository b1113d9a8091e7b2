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


# ---------------------------------------------------------------------------------------
# src/segment_utils.py: the remaining segmentation metrics and membership helpers
# ---------------------------------------------------------------------------------------
def mean_IOU_one_sample(pred, gt, C):
    """segment_utils.py:126-136: mean IoU of the label sets 0..C-1 (empty classes count as 1)."""
    pred, gt = np.asarray(pred), np.asarray(gt)
    eps = np.finfo(np.float32).eps
    total = 0.0
    for c in range(C):
        a, b = gt == c, pred == c
        total += (np.sum(a & b) + eps) / (np.sum(a | b) + eps)
    return total / C


def iou_segmentation(pred, gt):
    """segment_utils.py:267-280: primitive-type IoU after merging 0/6/7 -> 9 and 8 -> 2 (on copies;
    the reference rewrites its arguments in place)."""
    from .fitting import _merge_types
    return mean_IOU_one_sample(_merge_types(pred), _merge_types(gt), 6)


def matching_iou(matching, predicted_labels, labels):
    """segment_utils.py:295-324: mean IoU over the matched (predicted, ground-truth) label pairs."""
    per_shape = []
    for b in range(labels.shape[0]):
        rows, cols = matching[b]
        vals = []
        for r, c in zip(rows, cols):
            p, g = predicted_labels[b] == r, labels[b] == c
            if np.sum(g) == 0 and np.sum(p) == 0:
                continue
            vals.append(np.sum(p & g) / (np.sum(p | g) + 1e-8))
        per_shape.append(np.mean(vals))
    return np.mean(per_shape)


def SIOU(target, pred_labels):
    """fitting_utils.py:336-359: Hungarian matching on the relaxed IoU, then matching_iou."""
    from .fitting import _relaxed_iou_of_labels, solve_dense
    rids, cids = solve_dense(1.0 - _relaxed_iou_of_labels(pred_labels, target).astype(np.float64))
    return matching_iou([[rids, cids]], np.expand_dims(pred_labels, 0), np.expand_dims(target, 0))


def relaxed_iou(pred, gt, max_clusters=50):
    """segment_utils.py:327-353 (the loop form; same values as relaxed_iou_fast)."""
    from .fitting import relaxed_iou_fast
    return relaxed_iou_fast(pred, gt, max_clusters)


def primitive_type_segment(pred, weights):
    """segment_utils.py:245-253: arg-max over types of sum_n pred[n,l] * weights[n,k] (numpy)."""
    return np.argmax(np.asarray(pred).T @ np.asarray(weights), 0)


def primitive_type_segment_torch(pred, weights):
    """segment_utils.py:256-264."""
    return torch.max(pred.transpose(0, 1) @ weights, 0)[1]


def dot_product_from_cluster_centers(embedding, centers):
    return centers @ embedding.T


def cluster_prob(embedding, centers, band_width):
    """segment_utils.py:52-60 (the second, effective definition): Gaussian membership, numpy."""
    dist = 2 - 2 * centers @ embedding.T
    return np.exp(-dist / 2 / band_width) / np.sqrt(2 * np.pi * band_width)


def cluster_prob_mutual(embedding, centers, bandwidth, if_normalize=False):
    """segment_utils.py:63-76."""
    dist = np.exp(centers @ embedding.T / bandwidth)
    prob = dist / np.sum(dist, 0, keepdims=True)
    if if_normalize:
        prob = prob - np.min(prob, 1, keepdims=True)
        prob = prob / np.max(prob, 1, keepdims=True)
    return prob
