"""Differentiable fitting stage of the ParSeNet hot path: least squares, custom-gradient SVD,
membership normalisation, segment matching, PCA standardisation, weighted primitive fits,
SplineNet forward wrappers, residual losses and the end-to-end driver.

Same names, arguments and return values as the reference modules
  src/fitting_utils.py, src/primitive_forward.py, src/fitting_optimization.py,
  src/primitives.py, src/residual_utils.py, src/segment_utils.py (to_one_hot, relaxed_iou_fast)
with the numeric guards preserved (clamps, ridge search, minimum sizes).  Debugger traps of the
reference (ipdb) are exceptions here.  Everything runs on the GPU; the only host work is what the
reference also does on the host (Hungarian matching, the 3x3 PCA rotation, control flow on segment
sizes)."""
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from . import kernels as K
from ._lib import h2d, require_cuda
from .bspline import sample_points_from_control_points_, uniform_knot_bspline
from .chamfer import chamfer_distance_single_shape
from .mean_shift import MeanShift

EPS = float(np.finfo(np.float32).eps)


def guard_exp(x, max_value=75, min_value=-75):
    return torch.exp(torch.clamp(x, max=max_value, min=min_value))


def guard_sqrt(x, minimum=1e-5):
    return torch.sqrt(torch.clamp(x, min=minimum))


# ---------------------------------------------------------------------------------------
# small dense algebra
# ---------------------------------------------------------------------------------------
def _tall_gram(A, B=None):
    """A^T B for tall-skinny operands (n x 3 or so) as a broadcast product + reduction.  A
    library GEMM with a 3 x 3 output and n = 2 500..5 000 in the contraction runs on a single
    workgroup (0.14 ms per call in fp64 — 8 % of a seven-segment shape's GPU time)."""
    B = A if B is None else B
    return (A.unsqueeze(2) * B.unsqueeze(1)).sum(0)


def _gram_spectrum(A):
    """Singular values (descending, fp64) and right singular vectors (columns) of a tall (n,3)
    matrix from the eigen-decomposition of its fp64 Gram matrix (HIP Jacobi kernel)."""
    Ad = A.detach().double()
    G = _tall_gram(Ad).unsqueeze(0)
    evals, evecs = K.sym3_eig(G) if A.shape[1] == 3 else _eigh_desc(G)
    return torch.sqrt(torch.clamp(evals[0], min=0.0)), evecs[0]


def _eigh_desc(G):
    w, v = torch.linalg.eigh(G)
    return w.flip(-1), v.flip(-1)


class LeastSquares:
    """src/fitting_utils.py:32-65."""

    def lstsq(self, A, Y, lamb=0.0):
        """Differentiable least squares min ||A x - Y||.  Full column rank: the QR solution of
        the reference (evaluated through the fp64 normal equations, same minimiser); otherwise
        ridge regression with the smallest lambda in {1e-6 * 10^i} restoring full rank."""
        cols = A.shape[1]
        if cols == 3:
            sv, _ = _gram_spectrum(A)
        else:
            sv = torch.linalg.svdvals(A.detach().double())
        sv_host = sv.cpu().numpy()       # one download decides finiteness and rank
        if not np.isfinite(sv_host).all():
            raise RuntimeError("lstsq: non-finite entries in the design matrix")
        if cols == int((sv_host > sv_host.max() * max(A.shape) * EPS).sum()):
            Ad, Yd = A.double(), Y.double()
            if A.shape[0] == cols:     # square (the ridge system): solve it directly
                x = torch.linalg.solve(Ad, Yd)
            else:
                x = torch.linalg.solve(_tall_gram(Ad), _tall_gram(Ad, Yd))
            return x.to(A.dtype)
        AtA = _tall_gram(A) if A.shape[0] > 8 * cols else A.transpose(1, 0) @ A
        with torch.no_grad():
            lamb = best_lambda(AtA)
        A_dash = AtA + lamb * torch.eye(cols, device=A.device)
        Y_dash = _tall_gram(A, Y) if A.shape[0] > 8 * cols else A.transpose(1, 0) @ Y
        return self.lstsq(A_dash, Y_dash, 1)


def best_lambda(A):
    """src/fitting_utils.py:68-85: smallest lambda in {1e-6 * 10^i, i < 7} for which A + lambda I has
    full numerical rank (torch.matrix_rank convention).  A is the symmetric Gram matrix of the
    caller, so the singular values of A + lambda I are its eigenvalues shifted by lambda: one
    eigen-decomposition (fp64 Jacobi kernel for 3 x 3) and one download replace the reference's up
    to seven fp32 SVDs with a host synchronisation each.  Where lambda sits within fp32 noise of
    the rank tolerance the reference's choice is decided by that noise (case
    3); here it is decided by the exact spectrum."""
    cols = A.shape[0]
    Ad = A.detach().double()
    sym = 0.5 * (Ad + Ad.t())
    ev = (K.sym3_eig(sym.unsqueeze(0))[0][0] if cols == 3 else torch.linalg.eigvalsh(sym).flip(-1)).cpu().numpy()
    lamb = 1e-6
    for _ in range(7):
        sv = np.abs(ev + lamb)
        if cols == int((sv > sv.max() * cols * EPS).sum()):
            break
        lamb *= 10
    return lamb


def svd_grad_K(S):
    """src/fitting_utils.py:394-417."""
    N = S.shape[0]
    s1, s2 = S.view((1, N)), S.view((N, 1))
    diff, plus = s2 - s1, s2 + s1
    eps = torch.full((N, N), 1e-6, device=S.device, dtype=S.dtype)
    K_neg = torch.sign(diff) * torch.max(torch.abs(diff), eps)
    ar = torch.arange(N, device=S.device)
    K_neg[ar, ar] = 1e-6
    K_neg = 1 / K_neg
    K_pos = 1 / plus
    rm_diag = torch.ones((N, N), device=S.device, dtype=S.dtype) - torch.eye(N, device=S.device, dtype=S.dtype)
    return K_neg * K_pos * rm_diag


def compute_grad_V(U, S, V, grad_V):
    """src/fitting_utils.py:385-391."""
    N = S.shape[0]
    Kmat = svd_grad_K(S)
    Sd = torch.eye(N, device=S.device, dtype=S.dtype) * S.reshape((N, 1))
    inner = Kmat.T * (V.T @ grad_V)
    inner = (inner + inner.T) / 2.0
    return 2 * U @ Sd @ inner @ V.T


class CustomSVD(torch.autograd.Function):
    """SVD of a tall matrix whose backward only propagates grad_V with guarded denominators
    (src/fitting_utils.py:420-455).  Forward through the 3x3 Gram eigen kernel for n x 3 inputs;
    singular vectors are defined up to sign, here fixed by the kernel's convention."""

    @staticmethod
    def forward(ctx, input):
        if input.shape[1] == 3 and input.is_cuda:
            S64, V64 = _gram_spectrum(input)
            S, V = S64.to(input.dtype), V64.to(input.dtype)
            U = (input @ V) / torch.clamp(S, min=1e-30)
        else:
            U, S, Vh = torch.linalg.svd(input, full_matrices=False)
            V = Vh.transpose(-2, -1)
        ctx.save_for_backward(U, S, V)
        return U, S, V

    @staticmethod
    def backward(ctx, grad_U, grad_S, grad_V):
        U, S, V = ctx.saved_tensors
        return compute_grad_V(U, S, V, grad_V)


customsvd = CustomSVD.apply


def weights_normalize(weights, bw):
    """src/fitting_utils.py:306-325: soft memberships (C,N) from centre/point dot products.
    One item of the padded batch form the training path uses (fitting_batch.weights_normalize_batch:
    exponent, column normalisation over the clusters, per-cluster min-shift / max-scale)."""
    from .fitting_batch import weights_normalize_batch
    C = weights.shape[0]
    dev = weights.device
    bwt = bw if torch.is_tensor(bw) else torch.tensor(float(bw), device=dev)
    ncl = torch.full((1,), C, dtype=torch.int64, device=dev)
    # (a 0-dim bandwidth on another device — the reference's CPU scalar next to CUDA weights — follows the weights)
    return weights_normalize_batch(weights.unsqueeze(0), bwt.to(device=dev, dtype=weights.dtype).reshape(1), ncl)[0]


def to_one_hot(target, maxx=50, device_id=0):
    """src/segment_utils.py:283-292."""
    if isinstance(target, np.ndarray):
        target = h2d(target.astype(np.int64), torch.device("cuda", device_id if device_id is not None else
                                                          torch.cuda.current_device()))
    N = target.shape[0]
    one_hot = torch.zeros((N, maxx), device=target.device)
    return one_hot.scatter_(1, target.unsqueeze(1).long(), 1)


def relaxed_iou_fast(pred, gt, max_clusters=50):
    """src/segment_utils.py:356-374: (B,N,K) one-hots -> (B,K,K) IoU matrix."""
    norms_p = torch.sum(pred, 1).unsqueeze(2)
    norms_g = torch.sum(gt, 1).unsqueeze(1)
    dots = pred.transpose(2, 1) @ gt
    return dots / (norms_p + norms_g - dots + 1e-7)


def solve_dense(cost):
    """Hungarian assignment (the reference uses lapsolver.solve_dense)."""
    return linear_sum_assignment(cost)


def _relaxed_iou_of_labels(pred_labels, target, max_clusters=50):
    """relaxed_iou_fast(one_hot(pred), one_hot(target)) for two HOST label arrays, evaluated on
    the host: the one-hot products are integer counts (exact in fp32), so the confusion matrix
    from np.bincount followed by the same fp32 expression gives the reference's matrix bit for
    bit — without uploading the labels, two launches and a synchronising download per call."""
    p = np.asarray(pred_labels).astype(np.int64).ravel()
    g = np.asarray(target).astype(np.int64).ravel()
    if p.size and (p.min() < 0 or g.min() < 0 or p.max() >= max_clusters or g.max() >= max_clusters):
        raise ValueError("labels must lie in [0, %d) (one-hot width of the reference)" % max_clusters)
    dots = np.bincount(p * max_clusters + g, minlength=max_clusters * max_clusters)
    dots = dots.reshape(max_clusters, max_clusters).astype(np.float32)
    norms_p = dots.sum(1, keepdims=True, dtype=np.float32)
    norms_g = dots.sum(0, keepdims=True, dtype=np.float32)
    return dots / (norms_p + norms_g - dots + np.float32(1e-7))


def match(target, pred_labels):
    """src/fitting_utils.py:362-376: Hungarian matching of predicted to ground-truth segments on
    the relaxed IoU of their one-hot encodings."""
    cost_ = 1.0 - _relaxed_iou_of_labels(pred_labels, target)
    rids, cids = solve_dense(cost_)
    return rids, cids, np.unique(target), np.unique(pred_labels)


# ---------------------------------------------------------------------------------------
# PCA standardisation (src/fitting_utils.py:493-590)
# ---------------------------------------------------------------------------------------
def rotation_matrix_a_to_b(A, B):
    """Rotation taking unit vector A to B (numpy, float64), built the reference's way
    (fitting_utils.py:556-577): in the basis F = [A, (B - (A.B) A)^, (B x A)^] it is the planar
    rotation G by the angle between them, R = F G F^-1; singular F ((anti)parallel vectors) gives the
    identity.  Rodrigues' formula (data.rotation_matrix_a_to_b) gives the same matrix to 1e-7 (the
    reference normalises the basis with "+ EPS") — but the float32 BITS of R decide near-ties of
    the SplineNet's kNN graph on the standardised points (that difference moves the closed-spline
    fixture by 3e-5), so the fitting stage keeps the reference's construction."""
    cos, sin = np.dot(A, B), np.linalg.norm(np.cross(B, A))
    v = B - np.dot(A, B) * A
    w = np.cross(B, A)
    Fm = np.stack([A, v / (np.linalg.norm(v) + EPS), w / (np.linalg.norm(w) + EPS)], 1)
    G = np.array([[cos, -sin, 0], [sin, cos, 0], [0, 0, 1]])
    try:
        return Fm @ G @ np.linalg.inv(Fm)
    except np.linalg.LinAlgError:
        return np.eye(3, dtype=np.float32)


def pca_torch(X):
    """Eigen-decomposition of X^T X as (S (3,2) [real, imag], U (3,3)) like torch.eig.  The 3x3
    problem is solved on the host with the same LAPACK routine (geev) the reference ends up in,
    because the SIGN of the returned eigenvector decides the canonical frame of the spline
    patches (fitting_utils.py:532-540 moves it to the host anyway)."""
    cov = (torch.transpose(X, 1, 0) @ X).detach().cpu()
    w, v = torch.linalg.eig(cov)
    S = torch.stack([w.real, w.imag], 1)
    return S, v.real


def standardize_point_torch(point, weights):
    """src/fitting_utils.py:512-553: centre on the confident points, rotate the minor PCA axis to
    +x, scale by the weighted extent.  point (n,3), weights (n,1) -> (point, std (1,3), mean (3),
    R (3,3)).  The batched stage's routine with one segment (fitting_batch.standardize_segments:
    selection mask, weighted mean, covariance on the device, the reference's host geev for the
    axis sign, extent of the weighted selected points); every caller of the reference runs it
    under no_grad (src/primitive_forward.py:39-40, :358-359), so do the results here."""
    from .fitting_batch import standardize_segments
    require_cuda(point, weights)
    pts, std, mean, R = standardize_segments(point.detach().reshape(1, -1, 3).float(),
                                             weights.detach().reshape(1, -1).float())
    return pts[0], std[0].reshape((1, 3)), mean[0], R[0]


def standardize_points_torch(points, weights):
    Points, stds, Rs, means = [], [], [], []
    for i in range(points.shape[0]):
        point, std, mean, R = standardize_point_torch(points[i], weights)
        Points.append(point)
        stds.append(std)
        means.append(mean)
        Rs.append(R)
    return torch.stack(Points, 0), stds, means, Rs


def project_to_plane(points, a, d):
    a = a.reshape((3, 1))
    a = a / torch.norm(a, 2)
    projections = points - ((points @ a).permute(1, 0) * a).permute(1, 0)
    return projections + a.transpose(1, 0) * d


def _knn_points_by_differences(points, k):
    """k nearest neighbours of 3-D points from coordinate DIFFERENCES, (n,k) nearest first.
    The kNN kernels use the reference's GEMM form |x|^2 + |y|^2 - 2 x.y (src/model.py:9-22), whose
    rounding (~1e-7 absolute) reorders neighbours once spacings reach 1e-3 — which up-sampled
    segments do; the reference's up-sampling and open3d's KD-tree work on differences, so these
    two evaluation-only helpers do too: csrc/knn3.hip (one wave per query, the k-th value by bisection;
    float64 points -> float64 distances), the broadcast + topk only beyond the kernel's segment size."""
    require_cuda(points)
    n = points.shape[0]
    f64 = points.dtype == torch.float64
    if 0 < n <= 10240 and k <= 64:
        off = h2d(np.asarray([0, n], np.int32), points.device)
        return K.knn3_ragged(points.float(), off, n, k, f64=f64).long()
    out = []
    for s0 in range(0, points.shape[0], 2048):
        d = ((points[s0:s0 + 2048].unsqueeze(1) - points.unsqueeze(0)) ** 2).sum(2)
        out.append(torch.topk(d, k, 1, largest=False)[1])
    return torch.cat(out, 0)


def up_sample_points_torch(points, times=1):
    """src/fitting_utils.py:150-164: append the centroid of the 4 nearest neighbours of every
    point."""
    for _ in range(times):
        idx = _knn_points_by_differences(points, 5)
        centers = torch.mean(points[idx[:, 1:]], 1)
        points = torch.cat([points, centers])
    return points


def up_sample_points_in_range(points, weights, a_min, a_max):
    """src/fitting_utils.py:202-219."""
    N = points.shape[0]
    if N > a_max:
        L = np.random.choice(np.arange(N), a_max, replace=False)
        return points[L], weights[L]
    while True:
        points = up_sample_points_torch(points)
        weights = torch.cat([weights, weights], 0)
        if points.shape[0] >= a_max:
            break
    L = np.random.choice(np.arange(points.shape[0]), a_max, replace=False)
    return points[L], weights[L]


def up_sample_points_torch_memory_efficient(points, times=1):
    """src/fitting_utils.py:167-189: centroid of the 5 nearest neighbours INCLUDING the point."""
    for _ in range(times):
        idx = _knn_points_by_differences(points, 5)
        points = torch.cat([points, torch.mean(points[idx], 1)])
    return points


def up_sample_points(points, times=1):
    """src/fitting_utils.py:109-130: batched (B,3,N) -> (B,3,2^times N), centroid of the 3 nearest
    (the point itself included)."""
    pts = points.detach().permute(0, 2, 1)
    for _ in range(times):
        out = []
        for b in range(pts.shape[0]):
            idx = _knn_points_by_differences(pts[b], 3)
            out.append(torch.cat([pts[b], torch.mean(pts[b][idx], 1)]))
        pts = torch.stack(out, 0)
    return pts.permute(0, 2, 1)


def one_hot_normalization(weights):
    """src/fitting_utils.py:328-333: rows -> one-hot of their arg-max."""
    w = torch.as_tensor(weights)
    return to_one_hot(torch.argmax(w, 1), w.shape[1], device_id=w.device.index if w.is_cuda else None).float()


def pca_numpy(X):
    S, U = np.linalg.eig(X.T @ X)
    return S, U


def reverse_all_transformation(point, mean, std, R):
    """src/fitting_utils.py:601-606 (numpy): undo scale, rotation, centring."""
    return (np.linalg.inv(R) @ (point * std.reshape((1, 3))).T).T + mean


def reverse_all_transformations(points, means, stds, Rs):
    return np.stack([reverse_all_transformation(points[i], means[i], stds[i], Rs[i]) for i in range(len(Rs))], 0)


def project_to_point_cloud(points, surface):
    """src/fitting_utils.py:637-643: nearest surface point of every point (arg-min of the squared
    distance, smallest index on ties) through the Chamfer nearest-neighbour kernel instead of the
    reference's (n, m, 3) broadcast.  numpy in -> numpy out, tensors stay tensors."""
    as_numpy = isinstance(points, np.ndarray)
    dev = torch.device("cuda", torch.cuda.current_device())
    P = torch.as_tensor(points, dtype=torch.float32, device=dev if as_numpy else None)
    S = torch.as_tensor(surface, dtype=torch.float32, device=P.device)
    _, idx, _, _ = K.chamfer_nn(P.unsqueeze(0), S.unsqueeze(0), True, False)
    out = torch.as_tensor(surface, device=P.device)[idx[0]]
    return out.cpu().numpy() if as_numpy else out


def up_sample_points_torch_in_range(points, a_min, a_max):
    """src/fitting_utils.py:222-237."""
    N = points.shape[0]
    if N > a_max:
        L = np.random.choice(np.arange(N), a_max, replace=False)
        return points[L]
    while True:
        points = up_sample_points_torch(points)
        if points.shape[0] >= a_max:
            break
    L = np.random.choice(np.arange(points.shape[0]), a_max, replace=False)
    return points[L]


def remove_outliers(points, viz=False, nb_neighbors=20, std_ratio=0.50):
    """src/fitting_utils.py:704-710.  The reference hands the segment to open3d 0.9.0's
    ``remove_statistical_outlier(nb_neighbors=20, std_ratio=0.5)``; open3d is a third-party
    dependency without source in the reference tree, so its published algorithm is restated:
    mean distance of every point to its ``nb_neighbors`` nearest neighbours (the point itself
    included, as its KD-tree search returns it), threshold = mean + std_ratio * std (Bessel) of
    those means, keep 0 < mean < threshold; distances and statistics in float64 like open3d.
    numpy (n,3) in -> numpy float64 (m,3) out (the reference's types); tensors stay tensors."""
    as_numpy = isinstance(points, np.ndarray)
    dev = torch.device("cuda", torch.cuda.current_device())
    P = torch.as_tensor(points, dtype=torch.float32, device=dev if as_numpy else None)
    require_cuda(P)
    n = P.shape[0]
    k = min(nb_neighbors, n)
    Pd = P.double()
    idx = _knn_points_by_differences(Pd, k)                                    # (n,k), self first
    dist = (Pd[idx] - Pd.unsqueeze(1)).norm(dim=2)                             # (n,k)
    avg = dist.mean(1)
    valid = avg > 0
    nvalid = n                                      # every search returns >= 1 neighbour
    cloud_mean = avg[valid].sum() / nvalid
    sq = ((avg[valid] - cloud_mean) ** 2).sum()
    std = torch.sqrt(sq / (nvalid - 1)) if nvalid > 1 else torch.zeros((), dtype=torch.float64, device=P.device)
    keep = valid & (avg < cloud_mean + std_ratio * std)
    out = (Pd if as_numpy else torch.as_tensor(points))[keep]
    return out.cpu().numpy() if as_numpy else out


# ---------------------------------------------------------------------------------------
# SplineNet forward wrappers (src/primitive_forward.py:34-102, 347-415)
# ---------------------------------------------------------------------------------------
def _restore(points_std, scale, R, mean):
    """undo standardisation: x * std -> R^-1 -> + mean."""
    tmp = points_std * scale.reshape((1, 3))
    tmp = torch.inverse(R) @ torch.transpose(tmp, 1, 0)
    return torch.transpose(tmp, 1, 0) + mean


def _spline_segment(input_points_, control_decoder, nu, nv, weights, wrap):
    """The spline branch of the batched fitting stage with ONE segment (fitting_batch.py: standardise,
    SplineNet, pn_bspline_eval_f32 with the de-standardisation x * std -> R^-1 -> + mean folded into
    its affine map and, for closed surfaces, the first sample row appended again).  Returns
    (samples (1, G*G or (G+1)*G, 3), control grid (1,20,20,3) in the standardised frame, the affine
    map (1,3,4))."""
    from .fitting_batch import _BSplineEval, standardize_segments
    require_cuda(input_points_, weights)
    if input_points_.shape[0] != 1:
        raise ValueError("spline forward pass: one segment per call (got a batch of %d)" % input_points_.shape[0])
    dev = input_points_.device
    nu, nv = nu.to(dev), nv.to(dev)
    w = weights.reshape(1, -1)
    pts_std, std, mean, R = standardize_segments(input_points_.detach().float(), w.detach().float())
    affine = torch.cat([torch.linalg.inv(R) * std.unsqueeze(1), mean.unsqueeze(2)], 2).contiguous()
    ctrl = control_decoder(K.transpose12(pts_std), w).reshape(1, 20, 20, 3)
    return _BSplineEval.apply(ctrl, nu, nv, affine, wrap), ctrl, affine


def forward_pass_open_spline(input_points_, control_decoder, nu, nv, viz=False, weights=None,
                             if_optimize=True):
    """src/primitive_forward.py:34-85: standardise -> SplineNet (open, 20 x 20 grid) -> evaluate on
    (nu, nv) -> de-standardise.  input_points_ (1,n,3), weights (n,1).  Returns (samples, samples)."""
    rec, ctrl, affine = _spline_segment(input_points_, control_decoder, nu, nv, weights, False)
    if if_optimize:
        grid = ctrl.reshape(1, 400, 3) @ affine[:, :, :3].transpose(1, 2) + affine[:, :, 3].unsqueeze(1)
        rec = optimize_open_spline_kronecker(rec, input_points_, grid, deform=True)
    return rec, rec


def forward_closed_splines(input_points_, control_decoder, nu, nv, viz=False, weights=None,
                           if_optimize=True):
    """src/primitive_forward.py:347-397, closed (u-periodic) variant: the first sample row is appended
    again (31 x 30 = 930 points).  Returns (samples (1,930,3), None, samples)."""
    rec, ctrl, affine = _spline_segment(input_points_, control_decoder, nu, nv, weights, True)
    if if_optimize and input_points_.shape[1] > 200:
        # :389-410: the control grid is closed in u the same way (21 x 20) and restored to the input
        # frame before the refit
        grid = torch.cat([ctrl, ctrl[:, 0:1]], 1).reshape(1, 21 * 20, 3)
        grid = grid @ affine[:, :, :3].transpose(1, 2) + affine[:, :, 3].unsqueeze(1)
        rec = optimize_close_spline_kronecker(rec, input_points_, grid)
    return rec, None, rec


# ---------------------------------------------------------------------------------------
# evaluation-only LS refit of a predicted spline (src/primitive_forward.py:153-296)
# ---------------------------------------------------------------------------------------
def boundary_parameterization(grid_u):
    """src/curve_utils.py:211-221: parameters of the four boundary curves of the unit square."""
    u = np.arange(grid_u)
    zeros, ones = np.zeros(grid_u), np.ones(grid_u)
    parameters = [np.stack([zeros, u], 1),
                  np.stack([np.arange(1, grid_u), np.zeros(grid_u - 1)], 1),
                  np.stack([np.arange(1, grid_u), np.ones(grid_u - 1) * (grid_u - 1)], 1),
                  np.stack([np.ones(grid_u - 2) * (grid_u - 1), np.arange(1, grid_u - 1)], 1)]
    return np.concatenate(parameters, 0) / (grid_u - 1)


def regular_parameterization(grid_u, grid_v):
    """src/curve_utils.py:260-268."""
    xv, yv = np.meshgrid(np.linspace(0, 1, grid_u), np.linspace(0, 1, grid_v))
    return np.concatenate([xv.transpose().reshape(-1, 1), yv.transpose().reshape(-1, 1)], 1)


def _refit_spline(control_points, size_u, size_v, input_points, up_range, subsample, new_cp_size,
                  new_degree, boundary_grid):
    """Shared body of the two optimize_*_spline_kronecker functions:
    1600 parameters (random + boundary) -> samples of the PREDICTED degree-3 surface -> Hungarian
    matching to up-sampled input points -> LS control grid (new_cp_size^2, a27) -> its 30 x 30
    regular samples.  The surface evaluation replaces geomdl 5.2.9's ``evaluate_list`` by the
    tensor-product basis it implements (sum_ij N_i(u) N_j(v) P_ij).  The reference's ARAP
    deformation (deform=True) only rewrites a local variable that is never read again, so it has
    no effect on the result and is not reproduced."""
    from .approximation import fit_bezier_surface_fit_kronecker
    from .bspline import basis_matrix, uniform_knots
    dev = input_points.device
    parameters = boundary_parameterization(boundary_grid)
    parameters = np.concatenate([np.random.random((1600 - parameters.shape[0], 2)), parameters], 0)
    ku, kv = uniform_knots(size_u, 3), uniform_knots(size_v, 3)
    bu = torch.from_numpy(basis_matrix(parameters[:, 0], size_u, 3, ku)).to(dev)
    bv = torch.from_numpy(basis_matrix(parameters[:, 1], size_v, 3, kv)).to(dev)
    ctrl = control_points.reshape(size_u, size_v, 3).double()
    samples = torch.einsum("ni,nj,ijc->nc", bu, bv, ctrl)                      # (1600,3) fp64
    inp = up_sample_points_torch_in_range(input_points, up_range[0], up_range[1])
    if subsample is not None:
        L = np.random.choice(np.arange(inp.shape[0]), subsample, replace=False)
        inp = inp[L]
    inp = inp.double()
    # (1600, M) fp64 on the GPU, from coordinate differences (the GEMM form of cdist loses the
    # digits the assignment's near-ties depend on)
    dist = torch.cdist(samples, inp, compute_mode="donot_use_mm_for_euclid_dist")
    rids, cids = solve_dense(dist.cpu().numpy())
    matched = inp[torch.from_numpy(np.asarray(cids)).to(dev)]
    ku2, kv2 = uniform_knots(new_cp_size, new_degree), uniform_knots(new_cp_size, new_degree)
    NU = torch.from_numpy(basis_matrix(parameters[:, 0], new_cp_size, new_degree, ku2)).to(dev)
    NV = torch.from_numpy(basis_matrix(parameters[:, 1], new_cp_size, new_degree, kv2)).to(dev)
    new_ctrl = fit_bezier_surface_fit_kronecker(matched, NU, NV)               # (cp,cp,3) fp64
    reg = regular_parameterization(30, 30)
    RU = torch.from_numpy(basis_matrix(reg[:, 0], new_cp_size, new_degree, ku2)).to(dev)
    RV = torch.from_numpy(basis_matrix(reg[:, 1], new_cp_size, new_degree, kv2)).to(dev)
    return torch.einsum("ni,nj,ijc->nc", RU, RV, new_ctrl).float()


def optimize_open_spline_kronecker(reconstructed_points, input_points_, control_points, new_cp_size=10,
                                   new_degree=2, deform=False):
    """src/primitive_forward.py:229-296.  control_points (1,400,3) in the input frame;
    returns the refitted surface's 30 x 30 samples (1,900,3)."""
    pts = _refit_spline(control_points[0].detach(), 20, 20, input_points_[0].detach(), (1600, 2000), 1600,
                        new_cp_size, new_degree, 20)
    return pts.unsqueeze(0)


def optimize_close_spline_kronecker(reconstructed_points, input_points_, control_points, new_cp_size=10,
                                    new_degree=3, deform=True):
    """src/primitive_forward.py:153-226.  control_points (1,420,3) (u-closed 21 x 20 grid);
    returns 31 x 30 samples (1,930,3), first row repeated."""
    pts = _refit_spline(control_points[0].detach(), 21, 20, input_points_[0].detach(), (2000, 2100), None,
                        new_cp_size, new_degree, 30)
    pts = pts.reshape((30, 30, 3))
    pts = torch.cat([pts, pts[0:1]], 0).reshape((930, 3))
    return pts.unsqueeze(0)


def _load_splinenet(model_or_path, mode):
    from .encoders import DGCNNControlPoints
    if isinstance(model_or_path, torch.nn.Module):
        net = model_or_path
    else:
        net = DGCNNControlPoints(20, num_points=10, mode=mode)
        state = torch.load(model_or_path, map_location="cpu")
        state = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state.items()}
        net.load_state_dict(state)
    net.cuda(torch.cuda.current_device())
    net.eval()
    return net


def initialize_open_spline_model(modelname, mode):
    """Checkpoint path (saved through DataParallel, ``module.`` prefixes) or a ready module."""
    return _load_splinenet(modelname, mode)


def initialize_closed_spline_model(modelname, mode):
    return _load_splinenet(modelname, mode)


# ---------------------------------------------------------------------------------------
# weighted primitive fits (src/primitive_forward.py:708-843)
# ---------------------------------------------------------------------------------------
class _PrimitiveFit(torch.autograd.Function):
    """Closed-form fit of ONE analytic primitive to a weighted cloud: the batched stage's kernels
    (csrc/fitbatch.hip: 60 weighted moments in one pass over the points, the 3 x 3 algebra of
    src/primitive_forward.py:708-843 on dual numbers in fp64, the cone's second pass) with a table
    of one segment.  W (n,) weights -> params (16,) fp64; backward: the stored Jacobian
    d params / d moments folded into the adjoint moment pass.  Points and normals are data."""

    @staticmethod
    def forward(ctx, W, P, Nrm, code):
        dev = P.device
        n = P.shape[0]
        P3, N3, W3 = P.reshape(1, n, 3).contiguous(), Nrm.reshape(1, n, 3).contiguous(), W.reshape(1, 1, n).contiguous()
        zero = torch.zeros(1, dtype=torch.int32, device=dev)
        typ = torch.full((1,), int(code), dtype=torch.int32, device=dev)
        rows = torch.full((1,), n, dtype=torch.int32, device=dev)
        partial = K.weighted_moments(P3, N3, W3, zero, zero, 1, 0.0)
        params, jac, status = K.primitive_fit(partial, typ, rows)
        cone_direct = K.cone_angle(P3, W3, zero, zero, typ, status, params, jac, 1, 0.0)
        ctx.save_for_backward(P3, N3, W3, zero, typ, params, jac, cone_direct)
        ctx.mark_non_differentiable(status)
        return params[0], status

    @staticmethod
    def backward(ctx, gparams, _gs):
        P3, N3, W3, zero, typ, params, jac, cone_direct = ctx.saved_tensors
        one = torch.ones(1, dtype=torch.float32, device=P3.device)
        gW = K.weighted_moments_bwd(P3, N3, W3, zero, zero, typ, one, gparams.reshape(1, -1).double().contiguous(), jac,
                                    params, cone_direct, 1, 0.0)
        return gW.reshape(-1), None, None, None


class Fit:
    """src/primitive_forward.py:700-843.  points (n,3), normals (n,3), weights (n,1); the results are
    differentiable with respect to the weights (the only differentiable input on every path of the
    reference: points and normals are data)."""

    def __init__(self):
        self.lstsq = LeastSquares().lstsq
        self.parameters = {}

    @staticmethod
    def _fit(code, points, normals, weights):
        require_cuda(points, weights)
        if points.requires_grad or (normals is not None and normals.requires_grad):
            raise RuntimeError("Fit: points / normals are data on this path (no gradient with respect to them); "
                               "detach them")
        nrm = points if normals is None else normals
        params, status = _PrimitiveFit.apply(weights.reshape(-1).float(), points.detach().float(),
                                             nrm.detach().float(), code)
        if int(status.item()) & 1:
            raise RuntimeError("lstsq: non-finite design matrix / no full-rank ridge system")
        return params.float()

    def fit_plane_torch(self, points, normals, weights, ids=0, show_warning=False):
        """-> unit normal a (1,3), offset d: a.x = d."""
        p = self._fit(K.PRIM_PLANE, points, normals, weights)
        return p[0:3].reshape((1, 3)), p[3]

    def fit_sphere_torch(self, points, normals, weights, ids=0, show_warning=False):
        """-> centre (1,3), radius."""
        p = self._fit(K.PRIM_SPHERE, points, normals, weights)
        return p[0:3].reshape((1, 3)), p[3]

    def fit_cylinder_torch(self, points, normals, weights, ids=0, show_warning=False):
        """-> axis (3,1), centre (1,3) on the axis, radius."""
        p = self._fit(K.PRIM_CYLINDER, points, normals, weights)
        return p[0:3].reshape((3, 1)), p[3:6].reshape((1, 3)), p[6]

    def fit_cone_torch(self, points, normals, weights, ids=0, show_warning=False):
        """-> apex (3,1), axis (1,3), half angle; an ill-conditioned normal system gives the
        reference's null cone (apex 0, axis +x, angle 0)."""
        p = self._fit(K.PRIM_CONE, points, normals, weights)
        return p[0:3].reshape((3, 1)), p[3:6].reshape((1, 3)), p[6]


class FittingModule:
    """src/fitting_optimization.py:117-242: owns the two frozen SplineNets and a ``Fit``."""

    def __init__(self, closed_splinenet_path, open_splinenet_path):
        self.fitting = Fit()
        self.closed_splinenet_path = closed_splinenet_path
        self.open_splinenet_path = open_splinenet_path
        nu, nv = uniform_knot_bspline(20, 20, 3, 3, 30)
        self.nu = torch.from_numpy(nu.astype(np.float32))
        self.nv = torch.from_numpy(nv.astype(np.float32))
        self.open_control_decoder = initialize_open_spline_model(open_splinenet_path, 0)
        self.closed_control_decoder = initialize_closed_spline_model(closed_splinenet_path, 1)

    def forward_pass_open_spline(self, points, ids, weights, if_optimize=False):
        points = torch.unsqueeze(points, 0).detach()   # no gradient into the SplineNet encoder
        reconst_points = forward_pass_open_spline(points, self.open_control_decoder, self.nu, self.nv,
                                                  if_optimize=if_optimize, weights=weights)[1]
        self.fitting.parameters[ids] = ["open-spline", reconst_points]
        return reconst_points

    def forward_pass_closed_spline(self, points, ids, weights, if_optimize=False):
        points = torch.unsqueeze(points, 0).detach()
        reconst_points = forward_closed_splines(points, self.closed_control_decoder, self.nu, self.nv,
                                                if_optimize=if_optimize, weights=weights)[2]
        self.fitting.parameters[ids] = ["closed-spline", reconst_points]
        return reconst_points

    def forward_pass_plane(self, points, normals, weights, ids, sample_points=False):
        axis, distance = self.fitting.fit_plane_torch(points=points, normals=normals, weights=weights, ids=ids)
        self.fitting.parameters[ids] = ["plane", axis.reshape((3, 1)), distance]
        return None

    def forward_pass_cone(self, points, normals, weights, ids, sample_points=False):
        apex, axis, theta = self.fitting.fit_cone_torch(points, normals, weights=weights, ids=ids)
        self.fitting.parameters[ids] = ["cone", apex.reshape((1, 3)), axis.reshape((3, 1)), theta]
        return None

    def forward_pass_cylinder(self, points, normals, weights, ids, sample_points=False):
        a, center, radius = self.fitting.fit_cylinder_torch(points, normals, weights, ids=ids)
        self.fitting.parameters[ids] = ["cylinder", a, center, radius]
        return None

    def forward_pass_sphere(self, points, normals, weights, ids, sample_points=False):
        center, radius = self.fitting.fit_sphere_torch(points, normals, weights, ids=ids)
        self.fitting.parameters[ids] = ["sphere", center, radius]
        return None


def _mask_index(mask, device):
    """Row indices selected by a boolean numpy mask, as a device tensor."""
    return h2d(np.nonzero(np.asarray(mask))[0], device)


_CLOSED_TYPES, _OPEN_TYPES = (0, 9, 6, 7), (2, 8)
# analytic types -> FittingModule method (src/primitive_forward.py:1000-1018)
_ANALYTIC = {1: "forward_pass_plane", 3: "forward_pass_cone", 4: "forward_pass_cylinder", 5: "forward_pass_sphere"}
# evaluation mode re-samples a spline segment into the range its SplineNet was trained on (:989-996, :1030-1036)
_RESAMPLE = {"closed": (1400, 1800), "open": (1000, 1500)}


def fit_one_shape_torch(data, fitter, weights, bw, eval=False, sample_points=False, if_optimize=False,
                        if_visualize=False):
    """src/primitive_forward.py:925-1047, one matched segment after the other (the training loops use
    the stage-wise form of the same rules, fitting_batch.build_segment_table).
    Training mode: the segment's weight column over ALL points of the shape; every 2nd point is kept,
    every 4th for analytic primitives; at most 4 spline segments per shape; segments under 20 points
    (splines: 100) are recorded as None; the ground-truth modal type selects the fit.
    Evaluation mode: the segment's own points with their (hard) weights; spline segments lose their
    statistical outliers and are re-sampled into the SplineNet's range; ``if_optimize`` adds the LS
    refit (closed: only for segments of more than 200 points)."""
    if sample_points or if_visualize:
        raise NotImplementedError("sample_points / if_visualize build open3d meshes for the viewer: out of "
                                  "scope of the hot path (SURVEY section 8)")
    fitter.fitting.parameters = {}
    gt_points, reconstructed_shape = {}, []
    splines_seen = 0
    for points, normals, seg_type, gpoints, segment_indices, (part_index, label_index) in data:
        seg_type = int(seg_type)
        kind = "closed" if seg_type in _CLOSED_TYPES else "open" if seg_type in _OPEN_TYPES else "analytic"
        if kind == "analytic" and seg_type not in _ANALYTIC:
            raise ValueError("unknown primitive type %r" % (seg_type,))
        size_in = points.shape[0]
        if eval:
            weight = weights[_mask_index(segment_indices, weights.device), part_index:part_index + 1] + EPS
            keep = None
        else:
            weight = weights[:, part_index:part_index + 1] + EPS
            keep = slice(0, None, 2 if kind != "analytic" else 4)
        dropped = False
        if not eval and kind != "analytic":
            splines_seen += 1
            dropped = splines_seen > 4                          # memory guard of the reference (:957-963)
        if keep is not None and not dropped:
            points, normals, weight = points[keep], normals[keep], weight[keep]
        if dropped or points.shape[0] < (20 if kind == "analytic" else 100):
            reconstructed_shape.append(None)
            gt_points[label_index] = None
            fitter.fitting.parameters[label_index] = None
            continue
        if kind == "analytic":
            rec = getattr(fitter, _ANALYTIC[seg_type])(points, normals, weight, ids=label_index)
        else:
            if eval:
                size_in = points.shape[0]
                points = remove_outliers(points)
                points, weight = up_sample_points_in_range(points, weight[0:points.shape[0]], *_RESAMPLE[kind])
            if kind == "closed":
                rec = fitter.forward_pass_closed_spline(points, weights=weight, ids=label_index,
                                                        if_optimize=if_optimize and (size_in > 200))
            else:
                rec = fitter.forward_pass_open_spline(points, weights=weight, ids=label_index,
                                                      if_optimize=if_optimize)
        gt_points[label_index] = gpoints
        reconstructed_shape.append(rec)
    return gt_points, reconstructed_shape


# ---------------------------------------------------------------------------------------
# residuals (src/primitives.py:18-206)
# ---------------------------------------------------------------------------------------
class ComputePrimitiveDistance:
    def __init__(self, reduce=True, one_side=False):
        self.reduce = reduce
        self.one_side = one_side

    def _finish(self, distance, sqrt):
        if sqrt:
            distance = guard_sqrt(distance)
        return torch.mean(distance) if self.reduce else distance

    def distance_from_torus(self, points, params, sqrt=False):
        axis, center, major_radius, minor_radius = params
        axis = axis.reshape((3, 1)) / torch.norm(axis, p=2)
        c2p = points - center.reshape((1, 3))
        z_new = c2p @ axis
        x_new = guard_sqrt(torch.sum(c2p ** 2, 1, keepdim=True) - z_new ** 2)
        right = (guard_sqrt((x_new - major_radius) ** 2 + z_new ** 2) - minor_radius) ** 2
        left = (guard_sqrt((x_new + major_radius) ** 2 + z_new ** 2) - minor_radius) ** 2
        return self._finish(torch.min(right, left).squeeze(), sqrt)

    def distance_from_plane(self, points, params, sqrt=False):
        a, d = params
        return self._finish(torch.sum((points @ a.reshape((3, 1)) - d) ** 2, 1), sqrt)

    def distance_from_sphere(self, points, params, sqrt=False):
        center, radius = params
        return self._finish((torch.norm(points - center.reshape((1, 3)), p=2, dim=1) - radius) ** 2, sqrt)

    def distance_from_cylinder(self, points, params, sqrt=False):
        axis, center, radius = params
        v = points - center.reshape((1, 3))
        prj = (v @ axis.reshape((3, 1))) ** 2
        dist_from_surface = torch.clamp(torch.sum(v * v, 1) - prj[:, 0], min=1e-5)
        distance = (torch.sqrt(dist_from_surface) - radius) ** 2
        if sqrt:
            distance = guard_sqrt(distance)
        if bool(torch.isnan(distance).any()):
            raise RuntimeError("distance_from_cylinder produced NaN")
        return torch.mean(distance) if self.reduce else distance

    def distance_from_cone(self, points, params, sqrt=False):
        apex, axis, theta = params
        v = points - apex.reshape((1, 3)) + 1e-8
        mod_v = torch.norm(v, dim=1, p=2)
        alpha_x = torch.clamp((v @ axis.reshape((3, 1)))[:, 0] / (mod_v + 1e-7), min=-.999, max=0.999)
        alpha = torch.acos(alpha_x)
        dist_angle = torch.clamp(torch.abs(alpha - theta), max=3.142 / 2.0)
        return self._finish((mod_v * torch.sin(dist_angle)) ** 2, sqrt)

    def distance_from_bspline(self, points, params, sqrt=False):
        """Chamfer distance between the sampled spline (params[0][0]) and the segment's points."""
        return chamfer_distance_single_shape(params[0][0], points, one_side=self.one_side, sqrt=sqrt,
                                             reduce=self.reduce)


class ResidualLoss:
    def __init__(self, reduce=True, one_side=False):
        cp = ComputePrimitiveDistance(reduce, one_side=one_side)
        self.routines = {"torus": cp.distance_from_torus, "sphere": cp.distance_from_sphere,
                         "cylinder": cp.distance_from_cylinder, "cone": cp.distance_from_cone,
                         "plane": cp.distance_from_plane, "closed-spline": cp.distance_from_bspline,
                         "open-spline": cp.distance_from_bspline}

    def residual_loss(self, Points, parameters, sqrt=False):
        distances = {}
        for k, v in parameters.items():
            if v is None:
                continue
            distances[k] = [v[0], self.routines[v[0]](points=Points[k], params=v[1:], sqrt=sqrt)]
        return distances


# ---------------------------------------------------------------------------------------
# metrics returned by fitting_loss (src/segment_utils.py:139-255)
# ---------------------------------------------------------------------------------------
def _merge_types(p):
    p = np.array(p, copy=True)
    p[p == 0] = 9
    p[p == 6] = 9
    p[p == 7] = 9
    p[p == 8] = 2
    return p


def SIOU_matched_segments(target, pred_labels, primitives_pred, primitives, weights):
    """Segment IoU and primitive-type accuracy over Hungarian-matched segments.  (The reference
    rewrites the primitive-id arrays in place; copies are used here.)"""
    primitives, primitives_pred = _merge_types(primitives), _merge_types(primitives_pred)
    dev = weights.device.index
    rids, cids = solve_dense(1.0 - _relaxed_iou_of_labels(pred_labels, target))
    prim_hot = to_one_hot(primitives_pred, 10, dev).float()
    prim_pred = torch.max(torch.sum(prim_hot.unsqueeze(2) * weights.unsqueeze(1), 0), 0)[1].cpu().numpy()
    ious, prim_ok, pairs = [], [], []
    for r, c in zip(rids, cids):
        pi, gi = pred_labels == r, target == c
        if gi.sum() == 0 or pi.sum() == 0 or gi.sum() < 100:
            continue
        ious.append(np.sum(pi & gi) / (np.sum(pi | gi) + 1e-8))
        gt_type = primitives[gi][0]
        prim_ok.append(gt_type == prim_pred[r])
        pairs.append([gt_type, prim_pred[r]])
    return (np.mean(ious) if ious else float("nan"), np.mean(prim_ok) if prim_ok else float("nan"),
            [[rids, cids]], pairs)


# ---------------------------------------------------------------------------------------
# end-to-end driver (src/residual_utils.py:49-208, 333-378)
# ---------------------------------------------------------------------------------------
class Evaluation:
    def __init__(self, userspace=None, closed_path=None, open_path=None):
        """closed_path / open_path: SplineNet checkpoints (or ready DGCNNControlPoints modules)."""
        if closed_path is None:
            closed_path = "logs/pretrained_models/closed_spline.pth"
        if open_path is None:
            open_path = "logs/pretrained_models/open_spline.pth"
        self.res_loss = ResidualLoss()
        self.fitter = FittingModule(closed_path, open_path)
        for net in (self.fitter.closed_control_decoder, self.fitter.open_control_decoder):
            for p in net.parameters():
                p.requires_grad = False
        self.ms = MeanShift()
        # batched training path (fitting_batch.py) and counters for bench.py's ``segments_per_shape``
        self.batched = True
        self.stats = {"shapes": 0, "clusters": 0, "fitted": 0}

    def guard_mean_shift(self, embedding, quantile, iterations, kernel_type="gaussian"):
        """Re-run with a 1.2x larger quantile while more than 49 clusters come out."""
        while True:
            _, center, bandwidth, cluster_ids = self.ms.mean_shift(embedding, 10000, quantile, iterations,
                                                                   kernel_type=kernel_type)
            # labels index the centres: with <= 49 centres there cannot be more distinct labels
            if center.shape[0] > 49 and torch.unique(cluster_ids).shape[0] > 49:
                quantile *= 1.2
            else:
                break
        return center, bandwidth, cluster_ids

    def prefetch_clustering(self, embedding_b, quantile, iterations):
        """Queue the clustering of ONE shape (normalisation, bandwidth, mean-shift iterations) on
        the current stream without a host synchronisation and return a handle for
        ``fitting_loss(prefetched=[handle])``.  Lets a caller run shape b+1's iterations on a side
        stream underneath the launch- and sync-bound fitting stage of shape b."""
        emb = torch.nn.functional.normalize(embedding_b, p=2, dim=-1)
        if emb.shape[0] > 10000:      # the bandwidth would depend on the shuffle: synchronous path
            return {"emb": emb, "new_X": None, "bw": None, "flag": None}
        new_X, bw, flag = self.ms.shift_async(emb, 10000, quantile, iterations)
        return {"emb": emb, "new_X": new_X, "bw": bw, "flag": flag}

    def _clusters(self, emb_b, quantile, iterations, handle):
        if handle is not None and handle["new_X"] is not None:
            done = self.ms.finish(handle["emb"], handle["new_X"], handle["bw"], handle["flag"])
            if done is not None:
                _, center, bandwidth, cluster_ids = done
                if not (center.shape[0] > 49 and torch.unique(cluster_ids).shape[0] > 49):
                    return center, bandwidth, cluster_ids
                quantile *= 1.2          # the guard's retry, on the synchronous path
        return self.guard_mean_shift(emb_b, quantile, iterations, kernel_type="gaussian")

    def fitting_loss(self, embedding, points, normals, labels, primitives, primitives_log_prob,
                     quantile=0.125, iterations=5, lamb=1.0, debug=False, eval=False, prefetched=None):
        """embedding (B,N,128), points/normals (B,N,3) tensors; labels, primitives (B,N) integer
        arrays; primitives_log_prob (B,10,N).  Returns ([Loss, geometric mean, spline mean, s_iou,
        p_iou], [parameters, cluster ids, weights]) of the last shape, like the reference (which
        is written for B = 1).  ``prefetched``: per-shape handles of ``prefetch_clustering``.

        Training calls (not eval / debug / prefetched) go through the STAGE-WISE path by default
        (``self.batched``: fitting_batch.fitting_losses_train — same decisions and arithmetic,
        mean-shift products in the library's default arithmetic, see mean_shift.ARITH); set
        ``evaluation.batched = False`` for the segment-by-segment path.  With B > 1 every shape
        is fitted and only the last result returned (what the reference's loop leaves behind):
        use ``fitting_losses`` to get them all — a warning says so once."""
        if self.batched and eval and prefetched is None and not debug:
            # evaluation mode, stage by stage over all shapes and segments (fitting_eval.py)
            return self.fitting_losses_eval(embedding, points, normals, labels, primitives, primitives_log_prob,
                                            quantile=quantile, iterations=iterations, lamb=lamb)[-1]
        if self.batched and not eval and prefetched is None and not debug:
            if embedding.shape[0] > 1 and not getattr(self, "_warned_last_only", False):
                import warnings
                warnings.warn("Evaluation.fitting_loss with a batch of %d shapes returns the LAST shape's result "
                              "only (reference behaviour); call fitting_losses for all of them" % embedding.shape[0])
                self._warned_last_only = True
            return self.fitting_losses(embedding, points, normals, labels, primitives, primitives_log_prob,
                                       quantile=quantile, iterations=iterations, lamb=lamb)[-1]
        batch_size = embedding.shape[0]
        if prefetched is None:
            embedding = torch.nn.functional.normalize(embedding, p=2, dim=2)
        else:
            embedding = [h["emb"] for h in prefetched]
        prim_pred = torch.max(primitives_log_prob, 1)[1].data.cpu().numpy()
        labels = np.asarray(labels)
        primitives = np.asarray(primitives)
        loss = parameters = cluster_ids = weights = None
        for b in range(batch_size):
            center, bandwidth, cluster_ids = self._clusters(embedding[b], quantile, iterations,
                                                            prefetched[b] if prefetched is not None else None)
            weights = center @ torch.transpose(embedding[b], 1, 0)
            if not eval:
                loss, parameters, _, rows, cols, distance = self.residual_train_mode(
                    points[b], normals[b], labels[b], cluster_ids, primitives[b], weights, bandwidth, lamb=lamb)
            else:
                with torch.no_grad():
                    loss, parameters, _ = self.residual_eval_mode(
                        points[b], normals[b], labels[b], cluster_ids, primitives[b], prim_pred[b], weights,
                        bandwidth, lamb=lamb, sample_points=False, if_optimize=False)
                # in the eval mode the memberships are the hard selection
                ids_np = cluster_ids.data.cpu().numpy()
                weights = to_one_hot(ids_np, np.unique(ids_np).shape[0], device_id=points.device.index).T
            with torch.no_grad():
                s_iou, p_iou, _, _ = SIOU_matched_segments(labels[b], cluster_ids.data.cpu().numpy(),
                                                           prim_pred[b], primitives[b], weights.T)
            loss = loss + [s_iou, p_iou]
        return loss, [parameters, cluster_ids.data.cpu().numpy(), weights]

    def fitting_losses(self, embedding, points, normals, labels, primitives, primitives_log_prob,
                       quantile=0.125, iterations=5, lamb=1.0, defer_metrics=False):
        """Training-mode ``fitting_loss`` of EVERY shape of the batch (the reference's function
        returns the last shape's): a list of ([Loss, geometric mean, spline mean, s_iou, p_iou],
        [parameters, cluster ids, weights]), computed stage by stage over all shapes and segments
        (fitting_batch.py) instead of shape by shape and segment by segment.
        ``defer_metrics``: return (losses (B,) on the device, finish) instead; ``finish()`` downloads
        the metrics and returns the list — a training loop calls it after ``backward()`` has been
        queued, so that the device does not idle while the host waits for numbers it only logs."""
        from .fitting_batch import fitting_losses_train
        require_cuda(embedding, points, normals)
        return fitting_losses_train(self, embedding, points, normals, labels, primitives, primitives_log_prob,
                                    quantile, iterations, lamb, defer_metrics)

    def fitting_losses_eval(self, embedding, points, normals, labels, primitives, primitives_log_prob,
                            quantile=0.125, iterations=5, lamb=1.0, if_optimize=False):
        """Evaluation-mode ``fitting_loss`` (src/residual_utils.py:210-331: hard memberships, the modal
        predicted type, outlier removal and re-sampling of spline segments, sqrt residuals; ``if_optimize``:
        the LS refit of src/primitive_forward.py:153-296) of EVERY shape of the batch, stage by stage over
        all shapes and segments (fitting_eval.py).  A list of ([Loss, geometric mean, spline mean, s_iou,
        p_iou], [parameters, cluster ids, weights])."""
        from .fitting_eval import fitting_losses_eval
        require_cuda(embedding, points, normals)
        return fitting_losses_eval(self, embedding, points, normals, labels, primitives, primitives_log_prob,
                                   quantile, iterations, lamb, if_optimize)

    def fitting_losses_pipelined(self, embedding, points, normals, labels, primitives, primitives_log_prob,
                                 quantile=0.125, iterations=5, lamb=1.0, chunks=2):
        """``fitting_losses(defer_metrics=True)`` with the batch cut into ``chunks`` groups whose
        clustering is queued up front, so that the device works on the next group while the host
        matches the previous one (fitting_batch.fitting_losses_train_pipelined).  Returns
        (losses (B,) on the device, finish)."""
        from .fitting_batch import fitting_losses_train_pipelined
        require_cuda(embedding, points, normals)
        return fitting_losses_train_pipelined(self, embedding, points, normals, labels, primitives,
                                              primitives_log_prob, quantile, iterations, lamb, chunks)

    def residual_train_mode(self, points, normals, labels, cluster_ids, primitives, weights, bw, lamb=1.0):
        if not isinstance(cluster_ids, np.ndarray):
            cluster_ids = cluster_ids.data.cpu().numpy()
        rows, cols, unique_target, unique_pred = match(labels, cluster_ids)
        data = []
        for index, i in enumerate(unique_pred):
            gt_indices_i = labels == cols[i]
            pred_indices_i = cluster_ids == i
            if (np.sum(gt_indices_i) == 0) or (np.sum(pred_indices_i) == 0):
                continue
            # modal ground-truth primitive type of the matched segment (smallest on ties)
            seg_type = int(np.bincount(primitives[gt_indices_i].astype(np.int64)).argmax())
            gi = h2d(np.nonzero(gt_indices_i)[0], points.device)
            data.append([points, normals, seg_type, points[gi], None, (index, i)])
        w = torch.transpose(weights_normalize(weights, float(bw)), 1, 0)
        gt_points, recon_points = fit_one_shape_torch(data, self.fitter, w, bw, eval=False)
        distance = self.res_loss.residual_loss(gt_points, self.fitter.fitting.parameters)
        Loss = self.separate_losses(distance, gt_points, lamb=lamb)
        return Loss, self.fitter.fitting.parameters, None, rows, cols, distance

    def residual_eval_mode(self, points, normals, labels, cluster_ids, primitives, pred_primitives, weights,
                           bw, lamb=1.0, sample_points=False, if_optimize=False, if_visualize=False,
                           epsilon=None):
        """src/residual_utils.py:210-331: residual error with HARD memberships.  Every predicted
        segment is fitted on its own points with the modal predicted primitive type (smallest on
        ties, scipy.stats.mode), splines after outlier removal and re-sampling; distances are
        reported with sqrt=True.  Returns (Loss, parameters, None); the mesh output of
        ``sample_points`` / ``if_visualize`` is viewer code and not provided."""
        if sample_points or if_visualize:
            raise NotImplementedError("sample_points / if_visualize build open3d meshes for the viewer: out "
                                      "of scope of the hot path (SURVEY section 8)")
        if not isinstance(cluster_ids, np.ndarray):
            cluster_ids = cluster_ids.data.cpu().numpy()
        labels = np.asarray(labels)
        pred_primitives = np.asarray(pred_primitives)
        dev = points.device
        rows, cols, unique_target, unique_pred = match(labels, cluster_ids)
        data = []
        for index, i in enumerate(unique_pred):
            gt_indices_i = labels == cols[index]
            pred_indices_i = cluster_ids == i
            if (np.sum(gt_indices_i) == 0) or (np.sum(pred_indices_i) == 0):
                continue
            seg_type = int(np.bincount(pred_primitives[pred_indices_i].astype(np.int64)).argmax())
            pi = _mask_index(pred_indices_i, dev)
            gi = _mask_index(gt_indices_i, dev)
            data.append([points[pi], normals[pi], seg_type, points[gi], pred_indices_i, (index, i)])
        # hard memberships: one-hot of the cluster ids, pushed through the same normalisation and
        # arg-max as the reference does (a fixed point for one-hot input, kept for fidelity)
        w_first = to_one_hot(cluster_ids, np.unique(cluster_ids).shape[0], device_id=dev.index).T
        w = torch.transpose(weights_normalize(w_first, float(bw)), 1, 0)
        w = to_one_hot(torch.max(w, 1)[1], w.shape[1], device_id=dev.index)
        gt_points, recon_points = fit_one_shape_torch(data, self.fitter, w, bw, eval=True,
                                                      sample_points=False, if_optimize=if_optimize)
        distance = self.res_loss.residual_loss(gt_points, self.fitter.fitting.parameters, sqrt=True)
        Loss = self.separate_losses(distance, gt_points, lamb=lamb)
        return Loss, self.fitter.fitting.parameters, None

    def separate_losses(self, distance, gt_points, lamb=1.0):
        Loss, geometric_loss, spline_loss = [], [], []
        keys = [v for v in sorted(gt_points.keys()) if gt_points[v] is not None]
        # one download for all segments instead of a synchronising comparison + .item() each
        host = (torch.stack([distance[v][1].detach().reshape(()) for v in keys]).cpu().numpy()
                if keys else np.zeros(0, np.float32))
        for v, dv in zip(keys, host):
            if dv > 1:
                # most probably a degenerate case
                distance[v][1] = torch.ones(1, device=distance[v][1].device)[0] * 0.1
                dv = np.float32(0.1)
            if distance[v][0] in ["closed-spline", "open-spline"]:
                spline_loss.append(float(dv))
                Loss.append(distance[v][1] * lamb)
            else:
                geometric_loss.append(float(dv))
                Loss.append(distance[v][1])
        Loss = torch.mean(torch.stack(Loss)) if Loss else torch.zeros(1, device="cuda")
        geometric_loss = np.mean(geometric_loss) if geometric_loss else None
        spline_loss = np.mean(spline_loss) if spline_loss else None
        return [Loss, geometric_loss, spline_loss]
