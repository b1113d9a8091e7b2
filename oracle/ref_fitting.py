"""TEST INFRASTRUCTURE ONLY — torch-CPU restatement of the reference's fitting stage
(src/fitting_utils.py, src/primitive_forward.py, src/primitives.py, src/residual_utils.py,
src/loss.py, src/approximation.py), in the reference's own operation order (QR least squares,
LAPACK SVD with the custom backward, autograd everywhere).  Checker and CPU baseline only."""
import numpy as np
import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

from . import ref_torch as R

EPS = float(np.finfo(np.float32).eps)


# ---- src/loss.py:190-297 ----------------------------------------------------------------
def basis_function_one(p, U, i, u):
    """Cox-de Boor value N_{i,p}(u) by the triangular scheme of The NURBS Book A2.4."""
    if (i == 0 and u == U[0]) or (i == len(U) - p - 2 and u == U[-1]):
        return 1.0
    if u < U[i] or u >= U[i + p + 1]:
        return 0.0
    N = [1.0 if U[i + j] <= u < U[i + j + 1] else 0.0 for j in range(p + 1)] + [0.0] * i
    for k in range(1, p + 1):
        saved = 0.0 if N[0] == 0.0 else ((u - U[i]) * N[0]) / (U[i + k] - U[i])
        for j in range(p - k + 1):
            lo, hi = U[i + j + 1], U[i + j + k + 1]
            if N[j + 1] == 0.0:
                N[j], saved = saved, 0.0
            else:
                t = N[j + 1] / (hi - lo)
                N[j], saved = saved + (hi - u) * t, (u - lo) * t
    return N[0]


def uniform_knot_bspline(cu, cv, du, dv, grid_size=30):
    u = np.arange(0., 1, 1 / grid_size)
    ku = [0.0] * du + np.arange(0, 1.01, 1 / (cu - du)).tolist() + [1.0] * du
    kv = [0.0] * dv + np.arange(0, 1.01, 1 / (cv - dv)).tolist() + [1.0] * dv
    nu = np.array([[basis_function_one(du, ku, j, t) for j in range(cu)] for t in u])
    nv = np.array([[basis_function_one(dv, kv, j, t) for j in range(cv)] for t in u])
    return nu, nv


def sample_points_from_control_points_(nu, nv, outputs, batch_size, input_size_u=20, input_size_v=20):
    """src/fitting_utils.py:609-622."""
    B = outputs.shape[0]
    ctrl = outputs.reshape((B, input_size_u, input_size_v, 3))
    pts = torch.stack([torch.stack([nu @ ctrl[b, :, :, c] @ nv.t() for c in range(3)], 2) for b in range(B)], 0)
    return pts.view(B, nu.shape[0] * nv.shape[0], 3)


# ---- src/loss.py:13-239 -------------------------------------------------------------------
def _symmetries(a):
    f = [a, torch.flip(a, (1,)), torch.flip(a, (2,)), torch.flip(a, (1, 2))]
    return f, [torch.transpose(x, 2, 1) for x in f]


def control_points_permute_reg_loss(output, control_points, grid_size):
    B = output.shape[0]
    out = output.view(B, grid_size, grid_size, 3).unsqueeze(1)
    f, t = _symmetries(control_points)
    cands = torch.stack(f + t, 0).permute(1, 0, 2, 3, 4)
    diff = ((out - cands) ** 2).sum((2, 3, 4))
    loss, index = torch.min(diff, 1)
    return loss.mean() / (grid_size * grid_size * 3), cands[np.arange(B), index]


def control_points_permute_closed_reg_loss(output, control_points, gx, gy):
    B = output.shape[0]
    out = output.view(B, gx, gy, 3).unsqueeze(1)
    cands = []
    for i in range(gy):
        rolled = torch.roll(control_points, i, 1)
        f, _ = _symmetries(rolled)
        cands.append(torch.stack(f, 0).permute(1, 0, 2, 3, 4))
    cands = torch.cat(cands, 1)
    diff = ((out - cands) ** 2).sum((2, 3, 4))
    loss, index = torch.min(diff, 1)
    return loss.mean() / (gx * gy * 3), cands[np.arange(B), index]


def spline_reconstruction_loss_one_sided(nu, nv, output, points, batch_size, grid_size, side=1):
    out = output.view(batch_size, grid_size, grid_size, 3)
    rec = sample_points_from_control_points_(nu, nv, out, batch_size, grid_size, grid_size)
    return R.chamfer_distance_one_side(rec, points.permute(0, 2, 1), side), rec


def laplacian_loss(output, gt):
    lap = torch.tensor([[0.0, 0.25, 0.0], [0.25, -1.0, 0.25], [0.0, 0.25, 0.0]])
    w = torch.zeros(3, 3, 3, 3)
    for c in range(3):
        w[c, c] = -lap
    a = F.conv2d(output.permute(0, 3, 1, 2), w, padding=1)
    b = F.conv2d(gt.permute(0, 3, 1, 2), w, padding=1)
    return ((a - b) ** 2).sum(1).mean()


# ---- src/fitting_utils.py:32-85 -----------------------------------------------------------
def best_lambda(A):
    lamb = 1e-6
    n = A.shape[0]
    for _ in range(7):
        if n == torch.linalg.matrix_rank(A + lamb * torch.eye(n)):
            break
        lamb *= 10
    return lamb


def lstsq(A, Y, lamb=0.0):
    cols = A.shape[1]
    if cols == torch.linalg.matrix_rank(A):
        q, r = torch.linalg.qr(A)
        return torch.inverse(r) @ q.transpose(1, 0) @ Y
    AtA = A.transpose(1, 0) @ A
    with torch.no_grad():
        lamb = best_lambda(AtA)
    return lstsq(AtA + lamb * torch.eye(cols), A.transpose(1, 0) @ Y, 1)


# ---- src/fitting_utils.py:385-455 ---------------------------------------------------------
def svd_grad_K(S):
    n = S.shape[0]
    diff = S.view(n, 1) - S.view(1, n)
    plus = S.view(n, 1) + S.view(1, n)
    kn = torch.sign(diff) * torch.max(diff.abs(), torch.full((n, n), 1e-6))
    kn[torch.arange(n), torch.arange(n)] = 1e-6
    return (1 / kn) * (1 / plus) * (torch.ones(n, n) - torch.eye(n))


class CustomSVD(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp):
        U, S, Vh = torch.linalg.svd(inp, full_matrices=False)
        V = Vh.transpose(-2, -1)
        ctx.save_for_backward(U, S, V)
        return U, S, V

    @staticmethod
    def backward(ctx, gU, gS, gV):
        U, S, V = ctx.saved_tensors
        inner = svd_grad_K(S).T * (V.T @ gV)
        inner = (inner + inner.T) / 2.0
        return 2 * U @ torch.diag(S) @ inner @ V.T


customsvd = CustomSVD.apply


def weights_normalize(weights, bw):
    """src/fitting_utils.py:306-325."""
    prob = torch.exp(torch.clamp(weights / (bw ** 2) / 2, min=-75, max=75))
    prob = prob / prob.sum(0, keepdim=True)
    if weights.shape[0] == 1:
        return prob
    prob = prob - prob.min(1, keepdim=True)[0]
    return prob / (prob.max(1, keepdim=True)[0] + EPS)


def to_one_hot(t, maxx=50):
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(t.astype(np.int64))
    return torch.zeros((t.shape[0], maxx)).scatter_(1, t.unsqueeze(1).long(), 1)


def relaxed_iou_fast(pred, gt):
    dots = pred.transpose(2, 1) @ gt
    return dots / (pred.sum(1).unsqueeze(2) + gt.sum(1).unsqueeze(1) - dots + 1e-7)


def match(target, pred_labels):
    """src/fitting_utils.py:362-376 (lapsolver.solve_dense -> scipy linear_sum_assignment)."""
    cost = relaxed_iou_fast(to_one_hot(pred_labels).unsqueeze(0), to_one_hot(target).unsqueeze(0))
    r, c = linear_sum_assignment(1.0 - cost.numpy()[0])
    return r, c, np.unique(target), np.unique(pred_labels)


# ---- src/fitting_utils.py:493-590 ---------------------------------------------------------
def rotation_matrix_a_to_b(A, B):
    cos, sin = np.dot(A, B), np.linalg.norm(np.cross(B, A))
    v = B - np.dot(A, B) * A
    v = v / (np.linalg.norm(v) + EPS)
    w = np.cross(B, A)
    w = w / (np.linalg.norm(w) + EPS)
    Fm = np.stack([A, v, w], 1)
    G = np.array([[cos, -sin, 0], [sin, cos, 0], [0, 0, 1]])
    return Fm @ G @ np.linalg.inv(Fm)


def standardize_point_torch(point, weights):
    hi = weights[:, 0] > 0.8
    if hi.sum() < 400:
        n = weights.shape[0]
        hi = torch.topk(weights[:, 0], n // 4 if n >= 7500 else n // 2)[1]
    wp = point[hi] * weights[hi]
    mean = wp.sum(0) / (weights[hi].sum() + EPS)
    point = point - mean
    X = point[hi]
    w, v = torch.linalg.eig(X.t() @ X)
    ev = v.real[:, torch.min(w.real, 0)[1]].detach().numpy()
    Rm = torch.from_numpy(rotation_matrix_a_to_b(ev, np.array([1, 0, 0])).astype(np.float32))
    point = (Rm @ point.t()).t()
    wp = point[hi] * weights[hi]
    std = (wp.max(0)[0] - wp.min(0)[0]).abs().reshape((1, 3)).detach()
    return point / (std + EPS), std, mean, Rm


def _restore(p, s, Rm, mean):
    return (torch.inverse(Rm) @ (p * s.reshape((1, 3))).t()).t() + mean


def forward_pass_open_spline(points_, decoder, nu, nv, weights):
    """src/primitive_forward.py:34-85 (if_optimize=False)."""
    with torch.no_grad():
        p, s, m, Rm = standardize_point_torch(points_[0], weights)
    out = decoder(p.unsqueeze(0).permute(0, 2, 1), weights.T)
    rec = sample_points_from_control_points_(nu, nv, out, 1)
    return _restore(rec[0].clone(), s, Rm, m).unsqueeze(0)


def forward_closed_splines(points_, decoder, nu, nv, weights):
    """src/primitive_forward.py:347-397 (if_optimize=False)."""
    with torch.no_grad():
        p, s, m, Rm = standardize_point_torch(points_[0], weights)
    out = decoder(p.unsqueeze(0).permute(0, 2, 1), weights.T)
    rec = sample_points_from_control_points_(nu, nv, out, 1)
    t = _restore(rec[0].clone(), s, Rm, m).reshape((30, 30, 3))
    return torch.cat([t, t[0:1]], 0).reshape((1, 930, 3))


# ---- src/primitive_forward.py:708-843 -------------------------------------------------------
def fit_plane(points, weights):
    ws = weights.sum() + EPS
    X = points - (weights * points).sum(0).reshape((1, 3)) / ws
    _, _, V = customsvd(weights * X)
    a = V[:, -1].reshape((1, 3))
    d = (weights * (a @ points.t()).t()).sum() / ws
    return a, d


def fit_sphere(points, weights):
    N = weights.shape[0]
    sw = weights.sum() + EPS
    A = 2 * (-points + (points * weights).sum(0) / sw)
    dp = weights * (points * points).sum(1, keepdim=True)
    Y = (dp - dp.sum() / sw).reshape((N, 1))
    center = -lstsq(weights * A, weights * Y, 0.01).reshape((1, 3))
    r2 = torch.clamp((weights[:, 0] * ((points - center) ** 2).sum(1)).sum() / sw, min=1e-3)
    return center, torch.sqrt(torch.clamp(r2, min=1e-5))


def fit_cylinder(points, normals, weights):
    _, _, V = customsvd(weights * normals)
    a = V[:, -1].reshape((3, 1))
    a = a / (torch.norm(a, 2) + EPS)
    prj = points - ((points @ a).t() * a).t()
    c, r = fit_sphere(prj, weights)
    return a, c, r


def fit_cone(points, normals, weights):
    N = points.shape[0]
    A = weights * normals
    Y = weights * (normals * points).sum(1).reshape((N, 1))
    if np.linalg.cond(A.detach().numpy()) > 1e5:
        return torch.zeros((1, 3)), torch.tensor([[1.0, 0.0, 0.0]]), torch.zeros(1)
    c = lstsq(A, Y, 1e-3)
    a, _ = fit_plane(normals, weights)
    if (normals @ a.t()).sum() > 0:
        a = -1 * a
    diff = F.normalize(points - c.t(), p=2, dim=1) @ a.t()
    diff = torch.clamp(diff.abs(), max=0.999)
    theta = (weights * torch.acos(diff)).sum() / (weights.sum() + EPS)
    return c, a, torch.clamp(theta, min=1e-3, max=3.142 / 2 - 1e-3)


# ---- src/primitives.py:89-206 ---------------------------------------------------------------
def distance(kind, points, params, sqrt=False):
    """src/primitives.py:89-206; sqrt=True (evaluation) takes guard_sqrt of the squared residual
    of every point before the mean."""
    fin = (lambda d: R._guard_sqrt(d).mean()) if sqrt else (lambda d: d.mean())
    if kind == "plane":
        a, d = params
        return fin(((points @ a.reshape((3, 1)) - d) ** 2).sum(1))
    if kind == "sphere":
        c, r = params
        return fin((torch.norm(points - c.reshape((1, 3)), p=2, dim=1) - r) ** 2)
    if kind == "cylinder":
        a, c, r = params
        v = points - c.reshape((1, 3))
        prj = (v @ a.reshape((3, 1))) ** 2
        ds = torch.clamp((v * v).sum(1) - prj[:, 0], min=1e-5)
        return fin((torch.sqrt(ds) - r) ** 2)
    if kind == "cone":
        apex, a, theta = params
        v = points - apex.reshape((1, 3)) + 1e-8
        mv = torch.norm(v, dim=1, p=2)
        al = torch.acos(torch.clamp((v @ a.reshape((3, 1)))[:, 0] / (mv + 1e-7), min=-.999, max=0.999))
        return fin((mv * torch.sin(torch.clamp((al - theta).abs(), max=3.142 / 2.0))) ** 2)
    return R.chamfer_distance_single_shape(params[0][0], points, sqrt=sqrt)


# ---- evaluation-mode helpers ---------------------------------------------------------------------
def remove_outliers(points, nb_neighbors=20, std_ratio=0.5):
    """src/fitting_utils.py:704-710 calls open3d 0.9.0 ``remove_statistical_outlier``.  open3d is a
    third-party dependency that is not vendored in the reference and not installed here
    (PARITY UNPINNED for this function); this is its published algorithm
    (PointCloud::RemoveStatisticalOutliers): KD-tree kNN including the query point, mean of the
    Euclidean distances in double, threshold mean + std_ratio * std (Bessel) over the points,
    keep 0 < mean < threshold.  numpy (n,3) -> numpy float64 (m,3)."""
    P = np.asarray(points, dtype=np.float64)
    n = P.shape[0]
    k = min(nb_neighbors, n)
    avg = np.empty(n)
    for s0 in range(0, n, 1024):
        d = np.sqrt(((P[s0:s0 + 1024, None, :] - P[None, :, :]) ** 2).sum(2))
        avg[s0:s0 + 1024] = np.sort(d, 1)[:, :k].mean(1)
    valid = avg > 0
    cloud_mean = avg[valid].sum() / n
    std = np.sqrt(((avg[valid] - cloud_mean) ** 2).sum() / (n - 1)) if n > 1 else 0.0
    return P[valid & (avg < cloud_mean + std_ratio * std)]


def up_sample_points_torch(points, times=1):
    """src/fitting_utils.py:150-164: append the centroid of the 4 nearest neighbours."""
    for _ in range(times):
        idx = []
        for s0 in range(0, points.shape[0], 512):
            d = ((points[s0:s0 + 512].unsqueeze(1) - points.unsqueeze(0)) ** 2).sum(2)
            idx.append(torch.topk(d, 5, 1, largest=False)[1])
        idx = torch.cat(idx, 0)
        points = torch.cat([points, points[idx[:, 1:]].mean(1)])
    return points


def up_sample_points_in_range(points, weights, a_min, a_max):
    """src/fitting_utils.py:202-219 (same numpy RNG calls)."""
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


def up_sample_points_torch_in_range(points, a_min, a_max):
    """src/fitting_utils.py:222-237."""
    N = points.shape[0]
    if N > a_max:
        return points[np.random.choice(np.arange(N), a_max, replace=False)]
    while True:
        points = up_sample_points_torch(points)
        if points.shape[0] >= a_max:
            break
    return points[np.random.choice(np.arange(points.shape[0]), a_max, replace=False)]


def _basis_rows(params, n_ctrl, degree):
    ks = [0.0] * degree + np.arange(0, 1.01, 1 / (n_ctrl - degree)).tolist() + [1.0] * degree
    return np.array([[basis_function_one(degree, ks, j, t) for j in range(n_ctrl)] for t in params])


def refit_spline(control_points, size_u, size_v, input_points, up_range, subsample, new_cp_size, new_degree,
                 boundary_grid):
    """Body shared by optimize_open_spline_kronecker / optimize_close_spline_kronecker
    (src/primitive_forward.py:153-296), numpy float64 on the host like the reference.  geomdl
    5.2.9's ``evaluate_list`` (third party, absent: PARITY UNPINNED for this function) is
    restated as the tensor-product B-spline sum it evaluates.  Consumes numpy's RNG in the
    reference's order: random parameters, up-sampling choice, (open only) the 1600-subset."""
    from scipy.optimize import linear_sum_assignment
    g = boundary_grid
    u = np.arange(g)
    bnd = np.concatenate([np.stack([np.zeros(g), u], 1), np.stack([np.arange(1, g), np.zeros(g - 1)], 1),
                          np.stack([np.arange(1, g), np.ones(g - 1) * (g - 1)], 1),
                          np.stack([np.ones(g - 2) * (g - 1), np.arange(1, g - 1)], 1)], 0) / (g - 1)
    prm = np.concatenate([np.random.random((1600 - bnd.shape[0], 2)), bnd], 0)
    ctrl = np.asarray(control_points, dtype=np.float64).reshape(size_u, size_v, 3)
    samples = np.einsum("ni,nj,ijc->nc", _basis_rows(prm[:, 0], size_u, 3), _basis_rows(prm[:, 1], size_v, 3), ctrl)
    inp = up_sample_points_torch_in_range(input_points, up_range[0], up_range[1])
    if subsample is not None:
        inp = inp[np.random.choice(np.arange(inp.shape[0]), subsample, replace=False)]
    inp = inp.numpy().astype(np.float64)
    dist = np.linalg.norm(samples[:, None] - inp[None], axis=2)
    _, cids = linear_sum_assignment(dist)
    matched = inp[cids]
    NU, NV = _basis_rows(prm[:, 0], new_cp_size, new_degree), _basis_rows(prm[:, 1], new_cp_size, new_degree)
    new_ctrl = fit_bezier_surface_fit_kronecker(matched, NU, NV)
    xv, yv = np.meshgrid(np.linspace(0, 1, 30), np.linspace(0, 1, 30))
    ru, rv = xv.transpose().reshape(-1), yv.transpose().reshape(-1)
    out = np.einsum("ni,nj,ijc->nc", _basis_rows(ru, new_cp_size, new_degree),
                    _basis_rows(rv, new_cp_size, new_degree), new_ctrl)
    return torch.from_numpy(out.astype(np.float32))


# ---- src/residual_utils.py:86-208, 333-378 ----------------------------------------------------
class Evaluation:
    def __init__(self, closed_decoder, open_decoder):
        nu, nv = uniform_knot_bspline(20, 20, 3, 3, 30)
        self.nu, self.nv = torch.from_numpy(nu.astype(np.float32)), torch.from_numpy(nv.astype(np.float32))
        self.closed, self.open = closed_decoder.eval(), open_decoder.eval()
        for net in (self.closed, self.open):
            for p in net.parameters():
                p.requires_grad = False
        self.ms = R.MeanShift()

    def guard_mean_shift(self, emb, quantile, iterations):
        while True:
            _, center, bw, ids = self.ms.mean_shift(emb, 10000, quantile, iterations)
            if torch.unique(ids).shape[0] > 49:
                quantile *= 1.2
            else:
                return center, bw, ids

    def fitting_loss(self, embedding, points, normals, labels, primitives, quantile=0.125, iterations=5,
                     lamb=1.0, eval=False, primitives_log_prob=None, if_optimize=False):
        embedding = F.normalize(embedding, p=2, dim=2)
        b = 0
        center, bw, ids = self.guard_mean_shift(embedding[b], quantile, iterations)
        weights = center @ embedding[b].t()
        if not eval:
            loss, params = self.residual_train_mode(points[b], normals[b], labels[b], ids.numpy(), primitives[b],
                                                    weights, bw, lamb)
        else:
            pred = torch.max(primitives_log_prob, 1)[1].numpy()
            with torch.no_grad():
                loss, params = self.residual_eval_mode(points[b], normals[b], labels[b], ids.numpy(), pred[b], bw,
                                                       lamb, if_optimize)
        return loss, [params, ids.numpy(), weights]

    def residual_eval_mode(self, points, normals, labels, cluster_ids, pred_primitives, bw, lamb,
                           if_optimize=False):
        """src/residual_utils.py:210-331 + src/primitive_forward.py:925-1047 (eval branch):
        hard one-hot memberships, every predicted segment fitted on its own points with the modal
        predicted type, distances with sqrt=True."""
        rows, cols, _, unique_pred = match(labels, cluster_ids)
        C = np.unique(cluster_ids).shape[0]
        onehot = to_one_hot(torch.from_numpy(cluster_ids.astype(np.int64)), C).t()          # (C,N)
        w = weights_normalize(onehot, float(bw)).t()
        w = to_one_hot(torch.max(w, 1)[1], w.shape[1])                                      # (N,C)
        params, gts = {}, {}
        for index, i in enumerate(unique_pred):
            gi, pi = labels == cols[index], cluster_ids == i
            if gi.sum() == 0 or pi.sum() == 0:
                continue
            kind = int(np.bincount(pred_primitives[pi].astype(np.int64)).argmax())
            pi_t = torch.from_numpy(np.nonzero(pi)[0])
            p, n = points[pi_t], normals[pi_t]
            weight = w[pi_t, index:index + 1] + EPS
            Z = p.shape[0]
            if p.shape[0] < 20 or (kind in (0, 2, 6, 7, 8, 9) and p.shape[0] < 100):
                params[i], gts[i] = None, None
                continue
            if kind in (0, 6, 7, 9):
                p = torch.from_numpy(remove_outliers(p.numpy()).astype(np.float32))
                weight = weight[0:p.shape[0]]
                p, weight = up_sample_points_in_range(p, weight, 1400, 1800)
                params[i] = ["closed-spline", self._closed(p, weight, if_optimize and Z > 200)]
            elif kind in (2, 8):
                p = torch.from_numpy(remove_outliers(p.numpy()).astype(np.float32))
                weight = weight[0:p.shape[0]]
                p, weight = up_sample_points_in_range(p, weight, 1000, 1500)
                params[i] = ["open-spline", self._open(p, weight, if_optimize)]
            elif kind == 1:
                a, d = fit_plane(p, weight)
                params[i] = ["plane", a.reshape((3, 1)), d]
            elif kind == 3:
                c, a, t = fit_cone(p, n, weight)
                params[i] = ["cone", c.reshape((1, 3)), a.reshape((3, 1)), t]
            elif kind == 4:
                params[i] = ["cylinder"] + list(fit_cylinder(p, n, weight))
            elif kind == 5:
                params[i] = ["sphere"] + list(fit_sphere(p, weight))
            gts[i] = points[torch.from_numpy(np.nonzero(gi)[0])]
        losses, geo, spl = [], [], []
        for v in sorted(gts.keys()):
            if gts[v] is None:
                continue
            d = distance(params[v][0], gts[v], params[v][1:], sqrt=True)
            if d > 1:
                d = torch.ones(1)[0] * 0.1
            if params[v][0] in ("closed-spline", "open-spline"):
                spl.append(d.item())
                losses.append(d * lamb)
            else:
                geo.append(d.item())
                losses.append(d)
        L = torch.stack(losses).mean() if losses else torch.zeros(1)
        return [L, np.mean(geo) if geo else None, np.mean(spl) if spl else None], params

    def _open(self, p, weight, if_optimize):
        with torch.no_grad():
            ps, s, m, Rm = standardize_point_torch(p, weight)
        out = self.open(ps.unsqueeze(0).permute(0, 2, 1), weight.T)
        rec = _restore(sample_points_from_control_points_(self.nu, self.nv, out, 1)[0].clone(), s, Rm, m)
        if if_optimize:
            ctrl = _restore(out.view(400, 3), s, Rm, m)
            rec = refit_spline(ctrl.detach().numpy(), 20, 20, p.detach(), (1600, 2000), 1600, 10, 2, 20)
        return rec.unsqueeze(0)

    def _closed(self, p, weight, if_optimize):
        with torch.no_grad():
            ps, s, m, Rm = standardize_point_torch(p, weight)
        out = self.closed(ps.unsqueeze(0).permute(0, 2, 1), weight.T)
        t = _restore(sample_points_from_control_points_(self.nu, self.nv, out, 1)[0].clone(), s, Rm, m)
        t = t.reshape((30, 30, 3))
        rec = torch.cat([t, t[0:1]], 0).reshape((930, 3))
        if if_optimize:
            ctrl = _restore(out.view(400, 3), s, Rm, m).reshape((20, 20, 3))
            ctrl = torch.cat([ctrl, ctrl[0:1]], 0)
            r = refit_spline(ctrl.detach().numpy(), 21, 20, p.detach(), (2000, 2100), None, 10, 3, 30)
            r = r.reshape((30, 30, 3))
            rec = torch.cat([r, r[0:1]], 0).reshape((930, 3))
        return rec.unsqueeze(0)

    def residual_train_mode(self, points, normals, labels, cluster_ids, primitives, weights, bw, lamb):
        rows, cols, _, unique_pred = match(labels, cluster_ids)
        w = weights_normalize(weights, float(bw)).t()
        params, gts = {}, {}
        splines = 0
        for index, i in enumerate(unique_pred):
            gi = labels == cols[i]
            if gi.sum() == 0 or (cluster_ids == i).sum() == 0:
                continue
            kind = int(np.bincount(primitives[gi].astype(np.int64)).argmax())
            weight = w[:, index:index + 1] + EPS
            p, n, weight = points[0::2], normals[0::2], weight[0::2]
            if kind in (0, 2, 6, 7, 8, 9):
                splines += 1
                if splines > 4:
                    params[i], gts[i] = None, None
                    continue
            else:
                p, n, weight = p[0::2], n[0::2], weight[0::2]
            if p.shape[0] < 20 or (kind in (0, 2, 6, 7, 8, 9) and p.shape[0] < 100):
                params[i], gts[i] = None, None
                continue
            if kind in (0, 6, 7, 9):
                params[i] = ["closed-spline", forward_closed_splines(p.unsqueeze(0).detach(), self.closed,
                                                                      self.nu, self.nv, weight)]
            elif kind in (2, 8):
                params[i] = ["open-spline", forward_pass_open_spline(p.unsqueeze(0).detach(), self.open,
                                                                     self.nu, self.nv, weight)]
            elif kind == 1:
                a, d = fit_plane(p, weight)
                params[i] = ["plane", a.reshape((3, 1)), d]
            elif kind == 3:
                c, a, t = fit_cone(p, n, weight)
                params[i] = ["cone", c.reshape((1, 3)), a.reshape((3, 1)), t]
            elif kind == 4:
                params[i] = ["cylinder"] + list(fit_cylinder(p, n, weight))
            elif kind == 5:
                params[i] = ["sphere"] + list(fit_sphere(p, weight))
            gts[i] = points[gi]
        losses, geo, spl = [], [], []
        for v in sorted(gts.keys()):
            if gts[v] is None:
                continue
            d = distance(params[v][0], gts[v], params[v][1:])
            if d > 1:
                d = torch.ones(1)[0] * 0.1
            if params[v][0] in ("closed-spline", "open-spline"):
                spl.append(d.item())
                losses.append(d * lamb)
            else:
                geo.append(d.item())
                losses.append(d)
        L = torch.stack(losses).mean() if losses else torch.zeros(1)
        return [L, np.mean(geo) if geo else None, np.mean(spl) if spl else None], params


# ---- src/approximation.py:338-364 -------------------------------------------------------------
def fit_bezier_surface_fit_kronecker(points, basis_u, basis_v):
    N = basis_u.shape[0]
    n = basis_v.shape[1] - 1
    A = np.stack([np.outer(basis_u[i], basis_v[i]).reshape(-1) for i in range(N)], 0)
    ctrl = [np.linalg.lstsq(A, points[:, c], rcond=-1)[0].reshape((n + 1, n + 1)) for c in range(3)]
    return np.stack(ctrl, 2)
