"""Least-squares control-point solve of B-spline surfaces (src/approximation.py:338-364,
fit_bezier_surface_fit_kronecker): points P (N,3) with per-point basis rows NU (N,n+1), NV (N,m+1)
-> control grid (n+1, m+1, 3) minimising || A C - P ||, A[i,:] = vec(NU[i]^T NV[i]).

The reference solves it with numpy.linalg.lstsq on the host in float64 (evaluation only, not
differentiable).  Here the 1600 x 100 design matrix never leaves the GPU: its 100 x 100 normal
equations are accumulated and Cholesky-solved in fp64 (a per-segment small dense solve — not an
MFMA problem), and the result is differentiable w.r.t. the points.  ``BSpline.basis_functions``
and ``uniform_knot_bspline_`` are provided for callers that build the basis rows."""
import numpy as np
import torch

from .bspline import basis_function_one, uniform_knot_bspline_  # noqa: F401


class BSpline:
    """The basis-evaluation part of src/approximation.py:10-70."""

    def basis_function_one(self, degree, knot_vector, span, knot):
        return basis_function_one(degree, knot_vector, span, knot)

    def basis_functions(self, param, control_points_u, control_points_v, knot_vectors_u, knot_vectors_v,
                        degree_u, degree_v):
        nu = np.array([basis_function_one(degree_u, knot_vectors_u, j, param[0])
                       for j in range(control_points_u)]).reshape(control_points_u, 1)
        nv = np.array([basis_function_one(degree_v, knot_vectors_v, j, param[1])
                       for j in range(control_points_v)]).reshape(control_points_v, 1)
        return nu, nv


def fit_bezier_surface_fit_kronecker(points, basis_u, basis_v):
    """numpy in -> numpy (float64) out like the reference; torch (GPU) in -> torch out, fp64,
    differentiable w.r.t. ``points``."""
    as_numpy = isinstance(points, np.ndarray)
    dev = torch.device("cuda", torch.cuda.current_device())
    P = torch.as_tensor(points, dtype=torch.float64, device=dev if as_numpy else None)
    dev = P.device
    if not P.is_cuda:
        raise RuntimeError("fit_bezier_surface_fit_kronecker runs on the GPU (no CPU path)")
    bu = torch.as_tensor(basis_u, dtype=torch.float64).to(dev)
    bv = torch.as_tensor(basis_v, dtype=torch.float64).to(dev)
    N, nu1 = bu.shape
    nv1 = bv.shape[1]
    A = (bu.unsqueeze(2) * bv.unsqueeze(1)).reshape(N, nu1 * nv1)     # row-wise Kronecker product
    G = A.t() @ A
    rhs = A.t() @ P.double()
    L, info = torch.linalg.cholesky_ex(G)
    if int(info) != 0:      # rank-deficient sampling: minimum-norm solution like numpy.lstsq
        C = torch.linalg.lstsq(A, P.double()).solution
    else:
        C = torch.cholesky_solve(rhs, L)
    ctrl = C.reshape(nu1, nv1, 3)
    return ctrl.detach().cpu().numpy() if as_numpy else ctrl
