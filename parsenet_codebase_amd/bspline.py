"""Uniform clamped B-spline bases (src/loss.py:190-297, duplicated in src/approximation.py) and
surface evaluation from control grids (src/fitting_utils.py:609-622)."""
import numpy as np
import torch


def basis_function_one(degree, knot_vector, span, knot):
    """N_{span,degree}(knot): The NURBS Book, algorithm A2.4 (triangular table, zero detection)."""
    m = len(knot_vector) - 1
    if (span == 0 and knot == knot_vector[0]) or (span == m - degree - 1 and knot == knot_vector[m]):
        return 1.0
    if knot < knot_vector[span] or knot >= knot_vector[span + degree + 1]:
        return 0.0
    tab = [0.0] * (degree + span + 1)
    for j in range(degree + 1):
        if knot_vector[span + j] <= knot < knot_vector[span + j + 1]:
            tab[j] = 1.0
    for k in range(1, degree + 1):
        saved = 0.0
        if tab[0] != 0.0:
            saved = ((knot - knot_vector[span]) * tab[0]) / (knot_vector[span + k] - knot_vector[span])
        for j in range(degree - k + 1):
            left = knot_vector[span + j + 1]
            right = knot_vector[span + j + k + 1]
            if tab[j + 1] == 0.0:
                tab[j] = saved
                saved = 0.0
            else:
                temp = tab[j + 1] / (right - left)
                tab[j] = saved + (right - knot) * temp
                saved = (knot - left) * temp
    return tab[0]


def uniform_knots(num_ctrl, degree):
    return [0.0] * degree + np.arange(0, 1.01, 1 / (num_ctrl - degree)).tolist() + [1.0] * degree


def basis_matrix(params, num_ctrl, degree, knots=None):
    knots = uniform_knots(num_ctrl, degree) if knots is None else knots
    out = np.zeros((len(params), num_ctrl))
    for i, u in enumerate(params):
        for j in range(num_ctrl):
            out[i, j] = basis_function_one(degree, knots, j, u)
    return out


def uniform_knot_bspline(control_points_u, control_points_v, degree_u, degree_v, grid_size=30):
    """Basis matrices nu (grid x cu), nv (grid x cv) sampled at u = arange(0, 1, 1/grid)."""
    u = np.arange(0., 1, 1 / grid_size)
    return (basis_matrix(u, control_points_u, degree_u), basis_matrix(u, control_points_v, degree_v))


def uniform_knot_bspline_(control_points_u, control_points_v, degree_u, degree_v, grid_size=30):
    """src/approximation.py:494-514: as above, also returning the knot vectors."""
    ku, kv = uniform_knots(control_points_u, degree_u), uniform_knots(control_points_v, degree_v)
    u = np.arange(0., 1, 1 / grid_size)
    return basis_matrix(u, control_points_u, degree_u, ku), basis_matrix(u, control_points_v, degree_v, kv), ku, kv


def evaluate_surface(nu, nv, ctrl):
    """ctrl (B,cu,cv,3), nu (gu,cu), nv (gv,cv) -> (B, gu*gv, 3): nu @ P_c @ nv^T per coordinate."""
    pts = torch.einsum("ui,bijc,vj->buvc", nu, ctrl, nv)
    return pts.reshape(ctrl.shape[0], nu.shape[0] * nv.shape[0], 3)


def sample_points_from_control_points_(nu, nv, outputs, batch_size, input_size_u=20, input_size_v=20):
    """src/fitting_utils.py:609-622."""
    batch_size = outputs.shape[0]
    ctrl = outputs.reshape((batch_size, input_size_u, input_size_v, 3))
    return evaluate_surface(nu, nv, ctrl)
