"""SplineNet training losses (src/loss.py:13-239)."""
import numpy as np
import torch
import torch.nn.functional as F

from .bspline import basis_function_one, evaluate_surface, uniform_knot_bspline  # noqa: F401
from .chamfer import chamfer_distance, chamfer_distance_one_side


def regressions_loss_per_shape(output, points):
    return torch.mean(torch.sum((output - points) ** 2, 2))


def all_permutations(array):
    """The 8 symmetries of an open control grid: (B,G,G,3) -> (B,8,G,G,3)."""
    flips = [array, torch.flip(array, (1,)), torch.flip(array, (2,)), torch.flip(array, (1, 2))]
    return torch.stack(flips + [torch.transpose(f, 2, 1) for f in flips], 1)


def all_permutations_half(array):
    """The 4 flips used for closed (u-periodic) grids: (B,G,G,3) -> (B,4,G,G,3)."""
    return torch.stack([array, torch.flip(array, (1,)), torch.flip(array, (2,)), torch.flip(array, (1, 2))], 1)


def roll(x, shift, dim=-1, fill_pad=None):
    if shift == 0:
        return x
    return torch.roll(x, shifts=shift, dims=dim)


def _min_over_candidates(output, candidates, denom):
    diff = torch.sum((output.unsqueeze(1) - candidates) ** 2, (2, 3, 4))
    loss, index = torch.min(diff, 1)
    best = candidates[torch.arange(output.shape[0], device=output.device), index]
    return torch.mean(loss) / denom, best


def control_points_permute_reg_loss(output, control_points, grid_size):
    """min over the 8 grid symmetries of the squared error; returns (loss, best matching target)."""
    batch_size = output.shape[0]
    output = output.view(batch_size, grid_size, grid_size, 3)
    return _min_over_candidates(output, all_permutations(control_points), grid_size * grid_size * 3)


def control_points_permute_closed_reg_loss(output, control_points, grid_size_x, grid_size_y):
    """min over grid_size_y u-rolls x 4 flips (80 candidates for a 20 x 20 grid)."""
    batch_size = output.shape[0]
    output = output.view(batch_size, grid_size_x, grid_size_y, 3)
    cands = torch.cat([all_permutations_half(roll(control_points, i, 1)) for i in range(grid_size_y)], 1)
    return _min_over_candidates(output, cands, grid_size_x * grid_size_y * 3)


def control_points_loss(output, control_points, grid_size):
    batch_size = output.shape[0]
    output = output.view(batch_size, grid_size, grid_size, 3)
    return torch.mean(torch.sum((output - control_points) ** 2, (1, 2, 3))) / (grid_size * grid_size * 3)


def spline_reconstruction_loss_one_sided(nu, nv, output, points, config, side=1):
    """Evaluate the predicted grid on (nu, nv) and take the one-sided Chamfer distance to the
    input points (B,3,P)."""
    output = output.view(config.batch_size, config.grid_size, config.grid_size, 3)
    reconst_points = evaluate_surface(nu, nv, output)
    dist = chamfer_distance_one_side(reconst_points, points.permute(0, 2, 1), side)
    return dist, reconst_points


def spline_reconstruction_loss(nu, nv, output, points, config, sqrt=False):
    output = output.reshape(config.batch_size, nu.shape[1], nv.shape[1], 3)
    reconst_points = evaluate_surface(nu, nv, output)
    dist = chamfer_distance(reconst_points, points.permute(0, 2, 1), sqrt=sqrt)
    return dist, reconst_points


def laplacian_loss(output, gt, dist_type="l2"):
    """Difference of the discrete Laplacians of two control grids (B,G,G,3)."""
    lap = np.array([[0.0, 0.25, 0.0], [0.25, -1.0, 0.25], [0.0, 0.25, 0.0]], dtype=np.float32)
    filt = np.zeros((3, 3, 3, 3), dtype=np.float32)
    for c in range(3):
        filt[c, c] = -lap
    filt = torch.from_numpy(filt).to(output.device)
    lo = F.conv2d(output.permute(0, 3, 1, 2), filt, padding=1)
    li = F.conv2d(gt.permute(0, 3, 1, 2), filt, padding=1)
    dist = (lo - li) ** 2 if dist_type == "l2" else torch.abs(lo - li)
    return torch.mean(torch.sum(dist, 1))
