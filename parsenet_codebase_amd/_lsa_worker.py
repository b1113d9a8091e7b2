"""Worker side of the assignment pool (fitting_eval.assignment_pool): imports numpy and scipy ONLY — a spawned
worker never touches torch or the GPU.  src/primitive_forward.py:197-198, 272-273 call lapsolver.solve_dense on
a 1 600 x 1 600 ... 2 100 distance matrix per spline segment; scipy's linear_sum_assignment holds the GIL, so the
matrices of a batch go to worker PROCESSES."""
import numpy as np
from scipy.optimize import linear_sum_assignment


def solve(cost):
    rows, cols = linear_sum_assignment(np.asarray(cost))
    return np.asarray(cols)
