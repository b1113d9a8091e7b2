"""src/primitive_forward.py of the reference: SplineNet forward wrappers, the evaluation-only LS refit
and the per-shape fitting driver (training and evaluation mode; viewer/mesh output excluded)."""
from parsenet_codebase_amd.fitting import (Fit, fit_one_shape_torch, forward_closed_splines,  # noqa: F401
                                           forward_pass_open_spline, initialize_closed_spline_model,
                                           initialize_open_spline_model, optimize_close_spline_kronecker,
                                           optimize_open_spline_kronecker)
