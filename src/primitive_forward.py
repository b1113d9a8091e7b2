"""The training-path part of src/primitive_forward.py of the reference."""
from parsenet_codebase_amd.fitting import (Fit, fit_one_shape_torch, forward_closed_splines,  # noqa: F401
                                           forward_pass_open_spline, initialize_closed_spline_model,
                                           initialize_open_spline_model)
