"""The LS control-point solve of src/approximation.py of the reference."""
from parsenet_codebase_amd.approximation import (BSpline, fit_bezier_surface_fit_kronecker,  # noqa: F401
                                                 uniform_knot_bspline_)
from parsenet_codebase_amd.bspline import basis_function_one, uniform_knot_bspline  # noqa: F401
