"""src/fitting_optimization.py of the reference (FittingModule; the open3d ARAP part is
evaluation-only and out of scope)."""
from parsenet_codebase_amd.fitting import FittingModule  # noqa: F401
