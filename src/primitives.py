"""src/primitives.py of the reference (residual distances)."""
from parsenet_codebase_amd.fitting import ComputePrimitiveDistance, ResidualLoss  # noqa: F401
