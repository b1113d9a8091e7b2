"""src/residual_utils.py of the reference (end-to-end fitting loss: training mode and
evaluation mode with hard memberships)."""
from parsenet_codebase_amd.fitting import Evaluation  # noqa: F401
from parsenet_codebase_amd.fitting import one_hot_normalization as convert_to_one_hot  # noqa: F401,E402
