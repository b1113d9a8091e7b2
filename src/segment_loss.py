"""src/segment_loss.py of the reference."""
from parsenet_codebase_amd.losses import EmbeddingLoss, evaluate_miou, primitive_loss  # noqa: F401
