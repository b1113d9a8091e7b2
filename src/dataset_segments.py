"""src/dataset_segments.py of the reference: segmentation data generators (host side)."""
from parsenet_codebase_amd.data import EPS, Dataset  # noqa: F401
