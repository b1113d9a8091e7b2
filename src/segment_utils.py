"""The hot-path part of src/segment_utils.py of the reference (the matched
segment IoU and the coverage metrics; its other helpers belong to the out-of-scope evaluation scripts)."""
from parsenet_codebase_amd.fitting import SIOU_matched_segments, relaxed_iou_fast, to_one_hot  # noqa: F401
from parsenet_codebase_amd.metrics import continuous_labels, coverage_metrics  # noqa: F401
