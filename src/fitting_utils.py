"""The hot-path part of src/fitting_utils.py of the reference."""
from parsenet_codebase_amd.bspline import sample_points_from_control_points_  # noqa: F401
from parsenet_codebase_amd.fitting import (EPS, CustomSVD, LeastSquares, best_lambda,  # noqa: F401
                                           compute_grad_V, customsvd, match, pca_torch,
                                           project_to_plane, relaxed_iou_fast,
                                           rotation_matrix_a_to_b, standardize_point_torch,
                                           standardize_points_torch, svd_grad_K, to_one_hot,
                                           one_hot_normalization, pca_numpy, project_to_point_cloud,
                                           remove_outliers, reverse_all_transformation,
                                           reverse_all_transformations, up_sample_points,
                                           up_sample_points_in_range, up_sample_points_torch,
                                           up_sample_points_torch_in_range,
                                           up_sample_points_torch_memory_efficient, weights_normalize)
from parsenet_codebase_amd.metrics import matching_iou, relaxed_iou  # noqa: F401,E402  (src/fitting_utils.py:18-19)
