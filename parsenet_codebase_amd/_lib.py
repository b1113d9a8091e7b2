"""ctypes binding of ``libparsenet_hip.so`` (the C ABI of include/parsenet_hip.h).

There is deliberately no fallback: if the shared library is missing or a symbol
cannot be resolved, importing any compute entry point raises.  ``torch`` is
imported first so that the HIP runtime already mapped by PyTorch-ROCm
(SONAME ``libamdhip64.so.7``) is the one the kernels launch on; streams and
device pointers are then shared between torch and this library.
"""
import ctypes
import os

import torch  # noqa: F401  (must be loaded before the HIP library, see docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libparsenet_hip.so")

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_float = ctypes.c_float
c_size_t = ctypes.c_size_t
c_double = ctypes.c_double

# name -> (restype, argtypes); mirrors include/parsenet_hip.h one to one
ABI_VERSION = 18  # pn_abi_version() of the library these signatures describe

SIGNATURES = {
    "pn_last_error": (ctypes.c_char_p, []),
    "pn_abi_version": (c_int, []),
    "pn_prof_enable": (None, [c_int]),
    "pn_prof_reset": (None, []),
    "pn_prof_count": (c_int, []),
    "pn_prof_get": (c_int, [c_int, ctypes.c_char_p, c_int, ctypes.POINTER(c_double),
                            ctypes.POINTER(ctypes.c_longlong)]),
    "pn_knn_workspace": (c_size_t, [c_int, c_int, c_int, c_int]),
    "pn_knn_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pn_knn_pn_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pn_knn_graph_i32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pn_knn3_ragged": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pn_transpose_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "pn_edge_feature_fwd_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pn_edge_feature_bwd_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                        c_size_t, c_void_p]),
    "pn_edgeconv_reduce_workspace": (c_size_t, [c_int, c_int, c_int, c_int]),
    "pn_edgeconv_reduce_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                           c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                           c_void_p]),
    "pn_edgeconv_reduce_fwd_i32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                           c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                           c_void_p]),
    "pn_moments_f32": (c_int, [c_void_p, c_int, c_double, c_float, c_void_p, c_void_p, c_void_p]),
    "pn_edgeconv_finalize_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                             c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "pn_edgeconv_bwd_prep_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                         c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "pn_edgeconv_bwd_workspace": (c_size_t, [c_int, c_int, c_int]),
    "pn_edgeconv_csr_build": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "pn_edgeconv_bwd_prebuilt": (c_int, [c_void_p, c_void_p, c_int] + [c_void_p] * 6 + [c_int] * 7 +
                                 [c_void_p, c_void_p, c_size_t, c_void_p]),
    "pn_edgeconv_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                    c_void_p, c_size_t, c_void_p]),
    "pn_edgeconv_bwd_i32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                    c_void_p, c_size_t, c_void_p]),
    "pn_dot_select_workspace": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "pn_dot_select_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_size_t, c_void_p]),
    "pn_meanshift_slices": (c_int, [c_int, c_int]),
    "pn_meanshift_pack_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pn_meanshift_iter_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pn_meanshift_x3_image_bytes": (c_size_t, [c_int, c_int]),
    "pn_meanshift_x3_split_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pn_meanshift_x3_iter_fwd_f32": (c_int, [c_void_p] * 3 + [c_int, c_int, c_int] + [c_void_p] * 5 + [c_void_p]),
    "pn_meanshift_x3_iter_bwd_f32": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int] + [c_void_p] * 8 + [c_void_p]),
    "pn_meanshift_x3_exec_tiles": (c_int, [ctypes.POINTER(ctypes.c_ulonglong)]),
    "pn_meanshift_x3_nearest_f32": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pn_gemm_x3_weight_image_bytes": (c_size_t, [c_int, c_int]),
    "pn_gemm_x3_points_image_bytes": (c_size_t, [c_int, c_int, c_int]),
    "pn_gemm_x3_weight_image_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pn_gemm_x3_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t,
                               c_void_p]),
    "pn_standardize_select_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pn_standardize_scale_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p,
                                         c_void_p]),
    "pn_gather_flat_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "pn_adam_flat_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_longlong, ctypes.c_float,
                                 ctypes.c_float, ctypes.c_float, ctypes.c_float, c_int, c_void_p]),
    "pn_gemm_x3_cat_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                   c_size_t, c_void_p]),
    "pn_gemm_x3_wgrad_workspace": (c_size_t, [c_int, c_int, c_int, c_int]),
    "pn_gemm_x3_wgrad_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                     c_size_t, c_void_p]),
    "pn_meanshift_rows_bwd_workspace": (c_size_t, [c_int, c_int]),
    "pn_meanshift_rows_bwd_f32": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                          c_size_t, c_void_p]),
    "pn_meanshift_rows_scatter_add_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pn_meanshift_x3_plan_bytes": (c_size_t, [c_int, c_int]),
    "pn_meanshift_x3_plan_core_bytes": (c_size_t, [c_int, c_int]),
    "pn_gn_max_finish_f32": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int] + [c_void_p] * 5),
    "pn_gn_max_bwd_prep_f32": (c_int, [c_void_p] * 3 + [c_int, c_int] + [c_void_p] * 3),
    "pn_cell_order_i32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pn_kmeans_assign_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pn_kmeans_centres_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pn_meanshift_chain_order_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "pn_meanshift_x3_tileinfo_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pn_meanshift_x3_plan_f32": (c_int, [c_void_p] * 6 + [c_int, c_int, c_float, c_void_p, c_void_p]),
    "pn_meanshift_x3_iter_fwd_plan_f32": (c_int, [c_void_p] * 3 + [c_int, c_int, c_int] + [c_void_p] * 5 +
                                          [c_void_p, c_void_p]),
    "pn_meanshift_x3_iter_fwd_info_f32": (c_int, [c_void_p] * 3 + [c_int, c_int, c_int] + [c_void_p] * 5 +
                                          [c_void_p] * 5),
    "pn_meanshift_x3_iter_bwd_plan_f32": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int] + [c_void_p] * 8 +
                                          [c_void_p, c_void_p]),
    "pn_dot_kth_x3_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                  c_void_p, c_size_t, c_void_p]),
    "pn_dot_kth_unit_h2_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                       c_void_p, c_size_t, c_void_p]),
    "pn_meanshift_h2_image_bytes": (c_size_t, [c_int, c_int]),
    "pn_meanshift_h2_split_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pn_meanshift_h2_iter_fwd_f32": (c_int, [c_void_p] * 3 + [c_int, c_int, c_int] + [c_void_p] * 5 + [c_void_p]),
    "pn_meanshift_h2_iter_bwd_f32": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int] + [c_void_p] * 8 + [c_void_p]),
    "pn_meanshift_iter_bwd_f32": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int] + [c_void_p] * 9 + [c_void_p]),
    "pn_gn_rows_fwd_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "pn_gn_group_moments_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p,
                                        c_void_p, c_void_p]),
    "pn_gn_apply_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                    c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "pn_gn_rows_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                   c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "pn_gn_group_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                    c_void_p]),
    "pn_gn_apply_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                    c_void_p]),
    "pn_sym3_eig_f64": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "pn_chamfer_nn_workspace": (c_size_t, [c_int, c_int, c_int]),
    "pn_chamfer_nn_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pn_chamfer_nn_ragged_workspace": (c_size_t, [c_int, c_int]),
    "pn_chamfer_nn_ragged_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pn_chamfer_ragged_reduce_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "pn_chamfer_ragged_bwd_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_int, c_void_p, c_void_p]),
    "pn_gather_rows3_bwd_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pn_weighted_moments_chunks": (c_int, []),
    "pn_weighted_moments_count": (c_int, []),
    "pn_weighted_moments_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                        c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "pn_primitive_fit_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pn_cone_angle_f64": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pn_primitive_residual_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                          c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pn_weighted_moments_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                            c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_void_p]),
    "pn_bspline_eval_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                    c_int, c_void_p, c_void_p]),
    "pn_bspline_eval_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                        c_int, c_void_p, c_void_p]),
    "pn_edgeconv_bwd_stats_workspace": (c_size_t, [c_int, c_int, c_int]),
    "pn_edgeconv_bwd_stats_f32": (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_float] + [c_void_p] * 5 +
                                  [c_size_t, c_void_p]),
    "pn_triplet_fwd_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float,
                                   c_void_p, c_void_p, c_void_p, c_void_p]),
    "pn_triplet_bwd_workspace": (c_size_t, [c_int, c_int, c_int]),
    "pn_triplet_bwd_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                   c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pn_membership_fwd_f32": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_float] + [c_void_p] * 5 + [c_void_p]),
    "pn_membership_bwd_f32": (c_int, [c_void_p] * 6 + [c_int] * 3 + [c_void_p] * 2 + [c_void_p]),
    "pn_nms_occupied_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pn_nms_vote_f32": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p] * 3 + [c_void_p]),
    "pn_weighted_max_fwd_f32": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_float] + [c_void_p] * 3 + [c_void_p]),
    "pn_weighted_max_bwd_f32": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p, c_void_p]),
    "pn_affine_act_fwd_f32": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_float, c_void_p, c_void_p]),
    "pn_affine_act_bwd_f32": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_float, c_void_p, c_void_p]),
}

_lib = None


class HipExtensionError(RuntimeError):
    pass


def load():
    """Return the loaded library handle, raising loudly when it is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipExtensionError(
            "%s not found: build it with `python -m parsenet_codebase_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover
        raise HipExtensionError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise HipExtensionError("%s does not export %s (stale build?)" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    if lib.pn_abi_version() != ABI_VERSION:
        raise HipExtensionError("%s has ABI version %d, this package binds version %d: rebuild it "
                                "(python -m parsenet_codebase_amd.build)" % (LIB_PATH, lib.pn_abi_version(),
                                                                              ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().pn_last_error().decode("utf-8", "replace")
        raise RuntimeError("%s failed (rc=%d): %s" % (what, rc, msg))


def ptr(t):
    """Raw device pointer of a tensor as a plain int (None -> NULL); the argtypes table converts.
    (No ctypes object per argument: a training step marshals ~4 000 pointers.)"""
    if t is None:
        return None
    return t.data_ptr()


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(device):
    """``torch.cuda.device(device)`` only when ``device`` is not already current (one process per
    GPU: it always is): the context manager's two device switches and their Python frames cost more
    than the launch they guard, ~400 times per training step."""
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(device)


def current_stream(device):
    """torch's current HIP stream on ``device`` as a raw handle (an int for the c_void_p slot)."""
    idx = device.index
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if idx is None else idx)


def require_cuda(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "parsenet_codebase_amd runs on MI355X only: got a %s tensor; there is no CPU "
                "path in the product (the CPU restatement lives in oracle/ and is test-only)"
                % t.device)


class _PinnedRing:
    """Persistent page-locked staging buffers.  Allocating pinned memory is a driver call that can
    take MILLISECONDS (tens of them on a busy host: the kernel trace of a cfg5 step showed single
    40 ms holes in front of the first kernel after an upload), and the sizes of a step's tables
    change with the number of segments, which defeats torch's caching host allocator.  A ring of
    fixed-size slots allocated once: a slot is reused only after the event recorded behind its last
    copy has completed (normally long before its turn comes round again).

    Uploads are done with a slot once that event has completed.  A DOWNLOAD is not: the host reads
    the slot after the event, possibly much later (the fit status of a step is read after the
    backward pass has been queued) — such a slot is taken with ``hold=True`` and stays out of the
    rotation until the reader calls ``release`` (round-3 advisor finding: 32 further takes handed
    the slot to an upload whose host-side copy overwrote unread results).  Readers release in a
    ``finally``; a slot whose reader never came back at all (a stage generator abandoned by a dropped
    step, ``finish()`` never called after a rank-local failure) is reclaimed — with a warning — once
    ABANDONED_AFTER further takes have gone by: a live reader holds a slot for a fraction of a step, a
    few dozen takes at most (round-4 advisor finding: leaked slots silently turned every transfer into a
    fresh page-locked allocation, the multi-millisecond stall the ring exists to avoid).

    Every take hands out a HANDLE carrying the slot's generation number.  A reader that is merely late — not gone
    — when its slot is reclaimed still has its handle: ``release`` with a stale generation raises instead of clearing
    the hold of the slot's NEW owner, so the late reader fails loudly on data that may have been overwritten
    (round-5 advisor finding)."""

    ABANDONED_AFTER = 256

    class Handle:
        __slots__ = ("slot", "gen")

        def __init__(self, slot, gen):
            self.slot, self.gen = slot, gen

        def __getitem__(self, key):          # handle["event"]: the slot's event
            return self.slot[key]

    def __init__(self, slots=32, nbytes=1 << 20):
        self.nbytes, self.nslots, self.slots, self.next = nbytes, slots, None, 0
        self.takes, self.warned_full, self.warned_lost = 0, False, False

    def take(self, nbytes, hold=False):
        """(uint8 pinned view of ``nbytes``, slot) or (None, None) when the request exceeds a slot or
        every slot is held by a host reader."""
        if nbytes > self.nbytes:
            return None, None
        if self.slots is None:
            # ONE page-locked allocation for the whole ring, at the first use (not one per slot spread
            # over the first dozens of uploads — i.e. over somebody's timed steps)
            whole = torch.empty(self.nslots * self.nbytes, dtype=torch.uint8).pin_memory()
            self.slots = [{"buf": whole[i * self.nbytes:(i + 1) * self.nbytes], "event": torch.cuda.Event(),
                           "armed": False, "held": False, "gen": 0} for i in range(self.nslots)]
        self.takes += 1
        for _ in range(self.nslots):
            i = self.next
            self.next = (i + 1) % self.nslots
            slot = self.slots[i]
            if slot["held"]:
                if self.takes - slot.get("held_at", self.takes) <= self.ABANDONED_AFTER:
                    continue
                if not self.warned_lost:
                    import warnings
                    warnings.warn("parsenet_codebase_amd: a pinned download slot was never released by its reader "
                                  "(abandoned step?); reclaiming it")
                    self.warned_lost = True
            if slot["armed"]:
                slot["event"].synchronize()
            slot["held"] = bool(hold)
            slot["held_at"] = self.takes
            slot["gen"] += 1
            return slot["buf"][:max(nbytes, 1)], _PinnedRing.Handle(slot, slot["gen"])
        if not self.warned_full:
            import warnings
            warnings.warn("parsenet_codebase_amd: every pinned staging slot is held by a host reader; falling back "
                          "to a fresh page-locked allocation per transfer")
            self.warned_full = True
        return None, None

    @staticmethod
    def release(handle):
        """The host has read (or copied out) what the download left in the slot.  Raises if the slot was reclaimed
        and handed to another taker meanwhile (the data this reader just read may have been overwritten)."""
        if handle is None:
            return
        if handle.gen != handle.slot["gen"]:
            raise RuntimeError("parsenet_codebase_amd: a pinned download slot was reclaimed (%d further transfers) "
                               "before its reader came back: the downloaded data may have been overwritten"
                               % _PinnedRing.ABANDONED_AFTER)
        handle.slot["held"] = False

    @staticmethod
    def arm(handle):
        """Record the slot's event on the current stream: call right after queuing the copy."""
        handle.slot["event"].record()
        handle.slot["armed"] = True


_RING = _PinnedRing()


_SPIN_SECONDS = None


def wait_event(event):
    """Block the host until ``event`` has completed — by polling, for a bounded time.
    hipEventSynchronize puts the thread to sleep, and on a busy host (the pool's boxes are shared)
    the wake-up alone costs up to milliseconds while the device sits idle behind the very copy the
    host waits for; the three waits of a training step are short.  One process per GPU means one
    spinning core PER RANK next to its OMP threads, so with several ranks (WORLD_SIZE > 1) the
    spin is bounded — 300 us by default, PARSENET_SPIN_US overrides, 0 = always sleep — and the
    blocking wait takes over; a single rank spins up to 50 ms (a step's wait is 1-20 ms)."""
    global _SPIN_SECONDS
    if _SPIN_SECONDS is None:
        us = os.environ.get("PARSENET_SPIN_US")
        if us is None:
            us = 300 if int(os.environ.get("WORLD_SIZE", "1")) > 1 else 50000
        _SPIN_SECONDS = float(us) * 1e-6
    if event.query():
        return
    import time
    deadline = time.perf_counter() + _SPIN_SECONDS
    while time.perf_counter() < deadline:
        if event.query():
            return
    event.synchronize()


def pinned_like(shape, dtype, hold=False):
    """A pinned host tensor of ``shape`` / ``dtype`` from the ring (plus its slot, to be armed after
    the copy that fills or drains it), or a freshly pinned one (slot None) when it does not fit.
    ``hold=True`` for downloads: the slot is the caller's until _PinnedRing.release(slot)."""
    n = 1
    for d in shape:
        n *= int(d)
    nbytes = n * torch.empty((), dtype=dtype).element_size()
    raw, slot = _RING.take(nbytes, hold)
    if raw is None:
        return torch.empty(tuple(shape), dtype=dtype).pin_memory(), None
    return raw[:nbytes].view(dtype).reshape(tuple(shape)), slot


def h2d(array, device):
    """Host array -> device tensor through pinned memory, stream-ordered (non_blocking).  A
    pageable-memory copy makes the host wait for everything queued on the stream; the fitting
    stage issues several small table uploads per step, each of which would drain the GPU.  The
    pinned staging memory comes from a ring allocated once (see _PinnedRing)."""
    import numpy as np
    t = torch.from_numpy(np.ascontiguousarray(array))
    if device.type != "cuda":
        return t.to(device)
    host, slot = pinned_like(t.shape, t.dtype)
    if slot is None:
        host.copy_(t)
        return host.to(device, non_blocking=True)
    host.copy_(t)
    out = host.to(device, non_blocking=True)
    _PinnedRing.arm(slot)
    return out


def meanshift_exec_tiles():
    """(forward, row pass, column pass) tile pairs the bf16 x 3 mean-shift launches executed since the
    last call — counted inside the kernels; reading clears the counters (and synchronises)."""
    out = (ctypes.c_ulonglong * 3)()
    check(load().pn_meanshift_x3_exec_tiles(out), "pn_meanshift_x3_exec_tiles")
    return int(out[0]), int(out[1]), int(out[2])


def prof_enable(on=True):
    load().pn_prof_enable(1 if on else 0)


def prof_reset():
    load().pn_prof_reset()


def prof_results():
    """{kernel family: (total_ms, calls)} measured with HIP events on the launch stream."""
    lib = load()
    out = {}
    buf = ctypes.create_string_buffer(128)
    for i in range(lib.pn_prof_count()):
        ms = c_double()
        calls = ctypes.c_longlong()
        if lib.pn_prof_get(i, buf, 128, ctypes.byref(ms), ctypes.byref(calls)) == 0:
            out[buf.value.decode()] = (ms.value, calls.value)
    return out
