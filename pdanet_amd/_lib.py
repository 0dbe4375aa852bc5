"""ctypes binding of the C ABI declared in include/pda_pointnet2.h and include/pda_train.h.

The library is the product: if it is missing or a symbol is absent this module raises at
import of the first op -- there is no fallback path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PDA_LIB_PATH: load another build of the same ABI (A/B timing of kernel variants)
LIB_PATH = os.environ.get("PDA_LIB_PATH") or os.path.join(_HERE, "libpda_pointnet2.so")
ABI_VERSION = 20


class DensityNetScale(ctypes.Structure):
    """pda_densitynet_scale_t (include/pda_train.h)."""
    _fields_ = [("x", ctypes.c_void_p), ("grad_y", ctypes.c_void_p), ("params", ctypes.c_void_p), ("y", ctypes.c_void_p),
                ("stats", ctypes.c_void_p), ("scratch", ctypes.c_void_p), ("running", ctypes.c_void_p * 6),
                ("grad_params", ctypes.c_void_p), ("n", ctypes.c_int64), ("rowmap", ctypes.c_void_p),
                ("row_weight", ctypes.c_void_p), ("n_unique", ctypes.c_void_p), ("nsample", ctypes.c_int),
                ("eps", ctypes.c_float), ("momentum", ctypes.c_float)]


# Bumped by anything that writes parameters behind autograd's back (optimization.FlatAdamOneCycle.step updates the flat
# parameter buffer through a raw pointer, so tensor version counters do not move): caches of derived tensors (bf16 weight
# copies, BatchNorm folded into convolutions) key on it next to the version counters.
PARAM_EPOCH = [0]
# The same for the trained PARAMETERS only (bumped by the optimizer step, not by BatchNorm running statistics): packed weight
# planes made in a forward pass stay valid until the next optimizer step, i.e. through the backward pass of the same iteration.
WEIGHT_EPOCH = [0]

_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float

# name -> argtypes, in the order of include/pda_pointnet2.h
SIGNATURES = {
    "pda_furthest_point_sampling": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "pda_furthest_point_sampling_with_dist": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "pda_gather_points": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "pda_gather_points_grad": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "pda_ball_query": [_vp, _vp, _vp, _i, _i, _i, _f, _i, _vp],
    "pda_ball_query_dilated": [_vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp],
    "pda_ellipsoid_query": [_vp, _vp, _vp, _i, _i, _i, _f, _f, _f, _i, _vp],
    "pda_ball_query_cells_scratch_bytes": [_i, _i],
    "pda_ball_query_cells": [_vp, _vp, ctypes.POINTER(_vp), _i, _i, _i, _i, ctypes.POINTER(_f), ctypes.POINTER(ctypes.c_int32),
                             _vp, ctypes.c_int64, _vp],
    "pda_ball_query_multi": [_vp, _vp, ctypes.POINTER(_vp), _i, _i, _i, _i,
                             ctypes.POINTER(_f), ctypes.POINTER(ctypes.c_int32), _vp],
    "pda_group_points": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "pda_group_points_grad": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "pda_group_rows": [_vp, _vp, _vp, _i, _i, _i, ctypes.c_int64, _vp],
    "pda_group_rows_grad": [_vp, _vp, _vp, _i, _i, _i, ctypes.c_int64, _vp],
    "pda_three_nn": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "pda_three_interpolate": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "pda_three_interpolate_grad": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "pda_chamfer_forward": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "pda_chamfer_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "pda_group_attention_fwd": [_vp, _vp, _vp, ctypes.c_int64, _i, _i, _i, _vp],
    "pda_group_attention_bwd": [_vp, _vp, _vp, _vp, ctypes.c_int64, _i, _i, _i, _vp],
    "pda_group_attention_fwd_bf16": [_vp, _vp, _vp, ctypes.c_int64, _i, _i, _i, _vp],
    "pda_group_attention_bwd_bf16": [_vp, _vp, _vp, _vp, ctypes.c_int64, _i, _i, _i, _vp],
    "pda_sa_mlp_maxpool": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, ctypes.POINTER(ctypes.c_int32),
                           ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp],
    "pda_linear_cols_packed_size": [_i, _i],
    "pda_linear_cols_pack": [_vp, _vp, _i, _i, _i, _i, _vp],
    "pda_linear_cols": [_vp, _vp, _vp, ctypes.c_int64, _i, _i, _vp],
    "pda_sa_gather_linear": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "pda_linear_split_packed_bytes": [_i, _i],
    "pda_linear_split_pack": [_vp, _vp, _i, _i, _i, _vp],
    "pda_linear_split_pack_both": [_vp, _vp, _vp, _i, _i, _vp],
    "pda_linear_split": [_vp, _vp, _vp, _vp, ctypes.c_int64, _i, _i, _i, _vp],
    "pda_gemm_split": [_vp, _vp, _vp, _vp, ctypes.c_int64, _i, _i, _i, _i, _vp],
    "pda_sa_mlp_packed_size": [_i, _i, _i],
    "pda_sa_mlp_pack_weights": [_vp, _vp, _i, _i, _i, _vp],
    # include/pda_train.h
    "pda_grad_norm": [_vp, ctypes.c_int64, _vp, _vp, _vp],
    "pda_bn_relu_scratch_bytes": [_i],
    "pda_bn_relu_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _i, _f, _f, _vp],
    "pda_bn_relu_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _i, _vp],    "pda_bn_relu_fwd_mixed": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, ctypes.c_int64, _i, _f, _f, _vp],
    "pda_bn_relu_bwd_mixed": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _i, _vp],
    "pda_bn_relu_max_pool_fwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _i, _i, _f, _f, _vp],
    "pda_bn_relu_max_pool_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _i, _i, _vp],

    "pda_layer_norm_scratch_bytes": [_i],
    "pda_layer_norm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _i, _f, _vp],
    "pda_layer_norm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _i, _vp],
    "pda_layer_norm_fwd_mixed": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _i, _f, _vp],
    "pda_layer_norm_bwd_mixed": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _i, _vp],
    "pda_linear_wgrad_scratch_bytes": [ctypes.c_int64, _i, _i],
    "pda_linear_wgrad_form": [ctypes.c_int64, _i, _i],
    "pda_linear_wgrad": [_vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _i, _i, _vp],
    "pda_linear_wgrad_bn": [_vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _i, _i, _vp, _vp, _vp, _vp],
    "pda_gemm_split_bn_tiles": [ctypes.c_int64],
    "pda_gemm_split_bn": [_vp, _vp, _vp, ctypes.c_int64, _i, _i, _vp, _vp, _vp, _i, _vp, _vp],
    "pda_gemm_split_maxpool": [_vp, _vp, _vp, _vp, ctypes.c_int64, _i, _i, _i, _i, _vp],
    "pda_gemm_split_gather": [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "pda_bn_stats_fwd": [_vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _i, _f, _f, _vp],
    "pda_bn_finalize_fwd": [_vp, _i, _i, ctypes.c_int64, _f, _f, _vp, _vp, _vp, _vp],
    "pda_bn_relu_max_pool_apply": [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _i, _i, _vp],
    "pda_colsum_scratch_bytes": [_i],
    "pda_colsum_bf16": [_vp, _vp, _vp, ctypes.c_int64, _i, _vp],
    "pda_assemble_tokens": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "pda_assemble_tokens_grad": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "pda_add_max_pool": [_vp, _vp, _vp, _vp, ctypes.c_int64, _i, _i, _vp],
    "pda_max_pool_scatter": [_vp, _vp, _vp, ctypes.c_int64, _i, _i, _vp],
    "pda_add_max_pool_bf16": [_vp, _vp, _vp, _vp, ctypes.c_int64, _i, _i, _vp],
    "pda_max_pool_scatter_bf16": [_vp, _vp, _vp, _vp, ctypes.c_int64, _i, _i, _vp],
    "pda_ragged_plan": [_vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _i, _vp],
    "pda_assemble_tokens_ragged": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _i, _i, _i, _i, _i, _i, _vp],
    "pda_assemble_tokens_ragged_grad": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _i, _i, _i, _i, _i, _i, _vp],
    "pda_bn_relu_fwd_weighted": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _i, _f, _f, _vp, ctypes.c_int64, _vp],
    "pda_bn_relu_bwd_weighted": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _i, _vp, ctypes.c_int64, _vp],
    "pda_add_max_pool_ragged": [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _i, _vp],
    "pda_max_pool_scatter_ragged": [_vp, _vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, _i, _i, _vp],
    "pda_group_attention_ragged_fwd": [_vp, _vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, _i, _i, _i, _vp],
    "pda_group_attention_ragged_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, _i, _i, _i, _vp],
    "pda_densitynet_param_count": [],
    "pda_densitynet_eval_param_count": [],
    "pda_densitynet_scratch_bytes": [],
    "pda_densitynet_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _f, _f, _vp],
    "pda_densitynet_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _f, _vp],
    "pda_densitynet_eval": [_vp, _vp, _vp, ctypes.c_int64, _vp],
    "pda_densitynet_fwd_multi": [_vp, _i, _vp],
    "pda_densitynet_bwd_multi": [_vp, _i, _vp],
    "pda_densitynet_fwd_unique": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _vp, _vp, _vp, _i, _f, _f, _vp],
    "pda_densitynet_bwd_unique": [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _vp, _vp, _vp, _i, _f, _vp],
    "pda_pda_geometry": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp],
    "pda_points_in_boxes": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "pda_assign_point_targets": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "pda_sa_gaussian_mask": [_vp, _i, _i, _vp, _vp, _vp, ctypes.c_int64, _vp],
    "pda_head_assign_targets": [_vp, _i, _i, _vp, ctypes.POINTER(ctypes.c_float), _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "pda_head_cls_loss": [_vp, _i, _i, _i, _vp, _vp, ctypes.c_int64, _f, _vp, _vp, _vp],
    "pda_head_centerness": [_vp, _vp, _vp, _vp, ctypes.c_int64, _vp],
    "pda_head_box_loss": [_vp, _vp, _vp, _vp, _f, _i, _f, _f, ctypes.c_int64, _vp, _vp, _vp],
    "pda_head_vote_loss": [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _f, ctypes.c_int64, _vp, _vp, _vp],
    "pda_head_corner_loss": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _f, ctypes.c_int64, _vp, _vp, _vp, _vp],
    "pda_boxes_overlap_bev": [_vp, _vp, _vp, _i, _i, _vp],
    "pda_boxes_iou_bev": [_vp, _vp, _vp, _i, _i, _vp],
    "pda_nms_mask_words": [_i],
    "pda_nms_bev": [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp],
    "pda_sa_small_train_workspace_bytes": [],
    "pda_sa_small_train_supported": [_i, _i, _i, _i, _i, ctypes.c_int64],
    "pda_sa_small_train_fwd": [_vp] * 7 + [ctypes.POINTER(_vp)] * 4 + [ctypes.POINTER(_f)] * 2 + [_vp] * 4 + [_i] * 8 + [_vp],
    "pda_sa_small_train_bwd": [_vp] * 13 + [ctypes.POINTER(_vp)] * 2 + [_i] * 8 + [_vp],
    "pda_sa_xyz_grad_scratch_bytes": [_i],
    "pda_sa_point_gather": [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "pda_sa_xyz_grad": [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "pda_adam_onecycle_step": [_vp, _vp, _vp, _vp, ctypes.c_int64, _f, _f, _f, _f, _f, _i, _vp, _f, _vp],
    # include/pda_pointnet2_stack.h
    "pda_stack_ball_query": [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp],
    "pda_stack_group_points": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "pda_stack_group_points_grad": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "pda_stack_furthest_point_sampling": [_vp, _vp, _vp, _vp, _vp, _i, _vp],
    "pda_stack_three_nn": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "pda_stack_three_interpolate": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "pda_stack_three_interpolate_grad": [_vp, _vp, _vp, _vp, _i, _i, _vp],
}
INFO_SYMBOLS = ["pda_abi_version", "pda_last_error", "pda_fp_contract_mode", "pda_opt_n_threads",
                "pda_fps_coop_timeouts", "pda_debug_fps_spin_limit", "pda_debug_fps_exchange_nonzero"]

_LIB = None


class PdaError(RuntimeError):
    pass


def load():
    """Load libpda_pointnet2.so (after torch, so both share one HIP runtime)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    import torch  # noqa: F401  (loads libamdhip64 first; our NEEDED soname resolves to it)
    if not os.path.exists(LIB_PATH):
        raise PdaError(
            "%s not found: build it with `python -m pdanet_amd.build` (hipcc, gfx950). "
            "pdanet_amd has no fallback path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.argtypes = argtypes
        fn.restype = _i
    lib.pda_nms_mask_words.restype = ctypes.c_int64
    lib.pda_ball_query_cells_scratch_bytes.restype = ctypes.c_int64
    lib.pda_bn_relu_scratch_bytes.restype = ctypes.c_int64
    lib.pda_layer_norm_scratch_bytes.restype = ctypes.c_int64
    lib.pda_linear_wgrad_scratch_bytes.restype = ctypes.c_int64
    lib.pda_linear_split_packed_bytes.restype = ctypes.c_int64
    lib.pda_colsum_scratch_bytes.restype = ctypes.c_int64
    lib.pda_gemm_split_bn_tiles.restype = ctypes.c_int64
    lib.pda_densitynet_scratch_bytes.restype = ctypes.c_int64
    lib.pda_sa_small_train_workspace_bytes.restype = ctypes.c_int64
    lib.pda_sa_xyz_grad_scratch_bytes.restype = ctypes.c_int64
    lib.pda_abi_version.restype = _i
    lib.pda_last_error.restype = ctypes.c_char_p
    lib.pda_fp_contract_mode.restype = _i
    lib.pda_opt_n_threads.argtypes = [_i]
    lib.pda_opt_n_threads.restype = _i
    lib.pda_fps_coop_timeouts.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), _i]
    lib.pda_fps_coop_timeouts.restype = _i
    lib.pda_debug_fps_spin_limit.argtypes = [_i]
    lib.pda_debug_fps_spin_limit.restype = _i
    lib.pda_debug_fps_exchange_nonzero.argtypes = []
    lib.pda_debug_fps_exchange_nonzero.restype = ctypes.c_longlong
    if lib.pda_abi_version() != ABI_VERSION:
        raise PdaError("libpda_pointnet2.so ABI %d != binding ABI %d: rebuild"
                       % (lib.pda_abi_version(), ABI_VERSION))
    _LIB = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().pda_last_error().decode("utf-8", "replace")
        raise PdaError("%s failed with status %d: %s" % (what, status, msg))
