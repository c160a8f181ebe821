"""ctypes binding of the C-ABI in ``include/lora_amd.h`` (``csrc/liblora_amd.so``).

The library is the product: every CUDA/HIP-device code path of this package goes
through it and raises :class:`HipExtensionMissing` when it cannot be loaded —
there is no eager/PyTorch fallback for device tensors.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "liblora_amd.so")

OK = 0
F32, F16, BF16 = 0, 1, 2
FACTOR_RK, FACTOR_KR = 0, 1
ROUND_REFERENCE, ROUND_ONCE, ROUND_DITHER = 0, 1, 2
MAX_RANK = 64
ABI_VERSION = 7

_DT = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}

# every symbol include/lora_amd.h declares (tests check the .so exports them all)
SYMBOLS = (
    "lora_amd_abi_version", "lora_amd_last_error", "lora_amd_target_arch",
    "lora_amd_merge_plan", "lora_amd_merge_batched", "lora_amd_merge_set_tuning", "lora_amd_merge_step_set_tuning", "lora_amd_rank16_mfma", "lora_amd_factors_mfma_set_tuning",
    "lora_amd_merge_step_plan", "lora_amd_merge_step",
    "lora_amd_rowdot", "lora_amd_rowdot_masked", "lora_amd_rank_update",
    "lora_amd_colreduce_workspace", "lora_amd_colreduce",
    "lora_amd_rowdot_batched", "lora_amd_colreduce_batched", "lora_amd_chol_inverse_batched",
    "lora_amd_ragged_plan", "lora_amd_rowdot_ragged", "lora_amd_colreduce_ragged", "lora_amd_sub_ragged",
    "lora_amd_rowdot16_planes_plan", "lora_amd_rowdot16_planes", "lora_amd_rowdot16_planes_packed", "lora_amd_thin_pack",
    "lora_amd_split16_ragged", "lora_amd_split16_transpose",
    "lora_amd_split16_residual", "lora_amd_thin_gram", "lora_amd_thin_apply", "lora_amd_thin_rotate", "lora_amd_thin_select", "lora_amd_thin_clamp",
    "lora_amd_linear_plan", "lora_amd_linear_fwd", "lora_amd_linear_bwd_g", "lora_amd_linear_bwd_x",
    "lora_amd_linear_bwd_factors", "lora_amd_linear_bwd_factors_drop", "lora_amd_linear_bwd_factors_heads",
    "lora_amd_linear_factors_self_plan", "lora_amd_linear_factors_self_plan_rows", "lora_amd_linear_bwd_factors_self",
    "lora_amd_linear_factors_self_ragged_plan", "lora_amd_linear_bwd_factors_self_ragged",
    "lora_amd_factors_mfma_plan", "lora_amd_factor_pack_plan", "lora_amd_factor_pack",
    "lora_amd_factors_mfma_ragged_plan", "lora_amd_linear_bwd_factors_mfma_ragged", "lora_amd_factors_mfma_block_map",
    "lora_amd_linear_bwd_factors_mfma_ragged_mapped",
    "lora_amd_linear_gemm_fwd_heads",
    "lora_amd_reduce_batched", "lora_amd_linear_gemm_supported", "lora_amd_linear_gemm_fwd",
    "lora_amd_ws_config", "lora_amd_ws_packed_elems", "lora_amd_ws_pack", "lora_amd_linear_ws", "lora_amd_linear_ws_heads",
    "lora_amd_conv_plan", "lora_amd_conv_down_fwd", "lora_amd_conv_up_fwd", "lora_amd_conv_bwd_g", "lora_amd_conv_bwd_x",
    "lora_amd_conv3_nhwc_plan", "lora_amd_conv3_nhwc_pack", "lora_amd_conv3_nhwc_down_fwd", "lora_amd_conv3_nhwc_bwd_dx",
    "lora_amd_conv3_nhwc_bwd_down", "lora_amd_sum_parts",
    "lora_amd_conv3_nhwc_pack_plan", "lora_amd_conv3_nhwc_pack_batched", "lora_amd_conv3_nhwc_fwd_fused",
    "lora_amd_linear_bwd_g_blocks", "lora_amd_linear_bwd_g_folded",
    "lora_amd_sumsq_workspace", "lora_amd_sumsq", "lora_amd_clip_adamw", "lora_amd_clip_adamw_dev",
    "lora_amd_step_advance", "lora_amd_loss_scale_update", "lora_amd_ti_rows_step",
    "lora_amd_groupnorm_workspace", "lora_amd_groupnorm_supported", "lora_amd_groupnorm_fwd", "lora_amd_groupnorm_bwd",
    "lora_amd_geglu_fwd", "lora_amd_geglu_bwd",
    "lora_amd_layernorm_supported", "lora_amd_layernorm_fwd", "lora_amd_layernorm_bwd",
    "lora_amd_groupnorm_nhwc_workspace", "lora_amd_groupnorm_nhwc_fwd", "lora_amd_groupnorm_nhwc_bwd",
    "lora_amd_add_layernorm_fwd", "lora_amd_add_layernorm_bwd",
)


class HipExtensionMissing(RuntimeError):
    pass


class FactorsSelfPlan(C.Structure):
    _fields_ = [("supported", C.c_int32), ("rank_tile", C.c_int32), ("nparts", C.c_int32), ("reserved", C.c_int32),
                ("up_part_floats", C.c_int64), ("down_part_floats", C.c_int64)]


class SelfSite(C.Structure):
    """lora_amd_self_site (include/lora_amd.h): one adapter of the one-launch factor-gradient pass."""
    _fields_ = [
        ("g", C.c_void_p), ("x", C.c_void_p), ("down", C.c_void_p), ("up", C.c_void_p),
        ("up_part", C.c_void_p), ("down_part", C.c_void_p),
        ("ldg", C.c_int64), ("ldx", C.c_int64), ("M", C.c_int64),
        ("N", C.c_int32), ("K", C.c_int32), ("r", C.c_int32), ("scale", C.c_float),
        ("g_head_dim", C.c_int32), ("g_head_pad", C.c_int32), ("x_head_dim", C.c_int32), ("x_head_pad", C.c_int32),
        ("rows_per_block", C.c_int32), ("nsplit", C.c_int32), ("kt_g", C.c_int32), ("logL_g", C.c_int32),
        ("kt_x", C.c_int32), ("logL_x", C.c_int32), ("tile_g", C.c_int32), ("nct_g", C.c_int32),
        ("tile_x", C.c_int32), ("nct_x", C.c_int32),
        ("reserved0", C.c_int32), ("reserved", C.c_int32),
        ("block_begin", C.c_int64),
    ]


class FactorsMfmaPlan(C.Structure):
    _fields_ = [("supported", C.c_int32), ("lds_class", C.c_int32), ("rank_tile", C.c_int32),
                ("rows_per_block", C.c_int32), ("nparts", C.c_int32), ("lds_bytes", C.c_int32),
                ("blocks_per_wg", C.c_int32), ("reserved", C.c_int32),
                ("up_part_floats", C.c_int64), ("down_part_floats", C.c_int64),
                ("pack_up_elems", C.c_int64), ("pack_down_elems", C.c_int64)]


class PackSite(C.Structure):
    """lora_amd_pack_site: the f32 factors of one adapter -> MFMA fragment packs."""
    _fields_ = [("down", C.c_void_p), ("up", C.c_void_p), ("pk_down", C.c_void_p), ("pk_up", C.c_void_p),
                ("N", C.c_int32), ("K", C.c_int32), ("r", C.c_int32), ("reserved", C.c_int32), ("begin", C.c_int64)]


class FmSite(C.Structure):
    """lora_amd_fm_site: one adapter of the matrix-core factor-gradient pass."""
    _fields_ = [
        ("g", C.c_void_p), ("x", C.c_void_p), ("pk_up", C.c_void_p), ("pk_down", C.c_void_p),
        ("up_part", C.c_void_p), ("down_part", C.c_void_p),
        ("ldg", C.c_int64), ("ldx", C.c_int64), ("M", C.c_int64),
        ("N", C.c_int32), ("K", C.c_int32), ("r", C.c_int32), ("scale", C.c_float),
        ("g_head_dim", C.c_int32), ("g_head_pad", C.c_int32), ("x_head_dim", C.c_int32), ("x_head_pad", C.c_int32),
        ("rows_per_block", C.c_int32), ("blocks_per_wg", C.c_int32),
        ("resident_is_x", C.c_int32), ("cw", C.c_int32), ("nchunk", C.c_int32), ("x_head_magic", C.c_int32),
        ("g_head_magic", C.c_int32), ("lds_bytes", C.c_int32),
        ("block_begin", C.c_int64),
        ("dropout_p", C.c_float), ("reserved", C.c_int32), ("seed", C.c_uint64), ("offset", C.c_uint64),
        ("offset_dev", C.c_void_p),
    ]


class RaggedDesc(C.Structure):
    """lora_amd_ragged_desc (include/lora_amd.h): one stack of same-shape matrices of a ragged launch."""
    _fields_ = [
        ("x", C.c_void_p), ("f", C.c_void_p), ("out", C.c_void_p), ("partial", C.c_void_p),
        ("ldx", C.c_int64), ("stride_x", C.c_int64), ("stride_f", C.c_int64), ("stride_out", C.c_int64), ("M", C.c_int64),
        ("K", C.c_int32), ("batch", C.c_int32),
        ("stride_partial", C.c_int64), ("begin1", C.c_int64), ("begin2", C.c_int64),
        ("blocks1", C.c_int32), ("blocks2", C.c_int32), ("col_tiles", C.c_int32), ("kt_cols", C.c_int32),
        ("logL", C.c_int32), ("rows_per_block", C.c_int32),
    ]


class SubDesc(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("out", C.c_void_p), ("n", C.c_int64), ("begin", C.c_int64)]


class PlanesDesc(C.Structure):
    _fields_ = [("hi", C.c_void_p), ("lo", C.c_void_p), ("f", C.c_void_p), ("out", C.c_void_p), ("M", C.c_int64),
                ("C", C.c_int32), ("batch", C.c_int32), ("wps", C.c_int32), ("slabs_per_wg", C.c_int32),
                ("wg_begin", C.c_int64)]


class SplitTDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("hi", C.c_void_p), ("lo", C.c_void_p), ("thi", C.c_void_p), ("tlo", C.c_void_p),
                ("batch", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("reserved", C.c_int32), ("tile_begin", C.c_int64)]


class ResidDesc(C.Structure):
    _fields_ = [("tuned", C.c_void_p), ("base", C.c_void_p), ("hi", C.c_void_p), ("lo", C.c_void_p), ("thi", C.c_void_p),
                ("tlo", C.c_void_p), ("N", C.c_int32), ("K", C.c_int32), ("tile_begin", C.c_int64)]


class ThinSite(C.Structure):
    _fields_ = [("off", C.c_int64), ("rows", C.c_int64), ("block_begin", C.c_int64), ("blocks", C.c_int32),
                ("reserved", C.c_int32)]


class ThinQSite(C.Structure):
    _fields_ = [("off_u", C.c_int64), ("off_v", C.c_int64), ("n_u", C.c_int64), ("n_v", C.c_int64),
                ("block_begin", C.c_int64), ("blocks", C.c_int32), ("reserved", C.c_int32)]


class ThinFinishDesc(C.Structure):
    _fields_ = [("part", C.c_void_p), ("counters", C.c_void_p), ("mode", C.c_int32), ("shift_rel", C.c_float),
                ("linv_out", C.c_void_p), ("ritz_out", C.c_void_p), ("ubt", C.c_void_p), ("vb", C.c_void_p),
                ("s_out", C.c_void_p), ("rank", C.c_int32), ("reserved", C.c_int32)]


class SplitDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("hi", C.c_void_p), ("lo", C.c_void_p), ("n", C.c_int64), ("begin", C.c_int64)]


class MergeSite(C.Structure):
    _fields_ = [
        ("w_in", C.c_void_p), ("w_out", C.c_void_p), ("up", C.c_void_p), ("down", C.c_void_p),
        ("N", C.c_int32), ("K", C.c_int32), ("r", C.c_int32),
        ("rows_per_tile", C.c_int32), ("cols_per_tile", C.c_int32), ("tiles_k", C.c_int32),
        ("tile_begin", C.c_int64), ("flags", C.c_int32), ("out_heads", C.c_int32),
        ("transposed", C.c_int32), ("reserved", C.c_int32),
    ]


class MstepSite(C.Structure):
    """lora_amd_mstep_site (include/lora_amd.h): one adapter of the in-step merge (W_eff and W_eff^T from one read of W)."""
    _fields_ = [
        ("w", C.c_void_p), ("up", C.c_void_p), ("down", C.c_void_p), ("out", C.c_void_p), ("out_t", C.c_void_p),
        ("ld_out", C.c_int64), ("ld_out_t", C.c_int64),
        ("N", C.c_int32), ("K", C.c_int32), ("r", C.c_int32),
        ("row_d", C.c_int32), ("row_D", C.c_int32), ("col_d", C.c_int32), ("col_D", C.c_int32),
        ("dither_key", C.c_int32), ("tiles_k", C.c_int32), ("reserved", C.c_int32),
        ("tile_begin", C.c_int64),
    ]


class MergeSummary(C.Structure):
    _fields_ = [("total_tiles", C.c_int64), ("n_fast_sites", C.c_int32), ("rank_tile_fast", C.c_int32)]


class AdamWGroup(C.Structure):
    _fields_ = [("begin", C.c_int64), ("end", C.c_int64), ("lr", C.c_float), ("weight_decay", C.c_float)]


class LinearPlan(C.Structure):
    _fields_ = [("fused", C.c_int32), ("rank_tile", C.c_int32), ("nct_g", C.c_int32), ("nparts_up", C.c_int32),
                ("nparts_down", C.c_int32), ("reserved", C.c_int32), ("gt_part_floats", C.c_int64),
                ("up_part_floats", C.c_int64), ("down_part_floats", C.c_int64)]


class ConvPlan(C.Structure):
    _fields_ = [("native", C.c_int32), ("cpw_in", C.c_int32), ("ngroups_in", C.c_int32), ("ngroups_out", C.c_int32),
                ("split_in", C.c_int32), ("split_out", C.c_int32), ("rank_pad", C.c_int32), ("reserved", C.c_int32),
                ("t_part_floats", C.c_int64), ("gt_part_floats", C.c_int64), ("up_part_floats", C.c_int64),
                ("down_part_floats", C.c_int64)]


class Conv3NhwcPlan(C.Structure):
    _fields_ = [("native", C.c_int32), ("pt", C.c_int32), ("ksplit", C.c_int32), ("csplit", C.c_int32),
                ("ks", C.c_int32), ("pr", C.c_int32), ("nsplit", C.c_int32), ("rank_pad", C.c_int32),
                ("fwd_tiles", C.c_int32), ("reserved", C.c_int32),
                ("pf_elems", C.c_int64), ("pd_elems", C.c_int64), ("t_part_floats", C.c_int64),
                ("down_part_floats", C.c_int64)]


class Conv3PackSite(C.Structure):
    _fields_ = [("down", C.c_void_p), ("up", C.c_void_p), ("pf", C.c_void_p), ("pd", C.c_void_p), ("pu", C.c_void_p),
                ("r", C.c_int32), ("C_in", C.c_int32), ("C_out", C.c_int32), ("KS", C.c_int32), ("begin", C.c_int64)]


WS_MAX_SITES = 4


class WsSite(C.Structure):
    _fields_ = [("wp", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p), ("down", C.c_void_p), ("up", C.c_void_p),
                ("t_out", C.c_void_p), ("ldy", C.c_int64), ("N", C.c_int32), ("r", C.c_int32),
                ("panel_begin", C.c_int32), ("flayout", C.c_int32), ("scale", C.c_float), ("t_scale", C.c_float),
                ("dropout_p", C.c_float), ("y_heads", C.c_int32), ("seed", C.c_uint64), ("offset", C.c_uint64),
                ("offset_dev", C.c_void_p)]


class ReduceDesc(C.Structure):
    _fields_ = [("part", C.c_void_p), ("out", C.c_void_p), ("begin", C.c_int64), ("nparts", C.c_int32),
                ("RT", C.c_int32), ("C", C.c_int32), ("r", C.c_int32), ("layout", C.c_int32),
                ("reserved", C.c_int32), ("scale", C.c_float), ("beta", C.c_float)]


_lib: Optional[C.CDLL] = None
_load_error: Optional[str] = None


def _declare(lib: C.CDLL) -> None:
    vp, i32, i64, f32, u64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64, C.c_size_t
    lib.lora_amd_abi_version.restype = C.c_int
    lib.lora_amd_last_error.restype = C.c_char_p
    lib.lora_amd_target_arch.restype = C.c_char_p
    lib.lora_amd_merge_plan.argtypes = [C.POINTER(MergeSite), i32, i32, C.POINTER(MergeSummary)]
    lib.lora_amd_merge_batched.argtypes = [vp, i32, C.POINTER(MergeSummary), i32, i32, f32, i32, vp]
    lib.lora_amd_merge_step_plan.argtypes = [C.POINTER(MstepSite), i32, i32, C.POINTER(i64)]
    lib.lora_amd_merge_step.argtypes = [vp, i32, i64, i32, i32, f32, i32, vp]
    lib.lora_amd_merge_step_plan.restype = lib.lora_amd_merge_step.restype = C.c_int
    lib.lora_amd_merge_set_tuning.argtypes = [i64, i64]
    lib.lora_amd_merge_step_set_tuning.argtypes = [i32, i32]
    lib.lora_amd_rank16_mfma.argtypes = [i32]
    lib.lora_amd_rank16_mfma.restype = C.c_int
    lib.lora_amd_rowdot.argtypes = [vp, i64, vp, vp, i64, i32, i32, i32, i32, i32, f32, vp, i32, vp]
    lib.lora_amd_rowdot_masked.argtypes = [vp, i64, vp, vp, i64, i32, i32, i32, i32, i32, f32, vp, i32,
                                           f32, u64, u64, vp, vp]
    lib.lora_amd_rank_update.argtypes = [vp, i64, vp, vp, i64, i32, i32, i32, i32, i32, f32, f32, u64, u64, vp, vp]
    lib.lora_amd_colreduce_workspace.argtypes = [i64, i32, i32]
    lib.lora_amd_colreduce_workspace.restype = sz
    lib.lora_amd_colreduce.argtypes = [vp, i64, vp, vp, i64, i32, i32, i32, i32, f32, f32, f32, u64, u64, vp,
                                       vp, sz, vp]
    lib.lora_amd_rowdot_batched.argtypes = [vp, i64, i64, vp, i64, vp, i64, i32, i64, i32, i32, i32, i32, i32, f32, vp]
    lib.lora_amd_colreduce_batched.argtypes = [vp, i64, i64, vp, i64, vp, i64, i32, i64, i32, i32, i32, i32, f32, vp, sz, vp]
    lib.lora_amd_chol_inverse_batched.argtypes = [vp, vp, i32, i32, f32, vp]
    lib.lora_amd_rowdot_batched.restype = lib.lora_amd_colreduce_batched.restype = C.c_int
    lib.lora_amd_chol_inverse_batched.restype = C.c_int
    lib.lora_amd_ragged_plan.argtypes = [i32, vp, i32, i32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.lora_amd_rowdot_ragged.argtypes = [vp, i32, i64, i32, i32, f32, vp]
    lib.lora_amd_colreduce_ragged.argtypes = [vp, i32, i64, i64, i32, i32, f32, vp]
    lib.lora_amd_sub_ragged.argtypes = [vp, i32, i64, i32, vp]
    lib.lora_amd_rowdot16_planes_plan.argtypes = [vp, i32, vp]
    lib.lora_amd_rowdot16_planes.argtypes = [vp, i32, i64, i32, i32, vp]
    lib.lora_amd_split16_ragged.argtypes = [vp, i32, i64, i32, vp]
    lib.lora_amd_rowdot16_planes_plan.restype = lib.lora_amd_rowdot16_planes.restype = C.c_int
    lib.lora_amd_rowdot16_planes_packed.argtypes = [vp, i32, i64, i32, i32, vp]
    lib.lora_amd_thin_pack.argtypes = [vp, vp, i64, vp, vp, i32, vp]
    lib.lora_amd_rowdot16_planes_packed.restype = lib.lora_amd_thin_pack.restype = C.c_int
    lib.lora_amd_split16_ragged.restype = C.c_int
    lib.lora_amd_split16_transpose.argtypes = [vp, i32, i64, i32, vp]
    lib.lora_amd_split16_transpose.restype = C.c_int
    lib.lora_amd_split16_residual.argtypes = [vp, i32, i64, i32, i32, vp, vp]
    lib.lora_amd_split16_residual.restype = C.c_int
    lib.lora_amd_thin_gram.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    lib.lora_amd_thin_apply.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp]
    lib.lora_amd_thin_rotate.argtypes = [vp, vp, i64, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.lora_amd_thin_select.argtypes = [vp, vp, i64, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp]
    lib.lora_amd_thin_clamp.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp, i32, vp]
    lib.lora_amd_thin_gram.restype = lib.lora_amd_thin_apply.restype = lib.lora_amd_thin_rotate.restype = C.c_int
    lib.lora_amd_thin_select.restype = lib.lora_amd_thin_clamp.restype = C.c_int
    lib.lora_amd_ragged_plan.restype = lib.lora_amd_rowdot_ragged.restype = C.c_int
    lib.lora_amd_colreduce_ragged.restype = lib.lora_amd_sub_ragged.restype = C.c_int
    lib.lora_amd_sumsq_workspace.argtypes = [i64]
    lib.lora_amd_sumsq_workspace.restype = sz
    lib.lora_amd_sumsq.argtypes = [vp, i64, vp, vp, sz, vp]
    lib.lora_amd_clip_adamw.argtypes = [vp, vp, vp, vp, i64, vp, i32, vp, f32, f32, f32, f32, f32, i64, i32, vp]
    lib.lora_amd_clip_adamw_dev.argtypes = [vp, vp, vp, vp, i64, vp, i32, vp, f32, f32, f32, f32, f32, vp, vp, i32, vp]
    lib.lora_amd_loss_scale_update.argtypes = [vp, vp, vp, f32, f32, i32, vp]
    lib.lora_amd_loss_scale_update.restype = C.c_int
    lib.lora_amd_step_advance.argtypes = [vp, vp]
    lib.lora_amd_ti_rows_step.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp, f32, f32, f32, f32, f32, f32, i64, f32, f32,
                                          vp]
    lib.lora_amd_ti_rows_step.restype = C.c_int
    lib.lora_amd_clip_adamw_dev.restype = lib.lora_amd_step_advance.restype = C.c_int
    lib.lora_amd_linear_plan.argtypes = [i64, i32, i32, i32, C.POINTER(LinearPlan)]
    lib.lora_amd_linear_fwd.argtypes = [vp, i64, vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, i32, f32, vp, f32, u64,
                                        u64, vp, vp]
    lib.lora_amd_linear_bwd_g.argtypes = [vp, i64, vp, vp, vp, vp, i64, i32, i32, i32, i32, f32, f32, u64, u64, vp, vp]
    lib.lora_amd_linear_bwd_x.argtypes = [vp, i64, vp, i64, vp, i32, vp, vp, vp, i64, i32, i32, i32, i32, vp]
    lib.lora_amd_linear_bwd_factors.argtypes = [vp, i64, vp, vp, vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, f32, vp]
    lib.lora_amd_linear_bwd_factors.restype = C.c_int
    lib.lora_amd_linear_bwd_factors_drop.argtypes = [vp, i64, vp, vp, vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, f32,
                                                     f32, u64, u64, vp, vp]
    lib.lora_amd_linear_bwd_factors_drop.restype = C.c_int
    lib.lora_amd_linear_bwd_factors_heads.argtypes = [vp, i64, vp, vp, vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, f32,
                                                      i32, i32, i32, i32, vp]
    lib.lora_amd_linear_bwd_factors_heads.restype = C.c_int
    lib.lora_amd_linear_factors_self_plan.argtypes = [i64, i32, i32, i32, C.POINTER(FactorsSelfPlan)]
    lib.lora_amd_linear_bwd_factors_self.argtypes = [vp, i64, vp, i64, vp, vp, vp, vp, i64, i32, i32, i32, i32, f32,
                                                     i32, i32, i32, i32, vp]
    lib.lora_amd_linear_factors_self_plan.restype = lib.lora_amd_linear_bwd_factors_self.restype = C.c_int
    lib.lora_amd_linear_factors_self_plan_rows.argtypes = [i64, i32, i32, i32, i32, C.POINTER(FactorsSelfPlan)]
    lib.lora_amd_linear_factors_self_plan_rows.restype = C.c_int
    lib.lora_amd_linear_factors_self_ragged_plan.argtypes = [vp, i32, i32, C.POINTER(C.c_int64)]
    lib.lora_amd_linear_bwd_factors_self_ragged.argtypes = [vp, i32, i64, i32, i32, vp]
    lib.lora_amd_linear_factors_self_ragged_plan.restype = lib.lora_amd_linear_bwd_factors_self_ragged.restype = C.c_int
    lib.lora_amd_factors_mfma_plan.argtypes = [i64, i32, i32, i32, i32, i32, i32, C.POINTER(FactorsMfmaPlan)]
    lib.lora_amd_factor_pack_plan.argtypes = [C.POINTER(PackSite), i32, C.POINTER(i64)]
    lib.lora_amd_factor_pack.argtypes = [vp, i32, i64, i32, vp]
    lib.lora_amd_factors_mfma_ragged_plan.argtypes = [C.POINTER(FmSite), i32, i32, i32, C.POINTER(i64)]
    lib.lora_amd_linear_bwd_factors_mfma_ragged.argtypes = [vp, i32, i64, i32, i32, i32, i32, vp]
    lib.lora_amd_factors_mfma_block_map.argtypes = [C.POINTER(FmSite), i32, i64, C.POINTER(i32)]
    lib.lora_amd_linear_bwd_factors_mfma_ragged_mapped.argtypes = [vp, i32, i64, vp, i32, i32, i32, i32, vp]
    lib.lora_amd_factors_mfma_block_map.restype = lib.lora_amd_linear_bwd_factors_mfma_ragged_mapped.restype = C.c_int
    for name in ("lora_amd_factors_mfma_plan", "lora_amd_factor_pack_plan", "lora_amd_factor_pack",
                 "lora_amd_factors_mfma_ragged_plan", "lora_amd_linear_bwd_factors_mfma_ragged"):
        getattr(lib, name).restype = C.c_int
    lib.lora_amd_linear_gemm_fwd_heads.argtypes = [vp, i64, vp, i64, vp, vp, i64, vp, vp, vp, i64, i32, i32, i32, i32,
                                                   f32, f32, i32, i32, i32, i32, i32, i32, vp]
    lib.lora_amd_linear_gemm_fwd_heads.restype = C.c_int
    lib.lora_amd_reduce_batched.argtypes = [vp, i32, i64, vp]
    lib.lora_amd_linear_gemm_supported.argtypes = [i64, i32, i32, i32, i32]
    lib.lora_amd_linear_gemm_fwd.argtypes = [vp, i64, vp, i64, vp, vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, f32, f32, i32,
                                             i32, vp]
    lib.lora_amd_linear_gemm_supported.restype = lib.lora_amd_linear_gemm_fwd.restype = C.c_int
    lib.lora_amd_ws_config.argtypes = [i32, C.POINTER(i32), C.POINTER(i32)]
    lib.lora_amd_ws_packed_elems.argtypes = [i32, i32]
    lib.lora_amd_ws_packed_elems.restype = i64
    lib.lora_amd_ws_pack.argtypes = [vp, i64, i64, i32, i32, i32, vp, vp]
    lib.lora_amd_linear_ws.argtypes = [vp, i64, i64, i32, i32, C.POINTER(WsSite), i32, i32, vp]
    lib.lora_amd_linear_ws_heads.argtypes = [vp, i64, i64, i32, i32, i32, i32, C.POINTER(WsSite), i32, i32, vp]
    lib.lora_amd_ws_config.restype = lib.lora_amd_ws_pack.restype = lib.lora_amd_linear_ws.restype = C.c_int
    lib.lora_amd_linear_ws_heads.restype = C.c_int
    lib.lora_amd_conv_plan.argtypes = [i32, i32, i32, i32, i32, i32, i32, C.POINTER(ConvPlan)]
    lib.lora_amd_conv_down_fwd.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.lora_amd_conv_up_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, f32, u64, u64, vp, vp]
    lib.lora_amd_conv_bwd_g.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, f32, u64,
                                        u64, vp, vp]
    lib.lora_amd_conv_bwd_x.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.lora_amd_conv3_nhwc_plan.argtypes = [i32, i32, i32, i32, i32, C.POINTER(Conv3NhwcPlan)]
    lib.lora_amd_conv3_nhwc_pack.argtypes = [vp, i32, i32, i32, vp, vp, vp]
    lib.lora_amd_conv3_nhwc_down_fwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.lora_amd_conv3_nhwc_bwd_dx.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.lora_amd_conv3_nhwc_bwd_down.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.lora_amd_sum_parts.argtypes = [vp, i32, i64, vp, i64, vp]
    lib.lora_amd_conv3_nhwc_pack_plan.argtypes = [C.POINTER(Conv3PackSite), i32, C.POINTER(i64)]
    lib.lora_amd_conv3_nhwc_pack_batched.argtypes = [vp, i32, i64, i32, vp]
    lib.lora_amd_conv3_nhwc_fwd_fused.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, f32,
                                                  u64, u64, vp, vp]
    lib.lora_amd_linear_bwd_g_blocks.argtypes = [i64, i32, i32, C.POINTER(i64)]
    lib.lora_amd_linear_bwd_g_folded.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, f32, f32, u64,
                                                 u64, vp, vp]
    for name in ("lora_amd_conv3_nhwc_plan", "lora_amd_conv3_nhwc_pack", "lora_amd_conv3_nhwc_down_fwd",
                 "lora_amd_conv3_nhwc_bwd_dx", "lora_amd_conv3_nhwc_bwd_down", "lora_amd_sum_parts",
                 "lora_amd_conv3_nhwc_pack_plan", "lora_amd_conv3_nhwc_pack_batched", "lora_amd_conv3_nhwc_fwd_fused",
                 "lora_amd_linear_bwd_g_blocks", "lora_amd_linear_bwd_g_folded"):
        getattr(lib, name).restype = C.c_int
    lib.lora_amd_groupnorm_workspace.argtypes = [i32, i32, i32, i32]
    lib.lora_amd_groupnorm_workspace.restype = sz
    lib.lora_amd_groupnorm_supported.argtypes = [i32, i32, i32, i32]
    lib.lora_amd_groupnorm_fwd.argtypes = [vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, f32, i32, i32, vp]
    lib.lora_amd_groupnorm_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, i32, i32, vp]
    lib.lora_amd_geglu_fwd.argtypes = [vp, i64, vp, i64, i64, i32, i32, vp]
    lib.lora_amd_geglu_bwd.argtypes = [vp, i64, vp, i64, vp, i64, i64, i32, i32, vp]
    lib.lora_amd_groupnorm_nhwc_workspace.argtypes = [i32, i32, i32, i32]
    lib.lora_amd_groupnorm_nhwc_workspace.restype = sz
    lib.lora_amd_groupnorm_nhwc_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, f32, i32, i32, vp]
    lib.lora_amd_groupnorm_nhwc_bwd.argtypes = [vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, i32, i32, vp]
    lib.lora_amd_groupnorm_nhwc_fwd.restype = lib.lora_amd_groupnorm_nhwc_bwd.restype = C.c_int
    lib.lora_amd_add_layernorm_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, i32, vp]
    lib.lora_amd_add_layernorm_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, i32, vp]
    lib.lora_amd_add_layernorm_fwd.restype = lib.lora_amd_add_layernorm_bwd.restype = C.c_int
    lib.lora_amd_layernorm_supported.argtypes = [i32]
    lib.lora_amd_layernorm_fwd.argtypes = [vp, vp, vp, vp, vp, i64, i32, f32, i32, vp]
    lib.lora_amd_layernorm_bwd.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, vp]
    for name in ("lora_amd_groupnorm_supported", "lora_amd_groupnorm_fwd", "lora_amd_groupnorm_bwd",
                 "lora_amd_geglu_fwd", "lora_amd_geglu_bwd", "lora_amd_layernorm_supported",
                 "lora_amd_layernorm_fwd", "lora_amd_layernorm_bwd"):
        getattr(lib, name).restype = C.c_int
    for name in ("lora_amd_conv_plan", "lora_amd_conv_down_fwd", "lora_amd_conv_up_fwd", "lora_amd_conv_bwd_g",
                 "lora_amd_conv_bwd_x"):
        getattr(lib, name).restype = C.c_int
    for name in ("lora_amd_linear_plan", "lora_amd_linear_fwd", "lora_amd_linear_bwd_g", "lora_amd_linear_bwd_x",
                 "lora_amd_reduce_batched"):
        getattr(lib, name).restype = C.c_int
    for name in ("lora_amd_merge_plan", "lora_amd_merge_batched", "lora_amd_merge_set_tuning", "lora_amd_rowdot",
                 "lora_amd_rowdot_masked", "lora_amd_rank_update", "lora_amd_colreduce", "lora_amd_sumsq",
                 "lora_amd_clip_adamw"):
        getattr(lib, name).restype = C.c_int


def load() -> Optional[C.CDLL]:
    """Load the library once; returns None (and records why) if that fails."""
    global _lib, _load_error
    if _lib is not None or _load_error is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        _load_error = f"{LIB_PATH} not built (run `python -c 'import __graft_entry__ as g; g.build()'`)"
        return None
    try:
        lib = C.CDLL(LIB_PATH)
        _declare(lib)
        if lib.lora_amd_abi_version() != ABI_VERSION:
            raise OSError(f"ABI version {lib.lora_amd_abi_version()} != {ABI_VERSION} (rebuild: make -C lora_amd/csrc)")
        _lib = lib
    except OSError as e:  # pragma: no cover - environment dependent
        _load_error = f"cannot load {LIB_PATH}: {e}"
    return _lib


def available() -> bool:
    return load() is not None


def require() -> C.CDLL:
    lib = load()
    if lib is None:
        raise HipExtensionMissing(
            "lora_amd: the HIP extension is required for device tensors and is not loadable: " + str(_load_error))
    return lib


def _check(rc: int, what: str) -> None:
    if rc != OK:
        msg = require().lora_amd_last_error().decode()
        if rc == -2:
            raise ValueError(f"{what}: {msg}")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DT[dt]
    except KeyError:
        raise TypeError(f"lora_amd kernels support float32/float16/bfloat16, got {dt}") from None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _off(offset):
    """Dropout offset argument pair (scalar, device pointer): a python int, or a 1-element int64 device tensor whose
    value the kernel reads at run time (hipGraph-replay / recompute-safe, see ``ops.next_dropout_stream``)."""
    if torch.is_tensor(offset):
        if offset.dtype != torch.int64 or not offset.is_cuda or offset.numel() != 1:
            raise ValueError("dropout offset tensor must be a 1-element int64 device tensor")
        return 0, offset.data_ptr()
    return int(offset), None


def _dev_check(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise ValueError("lora_amd kernels take device tensors only")


# ----------------------------------------------------------------------------- merge (K3)
class MergePlan:
    """Planned descriptor table for one (w_dtype, ab_dtype) group of sites, resident on the device."""

    def __init__(self, sites: Sequence[Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]]):
        lib = require()
        if len(sites) == 0:
            raise ValueError("MergePlan: no sites")
        w0, _, up0, _ = sites[0][:4]
        self.w_dtype, self.ab_dtype = w0.dtype, up0.dtype
        self.device = w0.device
        n = len(sites)
        arr = (MergeSite * n)()
        self.keep = []  # keep tensors alive as long as the plan
        self.bytes_algorithmic = 0
        for i, site in enumerate(sites):
            w_in, w_out, up, down = site[:4]
            heads = site[4] if len(site) > 4 else None  # (d, D): w_out is [N, (K/d)*D], input columns head-padded
            transposed = bool(site[5]) if len(site) > 5 else False  # see lora_amd_merge_site.transposed
            _dev_check(w_in, w_out, up, down)
            if w_in.dtype != self.w_dtype or w_out.dtype != self.w_dtype:
                raise TypeError("MergePlan: mixed weight dtypes in one plan")
            if up.dtype != self.ab_dtype or down.dtype != self.ab_dtype:
                raise TypeError("MergePlan: mixed factor dtypes in one plan")
            if not (w_in.is_contiguous() and w_out.is_contiguous() and up.is_contiguous() and down.is_contiguous()):
                raise ValueError("MergePlan: tensors must be contiguous")
            N = w_in.shape[0]
            K = w_in.numel() // N
            r = up.shape[0] if transposed else down.shape[0]  # transposed: `up` is the original down [r, N]
            ko = K if heads is None else (K // heads[0]) * heads[1]
            if (up.shape[0] != (r if transposed else N) or up.numel() != N * r or down.numel() != r * K
                    or w_out.numel() != N * ko):
                raise ValueError(f"MergePlan: site {i} shape mismatch W{tuple(w_in.shape)} up{tuple(up.shape)} "
                                 f"down{tuple(down.shape)}")
            s = arr[i]
            s.w_in, s.w_out, s.up, s.down = w_in.data_ptr(), w_out.data_ptr(), up.data_ptr(), down.data_ptr()
            s.N, s.K, s.r = N, K, r
            s.out_heads = 0 if heads is None else int(heads[0]) | (int(heads[1]) << 16)
            s.transposed = int(transposed)
            self.keep.append((w_in, w_out, up, down))
            self.bytes_algorithmic += 2 * N * K * w_in.element_size() + (N + K) * r * up.element_size()
        self.summary = MergeSummary()
        _check(lib.lora_amd_merge_plan(arr, n, dtype_code(self.w_dtype), C.byref(self.summary)), "lora_amd_merge_plan")
        self.n_sites, self.total_tiles = n, self.summary.total_tiles
        self.host = arr
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self.table = raw.to(self.device)

    def launch(self, alpha: float = 1.0, rounding: int = ROUND_REFERENCE) -> None:
        lib = require()
        _check(lib.lora_amd_merge_batched(self.table.data_ptr(), self.n_sites, C.byref(self.summary),
                                          dtype_code(self.w_dtype), dtype_code(self.ab_dtype), float(alpha),
                                          int(rounding), _stream()), "lora_amd_merge_batched")


class MergeStepPlan:
    """Planned table of the in-step merge (``lora_amd_merge_step``): per site the frozen 16-bit weight, the f32 factors and
    where W_eff / W_eff^T go.  ``sites``: dicts with ``w`` [N, K], ``up`` [N, r], ``down`` [r, K], ``out`` (2-D view whose
    row stride is ld_out), ``out_t`` (2-D view or None), ``row_heads`` / ``col_heads`` = (d, D) or None, ``key`` (dither)."""

    def __init__(self, sites):
        if not sites:
            raise ValueError("MergeStepPlan: no sites")
        lib = require()
        self.w_dtype, self.device = sites[0]["w"].dtype, sites[0]["w"].device
        arr = (MstepSite * len(sites))()
        self.keep, self.bytes_algorithmic, self.rank_max = [], 0, 1
        for q, st in zip(arr, sites):
            w, up, down, out, out_t = st["w"], st["up"], st["down"], st["out"], st.get("out_t")
            _dev_check(w, up, down, out)
            if w.dtype != self.w_dtype or out.dtype != self.w_dtype or (out_t is not None and out_t.dtype != self.w_dtype):
                raise TypeError("MergeStepPlan: mixed weight dtypes in one plan")
            if up.dtype != torch.float32 or down.dtype != torch.float32:
                raise TypeError("MergeStepPlan: f32 factors expected")
            if not (w.is_contiguous() and up.is_contiguous() and down.is_contiguous()) or out.stride(1) != 1 or \
                    (out_t is not None and out_t.stride(1) != 1):
                raise ValueError("MergeStepPlan: contiguous W / factors and unit inner stride of the outputs expected")
            N, K = w.shape
            r = down.shape[0]
            rh, ch = st.get("row_heads"), st.get("col_heads")
            np_, kp = ((N // rh[0]) * rh[1] if rh else N), ((K // ch[0]) * ch[1] if ch else K)
            if tuple(up.shape) != (N, r) or tuple(down.shape) != (r, K) or tuple(out.shape) != (np_, kp) or \
                    (out_t is not None and tuple(out_t.shape) != (kp, np_)):
                raise ValueError(f"MergeStepPlan: shape mismatch W{tuple(w.shape)} up{tuple(up.shape)} down{tuple(down.shape)} "
                                 f"out{tuple(out.shape)}")
            q.w, q.up, q.down, q.out = w.data_ptr(), up.data_ptr(), down.data_ptr(), out.data_ptr()
            q.out_t = out_t.data_ptr() if out_t is not None else None
            q.ld_out, q.ld_out_t = out.stride(0), (out_t.stride(0) if out_t is not None else 0)
            q.N, q.K, q.r = N, K, r
            q.row_d, q.row_D = rh if rh else (0, 0)
            q.col_d, q.col_D = ch if ch else (0, 0)
            q.dither_key = int(st.get("key", 0)) & 0x7FFFFFFF
            self.rank_max = max(self.rank_max, r)
            self.keep.append((w, up, down, out, out_t))
            self.bytes_algorithmic += (2 + (out_t is not None)) * N * K * w.element_size() + (N + K) * r * 4
        total = C.c_int64(0)
        _check(lib.lora_amd_merge_step_plan(arr, len(sites), dtype_code(self.w_dtype), C.byref(total)),
               "lora_amd_merge_step_plan")
        self.n_sites, self.plan_value, self.host = len(sites), total.value, arr
        self.total_tiles = total.value & ((1 << 40) - 1)
        self.table = table_to_device(arr, self.device)

    def launch(self, alpha: float = 1.0, rounding: int = ROUND_ONCE) -> None:
        _check(require().lora_amd_merge_step(self.table.data_ptr(), self.n_sites, self.plan_value, self.rank_max,
                                             dtype_code(self.w_dtype), float(alpha), int(rounding), _stream()),
               "lora_amd_merge_step")


def merge_set_tuning(tile_elems: int = 0, blocks_per_cu: int = 0) -> None:
    require().lora_amd_merge_set_tuning(int(tile_elems), int(blocks_per_cu))


def factors_mfma_set_tuning(narrow: int = -1) -> int:
    """Kernel of class-1 tables of the matrix-core factor pass (0 wide, 1 narrow / ring 2, 2 narrow / ring 4); returns the previous."""
    lib = require()
    lib.lora_amd_factors_mfma_set_tuning.argtypes = [C.c_int32]
    lib.lora_amd_factors_mfma_set_tuning.restype = C.c_int
    return int(lib.lora_amd_factors_mfma_set_tuning(int(narrow)))


def rank16_mfma(enable: int = -1) -> int:
    """Matrix-core forms of the rank-9..16 kernels on / off (csrc/rank16_mfma.hip); returns the previous setting."""
    return int(require().lora_amd_rank16_mfma(int(enable)))


def merge_step_set_tuning(tile: int = -1, dither: int = -1) -> None:
    _check(require().lora_amd_merge_step_set_tuning(int(tile), int(dither)), "lora_amd_merge_step_set_tuning")


# ----------------------------------------------------------------------------- K1/K2 primitives
def _as2d(x: torch.Tensor) -> torch.Tensor:
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("expected a 2-D tensor with unit inner stride")
    return x


def rowdot(x: torch.Tensor, factor: torch.Tensor, layout: int, scale: float = 1.0,
           sel: Optional[torch.Tensor] = None, sel_transposed: bool = False, dropout_p: float = 0.0,
           seed: int = 0, offset: int = 0) -> torch.Tensor:
    """T[M,r] (f32) = scale * (mask*X)[M,K] @ F^T (@ S^T | @ S)."""
    lib = require()
    _dev_check(x, factor, sel)
    x = _as2d(x)
    M, K = x.shape
    factor = factor.contiguous()
    r = factor.shape[0] if layout == FACTOR_RK else factor.shape[1]
    if factor.numel() != r * K:
        raise ValueError(f"rowdot: factor {tuple(factor.shape)} does not match K={K}")
    if sel is not None:
        sel = sel.to(torch.float32).contiguous()
        if sel.shape != (r, r):
            raise ValueError("rowdot: selector must be [r, r]")
    t = torch.empty((M, r), dtype=torch.float32, device=x.device)
    _check(lib.lora_amd_rowdot_masked(x.data_ptr(), x.stride(0), factor.data_ptr(), t.data_ptr(), M, K, r,
                                      dtype_code(x.dtype), dtype_code(factor.dtype), layout, float(scale),
                                      sel.data_ptr() if sel is not None else None, int(bool(sel_transposed)),
                                      float(dropout_p), int(seed), *_off(offset), _stream()), "lora_amd_rowdot")
    return t


def rank_update_(y: torch.Tensor, t: torch.Tensor, factor: torch.Tensor, layout: int, scale: float = 1.0,
                 dropout_p: float = 0.0, seed: int = 0, offset: int = 0) -> torch.Tensor:
    """Y[M,N] += scale * mask * T[M,r] @ F (in place)."""
    lib = require()
    _dev_check(y, t, factor)
    y = _as2d(y)
    M, N = y.shape
    factor = factor.contiguous()
    r = factor.shape[0] if layout == FACTOR_RK else factor.shape[1]
    if t.dtype != torch.float32 or t.shape != (M, r) or not t.is_contiguous():
        raise ValueError(f"rank_update: T must be contiguous f32 [{M},{r}]")
    if factor.numel() != r * N:
        raise ValueError(f"rank_update: factor {tuple(factor.shape)} does not match N={N}")
    _check(lib.lora_amd_rank_update(y.data_ptr(), y.stride(0), t.data_ptr(), factor.data_ptr(), M, N, r,
                                    dtype_code(y.dtype), dtype_code(factor.dtype), layout, float(scale),
                                    float(dropout_p), int(seed), *_off(offset), _stream()), "lora_amd_rank_update")
    return y


def colreduce(x: torch.Tensor, t: torch.Tensor, layout: int, scale: float = 1.0,
              out: Optional[torch.Tensor] = None, beta: float = 0.0, dropout_p: float = 0.0, seed: int = 0,
              offset: int = 0) -> torch.Tensor:
    """D (f32, [r,K] or [K,r]) = beta*D + scale * T^T @ (mask*X)."""
    lib = require()
    _dev_check(x, t, out)
    x = _as2d(x)
    M, K = x.shape
    r = t.shape[1]
    if t.dtype != torch.float32 or t.shape[0] != M or not t.is_contiguous():
        raise ValueError("colreduce: T must be contiguous f32 [M, r]")
    shape = (r, K) if layout == FACTOR_RK else (K, r)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
        beta = 0.0
    elif out.dtype != torch.float32 or out.numel() != r * K or not out.is_contiguous():
        raise ValueError("colreduce: out must be contiguous f32 with r*K elements")
    ws_bytes = lib.lora_amd_colreduce_workspace(M, K, r)
    ws = torch.empty(max(ws_bytes // 4, 1), dtype=torch.float32, device=x.device)
    _check(lib.lora_amd_colreduce(x.data_ptr(), x.stride(0), t.data_ptr(), out.data_ptr(), M, K, r,
                                  dtype_code(x.dtype), layout, float(scale), float(beta), float(dropout_p),
                                  int(seed), *_off(offset), ws.data_ptr(), ws.numel() * 4, _stream()),
           "lora_amd_colreduce")
    return out


# ----------------------------------------------------------------------------- batched forms (SVD distillation)
def rowdot_batched(x: torch.Tensor, factor: torch.Tensor, layout: int = FACTOR_RK, scale: float = 1.0) -> torch.Tensor:
    """T [B, M, r] (f32) = scale * X [B, M, K] @ F^T for a stack of matrices: ONE launch (grid.y = B)."""
    lib = require()
    _dev_check(x, factor)
    if x.dim() != 3 or factor.dim() != 3 or not x.is_contiguous() or not factor.is_contiguous():
        raise ValueError("rowdot_batched: contiguous [B, M, K] and [B, r, K] / [B, K, r] stacks expected")
    B, M, K = x.shape
    r = factor.shape[1] if layout == FACTOR_RK else factor.shape[2]
    if factor.shape[0] != B or factor.numel() != B * r * K:
        raise ValueError("rowdot_batched: factor stack does not match")
    t = torch.empty((B, M, r), dtype=torch.float32, device=x.device)
    _check(lib.lora_amd_rowdot_batched(x.data_ptr(), K, M * K, factor.data_ptr(), r * K, t.data_ptr(), M * r, B, M, K, r,
                                       dtype_code(x.dtype), dtype_code(factor.dtype), layout, float(scale), _stream()),
           "lora_amd_rowdot_batched")
    return t


def colreduce_batched(x: torch.Tensor, t: torch.Tensor, layout: int = FACTOR_RK, scale: float = 1.0) -> torch.Tensor:
    """D [B, r, K] (or [B, K, r]) (f32) = scale * T^T @ X for stacks X [B, M, K], T [B, M, r]: ONE launch pair."""
    lib = require()
    _dev_check(x, t)
    if x.dim() != 3 or t.dim() != 3 or not x.is_contiguous() or not t.is_contiguous() or t.dtype != torch.float32:
        raise ValueError("colreduce_batched: contiguous [B, M, K] and f32 [B, M, r] stacks expected")
    B, M, K = x.shape
    r = t.shape[2]
    out = torch.empty((B, r, K) if layout == FACTOR_RK else (B, K, r), dtype=torch.float32, device=x.device)
    nbytes = lib.lora_amd_colreduce_workspace(M, K, r) * B
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=x.device)
    _check(lib.lora_amd_colreduce_batched(x.data_ptr(), K, M * K, t.data_ptr(), M * r, out.data_ptr(), r * K, B, M, K, r,
                                          dtype_code(x.dtype), layout, float(scale), ws.data_ptr(), ws.numel() * 4,
                                          _stream()), "lora_amd_colreduce_batched")
    return out


def chol_inverse_batched(gram: torch.Tensor, shift_rel: float = 0.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """L^{-1} [B, l, l] for G + shift_rel * tr(G)/l * I = L L^T (l <= 32): the small dense step of CholeskyQR."""
    lib = require()
    _dev_check(gram)
    if gram.dim() != 3 or gram.shape[1] != gram.shape[2] or gram.dtype != torch.float32 or not gram.is_contiguous():
        raise ValueError("chol_inverse_batched: contiguous f32 [B, l, l] expected")
    if out is None:
        out = torch.empty_like(gram)
    elif out.shape != gram.shape or out.dtype != torch.float32 or not out.is_contiguous() or not out.is_cuda:
        raise ValueError("chol_inverse_batched: out must match gram")
    _check(lib.lora_amd_chol_inverse_batched(gram.data_ptr(), out.data_ptr(), gram.shape[1], gram.shape[0],
                                             float(shift_rel), _stream()), "lora_amd_chol_inverse_batched")
    return out


RAGGED_ROWDOT, RAGGED_COLREDUCE = 0, 1


class RaggedProgram:
    """Descriptor tables of a sequence of ragged launches over persistent buffers (``cli_svd.distill_model``): every
    table is planned on the host as it is declared, ALL of them go to the device in one copy (``upload``), and a launch
    is then one C call with a pointer into that buffer — no per-launch allocation, planning or host-to-device traffic."""

    def __init__(self, device):
        self.device = device
        self._blobs, self._meta, self._size = [], [], 0
        self._dev = None

    def table(self, op: int, r: int, rows) -> int:
        """rows: per stack (x [B, M, K], f [B, ...], out [B, ...], partial or None).  Returns the table's handle."""
        lib = require()
        arr = (RaggedDesc * len(rows))()
        for d, (x, f, out, partial) in zip(arr, rows):
            B, M, K = x.shape
            if x.dtype != torch.float32 or f.dtype != torch.float32 or out.dtype != torch.float32:
                raise TypeError("ragged launches take f32 stacks")
            if not (x.is_contiguous() and f.is_contiguous() and out.is_contiguous()):
                raise ValueError("ragged launches take contiguous stacks")
            d.x, d.f, d.out = x.data_ptr(), f.data_ptr(), out.data_ptr()
            d.partial = partial.data_ptr() if partial is not None else None
            d.ldx, d.stride_x, d.M, d.K, d.batch = K, M * K, M, K, B
            d.stride_f, d.stride_out = f[0].numel(), out[0].numel()
            if partial is not None and partial.numel() * 4 < lib.lora_amd_colreduce_workspace(M, K, r) * B:
                raise ValueError("ragged colreduce: workspace too small")
        g1, g2 = C.c_int64(0), C.c_int64(0)
        _check(lib.lora_amd_ragged_plan(op, arr, len(rows), r, C.byref(g1), C.byref(g2)), "lora_amd_ragged_plan")
        off = self._size
        blob = bytes(arr)
        pad = (-len(blob)) % 64
        self._blobs.append(blob + b"\0" * pad)
        self._size += len(blob) + pad
        self._meta.append((op, r, len(rows), off, g1.value, g2.value))
        return len(self._meta) - 1

    def upload(self):
        self._dev = torch.frombuffer(bytearray(b"".join(self._blobs)), dtype=torch.uint8).to(self.device)

    def run(self, handle: int, layout: int, scale: float = 1.0) -> None:
        lib = require()
        op, r, n, off, g1, g2 = self._meta[handle]
        ptr = self._dev.data_ptr() + off
        if op == RAGGED_ROWDOT:
            _check(lib.lora_amd_rowdot_ragged(ptr, n, g1, r, layout, float(scale), _stream()), "lora_amd_rowdot_ragged")
        else:
            _check(lib.lora_amd_colreduce_ragged(ptr, n, g1, g2, r, layout, float(scale), _stream()),
                   "lora_amd_colreduce_ragged")


class PlanesProgram:
    """Descriptor tables of the matrix-core skinny products over (hi, lo) 16-bit planes (``lora_amd_rowdot16_planes``):
    declared once (``table``), uploaded in one copy, each run one C call.  rows: per shape group (hi [B, M, C], lo, f [B, C, r]
    f32, out [B, M, r] f32)."""

    def __init__(self, device, r: int, plane_dtype: torch.dtype = torch.bfloat16, packed: bool = False):
        """``packed``: the factor of every row is given as fragments (``thin_pack``): a 16-bit tensor [B, C * 32] (hi / lo
        1 KB blocks per 32 rows of the [C, 16] factor) instead of f32 [B, C, r]; r = 16."""
        self.device, self.r, self.dt, self.packed = device, int(r), plane_dtype, bool(packed)
        if packed and self.r != 16:
            raise ValueError("PlanesProgram: packed factors are 16 columns wide")
        self._blobs, self._meta, self._size, self._dev = [], [], 0, None

    def table(self, rows) -> int:
        lib = require()
        arr = (PlanesDesc * len(rows))()
        for d, (hi, lo, f, out) in zip(arr, rows):
            B, M, Cc = hi.shape
            fdt, fshape = (self.dt, (B, Cc * 32)) if self.packed else (torch.float32, (B, Cc, self.r))
            if hi.dtype != self.dt or lo.dtype != self.dt or f.dtype != fdt or out.dtype != torch.float32:
                raise TypeError("PlanesProgram: 16-bit planes, f32 (or packed 16-bit) factor and f32 output expected")
            if tuple(lo.shape) != (B, M, Cc) or tuple(f.shape) != fshape or tuple(out.shape) != (B, M, self.r):
                raise ValueError(f"PlanesProgram: shape mismatch {tuple(hi.shape)} {tuple(f.shape)} {tuple(out.shape)}")
            if not (hi.is_contiguous() and lo.is_contiguous() and f.is_contiguous() and out.is_contiguous()):
                raise ValueError("PlanesProgram: contiguous stacks expected")
            d.hi, d.lo, d.f, d.out, d.M, d.C, d.batch = hi.data_ptr(), lo.data_ptr(), f.data_ptr(), out.data_ptr(), M, Cc, B
        grid = C.c_int64(0)
        _check(lib.lora_amd_rowdot16_planes_plan(arr, len(rows), C.byref(grid)), "lora_amd_rowdot16_planes_plan")
        blob = bytes(arr)
        pad = (-len(blob)) % 64
        self._blobs.append(blob + b"\0" * pad)
        self._meta.append((len(rows), self._size, grid.value))
        self._size += len(blob) + pad
        return len(self._meta) - 1

    def upload(self):
        self._dev = torch.frombuffer(bytearray(b"".join(self._blobs)), dtype=torch.uint8).to(self.device)

    def run(self, handle: int, hi_only: bool = False) -> None:
        n, off, grid = self._meta[handle]
        if self.packed:
            _check(require().lora_amd_rowdot16_planes_packed(self._dev.data_ptr() + off, n, grid, dtype_code(self.dt),
                                                             1 if hi_only else 0, _stream()),
                   "lora_amd_rowdot16_planes_packed")
            return
        _check(require().lora_amd_rowdot16_planes(self._dev.data_ptr() + off, n, grid, self.r, dtype_code(self.dt), _stream()),
               "lora_amd_rowdot16_planes")


THIN_ROWS_PER_BLOCK, THIN_Q_ELEMS = 256, 8192


def split16_residual(groups, dims, dh, dl, th, tl) -> torch.Tensor:
    """Per site of every shape group: dW = tuned - base -> the (hi, lo) bf16 planes of dW (``dh[g][b]``, ``dl``) and of dW^T
    (``th``, ``tl``) from ONE read of the two weights (``lora_amd_split16_residual``; the f32 residual is never written).
    ``groups`` = [(tuned list, base list)], ``dims`` = [(B, N, K)].  Returns |dW|_F^2 per site [sum B] (f32, deterministic:
    per-tile partials of the launch summed in tile order)."""
    import numpy as np

    lib = require()
    nsites = sum(B for B, _, _ in dims)
    arr = (ResidDesc * nsites)()
    i, tiles, ends = 0, 0, []
    in_dt = groups[0][0][0].dtype
    for (tuned, base), (B, N, K), h, l, t_h, t_l in zip(groups, dims, dh, dl, th, tl):
        if N % 8 or K % 8:
            raise ValueError("split16_residual: N and K must be multiples of 8")
        hp, lp, thp, tlp = h.data_ptr(), l.data_ptr(), t_h.data_ptr(), t_l.data_ptr()
        nt = -(-N // 64) * -(-K // 64)
        for b in range(B):
            t_, b_ = tuned[b], base[b]
            if t_.dtype != in_dt or b_.dtype != in_dt or not (t_.is_contiguous() and b_.is_contiguous()) or t_.numel() != N * K:
                raise ValueError("split16_residual: contiguous weights of one dtype and of the group's size expected")
            d = arr[i]
            d.tuned, d.base = t_.data_ptr(), b_.data_ptr()
            o = b * N * K * 2
            d.hi, d.lo, d.thi, d.tlo, d.N, d.K, d.tile_begin = hp + o, lp + o, thp + o, tlp + o, N, K, tiles
            tiles += nt
            ends.append(tiles)
            i += 1
    dev = dh[0].device
    table = table_to_device(arr, dev)
    part = torch.empty(tiles, dtype=torch.float32, device=dev)
    _check(lib.lora_amd_split16_residual(table.data_ptr(), nsites, tiles, dtype_code(in_dt), dtype_code(dh[0].dtype),
                                         part.data_ptr(), _stream()), "lora_amd_split16_residual")
    c = torch.cumsum(part.double(), 0)
    e = torch.from_numpy(np.asarray(ends, dtype=np.int64) - 1).to(dev)
    tot = c[e]
    return torch.diff(tot, prepend=tot.new_zeros(1)).float()


class ThinTable:
    """Site table of the fused small steps of the subspace iteration (csrc/svd_small.hip): one entry per thin matrix
    [rows][16] f32 at element offset ``off`` of any flat buffer of that layout, 256-row blocks, a block -> site map."""

    def __init__(self, sites, device):
        """sites: [(off, rows)]."""
        import numpy as np

        arr = (ThinSite * len(sites))()
        blk, bm = 0, []
        for i, (d, (off, rows)) in enumerate(zip(arr, sites)):
            if off % 16:
                raise ValueError("ThinTable: site offsets must be multiples of 16 elements")
            nb = -(-int(rows) // THIN_ROWS_PER_BLOCK)
            d.off, d.rows, d.block_begin, d.blocks = int(off), int(rows), blk, nb
            bm += [i] * nb
            blk += nb
        self.n, self.total_blocks, self.device = len(sites), blk, device
        self.sites = table_to_device(arr, device)
        self.blockmap = torch.from_numpy(np.asarray(bm, dtype=np.int32)).to(device)
        self.part = torch.empty(blk * 256, dtype=torch.float32, device=device)
        self.counters = torch.zeros(len(sites), dtype=torch.int32, device=device)


class ThinQTable:
    """Per-site joint value ranges for the quantile selection / clamp: (off_u, n_u) in the `up` buffer, (off_v, n_v) in the
    `down` buffer; blocks of 8192 values."""

    def __init__(self, sites, device):
        import numpy as np

        arr = (ThinQSite * len(sites))()
        blk, bm = 0, []
        for i, (d, (off_u, n_u, off_v, n_v)) in enumerate(zip(arr, sites)):
            nb = -(-(int(n_u) + int(n_v)) // THIN_Q_ELEMS)
            d.off_u, d.off_v, d.n_u, d.n_v, d.block_begin, d.blocks = int(off_u), int(off_v), int(n_u), int(n_v), blk, nb
            bm += [i] * nb
            blk += nb
        self.n, self.total_blocks, self.device = len(sites), blk, device
        self.sites = table_to_device(arr, device)
        self.blockmap = torch.from_numpy(np.asarray(bm, dtype=np.int32)).to(device)
        self.hist = torch.zeros(len(sites) * 2048, dtype=torch.int32, device=device)
        self.counters = torch.zeros(len(sites), dtype=torch.int32, device=device)


def thin_finish(table: ThinTable, mode: int, rank: int, shift_rel: float = 0.0, linv_out=None, ritz_out=None, ubt=None,
                vb=None, s_out=None) -> ThinFinishDesc:
    f = ThinFinishDesc()
    f.part, f.counters, f.mode, f.shift_rel, f.rank = table.part.data_ptr(), table.counters.data_ptr(), mode, shift_rel, rank
    for name, t in (("linv_out", linv_out), ("ritz_out", ritz_out), ("ubt", ubt), ("vb", vb), ("s_out", s_out)):
        setattr(f, name, t.data_ptr() if t is not None else None)
    return f


def thin_gram(table: ThinTable, a: torch.Tensor, b, fin: ThinFinishDesc) -> None:
    _check(require().lora_amd_thin_gram(table.sites.data_ptr(), table.blockmap.data_ptr(), table.total_blocks, a.data_ptr(),
                                        b.data_ptr() if b is not None else None, C.byref(fin), _stream()), "lora_amd_thin_gram")


def thin_apply(table: ThinTable, src: torch.Tensor, mats: torch.Tensor, dst: torch.Tensor, fin=None) -> None:
    _check(require().lora_amd_thin_apply(table.sites.data_ptr(), table.blockmap.data_ptr(), table.total_blocks, src.data_ptr(),
                                         mats.data_ptr(), dst.data_ptr(), C.byref(fin) if fin is not None else None, _stream()),
           "lora_amd_thin_apply")


def thin_rotate(table: ThinTable, src, mats, rank: int, dst, scale_a=None, scale_b=None, sign_ws=None, sign_out=None) -> None:
    """sign_ws = (part [total_blocks * 32] f32, rows [total_blocks * 16] int32) when ``sign_out`` is wanted."""
    p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    _check(require().lora_amd_thin_rotate(table.sites.data_ptr(), table.blockmap.data_ptr(), table.total_blocks, src.data_ptr(),
                                          mats.data_ptr(), rank, p(scale_a), p(scale_b), dst.data_ptr(),
                                          p(sign_ws[0]) if sign_ws else None, p(sign_ws[1]) if sign_ws else None,
                                          table.counters.data_ptr() if sign_out is not None else None, p(sign_out), _stream()),
           "lora_amd_thin_rotate")


def thin_pack(table: ThinTable, src: torch.Tensor, dst: torch.Tensor) -> None:
    """src [rows][16] f32 per site -> dst: the (hi, lo) fragments of every site at 16-bit element offset 2 * off."""
    _check(require().lora_amd_thin_pack(table.sites.data_ptr(), table.blockmap.data_ptr(), table.total_blocks, src.data_ptr(),
                                        dst.data_ptr(), dtype_code(dst.dtype), _stream()), "lora_amd_thin_pack")


def thin_select(q: ThinQTable, u, v, sign, rank: int, pass_: int, state, out2) -> None:
    _check(require().lora_amd_thin_select(q.sites.data_ptr(), q.blockmap.data_ptr(), q.total_blocks, u.data_ptr(), v.data_ptr(),
                                          sign.data_ptr(), rank, pass_, q.hist.data_ptr(), q.counters.data_ptr(),
                                          state.data_ptr(), out2.data_ptr(), _stream()), "lora_amd_thin_select")


def thin_clamp(q: ThinQTable, u, v, sign, hi, down, rank: int) -> None:
    _check(require().lora_amd_thin_clamp(q.sites.data_ptr(), q.blockmap.data_ptr(), q.total_blocks, u.data_ptr(), v.data_ptr(),
                                         sign.data_ptr(), hi.data_ptr(), down.data_ptr(), rank, _stream()), "lora_amd_thin_clamp")


def split16_ragged(srcs, his, los) -> None:
    """(his[i], los[i]) = the 16-bit hi / lo planes of the flat f32 arrays srcs[i] (numel % 8 == 0), ONE launch."""
    lib = require()
    arr = (SplitDesc * len(srcs))()
    blocks, dt = 0, his[0].dtype
    for d, s_, h, l in zip(arr, srcs, his, los):
        _dev_check(s_, h, l)
        if s_.dtype != torch.float32 or h.dtype != dt or l.dtype != dt or s_.numel() != h.numel() or s_.numel() != l.numel() \
                or s_.numel() % 8 or not (s_.is_contiguous() and h.is_contiguous() and l.is_contiguous()):
            raise ValueError("split16_ragged: contiguous f32 sources and same-size 16-bit planes (numel % 8 == 0) expected")
        d.src, d.hi, d.lo, d.n, d.begin = s_.data_ptr(), h.data_ptr(), l.data_ptr(), s_.numel(), blocks
        blocks += (s_.numel() + 4095) // 4096
    dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(his[0].device)
    _check(lib.lora_amd_split16_ragged(dev.data_ptr(), len(srcs), blocks, dtype_code(dt), _stream()), "lora_amd_split16_ragged")


def split16_transpose(stacks, his, los, this, tlos) -> None:
    """Per f32 stack [B, N, K]: the (hi, lo) 16-bit planes [B, N, K] and those of the transposed matrices [B, K, N], ONE launch
    for all stacks, every stack read once."""
    lib = require()
    arr = (SplitTDesc * len(stacks))()
    tiles, dt = 0, his[0].dtype
    for d, x, h, l, th, tl in zip(arr, stacks, his, los, this, tlos):
        B, N, K = x.shape
        _dev_check(x, h, l, th, tl)
        if x.dtype != torch.float32 or any(t.dtype != dt for t in (h, l, th, tl)) or N % 8 or K % 8 \
                or tuple(h.shape) != (B, N, K) or tuple(l.shape) != (B, N, K) or tuple(th.shape) != (B, K, N) \
                or tuple(tl.shape) != (B, K, N) or not all(t.is_contiguous() for t in (x, h, l, th, tl)):
            raise ValueError("split16_transpose: contiguous f32 [B, N, K] (N, K multiples of 8) and matching 16-bit planes expected")
        d.src, d.hi, d.lo, d.thi, d.tlo = x.data_ptr(), h.data_ptr(), l.data_ptr(), th.data_ptr(), tl.data_ptr()
        d.batch, d.N, d.K, d.tile_begin = B, N, K, tiles
        tiles += B * (-(-N // 64)) * (-(-K // 64))
    dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(his[0].device)
    _check(lib.lora_amd_split16_transpose(dev.data_ptr(), len(stacks), tiles, dtype_code(dt), _stream()), "lora_amd_split16_transpose")


def colreduce_workspace_floats(M: int, K: int, r: int) -> int:
    return require().lora_amd_colreduce_workspace(M, K, r) // 4


def sub_ragged(pairs, outs) -> None:
    """outs[i] (f32, flat) = float(a_i) - float(b_i) for every (a_i, b_i) of ``pairs`` (same dtype) in ONE launch."""
    lib = require()
    arr = (SubDesc * len(pairs))()
    blocks = 0
    dt = pairs[0][0].dtype
    for d, (a, b), o in zip(arr, pairs, outs):
        _dev_check(a, b, o)
        if a.dtype != dt or b.dtype != dt or o.dtype != torch.float32 or a.numel() != b.numel() or a.numel() != o.numel():
            raise ValueError("sub_ragged: one input dtype, f32 outputs, matching sizes")
        if not (a.is_contiguous() and b.is_contiguous() and o.is_contiguous()):
            raise ValueError("sub_ragged: contiguous tensors expected")
        d.a, d.b, d.out, d.n, d.begin = a.data_ptr(), b.data_ptr(), o.data_ptr(), a.numel(), blocks
        blocks += (a.numel() + 4095) // 4096
    dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(outs[0].device)
    _check(lib.lora_amd_sub_ragged(dev.data_ptr(), len(pairs), blocks, dtype_code(dt), _stream()), "lora_amd_sub_ragged")
    # `dev` goes back to the caching allocator on return; the pool hands it out again only to work that is ordered after
    # this launch on the same stream


# ----------------------------------------------------------------------------- optimiser (C2/K6)
def sumsq(g: torch.Tensor, out: torch.Tensor, ws: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = require()
    _dev_check(g, out)
    if g.dtype != torch.float32 or not g.is_contiguous() or out.dtype != torch.float32:
        raise ValueError("sumsq: contiguous f32 tensors expected")
    nbytes = lib.lora_amd_sumsq_workspace(g.numel())
    if ws is None:
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=g.device)
    _check(lib.lora_amd_sumsq(g.data_ptr(), g.numel(), out.data_ptr(), ws.data_ptr(), ws.numel() * 4, _stream()),
           "lora_amd_sumsq")
    return out


def make_adamw_groups(groups: Sequence[Tuple[int, int, float, float]], device) -> torch.Tensor:
    arr = (AdamWGroup * len(groups))()
    for i, (b, e, lr, wd) in enumerate(groups):
        arr[i].begin, arr[i].end, arr[i].lr, arr[i].weight_decay = int(b), int(e), float(lr), float(wd)
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)


def clip_adamw(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, groups_dev: torch.Tensor,
               n_groups: int, sumsq_t: Optional[torch.Tensor], grad_scale: float, max_norm: float, beta1: float,
               beta2: float, eps: float, step, zero_grad: bool = True, scaler: Optional[torch.Tensor] = None) -> None:
    """``step``: python int (1-based), or a device int64 tensor (hipGraph-replayable; advance it with
    :func:`step_advance` or, with loss scaling, :func:`loss_scale_update`).  ``scaler``: the 4-float loss-scaling
    state (device-step form only)."""
    lib = require()
    _dev_check(p, g, m, v, groups_dev, sumsq_t)
    for t in (p, g, m, v):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != p.numel():
            raise ValueError("clip_adamw: flat contiguous f32 buffers of equal length expected")
    if torch.is_tensor(step):
        if step.dtype != torch.int64 or not step.is_cuda:
            raise ValueError("clip_adamw: device step must be an int64 device tensor")
        _check(lib.lora_amd_clip_adamw_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
                                           groups_dev.data_ptr(), int(n_groups),
                                           sumsq_t.data_ptr() if sumsq_t is not None else None, float(grad_scale),
                                           float(max_norm), float(beta1), float(beta2), float(eps), step.data_ptr(),
                                           _ptr(scaler), int(bool(zero_grad)), _stream()), "lora_amd_clip_adamw_dev")
        return
    if scaler is not None:
        raise ValueError("clip_adamw: loss scaling needs the device step counter")
    _check(lib.lora_amd_clip_adamw(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
                                   groups_dev.data_ptr(), int(n_groups),
                                   sumsq_t.data_ptr() if sumsq_t is not None else None, float(grad_scale),
                                   float(max_norm), float(beta1), float(beta2), float(eps), int(step),
                                   int(bool(zero_grad)), _stream()), "lora_amd_clip_adamw")


def loss_scale_update(state: torch.Tensor, sumsq_t: torch.Tensor, step_dev: Optional[torch.Tensor],
                      growth_factor: float = 2.0, backoff_factor: float = 0.5, growth_interval: int = 2000) -> None:
    """GradScaler.update() + the finite check of GradScaler.step(), on the device (see include/lora_amd.h)."""
    _dev_check(state, sumsq_t, step_dev)
    if state.dtype != torch.float32 or state.numel() != 4 or not state.is_contiguous():
        raise ValueError("loss_scale_update: state must be 4 contiguous f32 values")
    _check(require().lora_amd_loss_scale_update(state.data_ptr(), sumsq_t.data_ptr(), _ptr(step_dev),
                                                float(growth_factor), float(backoff_factor), int(growth_interval),
                                                _stream()), "lora_amd_loss_scale_update")


def step_advance(step_dev: torch.Tensor) -> None:
    _check(require().lora_amd_step_advance(step_dev.data_ptr(), _stream()), "lora_amd_step_advance")


# ----------------------------------------------------------------------------- fused K1/K2
_plan_cache = {}


def linear_plan(M: int, K: int, N: int, r: int) -> LinearPlan:
    key = (M, K, N, r)
    pl = _plan_cache.get(key)
    if pl is None:
        pl = LinearPlan()
        _check(require().lora_amd_linear_plan(M, K, N, r, C.byref(pl)), "lora_amd_linear_plan")
        _plan_cache[key] = pl
    return pl


def _rows_ok(t: torch.Tensor) -> bool:
    return (t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 8 == 0 and t.shape[1] % 8 == 0
            and t.data_ptr() % (32 if t.dtype == torch.float32 else 16) == 0)


def fused_ok(x: torch.Tensor, y_cols: int, r: int) -> bool:
    return r <= 16 and _rows_ok(x) and y_cols % 8 == 0 and bool(linear_plan(x.shape[0], x.shape[1], y_cols, r).fused)


def linear_fwd_(x: torch.Tensor, y: torch.Tensor, down: torch.Tensor, up: torch.Tensor, scale: float,
                sel: Optional[torch.Tensor], dropout_p: float, seed: int, offset: int) -> torch.Tensor:
    """y += scale*mask*((x @ down^T) @ S^T) @ up^T in place; returns T [M, r] f32."""
    lib = require()
    M, K = x.shape
    N = y.shape[1]
    r = down.shape[0]
    t = torch.empty((M, r), dtype=torch.float32, device=x.device)
    if sel is not None:
        sel = sel.to(torch.float32).contiguous()
    _check(lib.lora_amd_linear_fwd(x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), down.data_ptr(),
                                   up.data_ptr(), t.data_ptr(), M, K, N, r, dtype_code(x.dtype),
                                   dtype_code(down.dtype), float(scale), sel.data_ptr() if sel is not None else None,
                                   float(dropout_p), int(seed), *_off(offset), _stream()), "lora_amd_linear_fwd")
    return t


def linear_bwd_g(g: torch.Tensor, t: torch.Tensor, up: torch.Tensor, gt_part: Optional[torch.Tensor],
                 up_part: torch.Tensor, scale: float, dropout_p: float, seed: int, offset: int) -> None:
    """``gt_part`` None: dUp partials only."""
    M, N = g.shape
    _check(require().lora_amd_linear_bwd_g(g.data_ptr(), g.stride(0), t.data_ptr(), up.data_ptr(), _ptr(gt_part),
                                           up_part.data_ptr(), M, N, t.shape[1], dtype_code(g.dtype),
                                           dtype_code(up.dtype), float(scale), float(dropout_p), int(seed),
                                           *_off(offset), _stream()), "lora_amd_linear_bwd_g")


def linear_bwd_x(x: torch.Tensor, dx: Optional[torch.Tensor], gt_part: torch.Tensor, nct_g: int, down: torch.Tensor,
                 sel: Optional[torch.Tensor], down_part: torch.Tensor) -> None:
    M, K = x.shape
    if sel is not None:
        sel = sel.to(torch.float32).contiguous()
    _check(require().lora_amd_linear_bwd_x(x.data_ptr(), x.stride(0), dx.data_ptr() if dx is not None else None,
                                           dx.stride(0) if dx is not None else 0, gt_part.data_ptr(), int(nct_g),
                                           down.data_ptr(), sel.data_ptr() if sel is not None else None,
                                           down_part.data_ptr(), M, K, down.shape[0], dtype_code(x.dtype),
                                           dtype_code(down.dtype), _stream()), "lora_amd_linear_bwd_x")


def linear_bwd_factors(g: torch.Tensor, t: torch.Tensor, up_part: torch.Tensor, x: torch.Tensor, gt: torch.Tensor,
                       down_part: torch.Tensor, r: int, scale: float, sel: Optional[torch.Tensor] = None,
                       g_heads=None, x_heads=None, dropout=None) -> None:
    """dUp and dDown partials of a site in ONE launch (Gt already known: the fused MFMA backward produced it).
    ``g_heads`` / ``x_heads`` = (heads, d, D) when G / X rows are head-padded (logical N / K = heads * d);
    ``dropout`` = (p, seed, off) of the forward when the branch had nn.Dropout (dense rows only)."""
    _dev_check(g, t, up_part, x, gt, down_part, sel)
    if dropout is not None and dropout[0] > 0.0:
        if g_heads or x_heads:
            raise ValueError("linear_bwd_factors: dropout with head-padded rows is not supported")
        _check(require().lora_amd_linear_bwd_factors_drop(g.data_ptr(), g.stride(0), t.data_ptr(), up_part.data_ptr(),
                                                          x.data_ptr(), x.stride(0), gt.data_ptr(), _ptr(sel),
                                                          down_part.data_ptr(), g.shape[0], x.shape[1], g.shape[1], r,
                                                          dtype_code(g.dtype), float(scale), float(dropout[0]),
                                                          int(dropout[1]), *_off(dropout[2]), _stream()),
               "lora_amd_linear_bwd_factors_drop")
        return
    N = g_heads[0] * g_heads[1] if g_heads else g.shape[1]
    K = x_heads[0] * x_heads[1] if x_heads else x.shape[1]
    gd, gD = (g_heads[1], g_heads[2]) if g_heads else (0, 0)
    xd, xD = (x_heads[1], x_heads[2]) if x_heads else (0, 0)
    _check(require().lora_amd_linear_bwd_factors_heads(g.data_ptr(), g.stride(0), t.data_ptr(), up_part.data_ptr(),
                                                       x.data_ptr(), x.stride(0), gt.data_ptr(), _ptr(sel),
                                                       down_part.data_ptr(), g.shape[0], K, N, r,
                                                       dtype_code(g.dtype), float(scale), gd, gD, xd, xD, _stream()),
           "lora_amd_linear_bwd_factors")


_self_plan_cache = {}


def factors_self_plan(M: int, K: int, N: int, r: int, rows: int = 0) -> FactorsSelfPlan:
    """``rows`` > 0: rows per block chosen by the caller (the deferred one-launch pass: SELF_ROWS_DEFERRED)."""
    key = (M, K, N, r, rows)
    pl = _self_plan_cache.get(key)
    if pl is None:
        pl = FactorsSelfPlan()
        _check(require().lora_amd_linear_factors_self_plan_rows(M, K, N, r, rows, C.byref(pl)),
               "lora_amd_linear_factors_self_plan")
        _self_plan_cache[key] = pl
    return pl


# rows per block of the deferred one-launch factor-gradient pass: throughput-bound, so tall blocks (half the partial
# slabs of the per-site launch at M = 16384; 64 / 96 / 128 rows measured 1179 / 1141 / 1155 us for the pass and
# 86 / 63 / 49 us for the fold that follows it)
SELF_ROWS_DEFERRED = 128


def linear_bwd_factors_self(g: torch.Tensor, x: torch.Tensor, down: torch.Tensor, up: torch.Tensor,
                            up_part: torch.Tensor, down_part: torch.Tensor, scale: float, g_heads=None,
                            x_heads=None) -> None:
    """dUp and dDown partials of a site of the merged-weight path in ONE launch that needs neither T nor Gt:
    ``up_part[rb] = (s X down^T)^T G``, ``down_part[rb] = (s G up)^T X`` per row block (see include/lora_amd.h).
    ``g_heads`` / ``x_heads`` = (heads, d, D) when the rows of G / X are head-padded."""
    _dev_check(g, x, down, up, up_part, down_part)
    if down.dtype != torch.float32 or up.dtype != torch.float32 or not down.is_contiguous() or not up.is_contiguous():
        raise ValueError("linear_bwd_factors_self: contiguous f32 factors expected")
    N = g_heads[0] * g_heads[1] if g_heads else g.shape[1]
    K = x_heads[0] * x_heads[1] if x_heads else x.shape[1]
    r = down.shape[0]
    gd, gD = (g_heads[1], g_heads[2]) if g_heads else (0, 0)
    xd, xD = (x_heads[1], x_heads[2]) if x_heads else (0, 0)
    _check(require().lora_amd_linear_bwd_factors_self(g.data_ptr(), g.stride(0), x.data_ptr(), x.stride(0),
                                                      down.data_ptr(), up.data_ptr(), up_part.data_ptr(),
                                                      down_part.data_ptr(), g.shape[0], K, N, r, dtype_code(g.dtype),
                                                      float(scale), gd, gD, xd, xD, _stream()),
           "lora_amd_linear_bwd_factors_self")


def factors_self_ragged_table(sites, act_dtype: torch.dtype):
    """Host half of the one-launch factor-gradient pass: ``sites`` = [(g, x, down, up, up_part, down_part, scale,
    g_heads, x_heads)] (one activation dtype, one rank tile) -> (planned ctypes table, grid)."""
    lib = require()
    arr = (SelfSite * len(sites))()
    for q, (g, x, down, up, up_part, down_part, scale, g_heads, x_heads) in zip(arr, sites):
        q.rows_per_block = SELF_ROWS_DEFERRED
        q.g, q.x, q.down, q.up = g.data_ptr(), x.data_ptr(), down.data_ptr(), up.data_ptr()
        q.up_part, q.down_part = up_part.data_ptr(), down_part.data_ptr()
        q.ldg, q.ldx, q.M = g.stride(0), x.stride(0), g.shape[0]
        q.N = g_heads[0] * g_heads[1] if g_heads else g.shape[1]
        q.K = x_heads[0] * x_heads[1] if x_heads else x.shape[1]
        q.r, q.scale = down.shape[0], float(scale)
        q.g_head_dim, q.g_head_pad = (g_heads[1], g_heads[2]) if g_heads else (0, 0)
        q.x_head_dim, q.x_head_pad = (x_heads[1], x_heads[2]) if x_heads else (0, 0)
    grid = C.c_int64(0)
    _check(lib.lora_amd_linear_factors_self_ragged_plan(arr, len(sites), dtype_code(act_dtype), C.byref(grid)),
           "lora_amd_linear_factors_self_ragged_plan")
    return arr, grid.value


def linear_bwd_factors_self_ragged(table_dev: torch.Tensor, n: int, grid: int, rank: int, act_dtype: torch.dtype) -> None:
    _check(require().lora_amd_linear_bwd_factors_self_ragged(table_dev.data_ptr(), n, grid, rank, dtype_code(act_dtype),
                                                             _stream()), "lora_amd_linear_bwd_factors_self_ragged")


# ----------------------------------------------------------------------------- the matrix-core factor pass
# Which deferred sites take the matrix-core pass (csrc/factor_mfma.hip): "all" (default, also "1") = every 16-bit site —
# with the register-resident kernel the pass reads G and X once at 0.43-0.45 of the byte roof in the step (600-650 + 59 us fold + 11 us pack
# on the headline step's 144 sites against 938 + 28 us for the VALU pass, same call: profiles/r04_kbench_fm_register_form*.log);
# "masked" = the dropout sites only (their mask is regenerated inside the pass), the maskless ones keep the VALU pass;
# "0" = none
FACTORS_MFMA_MODE = {"1": "all", "0": "none"}.get(os.environ.get("LORA_AMD_FACTORS_MFMA", "all"),
                                                  os.environ.get("LORA_AMD_FACTORS_MFMA", "all"))
FACTORS_MFMA = FACTORS_MFMA_MODE != "none"
_mfma_plan_cache = {}


def factors_mfma_plan(M: int, K: int, N: int, r: int, act_dtype: torch.dtype, rows: int = 0) -> FactorsMfmaPlan:
    """Geometry of a site in the matrix-core factor pass (csrc/factor_mfma.hip): ``supported`` = 0 for f32 activations,
    N / K not multiples of 32, rank > 16 or a row block that does not fit the LDS."""
    key = (M, K, N, r, act_dtype, rows)
    pl = _mfma_plan_cache.get(key)
    if pl is None:
        pl = FactorsMfmaPlan()
        if act_dtype in (torch.float16, torch.bfloat16):
            _check(require().lora_amd_factors_mfma_plan(M, K, N, r, dtype_code(act_dtype), rows, 0, C.byref(pl)),
                   "lora_amd_factors_mfma_plan")
        _mfma_plan_cache[key] = pl
    return pl


def factor_pack_table(sites):
    """``sites`` = [(down f32 [r, K], up f32 [N, r], pk_down, pk_up)] -> (planned ctypes table, total work items)."""
    arr = (PackSite * len(sites))()
    for q, (down, up, pk_down, pk_up) in zip(arr, sites):
        if down.dtype != torch.float32 or up.dtype != torch.float32 or not down.is_contiguous() or not up.is_contiguous():
            raise ValueError("factor_pack: contiguous f32 factors expected")
        q.down, q.up, q.pk_down, q.pk_up = down.data_ptr(), up.data_ptr(), pk_down.data_ptr(), pk_up.data_ptr()
        q.N, q.K, q.r = up.shape[0], down.shape[1], down.shape[0]
    total = C.c_int64(0)
    _check(require().lora_amd_factor_pack_plan(arr, len(sites), C.byref(total)), "lora_amd_factor_pack_plan")
    return arr, total.value


def factor_pack(table_dev: torch.Tensor, n: int, total: int, act_dtype: torch.dtype) -> None:
    _check(require().lora_amd_factor_pack(table_dev.data_ptr(), n, total, dtype_code(act_dtype), _stream()),
           "lora_amd_factor_pack")


def factors_mfma_table(sites, act_dtype: torch.dtype, lds_class: int):
    """Host half of the matrix-core pass: ``sites`` = [(g, x, pk_down, pk_up, up_part, down_part, scale, g_heads,
    x_heads, r, plan[, (p, seed, offset)])] of one LDS class and rank tile (``plan`` = the site's ``factors_mfma_plan``;
    the optional last item = nn.Dropout on the branch: ``scale`` is then multiplied by 1 / (1 - p) here) -> (planned ctypes
    table, grid)."""
    arr = (FmSite * len(sites))()
    for q, site in zip(arr, sites):
        g, x, pk_down, pk_up, up_part, down_part, scale, g_heads, x_heads, r, plan = site[:11]
        drop = site[11] if len(site) > 11 else None
        if drop is not None and drop[0] > 0.0:
            p, seed, off = drop
            q.dropout_p, q.seed = float(p), int(seed)
            scale = float(scale) / (1.0 - float(p))
            if torch.is_tensor(off):
                q.offset, q.offset_dev = 0, off.data_ptr()
            else:
                q.offset, q.offset_dev = int(off), None
        q.g, q.x, q.pk_down, q.pk_up = g.data_ptr(), x.data_ptr(), pk_down.data_ptr(), pk_up.data_ptr()
        q.up_part, q.down_part = up_part.data_ptr(), down_part.data_ptr()
        q.ldg, q.ldx, q.M = g.stride(0), x.stride(0), g.shape[0]
        q.N = g_heads[0] * g_heads[1] if g_heads else g.shape[1]
        q.K = x_heads[0] * x_heads[1] if x_heads else x.shape[1]
        q.r, q.scale = int(r), float(scale)
        q.rows_per_block, q.blocks_per_wg = int(plan.rows_per_block), int(plan.blocks_per_wg)
        q.g_head_dim, q.g_head_pad = (g_heads[1], g_heads[2]) if g_heads else (0, 0)
        q.x_head_dim, q.x_head_pad = (x_heads[1], x_heads[2]) if x_heads else (0, 0)
    grid = C.c_int64(0)
    _check(require().lora_amd_factors_mfma_ragged_plan(arr, len(sites), dtype_code(act_dtype), lds_class, C.byref(grid)),
           "lora_amd_factors_mfma_ragged_plan")
    return arr, grid.value


def factors_mfma_block_map(arr, grid: int):
    """The block -> site map of a planned table (``factors_mfma_table``'s ``arr``): a ctypes int32 array of ``grid`` entries
    for ``linear_bwd_factors_mfma_ragged(..., block_map=)`` (uploaded behind the table by ``factors_mfma_table_bytes``)."""
    m = (C.c_int32 * int(grid))()
    _check(require().lora_amd_factors_mfma_block_map(arr, len(arr), int(grid), m), "lora_amd_factors_mfma_block_map")
    return m


def factors_mfma_table_bytes(arr, grid: int) -> Tuple[bytes, int]:
    """(table bytes + the block map behind them, byte offset of the map): one upload, one buffer."""
    raw = bytes(arr)
    pad = (-len(raw)) % 16
    return raw + b"\0" * pad + bytes(factors_mfma_block_map(arr, grid)), len(raw) + pad


def linear_bwd_factors_mfma_ragged(table_dev: torch.Tensor, n: int, grid: int, lds_class: int,
                                   act_dtype: torch.dtype, masked: bool = False, rows: int = 64, map_offset: int = 0) -> None:
    """``rows``: the block height (``plan.rows_per_block``) of EVERY site of the table (one per table, ABI 6).
    ``map_offset`` > 0: the table buffer holds the block -> site map at that byte offset (``factors_mfma_table_bytes``)."""
    if map_offset:
        _check(require().lora_amd_linear_bwd_factors_mfma_ragged_mapped(table_dev.data_ptr(), n, grid,
                                                                        table_dev.data_ptr() + int(map_offset), lds_class,
                                                                        int(rows), dtype_code(act_dtype), int(bool(masked)),
                                                                        _stream()),
               "lora_amd_linear_bwd_factors_mfma_ragged_mapped")
        return
    _check(require().lora_amd_linear_bwd_factors_mfma_ragged(table_dev.data_ptr(), n, grid, lds_class, int(rows),
                                                             dtype_code(act_dtype), int(bool(masked)), _stream()),
           "lora_amd_linear_bwd_factors_mfma_ragged")


def table_to_device(arr, device) -> torch.Tensor:
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)


def make_reduce_table(rows: Sequence[Tuple[torch.Tensor, torch.Tensor, int, int, int, int, int, float, float]],
                      device) -> Tuple[torch.Tensor, int, int]:
    """rows: (part, out, nparts, RT, C, r, layout, scale, beta) -> (device table, n, total work items)."""
    arr = (ReduceDesc * len(rows))()
    begin = 0
    for d, (part, out, nparts, RT, Cc, r, layout, scale, beta) in zip(arr, rows):
        d.part, d.out, d.begin = part.data_ptr(), out.data_ptr(), begin
        d.nparts, d.RT, d.C, d.r, d.layout, d.scale, d.beta = nparts, RT, Cc, r, layout, scale, beta
        begin += r * Cc
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device), len(rows), begin


def reduce_batched(table: torch.Tensor, n: int, total: int) -> None:
    _check(require().lora_amd_reduce_batched(table.data_ptr(), int(n), int(total), _stream()),
           "lora_amd_reduce_batched")


# ----------------------------------------------------------------------------- K4 conv adapter
_conv_plan_cache = {}


def conv_plan(B: int, C_in: int, C_out: int, H: int, W: int, ks: int, r: int) -> ConvPlan:
    key = (B, C_in, C_out, H, W, ks, r)
    pl = _conv_plan_cache.get(key)
    if pl is None:
        pl = ConvPlan()
        _check(require().lora_amd_conv_plan(B, C_in, C_out, H, W, ks, r, C.byref(pl)), "lora_amd_conv_plan")
        _conv_plan_cache[key] = pl
    return pl


def _ptr(t: Optional[torch.Tensor]):
    return t.data_ptr() if t is not None else None


def conv_down_fwd(x: torch.Tensor, down: torch.Tensor, sel: Optional[torch.Tensor], t_part: torch.Tensor,
                  t_out: torch.Tensor, ks: int) -> None:
    """t_out[B,r,H,W] (f32) = S . conv_kxk(x; down)."""
    B, Ci, H, W = x.shape
    _check(require().lora_amd_conv_down_fwd(x.data_ptr(), down.data_ptr(), _ptr(sel), t_part.data_ptr(),
                                            t_out.data_ptr(), B, Ci, H, W, ks, down.shape[0], dtype_code(x.dtype),
                                            dtype_code(down.dtype), _stream()), "lora_amd_conv_down_fwd")


def conv_up_fwd_(y: torch.Tensor, t: torch.Tensor, up: torch.Tensor, scale: float, dropout_p: float, seed: int,
                 offset: int) -> None:
    """y[B,Co,H,W] += scale * mask * conv_1x1(t; up) in place."""
    B, Co, H, W = y.shape
    _check(require().lora_amd_conv_up_fwd(y.data_ptr(), t.data_ptr(), up.data_ptr(), B, Co, H, W, t.shape[1],
                                          dtype_code(y.dtype), dtype_code(up.dtype), float(scale), float(dropout_p),
                                          int(seed), *_off(offset), _stream()), "lora_amd_conv_up_fwd")


def conv_bwd_g(g: torch.Tensor, t: torch.Tensor, up: torch.Tensor, sel: Optional[torch.Tensor], gt_part: torch.Tensor,
               gt_out: torch.Tensor, up_part: torch.Tensor, scale: float, dropout_p: float, seed: int,
               offset: int) -> None:
    B, Co, H, W = g.shape
    _check(require().lora_amd_conv_bwd_g(g.data_ptr(), t.data_ptr(), up.data_ptr(), _ptr(sel), gt_part.data_ptr(),
                                         gt_out.data_ptr(), up_part.data_ptr(), B, Co, H, W, t.shape[1],
                                         dtype_code(g.dtype), dtype_code(up.dtype), float(scale), float(dropout_p),
                                         int(seed), *_off(offset), _stream()), "lora_amd_conv_bwd_g")


def conv_bwd_x(x: torch.Tensor, dx: Optional[torch.Tensor], gt: torch.Tensor, down: torch.Tensor,
               down_part: torch.Tensor, ks: int) -> None:
    B, Ci, H, W = x.shape
    _check(require().lora_amd_conv_bwd_x(x.data_ptr(), _ptr(dx), gt.data_ptr(), down.data_ptr(),
                                         down_part.data_ptr(), B, Ci, H, W, ks, down.shape[0], dtype_code(x.dtype),
                                         dtype_code(down.dtype), _stream()), "lora_amd_conv_bwd_x")


# ----------------------------------------------------------------------------- K4, channels-last 3x3 (csrc/conv_nhwc.hip)
_conv3_nhwc_plan_cache = {}


def conv3_nhwc_plan(B: int, C_in: int, H: int, W: int, r: int) -> Conv3NhwcPlan:
    key = (B, C_in, H, W, r)
    pl = _conv3_nhwc_plan_cache.get(key)
    if pl is None:
        pl = Conv3NhwcPlan()
        _check(require().lora_amd_conv3_nhwc_plan(B, C_in, H, W, r, C.byref(pl)), "lora_amd_conv3_nhwc_plan")
        _conv3_nhwc_plan_cache[key] = pl
    return pl


def _nhwc_dims(x: torch.Tensor):
    """(B, C, H, W) of a logically-NCHW tensor whose memory is [B, H, W, C] contiguous."""
    B, Ci, H, W = x.shape
    if x.dim() != 4 or not x.is_contiguous(memory_format=torch.channels_last):  # (size-1 dims may carry any stride)
        raise ValueError("lora_amd: expected a channels_last-contiguous [B, C, H, W] tensor")
    return B, Ci, H, W


def conv3_nhwc_pack(down: torch.Tensor, act_dtype: torch.dtype, plan: Conv3NhwcPlan) -> Tuple[torch.Tensor, torch.Tensor]:
    """down [r, C_in, 3, 3] f32 -> (pf, pd): MFMA fragment order, activation dtype (forward / input-gradient operand)."""
    r, Ci = down.shape[0], down.shape[1]
    pf = torch.empty(int(plan.pf_elems), dtype=act_dtype, device=down.device)
    pd = torch.empty(int(plan.pd_elems), dtype=act_dtype, device=down.device)
    _check(require().lora_amd_conv3_nhwc_pack(down.data_ptr(), r, Ci, dtype_code(act_dtype), pf.data_ptr(),
                                              pd.data_ptr(), _stream()), "lora_amd_conv3_nhwc_pack")
    return pf, pd


def conv3_nhwc_down_fwd(x: torch.Tensor, pf: torch.Tensor, r: int, t_part: Optional[torch.Tensor] = None) -> torch.Tensor:
    """T [B*H*W, r] f32 = conv3x3(x; down) for a channels_last x (``t_part``: plan.t_part_floats floats of workspace,
    allocated here when the geometry needs it and none is given)."""
    B, Ci, H, W = _nhwc_dims(x)
    t = torch.empty((B * H * W, r), dtype=torch.float32, device=x.device)
    need = int(conv3_nhwc_plan(B, Ci, H, W, r).t_part_floats)
    if need and (t_part is None or t_part.numel() < need):
        t_part = torch.empty(need, dtype=torch.float32, device=x.device)
    _check(require().lora_amd_conv3_nhwc_down_fwd(x.data_ptr(), pf.data_ptr(), _ptr(t_part) if need else None,
                                                  t.data_ptr(), B, Ci, H, W, r, dtype_code(x.dtype), _stream()),
           "lora_amd_conv3_nhwc_down_fwd")
    return t


def conv3_nhwc_pack_table(sites):
    """``sites`` = [(down f32 [r, C_in, 3, 3], up f32 [C_out, r], pf, pd, pu)] -> (planned ctypes table, total pieces): the
    fragment packs of every conv site of a model in ONE launch (round 6; once per optimiser step)."""
    arr = (Conv3PackSite * len(sites))()
    for q, (down, up, pf, pd, pu) in zip(arr, sites):
        if down.dtype != torch.float32 or up.dtype != torch.float32 or not down.is_contiguous() or not up.is_contiguous():
            raise ValueError("conv3_nhwc_pack_table: contiguous f32 factors expected")
        q.down, q.up, q.pf, q.pd, q.pu = down.data_ptr(), up.data_ptr(), pf.data_ptr(), pd.data_ptr(), pu.data_ptr()
        q.r, q.C_in, q.C_out = down.shape[0], down.shape[1], up.shape[0]
    total = C.c_int64(0)
    _check(require().lora_amd_conv3_nhwc_pack_plan(arr, len(sites), C.byref(total)), "lora_amd_conv3_nhwc_pack_plan")
    return arr, total.value


def conv3_nhwc_pack_batched(table_dev: torch.Tensor, n: int, total: int, act_dtype: torch.dtype) -> None:
    _check(require().lora_amd_conv3_nhwc_pack_batched(table_dev.data_ptr(), n, total, dtype_code(act_dtype), _stream()),
           "lora_amd_conv3_nhwc_pack_batched")


def conv3_nhwc_fused_ok(x: torch.Tensor, C_out: int, r: int) -> bool:
    """Does ``lora_amd_conv3_nhwc_fwd_fused`` take this site?  (bf16 rows, C_out a multiple of 32, a native geometry.)"""
    B, Ci, H, W = x.shape
    return x.dtype == torch.bfloat16 and C_out % 32 == 0 and bool(conv3_nhwc_plan(B, Ci, H, W, r).native)


def conv3_nhwc_fwd_fused_(x: torch.Tensor, pf: torch.Tensor, pu: torch.Tensor, y: torch.Tensor, r: int, scale: float,
                          t_part: Optional[torch.Tensor], counters: Optional[torch.Tensor], dropout_p: float = 0.0,
                          seed: int = 0, offset=0) -> torch.Tensor:
    """y (channels_last, in place) += scale * mask o (conv3x3(x; down) up^T) in ONE launch; returns T [B*H*W, r] f32."""
    B, Ci, H, W = _nhwc_dims(x)
    Co = y.shape[1]
    _nhwc_dims(y)
    t = torch.empty((B * H * W, r), dtype=torch.float32, device=x.device)
    _check(require().lora_amd_conv3_nhwc_fwd_fused(x.data_ptr(), pf.data_ptr(), pu.data_ptr(), y.data_ptr(), t.data_ptr(),
                                                   _ptr(t_part), _ptr(counters), B, Ci, Co, H, W, r, dtype_code(x.dtype),
                                                   float(scale), float(dropout_p), int(seed), *_off(offset), _stream()),
           "lora_amd_conv3_nhwc_fwd_fused")
    return t


def linear_bwd_g_blocks(M: int, N: int, r: int) -> int:
    nb = C.c_int64(0)
    _check(require().lora_amd_linear_bwd_g_blocks(int(M), int(N), int(r), C.byref(nb)), "lora_amd_linear_bwd_g_blocks")
    return int(nb.value)


def linear_bwd_g_folded_ok(g: torch.Tensor, up: torch.Tensor, r: int) -> bool:
    return g.dtype == torch.bfloat16 and up.dtype == torch.float32 and 8 < r <= 16


def linear_bwd_g_folded(g: torch.Tensor, t: torch.Tensor, up: torch.Tensor, gt_part: torch.Tensor, gt_out: torch.Tensor,
                        counters: torch.Tensor, up_part: torch.Tensor, scale: float, dropout_p: float, seed: int,
                        offset) -> None:
    """:func:`linear_bwd_g` with the fold of the Gt column-tile partials inside the launch (``gt_out`` [M, r] complete)."""
    M, N = g.shape
    _check(require().lora_amd_linear_bwd_g_folded(g.data_ptr(), g.stride(0), t.data_ptr(), up.data_ptr(), gt_part.data_ptr(),
                                                  gt_out.data_ptr(), counters.data_ptr(), up_part.data_ptr(), M, N,
                                                  t.shape[1], dtype_code(g.dtype), dtype_code(up.dtype), float(scale),
                                                  float(dropout_p), int(seed), *_off(offset), _stream()),
           "lora_amd_linear_bwd_g_folded")


def conv3_nhwc_bwd_dx_(dx: torch.Tensor, gt: torch.Tensor, pd: torch.Tensor) -> torch.Tensor:
    """dx (channels_last, in place) += conv_transpose3x3(gt; down)."""
    B, Ci, H, W = _nhwc_dims(dx)
    _check(require().lora_amd_conv3_nhwc_bwd_dx(dx.data_ptr(), gt.data_ptr(), pd.data_ptr(), B, Ci, H, W, gt.shape[1],
                                                dtype_code(dx.dtype), _stream()), "lora_amd_conv3_nhwc_bwd_dx")
    return dx


def conv3_nhwc_bwd_down(x: torch.Tensor, gt: torch.Tensor, down_part: torch.Tensor) -> None:
    B, Ci, H, W = _nhwc_dims(x)
    _check(require().lora_amd_conv3_nhwc_bwd_down(x.data_ptr(), gt.data_ptr(), down_part.data_ptr(), B, Ci, H, W,
                                                  gt.shape[1], dtype_code(x.dtype), _stream()),
           "lora_amd_conv3_nhwc_bwd_down")


def sum_parts(part: torch.Tensor, nparts: int, n: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[n] = sum of the `nparts` consecutive length-n slices of `part` (f32)."""
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=part.device)
    _check(require().lora_amd_sum_parts(part.data_ptr(), int(nparts), int(n), out.data_ptr(), int(n), _stream()),
           "lora_amd_sum_parts")
    return out


# ----------------------------------------------------------------------------- K1 fully fused (MFMA GEMM + LoRA)
def gemm_supported(x: torch.Tensor, weight: torch.Tensor, y_cols: int, r: int) -> bool:
    return (x.dtype in (torch.bfloat16, torch.float16) and weight.dtype == x.dtype and weight.is_contiguous()
            and x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0
            and weight.data_ptr() % 16 == 0
            and bool(require().lora_amd_linear_gemm_supported(x.shape[0], x.shape[1], y_cols, r, dtype_code(x.dtype))))


Heads = Optional[Tuple[int, int, int]]  # (heads, d, D): `heads` runs of d elements, each stored padded to D


def heads_width(cols: int, lay: Heads) -> int:
    """Physical row length of a [*, cols] tensor stored with the head layout ``lay`` (None: dense)."""
    if lay is None:
        return cols
    h, d, D = lay
    if h * d != cols:
        raise ValueError(f"head layout {lay} does not describe {cols} columns")
    return h * D


def linear_gemm_fwd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], down: torch.Tensor,
                    up: torch.Tensor, scale: float, tile: int = 0, t_scale: float = 1.0, factor_layout: int = 0,
                    x_heads: Heads = None, y_heads: Heads = None):
    """(y [M,N], t [M,r] f32) = fused frozen GEMM + LoRA branch in one launch (see include/lora_amd.h).  With
    ``x_heads`` the rows of x are head-padded (logical K = weight.shape[1]); with ``y_heads`` y comes back head-padded
    ([M, heads*D], pad zeroed)."""
    M = x.shape[0]
    N, K = weight.shape
    if x.shape[1] != heads_width(K, x_heads):
        raise ValueError(f"linear_gemm_fwd: x has {x.shape[1]} columns, expected {heads_width(K, x_heads)}")
    r = down.shape[1] if factor_layout & 1 else down.shape[0]
    y = torch.empty((M, heads_width(N, y_heads)), dtype=x.dtype, device=x.device)
    t = torch.empty((M, r), dtype=torch.float32, device=x.device)
    xd, xD = (x_heads[1], x_heads[2]) if x_heads else (0, 0)
    yd, yD = (y_heads[1], y_heads[2]) if y_heads else (0, 0)
    _check(require().lora_amd_linear_gemm_fwd_heads(x.data_ptr(), x.stride(0), weight.data_ptr(), weight.stride(0),
                                                    _ptr(bias), y.data_ptr(), y.stride(0), down.data_ptr(),
                                                    up.data_ptr(), t.data_ptr(), M, K, N, r, dtype_code(x.dtype),
                                                    float(scale), float(t_scale), int(factor_layout), int(tile),
                                                    xd, xD, yd, yD, _stream()), "lora_amd_linear_gemm_fwd")
    return y, t


def linear_gemm_dx(g: torch.Tensor, weight_t: torch.Tensor, down: torch.Tensor, up: torch.Tensor, scale: float,
                   tile: int = 0, g_heads: Heads = None, dx_heads: Heads = None):
    """(dX [M,K], Gt [M,r] f32): dX = G W + scale (G up) down, Gt = scale G up, ONE launch of the same MFMA kernel on
    the resident transposed weight ``weight_t`` [K, N] (factors read in place: up [N,r] k-major, down [r,K]).
    ``g_heads``: G arrives head-padded (the site's output was); ``dx_heads``: dX is written head-padded (its input was)."""
    return linear_gemm_fwd(g, weight_t, None, up, down, scale, tile, t_scale=scale, factor_layout=3,
                           x_heads=g_heads, y_heads=dx_heads)


def heads_tile_ok(lay: Heads) -> bool:
    """Output head layouts the fused kernel can write: an output tile (160 or 320 columns) must own whole heads."""
    return lay is None or (lay[1] % 8 == 0 and lay[2] % 8 == 0 and 160 % lay[1] == 0)


GEMM_TILES = (22, 23, 24, 21)  # stages*10 + shape (see lora_amd_linear_gemm_fwd)


class _TuneCache(dict):
    """Per-shape kernel choices; with ``LORA_AMD_TUNE_CACHE=<file.json>`` they are loaded at start and saved when they
    change, so a second process (a profiled re-run, the next training job) does not time the candidates again."""

    def __init__(self, section: str):
        super().__init__()
        self.section, self.path = section, os.environ.get("LORA_AMD_TUNE_CACHE")
        if self.path and os.path.exists(self.path):
            try:
                import json

                for k, v in json.load(open(self.path)).get(section, {}).items():
                    dict.__setitem__(self, k, v)
            except (OSError, ValueError):
                pass

    def __setitem__(self, k, v):
        dict.__setitem__(self, k, v)
        if self.path:
            import json

            try:
                data = json.load(open(self.path)) if os.path.exists(self.path) else {}
            except (OSError, ValueError):
                data = {}
            data.setdefault(self.section, {})[k] = v
            tmp = f"{self.path}.{os.getpid()}.tmp"
            with open(tmp, "w") as f:
                json.dump(data, f)
            os.replace(tmp, self.path)


_gemm_choice = _TuneCache("gemm_fwd")


def _gpu_time(fn, inner: int = 5) -> float:
    """Device time of one ``fn()`` in ms: ``inner`` calls captured into a hipGraph and replayed between two events, so
    the host's launch cost (which would favour whichever candidate has fewer launches, but vanishes under the
    hipGraph-replayed training step) is not part of the comparison.  Falls back to eager timing if capture fails."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    try:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(inner):
                fn()
        graph.replay()
        best = float("inf")
        for _ in range(3):  # minimum over repeats: one stall must not decide the kernel of a whole training run
            torch.cuda.synchronize()
            a.record()
            graph.replay()
            graph.replay()
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / (2 * inner))
        return best
    except Exception:  # noqa: BLE001 - e.g. a library call that cannot be captured: time it eagerly instead
        torch.cuda.synchronize()
        a.record()
        for _ in range(inner):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / inner


WS_TILE = 50  # kernel code of the weight-stationary kernel next to the LDS-ring tiles (stages*10 + shape) and 0
AUTOTUNE = os.environ.get("LORA_AMD_AUTOTUNE", "0") == "1"


def static_fwd_choice(M: int, K: int, N: int, ws_ok: bool, ring_ok: bool) -> int:
    """Which kernel runs the forward of an (M, K, N) site: a fixed function of the shape (round 1 timed the candidates
    on every box, which made numerics — the fused kernels round T and scale*up to the activation dtype, the two-launch
    path does not — depend on a timing race).  Table from scripts/kbench.py --what ws on MI355X (profiles/r02_kbench_ws.log),
    us per launch WS / LDS-ring / library GEMM + linear_fwd:
        (16384,320,320) 13.5 / 14.6 / 25    (16384,320,2560) 61 / 68 / 81     (4096,640,640) 16.8 / 16.5 / 18.6
        (4096,640,5120) 78 / 52 / 65        (1024,1280,1280) 23.4 / 23.8 / 19.0   (1024,1280,10240) 129 / 46 / 49
        (256,1280,1280) 9.5 / 23 / 13.8     (308,768,320) 8.1 / 15.9 / 9.8     (308,768,1280) 8.3 / 16 / 12.4
    i.e. weight-stationary while a panel is amortised over few enough columns, the LDS ring for the wide GEGLU
    projections at K >= 640, the library GEMM + one fused launch for the square 1280 sites."""
    wide = N >= 4 * K
    if ws_ok:
        if K <= 320 or K == 768:
            return WS_TILE
        if K == 640 and not wide:
            return WS_TILE
        if K == 1280 and not wide and M <= 512:
            return WS_TILE
    if ring_ok and wide and K >= 640:
        return 24 if K == 640 else 0
    if ring_ok and K < 640 and M >= 4096:
        return 24 if wide else 22
    return 0


def static_bwd_choice(M: int, K: int, N: int, ws_ok: bool) -> int:
    """Backward of a site (dX = G W + ..., contraction over N): weight-stationary on W^T for N = 320 / 640 (13.8 / 22 us
    against library GEMM + two streaming passes at 14 + 16 / 12.7 + 16), the three-launch path elsewhere."""
    return WS_TILE if ws_ok and N in (320, 640) and M >= 2048 else 0


def gemm_choice(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], down: torch.Tensor,
                up: torch.Tensor, scale: float) -> int:
    """Which forward runs this (M, K, N, r, dtype) site: 0 = library GEMM + ``linear_fwd`` (two launches), WS_TILE =
    weight-stationary kernel, else the tile code of the LDS-ring MFMA kernel.  Default: :func:`static_fwd_choice`
    (deterministic).  ``LORA_AMD_GEMM=<code>`` pins one; ``LORA_AMD_AUTOTUNE=1`` times the candidates once per shape on
    the live tensors instead (never inside a hipGraph capture: there the answer is the cached one, or the static one)."""
    env = os.environ.get("LORA_AMD_GEMM")
    if env is not None:
        return int(env)
    M, K = x.shape
    N, r = weight.shape[0], down.shape[0]
    ws_ok = ws_supported(x, K, N, r) and weight.dtype == x.dtype
    ring_ok = gemm_supported(x, weight, N, r)
    if not AUTOTUNE:
        return static_fwd_choice(M, K, N, ws_ok, ring_ok)
    key = repr((M, K, N, r, str(x.dtype), bias is not None))
    c = _gemm_choice.get(key)
    if c is not None:
        return c
    if torch.cuda.is_current_stream_capturing():
        return static_fwd_choice(M, K, N, ws_ok, ring_ok)
    import torch.nn.functional as F

    def run(tile):
        if tile == 0:
            y = F.linear(x, weight, bias)
            linear_fwd_(x, y, down, up, scale, None, 0.0, 0, 0)
        elif tile == WS_TILE:
            linear_ws_fwd(x, weight, bias, down, up, scale)
        else:
            linear_gemm_fwd(x, weight, bias, down, up, scale, tile)

    best, best_t = 0, float("inf")
    fused0 = fused_ok(x, N, r)
    for tile in ((0,) if fused0 else ()) + (GEMM_TILES if ring_ok else ()) + ((WS_TILE,) if ws_ok else ()):
        t = _gpu_time(lambda: run(tile))
        if t < best_t:
            best, best_t = tile, t
    _gemm_choice[key] = best
    return best


_wt_cache = {}


def gemm_choice_cached(M: int, K: int, N: int, r: int, dtype: torch.dtype, has_bias: bool) -> Optional[int]:
    """The forward tile :func:`gemm_choice` settled on for this shape, or None if it has not been timed yet."""
    env = os.environ.get("LORA_AMD_GEMM")
    if env is not None:
        return int(env)
    c = _gemm_choice.get(repr((M, K, N, r, str(dtype), has_bias)))
    if c is None and not AUTOTUNE:  # deterministic policy: the head-padded layouts live in the LDS-ring kernel only
        c = static_fwd_choice(M, K, N, False, K % 64 == 0 and N % 8 == 0 and r <= 16)
    return None if c == WS_TILE else c


def gemm_choice_bwd_cached(M: int, K: int, N: int, r: int, dtype: torch.dtype) -> Optional[int]:
    env = os.environ.get("LORA_AMD_GEMM_BWD", os.environ.get("LORA_AMD_GEMM"))
    if env is not None:
        return int(env)
    c = _gemm_choice_bwd.get(repr((M, K, N, r, str(dtype))))
    if c is None and not AUTOTUNE:
        c = 22 if (N in (320, 640) and M >= 2048) else 0  # the shapes static_bwd_choice fuses, on the LDS-ring kernel
    return None if c == WS_TILE else c


def _tensor_version(t: torch.Tensor) -> int:
    """``t._version``, or 0 for inference tensors (created under ``torch.inference_mode()``: they do not track one and
    reading it raises; such a tensor cannot be modified in place outside inference mode either)."""
    try:
        return t._version
    except RuntimeError:
        return 0


def weight_t(weight: torch.Tensor) -> torch.Tensor:
    """Resident transposed copy [K, N] of a frozen [N, K] weight (the fused dX kernel contracts over N and wants it
    contiguous); built once per weight — frozen weights do not change during training, 288 GB of HBM make the second
    layout of the adapted sites (385 MB for the SD1.5 UNet) a non-issue.

    The entry keeps the SOURCE tensor alive: as long as it is cached its memory cannot be handed to another tensor by
    the caching allocator, so "same data_ptr / version / dtype / shape" really means "same weight" (an entry keyed on
    the address alone could return another site's transpose after a free + re-allocation).  In-place edits through
    ``.data`` do not bump ``_version``; code that rewrites a frozen weight in place must call
    :func:`invalidate_weight_caches` (``collapse_lora``, ``monkeypatch_*`` and ``Module._apply`` of the adapters do)."""
    key = (weight.data_ptr(), _tensor_version(weight), weight.dtype, tuple(weight.shape))
    hit = _wt_cache.get(key)
    if hit is None:
        if len(_wt_cache) >= 2048:
            _wt_cache.clear()
        src = weight.detach()
        hit = _wt_cache[key] = (src, src.t().contiguous())
    return hit[1]


def invalidate_weight_caches() -> None:
    """Drop every derived layout of frozen weights (transposes and fragment-order packs for the fused kernels).
    A captured hipGraph (``trainer.GraphedForwardBackward``) holds raw pointers into these buffers: re-capture after
    anything that calls this (``collapse_lora``, ``monkeypatch_*``, ``Module.to``) — frozen weights do not change while
    a training graph is alive, which is the only place graphs are used."""
    _wt_cache.clear()
    _ws_cache.clear()


# ----------------------------------------------------------------------------- K1/K2 weight-stationary (gemm_ws.hip)
_WS_K = (320, 640, 768, 1280)
_ws_cache = {}


def ws_supported(x: torch.Tensor, K: int, N: int, r: int) -> bool:
    """Can the weight-stationary kernel run a site of contraction length K, width N, rank r on input rows ``x``?"""
    return (K in _WS_K and x.dtype in (torch.bfloat16, torch.float16) and r <= 16 and N % 4 == 0 and x.dim() == 2
            and x.stride(1) == 1 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0)


def ws_pack(weight: torch.Tensor, transposed: bool = False) -> torch.Tensor:
    """Frozen weight [N, K] in MFMA fragment order (``lora_amd_ws_pack``), built once and kept resident like
    :func:`weight_t` (same keep-the-source-alive cache discipline).  ``transposed``: pack W^T — the operand of the
    input gradient dX = G W (contraction over N) — straight from the [N, K] storage."""
    key = (weight.data_ptr(), _tensor_version(weight), weight.dtype, tuple(weight.shape), tuple(weight.stride()),
           bool(transposed))
    hit = _ws_cache.get(key)
    if hit is None:
        if len(_ws_cache) >= 4096:
            _ws_cache.clear()
        lib = require()
        src = weight.detach()
        if src.dim() != 2 or src.dtype not in (torch.bfloat16, torch.float16):
            raise ValueError("ws_pack: 2-D bf16/f16 weight expected")
        n_out, k_c = (src.shape[1], src.shape[0]) if transposed else (src.shape[0], src.shape[1])
        sn, sk = (src.stride(1), src.stride(0)) if transposed else (src.stride(0), src.stride(1))
        elems = lib.lora_amd_ws_packed_elems(n_out, k_c)
        if elems == 0:
            raise ValueError(f"ws_pack: contraction length {k_c} has no weight-stationary kernel")
        out = torch.empty(elems, dtype=src.dtype, device=src.device)
        _check(lib.lora_amd_ws_pack(src.data_ptr(), sn, sk, n_out, k_c, dtype_code(src.dtype), out.data_ptr(), _stream()),
               "lora_amd_ws_pack")
        hit = _ws_cache[key] = (src, out)
    return hit[1]


def ws_heads_ok(K: int, N: int, x_heads: Heads = None, y_heads: Heads = None) -> bool:
    """Can ``linear_ws`` (dropout sites) read an input / write an output of contraction length ``K`` / width ``N`` in these head
    layouts: input heads in 16-byte chunks; output pad no wider than the head and at least half of it (one zero group per live
    group), whole panels, never both in one launch."""
    if x_heads and y_heads:
        return False
    if x_heads:
        h, d, D = x_heads
        if d % 8 or D % 8 or D < d or h * d != K:
            return False
    if y_heads:
        h, d, D = y_heads
        bn = C.c_int32(0)
        if not require().lora_amd_ws_config(int(K), C.byref(bn), None):
            return False
        if d % 4 or D % 4 or not (0 < D - d <= d <= 2 * (D - d)) or h * d != N or N % bn.value or D >= 65536:
            return False
    return True


def linear_ws(x: torch.Tensor, sites, row_groups: int = 0, x_heads: Heads = None):
    """One launch for every site in ``sites`` (all reading ``x`` [M, K]); each site is a dict with ``wp`` (packed
    weight), ``N``, ``down``, ``up``, ``scale`` and optionally ``bias``, ``y`` (output buffer, allocated if absent),
    ``want_t`` (default True), ``t_scale``, ``flayout``, and ``p`` / ``seed`` / ``off`` for nn.Dropout on the branch (every
    site of a launch or none; ``off`` an int or a 1-element int64 device tensor).  Returns [(y, t), ...].
    ``x_heads`` = (heads, d, D): the rows of x are head-padded (K = heads * d); a site's ``y_heads``: its output leaves
    head-padded, pad zeroed (dropout sites only: ``ws_heads_ok``)."""
    lib = require()
    M = x.shape[0]
    K = x_heads[0] * x_heads[1] if x_heads else x.shape[1]
    if x.shape[1] != heads_width(K, x_heads):
        raise ValueError(f"linear_ws: x has {x.shape[1]} columns, expected {heads_width(K, x_heads)}")
    if not 1 <= len(sites) <= WS_MAX_SITES:
        raise ValueError(f"linear_ws: 1..{WS_MAX_SITES} sites")
    arr = (WsSite * len(sites))()
    outs, keep = [], []
    for d, s in zip(arr, sites):
        N, fl = int(s["N"]), int(s.get("flayout", 0))
        down, up = s["down"], s["up"]
        r = down.shape[1] if fl & 1 else down.shape[0]
        if down.dtype != torch.float32 or up.dtype != torch.float32 or not down.is_contiguous() or not up.is_contiguous():
            raise ValueError("linear_ws: contiguous f32 factors expected")
        y = s.get("y")
        yh = s.get("y_heads")
        if y is None:
            y = torch.empty((M, heads_width(N, yh)), dtype=x.dtype, device=x.device)
        d.y_heads = (int(yh[1]) | (int(yh[2]) << 16)) if yh else 0
        t = torch.empty((M, r), dtype=torch.float32, device=x.device) if s.get("want_t", True) else None
        bias = s.get("bias")
        d.wp, d.bias, d.y, d.down, d.up, d.t_out = (s["wp"].data_ptr(), _ptr(bias), y.data_ptr(), down.data_ptr(),
                                                    up.data_ptr(), _ptr(t))
        d.ldy, d.N, d.r, d.panel_begin, d.flayout = y.stride(0), N, r, 0, fl
        d.scale, d.t_scale = float(s["scale"]), float(s.get("t_scale", 1.0))
        off_s, off_p = _off(s.get("off", 0))
        d.dropout_p, d.seed, d.offset, d.offset_dev = float(s.get("p", 0.0)), int(s.get("seed", 0)), off_s, off_p
        outs.append((y, t))
        keep.append((down, up, bias))
    xd, xD = (x_heads[1], x_heads[2]) if x_heads else (0, 0)
    _check(lib.lora_amd_linear_ws_heads(x.data_ptr(), x.stride(0), M, K, xd, xD, dtype_code(x.dtype), arr, len(sites),
                                        int(row_groups), _stream()), "lora_amd_linear_ws")
    return outs


def linear_ws_fwd(x, weight, bias, down, up, scale, row_groups: int = 0, p: float = 0.0, seed: int = 0, off=0,
                  x_heads: Heads = None, y_heads: Heads = None):
    """(y, t) of ONE site through the weight-stationary kernel (same contract as :func:`linear_gemm_fwd`);
    ``p`` > 0: nn.Dropout on the low-rank branch, mask indexed as :func:`linear_fwd_` does (logical columns, whatever
    the head layouts)."""
    return linear_ws(x, [dict(wp=ws_pack(weight), N=weight.shape[0], bias=bias, down=down, up=up, scale=scale, p=p,
                              seed=seed, off=off, y_heads=y_heads)], row_groups, x_heads)[0]


def linear_ws_dx(g, weight, down, up, scale, row_groups: int = 0, p: float = 0.0, seed: int = 0, off=0,
                 g_heads: Heads = None, dx_heads: Heads = None):
    """(dX [M,K], Gt [M,r] f32) = (G W + scale ((mask*G) up) down, scale (mask*G) up) of one site, weight-stationary on
    W^T (mask = the forward's dropout mask when ``p`` > 0, all ones otherwise).  ``g_heads``: G arrives head-padded (the
    site's output was); ``dx_heads``: dX leaves head-padded (its input was)."""
    return linear_ws(g, [dict(wp=ws_pack(weight, True), N=weight.shape[1], down=up, up=down, scale=scale,
                              t_scale=scale, flayout=3, p=p, seed=seed, off=off, y_heads=dx_heads)], row_groups, g_heads)[0]


_gemm_choice_bwd = _TuneCache("gemm_bwd")


def gemm_choice_bwd(g: torch.Tensor, x: torch.Tensor, weight: torch.Tensor, t: torch.Tensor, down: torch.Tensor,
                    up: torch.Tensor, scale: float, bufs) -> int:
    """As :func:`gemm_choice` for the backward of a site: 0 = G pass + library GEMM + X/dX pass, WS_TILE = fused dX/Gt
    on the weight-stationary kernel, else the tile of the LDS-ring kernel on the resident W^T (then: fused dX/Gt launch
    + ONE launch for the dUp / dDown partials)."""
    env = os.environ.get("LORA_AMD_GEMM_BWD", os.environ.get("LORA_AMD_GEMM"))
    if env is not None:
        return int(env)
    M, N = g.shape
    K, r = x.shape[1], down.shape[0]
    ws_ok = ws_supported(g, N, K, r) and weight.dtype == g.dtype and weight.is_contiguous()
    if not AUTOTUNE:
        return static_bwd_choice(M, K, N, ws_ok)
    key = repr((M, K, N, r, str(g.dtype)))
    c = _gemm_choice_bwd.get(key)
    if c is not None:
        return c
    if torch.cuda.is_current_stream_capturing():
        return static_bwd_choice(M, K, N, ws_ok)
    gt_part, up_part, down_part = bufs() if callable(bufs) else bufs
    plan = linear_plan(M, K, N, r)
    ring_ok = gemm_supported(g, weight_t(weight), K, r)

    def run(tile):
        if tile == 0:
            linear_bwd_g(g, t, up, gt_part, up_part, scale, 0.0, 0, 0)
            dx = g @ weight
            linear_bwd_x(x, dx, gt_part, plan.nct_g, down, None, down_part)
        else:
            if tile == WS_TILE:
                dx, gt = linear_ws_dx(g, weight, down, up, scale)
            else:
                dx, gt = linear_gemm_dx(g, weight_t(weight), down, up, scale, tile)
            linear_bwd_factors(g, t, up_part, x, gt, down_part, r, scale)

    best, best_t = 0, float("inf")
    for tile in (0,) + (GEMM_TILES if ring_ok else ()) + ((WS_TILE,) if ws_ok else ()):
        tt = _gpu_time(lambda: run(tile))
        if tt < best_t:
            best, best_t = tile, tt
    _gemm_choice_bwd[key] = best
    return best


def ti_rows_step(table: torch.Tensor, table_grad: torch.Tensor, ids: torch.Tensor, rows: torch.Tensor, m: torch.Tensor,
                 v: torch.Tensor, lr: float, step: int, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 grad_scale: float = 1.0, decay_lambda: float = -1.0, target_norm: float = 0.4) -> None:
    """Placeholder-row AdamW + norm decay + scatter in one launch (see include/lora_amd.h)."""
    _dev_check(table, table_grad, ids, rows, m, v)
    if not (table.is_contiguous() and table_grad.is_contiguous() and rows.is_contiguous()):
        raise ValueError("ti_rows_step: contiguous tensors expected")
    if ids.dtype != torch.int64 or rows.dtype != torch.float32 or table_grad.dtype != table.dtype:
        raise ValueError("ti_rows_step: ids int64, rows f32, grad in the table's dtype expected")
    _check(require().lora_amd_ti_rows_step(table.data_ptr(), table_grad.data_ptr(), ids.data_ptr(), ids.numel(),
                                           table.shape[1], dtype_code(table.dtype), rows.data_ptr(), m.data_ptr(),
                                           v.data_ptr(), float(lr), float(betas[0]), float(betas[1]), float(eps),
                                           float(weight_decay), float(grad_scale), int(step), float(decay_lambda),
                                           float(target_norm), _stream()), "lora_amd_ti_rows_step")


# ----------------------------------------------------------------------------- frozen host-model fusions (hostops.hip)
_gn_ws_cache: Dict[Tuple[int, int, int, int], int] = {}


def groupnorm_workspace(B: int, C_: int, HW: int, groups: int) -> int:
    """Bytes of f32 slice statistics the two GroupNorm launches exchange; 0 = geometry not supported."""
    key = (B, C_, HW, groups)
    n = _gn_ws_cache.get(key)
    if n is None:
        n = _gn_ws_cache[key] = int(require().lora_amd_groupnorm_workspace(B, C_, HW, groups))
    return n


def groupnorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float,
                  act: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """y = act(group_norm(x) * gamma + beta) for NCHW-contiguous x; returns (y, stats [B*groups, 2] = mean, rstd)."""
    _dev_check(x, gamma, beta)
    B, C_ = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C_)
    nbytes = groupnorm_workspace(B, C_, HW, groups)
    if nbytes == 0:
        raise ValueError(f"lora_amd_groupnorm: geometry {tuple(x.shape)} / {groups} groups not supported")
    y = torch.empty_like(x)
    stats = torch.empty(B * groups, 2, dtype=torch.float32, device=x.device)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    _check(require().lora_amd_groupnorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                            stats.data_ptr(), ws.data_ptr(), nbytes, B, C_, HW, groups, eps,
                                            1 if act else 0, dtype_code(x.dtype), _stream()), "lora_amd_groupnorm_fwd")
    return y, stats


def groupnorm_bwd(x: torch.Tensor, gout: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, stats: torch.Tensor,
                  groups: int, act: bool) -> torch.Tensor:
    _dev_check(x, gout, gamma, beta, stats)
    B, C_ = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C_)
    nbytes = groupnorm_workspace(B, C_, HW, groups)
    dx = torch.empty_like(x)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    _check(require().lora_amd_groupnorm_bwd(x.data_ptr(), gout.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                            stats.data_ptr(), dx.data_ptr(), ws.data_ptr(), nbytes, B, C_, HW, groups,
                                            1 if act else 0, dtype_code(x.dtype), _stream()), "lora_amd_groupnorm_bwd")
    return dx


def geglu_fwd(y: torch.Tensor) -> torch.Tensor:
    """out [..., inner] = y[..., :inner] * gelu(y[..., inner:]) for a row-contiguous y [..., 2*inner]."""
    _dev_check(y)
    y2 = _as2d(y.reshape(-1, y.shape[-1]))
    inner = y2.shape[1] // 2
    out = torch.empty(*y.shape[:-1], inner, dtype=y.dtype, device=y.device)
    _check(require().lora_amd_geglu_fwd(y2.data_ptr(), y2.stride(0), out.data_ptr(), inner, y2.shape[0], inner,
                                        dtype_code(y.dtype), _stream()), "lora_amd_geglu_fwd")
    return out


def geglu_bwd(y: torch.Tensor, gout: torch.Tensor) -> torch.Tensor:
    """Gradient w.r.t. y [..., 2*inner] given gout [..., inner] (one pass, both halves written in place of a cat)."""
    _dev_check(y, gout)
    y2, g2 = _as2d(y.reshape(-1, y.shape[-1])), _as2d(gout.reshape(-1, gout.shape[-1]))
    inner = y2.shape[1] // 2
    gy = torch.empty(y.shape, dtype=y.dtype, device=y.device)
    _check(require().lora_amd_geglu_bwd(y2.data_ptr(), y2.stride(0), g2.data_ptr(), g2.stride(0), gy.data_ptr(),
                                        2 * inner, y2.shape[0], inner, dtype_code(y.dtype), _stream()),
           "lora_amd_geglu_bwd")
    return gy


def layernorm_supported(K: int) -> bool:
    return bool(require().lora_amd_layernorm_supported(K))


def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """LayerNorm over the last dimension of a contiguous x; returns (y, stats [rows, 2] = mean, rstd)."""
    _dev_check(x, gamma, beta)
    K = x.shape[-1]
    M = x.numel() // K
    y = torch.empty_like(x)
    stats = torch.empty(M, 2, dtype=torch.float32, device=x.device)
    _check(require().lora_amd_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                            stats.data_ptr(), M, K, eps, dtype_code(x.dtype), _stream()),
           "lora_amd_layernorm_fwd")
    return y, stats


def layernorm_bwd(x: torch.Tensor, gout: torch.Tensor, gamma: torch.Tensor, stats: torch.Tensor) -> torch.Tensor:
    _dev_check(x, gout, gamma, stats)
    K = x.shape[-1]
    dx = torch.empty_like(x)
    _check(require().lora_amd_layernorm_bwd(x.data_ptr(), gout.data_ptr(), gamma.data_ptr(), stats.data_ptr(),
                                            dx.data_ptr(), x.numel() // K, K, dtype_code(x.dtype), _stream()),
           "lora_amd_layernorm_bwd")
    return dx


_gn_nhwc_ws_cache: Dict[Tuple[int, int, int, int], int] = {}


def groupnorm_nhwc_workspace(B: int, C_: int, HW: int, groups: int) -> int:
    key = (B, C_, HW, groups)
    n = _gn_nhwc_ws_cache.get(key)
    if n is None:
        n = _gn_nhwc_ws_cache[key] = int(require().lora_amd_groupnorm_nhwc_workspace(B, C_, HW, groups))
    return n


def groupnorm_nhwc_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float,
                       act: bool, addend: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """GroupNorm (+SiLU) of a channels_last x [B, C, H, W] (+ addend [B, C] f32 before the normalisation); returns
    (y channels_last, aff [B, 4, C] f32)."""
    _dev_check(x, gamma, beta, addend)
    if addend is not None and (addend.dtype != torch.float32 or not addend.is_contiguous()
                               or tuple(addend.shape) != (x.shape[0], x.shape[1])):
        raise ValueError("groupnorm_nhwc_fwd: addend must be a contiguous float32 [B, C] tensor")
    B, C_ = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C_)
    nbytes = groupnorm_nhwc_workspace(B, C_, HW, groups)
    if nbytes == 0:
        raise ValueError(f"lora_amd_groupnorm_nhwc: geometry {tuple(x.shape)} / {groups} groups not supported")
    y = torch.empty_like(x)  # preserves the channels_last strides
    aff = torch.empty(B, 4, C_, dtype=torch.float32, device=x.device)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    _check(require().lora_amd_groupnorm_nhwc_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(addend),
                                                 y.data_ptr(), aff.data_ptr(), ws.data_ptr(), nbytes, B, C_, HW, groups, eps,
                                                 1 if act else 0, dtype_code(x.dtype), _stream()),
           "lora_amd_groupnorm_nhwc_fwd")
    return y, aff


def groupnorm_nhwc_bwd(x: torch.Tensor, gout: torch.Tensor, gamma: torch.Tensor, aff: torch.Tensor, groups: int,
                       act: bool) -> torch.Tensor:
    _dev_check(x, gout, gamma, aff)
    B, C_ = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C_)
    nbytes = groupnorm_nhwc_workspace(B, C_, HW, groups)
    dx = torch.empty_like(x)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    _check(require().lora_amd_groupnorm_nhwc_bwd(x.data_ptr(), gout.data_ptr(), gamma.data_ptr(), aff.data_ptr(),
                                                 dx.data_ptr(), ws.data_ptr(), nbytes, B, C_, HW, groups,
                                                 1 if act else 0, dtype_code(x.dtype), _stream()),
           "lora_amd_groupnorm_nhwc_bwd")
    return dx


def add_layernorm_fwd(x: torch.Tensor, res: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                      eps: float) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(x + res, layernorm(x + res), stats) in one pass over contiguous x / res of the same shape."""
    _dev_check(x, res, gamma, beta)
    K = x.shape[-1]
    M = x.numel() // K
    s, y = torch.empty_like(x), torch.empty_like(x)
    stats = torch.empty(M, 2, dtype=torch.float32, device=x.device)
    _check(require().lora_amd_add_layernorm_fwd(x.data_ptr(), res.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                s.data_ptr(), y.data_ptr(), stats.data_ptr(), M, K, eps,
                                                dtype_code(x.dtype), _stream()), "lora_amd_add_layernorm_fwd")
    return s, y, stats


def add_layernorm_bwd(s: torch.Tensor, gout: torch.Tensor, gsum: Optional[torch.Tensor], gamma: torch.Tensor,
                      stats: torch.Tensor) -> torch.Tensor:
    """Gradient of both addends: layernorm backward at the saved sum + the gradient of the residual stream."""
    _dev_check(s, gout, gsum, gamma, stats)
    K = s.shape[-1]
    dx = torch.empty_like(s)
    _check(require().lora_amd_add_layernorm_bwd(s.data_ptr(), gout.data_ptr(), _ptr(gsum), gamma.data_ptr(),
                                                stats.data_ptr(), dx.data_ptr(), s.numel() // K, K,
                                                dtype_code(s.dtype), _stream()), "lora_amd_add_layernorm_bwd")
    return dx
