"""ctypes binding of libmdvit_hip.so (the C ABI declared in include/mdvit_hip.h).

There is NO fallback: if the shared library is missing or a call fails, the op raises.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import List

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MDVIT_HIP_LIB") or os.path.join(_HERE, "lib", "libmdvit_hip.so")      # env: a variant build (tuning experiments)
HEADER_PATH = os.path.join(_HERE, "..", "include", "mdvit_hip.h")

EPI_NONE, EPI_GELU_DUAL, EPI_DGELU = 0, 1, 2
ACT_NONE, ACT_HSWISH, ACT_RELU = 0, 1, 2

f32p = C.c_void_p
vp = C.c_void_p
i32 = C.c_int32
i64 = C.c_int64
u32 = C.c_uint32
f32 = C.c_float


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", vp), ("B", vp), ("C", vp), ("C2", vp),
        ("lda", i64), ("ldb", i64), ("ldc", i64),
        ("M", i32), ("N", i32), ("K", i32),
        ("trans_a", i32), ("trans_b", i32),
        ("bias", vp),
        ("epi", i32),
        ("e_drop_p", f32), ("e_key0", u32), ("e_key1", u32),
        ("e_rowscale", vp), ("e_rows_per_scale", i32),
        ("residual", vp), ("ldr", i64),
        ("gelu_u", vp), ("ldu", i64),
        ("allow_split", i32),
        ("ws", vp), ("ws_bytes", C.c_uint64),
        ("accumulate", i32),
        ("colsum_a", vp),
        ("precision", i32),
        ("drop_seed", vp),
        ("rc_a", vp), ("rc_lda", i64), ("rc_b", vp), ("rc_ldb", i64), ("rc_bias", vp), ("rc_k", i32),
        ("conv_c", i32), ("conv_h", i32), ("conv_w", i32), ("conv_ho", i32), ("conv_wo", i32), ("conv_stride", i32), ("conv_dilation", i32),
        ("conv_up", i32),
        ("a_bf16", i32), ("b_bf16", i32), ("conv_wgrad_nchw", i32),
    ]


class PlaneGemmDesc(C.Structure):
    _fields_ = [
        ("A", vp), ("lda", i64), ("a_plane", i64), ("a_f32", i32),
        ("B", vp), ("ldb", i64), ("b_plane", i64),
        ("planes", i32), ("trans", i32),
        ("M", i32), ("N", i32), ("K", i32),
        ("C", vp), ("ldc", i64),
        ("Cp", vp), ("ldcp", i64), ("c_plane", i64),
        ("U", vp), ("ldu_out", i64),
        ("bias", vp),
        ("epi", i32),
        ("e_drop_p", f32), ("e_key0", u32), ("e_key1", u32),
        ("e_rowscale", vp), ("e_rows_per_scale", i32),
        ("residual", vp), ("ldr", i64),
        ("gelu_u", vp), ("ldu", i64),
        ("rc_a", vp), ("rc_lda", i64), ("rc_a_plane", i64), ("rc_b", vp), ("rc_ldb", i64), ("rc_b_plane", i64), ("rc_bias", vp), ("rc_k", i32),
        ("allow_split", i32), ("ws", vp), ("ws_bytes", C.c_uint64),
        ("accumulate", i32),
        ("drop_seed", vp),
    ]


class BlockDesc(C.Structure):
    _fields_ = ([(n, i32) for n in ("B", "H", "W", "C", "heads", "hidden", "s3", "s5", "s7", "ln_groups", "D", "da_hidden", "precision")]
                + [("eps", f32), ("drop_p", f32), ("key_proj", u32 * 2), ("key_fc1", u32 * 2), ("key_fc2", u32 * 2), ("drop_seed", vp),
                   ("rowscale1", vp), ("rowscale2", vp), ("label", vp)]
                + [(n, vp) for n in ("cpe_w", "cpe_b", "n1_g", "n1_b", "qkv_w", "qkv_b", "w3", "b3", "w5", "b5", "w7", "b7", "da_w1", "da_b1", "da_w2", "da_b2",
                                     "proj_w", "proj_b", "n2_g", "n2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")]
                + [(n, vp) for n in ("qkv_wt", "proj_wt", "fc1_wt", "fc2_wt", "fc1_p", "fc2_p", "fc2t_p", "fc1t_p", "qkv_p", "proj_p", "projt_p", "qkvt_p")]
                + [("store_bf16", i32), ("a_pre", vp), ("attn_kind", i32)])


BLOCK_PARAMS = ("cpe_w", "cpe_b", "n1_g", "n1_b", "qkv_w", "qkv_b", "w3", "b3", "w5", "b5", "w7", "b7", "da_w1", "da_b1", "da_w2", "da_b2",
                "proj_w", "proj_b", "n2_g", "n2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")


DA_MANY_MAX = 32


class DaMany(C.Structure):
    _fields_ = [("n", i32), ("hid", i32 * DA_MANY_MAX), ("C", i32 * DA_MANY_MAX), ("heads", i32 * DA_MANY_MAX), ("W1", vp * DA_MANY_MAX), ("b1", vp * DA_MANY_MAX),
                ("W2", vp * DA_MANY_MAX), ("b2", vp * DA_MANY_MAX), ("a", vp * DA_MANY_MAX)]


class DaManyGrads(C.Structure):
    _fields_ = [("e", vp * DA_MANY_MAX), ("dW1", vp * DA_MANY_MAX), ("db1", vp * DA_MANY_MAX), ("dW2", vp * DA_MANY_MAX), ("db2", vp * DA_MANY_MAX)]


class BlockGrads(C.Structure):
    _fields_ = [(n, vp) for n in BLOCK_PARAMS] + [("accumulate", i32), ("dgrad_only", i32), ("aux_first", i32), ("ln_accumulate", i32), ("e_out", vp)]


class BlockStreams(C.Structure):
    _fields_ = [("main", vp), ("side", vp), ("events", C.POINTER(vp)), ("n_events", i32), ("next_event", C.POINTER(i32))]


_SIGS = {
    "mdvit_block_fwd": [C.POINTER(BlockDesc), vp, vp, vp, C.c_size_t, vp, C.c_size_t, vp],
    "mdvit_block_bwd": [C.POINTER(BlockDesc), C.POINTER(BlockGrads), C.POINTER(BlockStreams), vp, vp, C.c_size_t, vp, vp, vp, C.c_size_t, vp, C.c_size_t],
    "mdvit_gemm_planes": [C.POINTER(PlaneGemmDesc), vp],
    "mdvit_gemm_planes_plan": [C.POINTER(PlaneGemmDesc), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)],
    "mdvit_gemm_planes_force_plan": [i32, i32],
    "mdvit_gemm_ph_config": [i32],
    "mdvit_gemm_pm_config": [i32],
    "mdvit_da_fwd_many": [C.POINTER(DaMany), vp, i32, i32, vp],
    "mdvit_da_bwd_many": [C.POINTER(DaMany), C.POINTER(DaManyGrads), vp, f32, vp, C.c_size_t, i32, i32, vp],
    "mdvit_gemm_pm_prefers": [i32, i32, i32, i32, i32],
    "mdvit_gemm_pm_splits": [i32, i32, i32, i32, i32],
    "mdvit_gemm_f32_grouped": [C.POINTER(GemmDesc), i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp],
    "mdvit_gemm_f32_grouped_bias": [C.POINTER(GemmDesc), i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp],
    "mdvit_transpose_batch": [i32, C.POINTER(vp), C.POINTER(i64), C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), vp],
    "mdvit_compose_bias": [i32, C.POINTER(vp), i64, C.POINTER(vp), C.POINTER(vp), i32, i32, vp],
    "mdvit_compose_bias_bwd": [i32, C.POINTER(vp), i64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i64, C.POINTER(vp), i32, i32, i32, vp],
    "mdvit_gemm_ph_prefers": [i32, i32, i32, i32],
    "mdvit_gemm_ph_prefers_epi": [i32, i32, i32, i32, i32],
    "mdvit_mlp_config": [i32, i32],
    "mdvit_mlp_rc_config": [i32],
    "mdvit_mlp_rc_planes": [i32],
    "mdvit_block_config": [i32],
    "mdvit_gemm_ledger": [i32],
    "mdvit_gemm_ledger_read": [i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "mdvit_gemm_sampler": [C.c_char_p, i32],
    "mdvit_gemm_sampler_read": [i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(i64), C.POINTER(C.c_double)],
    "mdvit_factoratt_config": [i32, i32],
    "mdvit_mlp_rc_fwd": [vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, f32, u32, u32, u32, u32, vp, vp],
    "mdvit_linear_rc": [vp, i64, vp, i64, vp, vp, i64, i32, i32, i32, f32, u32, u32, vp, i32, vp, i64, vp, vp],
    "mdvit_linear_rc_ln": [vp, vp, vp, i32, f32, vp, vp, vp, vp, i64, vp, vp, i64, i32, i32, i32, vp],
    "mdvit_mlp_rc_fwd_ln": [vp, vp, vp, i32, f32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, f32, u32, u32, u32, u32, vp, vp],
    "mdvit_mlp_rc16_fwd": [vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, f32, u32, u32, u32, u32, vp, vp],
    "mdvit_mlp_rc_dgrad": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, u32, u32, vp, vp],
    "mdvit_mlp_rc16_dgrad": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, u32, u32, vp, vp],
    "mdvit_mlp_rc_fwd_ln_hbf16": [vp, vp, vp, i32, f32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, f32, u32, u32, u32, u32, vp, vp],
    "mdvit_mlp_rc16_fwd_hbf16": [vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, f32, u32, u32, u32, u32, vp, vp],
    "mdvit_mlp_rc16_dgrad_hbf16": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, u32, u32, vp, vp],
    "mdvit_mlp_rc_wgrad": [vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, i32, i32, i32, f32, u32, u32, vp, i32, vp],
    "mdvit_mlp_rc_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, i32, i32, i32, f32, u32, u32, vp, i32, vp],
    "mdvit_imgconv_fwd": [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "mdvit_imgconv_wgrad": [vp, vp, vp, vp, C.c_size_t, i32, i32, i32, i32, i32, i32, i32, vp],
    "mdvit_maxpool3x3s2_fwd": [vp, vp, vp, i32, i32, i32, i32, vp],
    "mdvit_maxpool3x3s2_bwd": [vp, vp, vp, i32, i32, i32, i32, vp],
    "mdvit_resize_ac_fwd": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "mdvit_resize_ac_bwd": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "mdvit_ew": [vp, vp, vp, i64, i32, vp],
    "mdvit_add3": [vp, vp, vp, vp, i64, vp],
    "mdvit_add_parts": [vp, vp, C.POINTER(vp), i32, i64, vp, vp],
    "mdvit_add_bcast": [vp, vp, vp, i32, i64, vp],
    "mdvit_sum_batch": [vp, vp, i32, i64, vp],
    "mdvit_gate_fwd": [vp, vp, vp, i32, i64, i32, i32, vp],
    "mdvit_gate_bwd": [vp, vp, vp, vp, vp, vp, C.c_size_t, i32, i64, i32, i32, vp],
    "mdvit_imgconv_im2col": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "mdvit_chanpool_fwd": [vp, vp, vp, i64, i32, vp],
    "mdvit_chanpool_bwd": [vp, vp, vp, i64, i32, vp],
    "mdvit_conv7x7_2to1_fwd": [vp, vp, vp, i32, i32, i32, vp],
    "mdvit_conv7x7_2to1_bwd": [vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "mdvit_bn1_fwd": [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, f32, f32, vp],
    "mdvit_bn1_bwd": [vp, vp, vp, vp, vp, vp, i64, i32, i32, vp],
    "mdvit_subsample2": [vp, vp, i32, i32, i32, i32, i32, vp],
    "mdvit_patchify": [vp, vp, i32, i32, i32, i32, i32, vp],
    "mdvit_dropout2d": [vp, vp, i32, i64, i32, f32, u32, u32, vp, vp],
    "mdvit_sdpa_fwd": [vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "mdvit_sdpa_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "mdvit_sdpa_mfma_fwd": [vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "mdvit_sdpa_mfma_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "mdvit_structure_weight": [vp, vp, vp, i32, i32, i32, vp],
    "mdvit_structure_loss_fwd": [vp, vp, vp, vp, vp, i32, i64, vp],
    "mdvit_structure_loss_bwd": [vp, vp, vp, vp, vp, vp, i32, i64, vp],
    "mdvit_split_planes": [vp, i64, vp, i64, i64, i64, i32, i32, vp],
    "mdvit_split_planes_many": [vp, i32, i32, i32, vp],
    "mdvit_split_planes_t": [vp, i64, vp, i64, i64, i32, i32, i32, i32, vp],
    "mdvit_event_create": [C.POINTER(vp)],
    "mdvit_event_destroy": [vp],
    "mdvit_event_elapsed_ms": [vp, vp, C.POINTER(f32)],
    "mdvit_timing_arm": [vp, vp],
    "mdvit_gemm_f32": [C.POINTER(GemmDesc), vp],
    "mdvit_gemm_plan": [C.POINTER(GemmDesc), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)],
    "mdvit_gemm_force_plan": [i32, i32],
    "mdvit_conv_weight_relayout": [vp, vp, i32, i32, i32, vp],
    "mdvit_conv_weight_relayout_many": [vp, i32, i32, vp],
    "mdvit_gemm_kernel_name": [vp, C.c_char_p, i32],
    "mdvit_gemm_tn_config": [i32, i32, i32],
    "mdvit_gemm_tn_grid_order": [i32],
    "mdvit_transpose_f32": [vp, i64, vp, i32, i32, vp],
    "mdvit_mlp_bwd_dgrad_f32": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, u32, u32, vp, vp],
    "mdvit_mlp_fwd_f32": [vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, f32, u32, u32, u32, u32, vp, vp],
    "mdvit_transpose_many": [vp, i32, i32, vp],
    "mdvit_rowdot_fwd": [vp, i64, vp, vp, vp, i32, i32, i32, vp],
    "mdvit_rowdot_bwd": [vp, i64, vp, vp, vp, i64, vp, vp, vp, C.c_size_t, i32, i32, vp],
    "mdvit_colsum_f32": [vp, i64, vp, vp, vp, C.c_size_t, i32, i32, f32, u32, u32, vp, i32, i32, vp, vp],
    "mdvit_layernorm_fwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp],
    "mdvit_layernorm_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, i32, i32, i32, vp],
    "mdvit_layernorm_bwd_masked": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, i32, i32, i32, f32, u32, u32, vp, i32, vp, vp],
    "mdvit_dwconv3x3_fwd": [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "mdvit_dwconv3x3_bwd": [vp, vp, vp, vp, vp, vp, vp, C.c_size_t, i32, i32, i32, i32, i32, i32, i32, vp],
    "mdvit_gconv2_3x3_fwd": [vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "mdvit_gconv2_3x3_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, i32, i32, i32, i32, i32, vp],
    "mdvit_im2col3x3": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "mdvit_col2im3x3": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "mdvit_dropout_f32": [vp, vp, i64, f32, u32, u32, vp, vp],
    "mdvit_stemconv_fwd": [vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "mdvit_stemconv_wgrad": [vp, vp, vp, vp, C.c_size_t, i32, i32, i32, i32, i32, i32, vp],
    "mdvit_bn_rowdot_fwd": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, u32, u32, vp, i32, vp],
    "mdvit_bn_rowdot_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, i32, i32, i32, i32, f32, u32, u32, vp, i32, vp],
    "mdvit_bn_stats": [vp, vp, C.c_size_t, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, vp],
    "mdvit_bn_eval_prep": [vp, vp, vp, vp, i32, f32, vp],
    "mdvit_bn_apply": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, u32, u32, vp, i32, vp],
    "mdvit_bn_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, i32, i32, i32, i32, i32, i32, f32, u32, u32, vp, i32, vp],
    "mdvit_upsample_multi_fwd": [vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, vp],
    "mdvit_upsample_multi_bwd": [vp, vp, vp, vp, i32, vp, C.c_size_t, i32, i32, i32, i32, vp],
    "mdvit_upsample_fwd": [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "mdvit_upsample_bwd": [vp, vp, vp, C.c_size_t, i32, i32, i32, i32, i32, i32, vp],
    "mdvit_da_fwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "mdvit_da_bwd": [vp] * 7 + [f32] + [vp] * 4 + [vp, C.c_size_t] + [i32] * 5 + [vp],
    "mdvit_factoratt_fwd": [vp] * 13 + [vp, C.c_size_t] + [i32] * 8 + [vp],
    "mdvit_factoratt_wgrad": [vp, vp, C.c_size_t] + [vp] * 6 + [i32] * 9 + [vp],
    "mdvit_factoratt_bwd": [vp] * 22 + [vp, C.c_size_t] + [i32] * 8 + [vp],
    "mdvit_seg_losses_fwd": [vp, vp, vp, vp, vp, i64, vp],
    "mdvit_adamw_step": [vp, i32, i32, vp, vp, f32, f32, f32, f32, i32, vp],
    "mdvit_seg_metrics": [vp, vp, vp, vp, vp, i64, vp],
    "mdvit_image_normalize_u8": [vp, vp, i32, i32, i32, vp],
    "mdvit_seg_losses_sums": [vp, vp, vp, vp, i64, vp],
    "mdvit_seg_losses_final": [vp, vp, i64, i32, vp],
    "mdvit_seg_losses_bwd": [vp, vp, vp, vp, vp, vp, vp, i64, f32, vp],
    "mdvit_seg_losses_bwd3": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, f32, vp],
    "mdvit_seg_losses_groups_sums": [vp, vp, vp, vp, i64, i32, vp],
    "mdvit_seg_losses_groups_final": [vp, vp, vp, i64, i32, i32, vp],
    "mdvit_seg_losses_groups_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, vp],
}

_lib = None


class MdvitHipError(RuntimeError):
    pass


def declared_symbols() -> List[str]:
    """Every function the public header declares (used by the symbol-export test)."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mdvit_[a-z0-9_]+)\s*\(", text)))


_on_load = []          # process-wide configuration calls queued by modules imported BEFORE the library exists (a fresh tree: `python -m mdvit_amd.build` imports the package)


def on_load(fn):
    """run fn(lib) now if the library is ALREADY loaded, else right after the first explicit load().  Importing mdvit_amd never dlopens anything (a
    stale git-ignored .so from an earlier tree made `import mdvit_amd` -- and therefore `python -m mdvit_amd.build`, the command that rebuilds it -- raise)"""
    if _lib is not None:
        fn(_lib)
    else:
        _on_load.append(fn)


def load():
    """dlopen the library (works without a GPU) and attach prototypes.  Raises if it is not built or was built from an earlier tree."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MdvitHipError(
            f"{LIB_PATH} is missing: build it with `python -m mdvit_amd.build` (hipcc, gfx950). "
            "mdvit_amd has no CPU or PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    try:
        _attach_prototypes(lib)
    except AttributeError as e:
        raise MdvitHipError(f"{LIB_PATH} is stale ({e}): rebuild it with `python -m mdvit_amd.build`.") from None
    _lib = lib
    while _on_load:
        _on_load.pop(0)(lib)
    return lib


def _attach_prototypes(lib):
    lib.mdvit_last_error.restype = C.c_char_p
    lib.mdvit_last_error.argtypes = []
    lib.mdvit_version.restype = C.c_int
    lib.mdvit_version.argtypes = []
    lib.mdvit_factoratt_ws_bytes.restype = C.c_size_t
    lib.mdvit_factoratt_ws_bytes.argtypes = [i32, i32, i32, i32]
    lib.mdvit_gemm_ws_bytes.restype = C.c_size_t
    lib.mdvit_gemm_ws_bytes.argtypes = [C.POINTER(GemmDesc)]
    lib.mdvit_gate_bwd_ws_bytes.restype = C.c_size_t
    lib.mdvit_gate_bwd_ws_bytes.argtypes = [i32, i64, i32, i32]
    lib.mdvit_gemm_planes_ws_bytes.restype = C.c_size_t
    lib.mdvit_gemm_planes_ws_bytes.argtypes = [C.POINTER(PlaneGemmDesc)]
    lib.mdvit_bn_rowdot_ws_bytes.restype = C.c_size_t
    lib.mdvit_bn_rowdot_ws_bytes.argtypes = [i32, i32]
    lib.mdvit_bn_ws_bytes.restype = C.c_size_t
    lib.mdvit_bn_ws_bytes.argtypes = [i32, i32, i32]
    lib.mdvit_upsample_multi_bwd_ws_bytes.restype = C.c_size_t
    lib.mdvit_upsample_multi_bwd_ws_bytes.argtypes = [vp, i32, i32, i32, i32]
    lib.mdvit_upsample_bwd_ws_bytes.restype = C.c_size_t
    lib.mdvit_upsample_bwd_ws_bytes.argtypes = [i32] * 6
    lib.mdvit_partials_ws_bytes.restype = C.c_size_t
    lib.mdvit_partials_ws_bytes.argtypes = [i32]
    lib.mdvit_mlp_rc_wgrad_ws_bytes.restype = C.c_size_t
    lib.mdvit_mlp_rc_wgrad_ws_bytes.argtypes = [i32, i32, i32]
    lib.mdvit_block_save_bytes.restype = C.c_size_t
    lib.mdvit_block_save_bytes.argtypes = [C.POINTER(BlockDesc)]
    lib.mdvit_block_fwd_ws_bytes.restype = C.c_size_t
    lib.mdvit_block_fwd_ws_bytes.argtypes = [C.POINTER(BlockDesc)]
    lib.mdvit_block_bwd_ws_bytes.restype = C.c_size_t
    lib.mdvit_block_bwd_ws_bytes.argtypes = [C.POINTER(BlockDesc), C.POINTER(BlockGrads), i32, C.POINTER(C.c_size_t)]
    lib.mdvit_da_ws_bytes.restype = C.c_size_t
    lib.mdvit_da_ws_bytes.argtypes = [i32, i32, i32]
    lib.mdvit_da_many_ws_bytes.restype = C.c_size_t
    lib.mdvit_da_many_ws_bytes.argtypes = [C.POINTER(DaMany), i32]
    for name, sig in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise MdvitHipError(f"{LIB_PATH} is stale: it does not export `{name}` (built from an earlier tree). "
                                "Rebuild it with `python -m mdvit_amd.build`.") from None
        fn.restype = C.c_int
        fn.argtypes = sig


def check(code: int, what: str):
    if code != 0:
        msg = load().mdvit_last_error().decode(errors="replace")
        raise MdvitHipError(f"{what} failed (code {code}): {msg}")


def call(name: str, *args):
    lib = load()
    check(getattr(lib, name)(*args), name)
