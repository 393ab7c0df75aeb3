"""ctypes binding of libacm_hip.so (C ABI declared in include/acm_hip.h).

There is no CPU or torch fallback behind this module: if the library cannot be
loaded, or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os
import threading

from . import build as _build

c_f32p = C.POINTER(C.c_float)
c_i32p = C.POINTER(C.c_int32)

ACM_OK = 0
STATUS_NAMES = {1: "ACM_EINVAL", 2: "ACM_ESHAPE", 3: "ACM_EHIP", 4: "ACM_EUNSUPPORTED", 5: "ACM_ENOMEM"}
ABI_VERSION = 28

# every symbol include/acm_hip.h declares
EXPORTED_SYMBOLS = (
    "acm_version", "acm_last_error", "acm_tuning_get", "acm_tuning_set", "acm_csr_create", "acm_csr_transpose", "acm_csr_slice_rows",
    "acm_csr_destroy", "acm_csr_info", "acm_csr_build_streams", "acm_csr_build_item_streams", "acm_spmm_workspace_bytes", "acm_gemm_workspace_bytes",
    "acm_gemm", "acm_gemm_blocks", "acm_gemm_drop", "acm_proj3", "acm_gemm_split", "acm_proj_fwd", "acm_proj_fwd_at", "acm_proj_bwd_workspace_bytes", "acm_proj_bwd", "acm_spmm", "acm_spmm_v", "acm_spmm_ex", "acm_cast_bf16", "acm_conv_fwd", "acm_conv_bwd_local_workspace_bytes",
    "acm_conv_bwd_local", "acm_conv_bwd_spmm", "acm_conv_agg_fwd", "acm_conv_agg_bwd_workspace_bytes",
    "acm_conv_agg_bwd", "acm_nll_loss_workspace_bytes", "acm_nll_loss", "acm_adam_step", "acm_dropout",
    "acm_reduce_flush", "acm_conv_fwd_tail_workspace_bytes", "acm_conv_fwd_tail", "acm_shard_plan",
    "acm_conv_acmii_fwd_workspace_bytes", "acm_conv_acmii_fwd", "acm_linear_fwd", "acm_bias_act", "acm_bias_act_bwd_workspace_bytes", "acm_bias_act_bwd",
    "acm_acmii_table_bytes", "acm_acmii_table", "acm_conv_acmii_v_fwd", "acm_conv_acmii_v_bwd_workspace_bytes", "acm_conv_acmii_v_bwd",
    "acm_linear_bwd_workspace_bytes", "acm_linear_bwd", "acm_linear_fwd_add", "acm_linear_bwd_recompute",
    "acm_small_step_workspace_bytes", "acm_small_step", "acm_conv_head_fwd", "acm_conv_aggw_fwd", "acm_conv_aggw_bwd_workspace_bytes", "acm_conv_aggw_bwd",
    "acm_eval_metrics_workspace_bytes", "acm_eval_metrics",
)


class Dropout(C.Structure):
    _fields_ = [("p", C.c_float), ("tag", C.c_int32), ("seed", C.c_uint64), ("step", C.c_void_p),
                ("row_offset", C.c_int64), ("step_offset", C.c_int64)]


class CsrInfo(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("n_cols", C.c_int64), ("nnz", C.c_int64),
                ("n_items", C.c_int64), ("n_long_rows", C.c_int64), ("n_partial_slots", C.c_int64),
                ("chunk", C.c_int32), ("max_degree", C.c_int32),
                ("indptr", C.c_void_p), ("indices", C.c_void_p), ("vals", C.c_void_p),
                ("src_pos", C.c_void_p),
                ("stream_steps", C.c_int64), ("stream_slices", C.c_int64),
                ("stream_waves", C.c_int32), ("stream_long_rows", C.c_int32),
                ("item_stream_waves", C.c_int32), ("reserved", C.c_int32), ("item_stream_batches", C.c_int64)]


class ConvFwd(C.Structure):
    _fields_ = [("f_out", C.c_int32), ("n_channels", C.c_int32), ("relu_after", C.c_int32),
                ("relu_mlp", C.c_int32), ("layernorm", C.c_int32), ("scale", C.c_float),
                ("row_offset", C.c_int64),
                ("g_low", C.c_void_p), ("ld_g_low", C.c_int64),
                ("g_high", C.c_void_p), ("ld_g_high", C.c_int64),
                ("g_struc", C.c_void_p), ("ld_g_struc", C.c_int64),
                ("s_high", C.c_void_p), ("ld_s_high", C.c_int64),
                ("s_mlp", C.c_void_p), ("ld_s_mlp", C.c_int64),
                ("s_struc", C.c_void_p), ("ld_s_struc", C.c_int64),
                ("deg", C.c_void_p),
                ("att_vec", C.c_void_p * 4), ("ln_weight", C.c_void_p * 4), ("ln_bias", C.c_void_p * 4),
                ("att_mix", C.c_void_p),
                ("out", C.c_void_p), ("ld_out", C.c_int64),
                ("pre", C.c_void_p), ("ld_pre", C.c_int64),
                ("att", C.c_void_p), ("post_scale", C.c_void_p), ("ld_post_scale", C.c_int64), ("post_relu", C.c_int32),
                ("gather_bf16", C.c_int32), ("row_scale", C.c_void_p), ("post_drop", Dropout)]


class ConvBwdLocal(C.Structure):
    _fields_ = [("f_out", C.c_int32), ("n_channels", C.c_int32), ("relu_after", C.c_int32),
                ("relu_mlp", C.c_int32), ("layernorm", C.c_int32), ("scale", C.c_float),
                ("grad_out", C.c_void_p), ("ld_grad_out", C.c_int64),
                ("pre", C.c_void_p), ("ld_pre", C.c_int64),
                ("s_mlp", C.c_void_p), ("ld_s_mlp", C.c_int64),
                ("deg", C.c_void_p),
                ("att_vec", C.c_void_p * 4), ("ln_weight", C.c_void_p * 4), ("ln_bias", C.c_void_p * 4),
                ("att_mix", C.c_void_p),
                ("g_low", C.c_void_p), ("ld_g_low", C.c_int64),
                ("g_high", C.c_void_p), ("ld_g_high", C.c_int64),
                ("g_mlp", C.c_void_p), ("ld_g_mlp", C.c_int64),
                ("g_struc", C.c_void_p), ("ld_g_struc", C.c_int64),
                ("d_att_vec", C.c_void_p * 4), ("d_ln_weight", C.c_void_p * 4),
                ("d_ln_bias", C.c_void_p * 4), ("d_att_mix", C.c_void_p),
                ("post_scale", C.c_void_p), ("ld_post_scale", C.c_int64), ("post_relu", C.c_int32),
                ("g_scale", C.c_void_p), ("post_drop", Dropout), ("defer", C.c_void_p)]


class ConvBwdSpmm(C.Structure):
    _fields_ = [("f_out", C.c_int32), ("row_offset", C.c_int64),
                ("g_low", C.c_void_p), ("ld_g_low", C.c_int64),
                ("g_high", C.c_void_p), ("ld_g_high", C.c_int64),
                ("g_struc", C.c_void_p), ("ld_g_struc", C.c_int64),
                ("s_high", C.c_void_p), ("ld_s_high", C.c_int64),
                ("s_struc", C.c_void_p), ("ld_s_struc", C.c_int64),
                ("inv_deg", C.c_void_p),
                ("mask_low", C.c_void_p), ("ld_mask_low", C.c_int64),
                ("mask_high", C.c_void_p), ("ld_mask_high", C.c_int64),
                ("dz_low", C.c_void_p), ("ld_dz_low", C.c_int64),
                ("dz_high", C.c_void_p), ("ld_dz_high", C.c_int64),
                ("d_struc", C.c_void_p), ("ld_d_struc", C.c_int64), ("self_scale", C.c_void_p),
                ("gather_bf16", C.c_int32)]


class ConvAggFwd(C.Structure):
    _fields_ = [("f_in", C.c_int32), ("f_pad", C.c_int32), ("f_out", C.c_int32), ("relu_after", C.c_int32),
                ("relu_mlp", C.c_int32), ("layernorm", C.c_int32), ("scale", C.c_float),
                ("xg", C.c_void_p), ("ld_xg", C.c_int64),
                ("xs", C.c_void_p), ("ld_xs", C.c_int64),
                ("w_low", C.c_void_p), ("w_high", C.c_void_p), ("w_mlp", C.c_void_p), ("ld_w", C.c_int64),
                ("att_vec", C.c_void_p * 4), ("ln_weight", C.c_void_p * 4), ("ln_bias", C.c_void_p * 4),
                ("att_mix", C.c_void_p),
                ("out", C.c_void_p), ("ld_out", C.c_int64),
                ("agg", C.c_void_p), ("ld_agg", C.c_int64),
                ("att", C.c_void_p), ("post_scale", C.c_void_p), ("ld_post_scale", C.c_int64), ("post_relu", C.c_int32),
                ("n_channels", C.c_int32), ("sg", C.c_void_p), ("ld_sg", C.c_int64), ("sg_bf16", C.c_int32),
                ("ss", C.c_void_p), ("ld_ss", C.c_int64), ("deg", C.c_void_p),
                ("ps", C.c_void_p), ("ld_ps", C.c_int64), ("row_scale", C.c_void_p), ("post_drop", Dropout),
                ("head_stats", C.c_void_p), ("ld_head_stats", C.c_int64),
                ("next_w_low", C.c_void_p), ("next_w_high", C.c_void_p), ("next_w_mlp", C.c_void_p), ("next_ld_w", C.c_int64),
                ("next_f", C.c_int32), ("next_relu", C.c_int32),
                ("next_zlh", C.c_void_p), ("ld_next_zlh", C.c_int64), ("next_zi", C.c_void_p), ("ld_next_zi", C.c_int64),
                ("agg_given", C.c_int32), ("reserved0", C.c_int32),
                ("agg_copy", C.c_void_p), ("ld_agg_copy", C.c_int64), ("xs_copy", C.c_void_p), ("ld_xs_copy", C.c_int64),
                ("next_x", C.c_void_p), ("ld_next_x", C.c_int64), ("next_drop", Dropout)]


class ConvAggBwd(C.Structure):
    _fields_ = [("f_in", C.c_int32), ("f_pad", C.c_int32), ("f_out", C.c_int32), ("relu_after", C.c_int32),
                ("relu_mlp", C.c_int32), ("layernorm", C.c_int32), ("scale", C.c_float),
                ("grad_out", C.c_void_p), ("ld_grad_out", C.c_int64),
                ("agg", C.c_void_p), ("ld_agg", C.c_int64),
                ("xs", C.c_void_p), ("ld_xs", C.c_int64),
                ("w_low", C.c_void_p), ("w_high", C.c_void_p), ("w_mlp", C.c_void_p), ("ld_w", C.c_int64),
                ("att_vec", C.c_void_p * 4), ("ln_weight", C.c_void_p * 4), ("ln_bias", C.c_void_p * 4),
                ("att_mix", C.c_void_p),
                ("d_params", C.c_void_p), ("post_scale", C.c_void_p), ("ld_post_scale", C.c_int64), ("post_relu", C.c_int32),
                ("n_channels", C.c_int32), ("ps", C.c_void_p), ("ld_ps", C.c_int64),
                ("ss", C.c_void_p), ("ld_ss", C.c_int64), ("deg", C.c_void_p),
                ("g_struc", C.c_void_p), ("ld_g_struc", C.c_int64), ("g_struc_scale", C.c_void_p),
                ("post_drop", Dropout), ("defer", C.c_void_p),
                ("head_stats", C.c_void_p), ("ld_head_stats", C.c_int64),
                ("out", C.c_void_p), ("ld_out", C.c_int64),
                ("next_a", C.c_void_p), ("next_xg", C.c_void_p), ("ld_next_xg", C.c_int64),
                ("next_row_scale", C.c_void_p), ("next_agg", C.c_void_p), ("ld_next_agg", C.c_int64),
                ("proj_dz", C.c_void_p), ("ld_proj_dz", C.c_int64),
                ("proj_w_low", C.c_void_p), ("proj_w_high", C.c_void_p), ("proj_w_mlp", C.c_void_p), ("proj_ld_w", C.c_int64),
                ("proj_f", C.c_int32), ("proj_d_w", C.c_void_p)]


class ConvAcmiiFwd(C.Structure):
    _fields_ = [("f_in", C.c_int32), ("f_pad", C.c_int32), ("f_out", C.c_int32), ("layernorm", C.c_int32),
                ("scale", C.c_float),
                ("xg", C.c_void_p), ("ld_xg", C.c_int64), ("xs", C.c_void_p), ("ld_xs", C.c_int64),
                ("w_low", C.c_void_p), ("w_high", C.c_void_p), ("w_mlp", C.c_void_p), ("ld_w", C.c_int64),
                ("att_vec", C.c_void_p * 4), ("ln_weight", C.c_void_p * 4), ("ln_bias", C.c_void_p * 4),
                ("att_mix", C.c_void_p),
                ("out", C.c_void_p), ("ld_out", C.c_int64), ("pre", C.c_void_p), ("ld_pre", C.c_int64),
                ("att", C.c_void_p), ("zlh", C.c_void_p), ("ld_zlh", C.c_int64), ("zi", C.c_void_p), ("ld_zi", C.c_int64),
                ("post_scale", C.c_void_p), ("ld_post_scale", C.c_int64), ("post_relu", C.c_int32),
                ("row_scale", C.c_void_p), ("post_drop", Dropout),
                ("n_channels", C.c_int32), ("ps", C.c_void_p), ("ld_ps", C.c_int64), ("ss", C.c_void_p), ("ld_ss", C.c_int64),
                ("deg", C.c_void_p)]


class ConvAcmiiBwd(C.Structure):
    _fields_ = [("f_in", C.c_int32), ("table", C.c_void_p),
                ("g_low", C.c_void_p), ("ld_g_low", C.c_int64), ("g_high", C.c_void_p), ("ld_g_high", C.c_int64),
                ("g_mlp", C.c_void_p), ("ld_g_mlp", C.c_int64), ("x", C.c_void_p), ("ld_x", C.c_int64),
                ("self_offset", C.c_int64), ("row_scale", C.c_void_p),
                ("d_w_low", C.c_void_p), ("d_w_high", C.c_void_p), ("d_w_mlp", C.c_void_p), ("ld_dw", C.c_int64),
                ("defer", C.c_void_p)]


class Loss(C.Structure):
    _fields_ = [("n_classes", C.c_int32), ("labels", C.c_void_p), ("row_weight", C.c_void_p),
                ("loss", C.c_void_p), ("dlogits", C.c_void_p), ("ld_dlogits", C.c_int64)]


class ReduceSeg(C.Structure):
    _fields_ = [("partial", C.c_void_p), ("nblk", C.c_int32), ("row_stride", C.c_int32),
                ("q0", C.c_int32), ("len", C.c_int32), ("dst", C.c_void_p),
                ("inner", C.c_int32), ("col_block", C.c_int32),
                ("outer_stride", C.c_int64), ("block_stride", C.c_int64),
                ("elem_stride", C.c_int32), ("reserved", C.c_int32)]


class ReduceList(C.Structure):
    _fields_ = [("n", C.c_int32), ("cap", C.c_int32), ("segs", C.POINTER(ReduceSeg))]


class SpmmOpts(C.Structure):
    _fields_ = [("vals", C.c_void_p), ("row_scale", C.c_void_p), ("sub", C.c_void_p), ("ld_sub", C.c_int64),
                ("sub_scale", C.c_void_p), ("relu", C.c_int32), ("g_bf16", C.c_int32)]


class AdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("step", C.c_void_p), ("numel", C.c_int64)]


class AdamConfig(C.Structure):
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("weight_decay", C.c_double), ("decoupled", C.c_int32), ("also_advance", C.c_void_p),
                ("arrive", C.c_void_p), ("pending", C.c_void_p)]


SMALL_ROLES = 17
# acm_small_step_t.t[layer][role]: role indices (include/acm_hip.h: ACM_SR_*)
SR_W_LOW, SR_W_HIGH, SR_W_MLP, SR_V_LOW, SR_V_HIGH, SR_V_MLP, SR_V_STRUC = range(7)
SR_LNW_LOW, SR_LNW_HIGH, SR_LNW_MLP, SR_LNW_STRUC, SR_LNB_LOW, SR_LNB_HIGH, SR_LNB_MLP, SR_LNB_STRUC, SR_MIX, SR_STRUC = range(7, 17)


class SmallStep(C.Structure):
    _fields_ = [("n_classes", C.c_int32), ("n_channels", C.c_int32), ("relu_before", C.c_int32), ("layernorm", C.c_int32),
                ("scale", C.c_float), ("train", C.c_int32), ("update", C.c_int32), ("reserved0", C.c_int32),
                ("t", (AdamTensor * SMALL_ROLES) * 2),
                ("x_vals", C.c_void_p), ("xt_src_pos", C.c_void_p), ("xt_vals", C.c_void_p), ("f_in", C.c_int32), ("reserved1", C.c_int32), ("drop_in", Dropout), ("drop_hidden", Dropout),
                ("row_scale", C.c_void_p), ("labels", C.c_void_p), ("row_weight", C.c_void_p), ("loss", C.c_void_p),
                ("logits", C.c_void_p), ("att1", C.c_void_p), ("att2", C.c_void_p),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("weight_decay", C.c_double), ("decoupled", C.c_int32), ("also_advance", C.c_void_p), ("arrive", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


_lib = None
_lock = threading.Lock()


def _declare(lib):
    vp, i64, i32, sz = C.c_void_p, C.c_int64, C.c_int, C.c_size_t
    lib.acm_version.restype = C.c_int
    lib.acm_version.argtypes = []
    lib.acm_last_error.restype = C.c_char_p
    lib.acm_last_error.argtypes = []
    lib.acm_csr_create.argtypes = [i64, i64, i64, vp, vp, vp, i32, C.POINTER(vp)]
    lib.acm_csr_transpose.argtypes = [vp, i32, C.POINTER(vp)]
    lib.acm_csr_slice_rows.argtypes = [vp, i64, i64, i32, C.POINTER(vp)]
    lib.acm_csr_destroy.argtypes = [vp]
    lib.acm_csr_destroy.restype = None
    lib.acm_csr_info.argtypes = [vp, C.POINTER(CsrInfo)]
    lib.acm_csr_build_streams.argtypes = [vp, i32, i32]
    lib.acm_csr_build_item_streams.argtypes = [vp, i32]
    lib.acm_shard_plan.argtypes = [i64, vp, i32, i64, vp]
    lib.acm_conv_acmii_fwd_workspace_bytes.argtypes = [vp, C.POINTER(sz)]
    lib.acm_conv_acmii_fwd.argtypes = [vp, C.POINTER(ConvAcmiiFwd), vp, sz, vp]
    lib.acm_acmii_table_bytes.argtypes = [i64, C.POINTER(sz)]
    lib.acm_acmii_table.argtypes = [i64, i32, vp, i64, vp, vp, i64, vp, sz, vp]
    lib.acm_conv_acmii_v_fwd.argtypes = [vp, C.POINTER(ConvAcmiiFwd), vp, vp, sz, vp]
    lib.acm_conv_acmii_v_bwd_workspace_bytes.argtypes = [vp, C.POINTER(sz)]
    lib.acm_conv_acmii_v_bwd.argtypes = [vp, C.POINTER(ConvAcmiiBwd), vp, sz, vp]
    lib.acm_linear_fwd.argtypes = [i64, i64, i64, vp, i64, vp, i64, vp, i32, vp, vp, i64, vp, sz, vp]
    lib.acm_bias_act.argtypes = [i64, i32, vp, i64, vp, i32, vp, vp]
    lib.acm_linear_bwd_workspace_bytes.argtypes = [i64, i32, i32, C.POINTER(sz)]
    lib.acm_linear_bwd.argtypes = [i64, i32, i32, vp, i64, vp, i64, vp, i64, C.c_float, i32, vp, i64, vp, vp, sz, vp, vp]
    lib.acm_linear_fwd_add.argtypes = [i64, i32, i32, vp, i64, vp, i64, vp, i32, vp, vp, i64, vp, i64, vp]
    lib.acm_linear_bwd_recompute.argtypes = [i64, i32, i32, vp, i64, vp, i64, vp, i32, vp, vp, i64, vp, i64, vp, vp, sz, vp, vp]
    lib.acm_bias_act_bwd_workspace_bytes.argtypes = [i64, i32, C.POINTER(sz)]
    lib.acm_bias_act_bwd.argtypes = [i64, i32, vp, i64, vp, i64, C.c_float, i32, vp, i64, vp, vp, sz, vp, vp]
    lib.acm_spmm_workspace_bytes.argtypes = [vp, i32, C.POINTER(sz)]
    lib.acm_gemm_workspace_bytes.argtypes = [i32, i32, i64, i64, i64, C.POINTER(sz)]
    lib.acm_gemm.argtypes = [i32, i32, i64, i64, i64, vp, i64, vp, i64, vp, i64, i32, vp, sz, vp]
    lib.acm_proj_bwd_workspace_bytes.argtypes = [i64, i64, i32, C.POINTER(sz)]
    lib.acm_proj_fwd.argtypes = [i64, i64, i32, vp, i64, vp, vp, vp, i64, i32, vp, i64, vp, i64, vp]
    lib.acm_proj_fwd_at.argtypes = [i64, i64, i32, vp, i64, vp, vp, vp, i64, i32, vp, i64, i64, vp, i64, vp]
    lib.acm_proj_bwd.argtypes = [i64, i64, i32, vp, i64, vp, i64, vp, vp, vp, i64, vp, i64, vp, i64, i64, i64, vp, sz, vp, vp]
    lib.acm_gemm_split.argtypes = [i32, i32, i64, i64, i64, vp, i64, vp, i64, vp, i64, i64, vp, i64, i32, vp, sz, vp]
    lib.acm_gemm_blocks.argtypes = [i32, i32, i64, i64, i64, vp, i64, vp, i64, vp, i64, i64, i64, i32, vp, sz, vp]
    lib.acm_gemm_drop.argtypes = [i32, i32, i64, i64, i64, vp, i64, vp, i64, vp, i64, i64, i64, i32, C.POINTER(Dropout), vp, sz, vp]
    lib.acm_proj3.argtypes = [i64, i64, vp, i64, vp, vp, vp, i64, i64, i64, vp, i64, i64, vp, i64, i32, C.POINTER(Dropout), vp]
    lib.acm_spmm.argtypes = [vp, vp, i64, i32, vp, i64, vp, sz, vp]
    lib.acm_spmm_v.argtypes = [vp, vp, vp, i64, i32, vp, i64, i32, vp, sz, vp]
    lib.acm_adam_step.argtypes = [i32, vp, vp, vp]
    lib.acm_dropout.argtypes = [i64, i64, vp, i64, vp, i64, i64, vp, vp]
    lib.acm_spmm_ex.argtypes = [vp, vp, i64, i32, vp, i64, vp, vp, sz, vp]
    lib.acm_cast_bf16.argtypes = [i64, i64, vp, i64, vp, i64, vp]
    lib.acm_conv_fwd.argtypes = [vp, C.POINTER(ConvFwd), vp, sz, vp]
    lib.acm_conv_head_fwd.argtypes = [i64, C.POINTER(ConvFwd), vp]
    lib.acm_conv_aggw_bwd_workspace_bytes.argtypes = [i64, i64, C.POINTER(C.c_size_t)]
    lib.acm_conv_aggw_bwd.argtypes = [i64, i64, i64, vp, i64, vp, i64, C.POINTER(ConvBwdLocal), vp, vp, vp, i64, vp, C.c_size_t, vp]
    lib.acm_conv_aggw_fwd.argtypes = [i64, i64, i64, vp, i64, vp, i64, vp, vp, vp, i64, vp, i64, C.POINTER(ConvFwd), vp]
    lib.acm_conv_bwd_local_workspace_bytes.argtypes = [i64, i32, i32, C.POINTER(sz)]
    lib.acm_conv_bwd_local.argtypes = [i64, C.POINTER(ConvBwdLocal), vp, sz, vp]
    lib.acm_conv_bwd_spmm.argtypes = [vp, C.POINTER(ConvBwdSpmm), vp, sz, vp]
    lib.acm_conv_agg_fwd.argtypes = [vp, C.POINTER(ConvAggFwd), vp, sz, vp]
    lib.acm_conv_agg_bwd_workspace_bytes.argtypes = [i64, i32, i32, C.POINTER(sz)]
    lib.acm_conv_agg_bwd.argtypes = [i64, C.POINTER(ConvAggBwd), vp, sz, vp]
    lib.acm_nll_loss_workspace_bytes.argtypes = [i64, C.POINTER(sz)]
    lib.acm_nll_loss.argtypes = [i64, i32, vp, i64, vp, vp, vp, vp, i64, vp, sz, vp, vp]
    lib.acm_eval_metrics_workspace_bytes.argtypes = [i64, i32, C.POINTER(sz)]
    lib.acm_eval_metrics.argtypes = [i64, i32, vp, i64, vp, vp, i64, i32, i32, vp, vp, sz, vp]
    lib.acm_reduce_flush.argtypes = [vp, vp]
    lib.acm_conv_fwd_tail_workspace_bytes.argtypes = [i64, i32, i32, C.POINTER(sz)]
    lib.acm_conv_fwd_tail.argtypes = [vp, C.POINTER(ConvFwd), C.POINTER(Loss), C.POINTER(ConvBwdLocal), vp, sz, vp, sz, vp]
    lib.acm_small_step_workspace_bytes.argtypes = [vp, vp, vp, C.POINTER(sz)]
    lib.acm_small_step.argtypes = [vp, vp, vp, C.POINTER(SmallStep), vp]
    for name in EXPORTED_SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("acm_version", "acm_last_error", "acm_csr_destroy"):
            fn.restype = C.c_int


def library_path():
    return os.environ.get("ACM_HIP_LIBRARY", _build.LIB_PATH)


def load(build_if_missing=True):
    """Load (building first if the in-tree .so is missing/stale and hipcc exists)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if "ACM_HIP_LIBRARY" not in os.environ and build_if_missing:
            try:
                if _build.is_stale():
                    _build.build_library()
            except Exception as exc:  # no hipcc on this box: fall through to the existing file
                if not os.path.exists(path):
                    raise RuntimeError(f"libacm_hip.so is not built and cannot be built here: {exc}")
        if not os.path.exists(path):
            raise RuntimeError(f"libacm_hip.so not found at {path}; run `python -m acm_gnn_amd.build`")
        lib = C.CDLL(path)
        _declare(lib)
        ver = lib.acm_version()
        if ver != ABI_VERSION:
            raise RuntimeError(f"libacm_hip.so ABI version {ver} != expected {ABI_VERSION}")
        _lib = lib
        return _lib


def check(status, what=""):
    if status != ACM_OK:
        msg = load().acm_last_error().decode(errors="replace")
        raise RuntimeError(f"{what or 'libacm_hip'} failed with {STATUS_NAMES.get(status, status)}: {msg}")
