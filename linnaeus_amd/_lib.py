"""ctypes binding of liblnx_hip.so (the C ABI declared in include/lnx.h).

The library is the product: there is no Python/CPU fallback.  If it is missing or does not
load, importing anything that needs a kernel raises immediately.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LNX_LIB_PATH") or os.path.join(_HERE, "liblnx_hip.so")  # override: A/B runs against another build

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_RELU, ACT_GELU_BWD, ACT_RELU_BWD, ACT_GELU_D, ACT_MUL_AUX = 0, 1, 2, 3, 4, 5, 6
TN_WS_FLOATS = 256 * (256 * 128 + 256)  # == LNX_TN_WS_FLOATS
ADDR_PLAIN, ADDR_PATCH2 = 0, 1


class LnxError(RuntimeError):
    pass


class RowMap(C.Structure):
    _fields_ = [("group", C.c_int), ("pad", C.c_int), ("off", C.c_int)]


class GemmArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("A", C.c_void_p), ("lda", C.c_int64),
        ("W", C.c_void_p), ("ldw", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int64),
        ("out_f32", C.c_int),
        ("a_mode", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("Cin", C.c_int),
        ("c_mode", C.c_int), ("c_map", RowMap),
        ("bias", C.c_void_p),
        ("c2", C.c_void_p), ("ldc2", C.c_int64),
        ("act", C.c_int),
        ("aux", C.c_void_p), ("ldaux", C.c_int64),
        ("gamma", C.c_void_p), ("rowscale", C.c_void_p), ("rows_per_sample", C.c_int),
        ("res", C.c_void_p), ("ldres", C.c_int64),
        ("c8", C.c_void_p), ("ldc8", C.c_int64), ("c8_scales", C.c_void_p),
    ]


class WgradArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("dY", C.c_void_p), ("lddy", C.c_int64),
        ("A", C.c_void_p), ("lda", C.c_int64),
        ("a_mode", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("Cin", C.c_int),
        ("dW", C.c_void_p), ("lddw", C.c_int64),
        ("k_perm_c", C.c_int),
        ("db", C.c_void_p),
        ("splits", C.c_int),
        ("k_store", C.c_int),
        ("ws", C.c_void_p), ("ws_floats", C.c_int64),
        ("defer", C.c_int),
    ]


class SoftCEArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("C", C.c_int),
        ("logits", C.c_void_p), ("ld", C.c_int64),
        ("target", C.c_void_p),
        ("soft", C.c_void_p), ("smoothing", C.c_float),
        ("class_weight", C.c_void_p),
        ("ignore_index", C.c_int64),
        ("row_scale", C.c_void_p), ("scale", C.c_float),
        ("loss", C.c_void_p), ("loss_sum", C.c_void_p),
        ("dlogits", C.c_void_p), ("ldd", C.c_int64),
    ]


ADAMW_MAX_GROUPS = 16


class AdamWDesc(C.Structure):
    _fields_ = [("p", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("g", C.c_void_p), ("n", C.c_int64), ("group", C.c_int), ("block_start", C.c_int)]


class AdamWHyper(C.Structure):
    _fields_ = [("ngroups", C.c_int)] + [(k, C.c_float * ADAMW_MAX_GROUPS) for k in ("lr", "beta1", "beta2", "eps", "weight_decay", "bias_c1", "bias_c2", "omb1", "omb2")]


_lib = None


def lib() -> C.CDLL:
    """Load the HIP library or fail loudly (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LnxError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C linnaeus_amd/csrc`). linnaeus_amd has no CPU fallback."
            )
        _lib = C.CDLL(LIB_PATH)
        _lib.lnx_last_error.restype = C.c_char_p
        _lib.lnx_nt_kernel_launches.restype = C.c_int64
        _lib.lnx_convmlp_bwd_ws_floats.restype = C.c_int64
        _lib.lnx_meta_heads_bwd_part_floats.restype = C.c_int64
        # A/B runs against an OLDER build (LNX_LIB_PATH=... LNX_LIB_OLDER=1): entry points added since are allowed to be missing; calling
        # one then fails with ctypes' AttributeError
        older = bool(os.environ.get("LNX_LIB_PATH")) and os.environ.get("LNX_LIB_OLDER") == "1"
        for name in EXPORTS:
            if not hasattr(_lib, name) and not older:
                raise LnxError(f"{LIB_PATH} does not export {name}; rebuild it")
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise LnxError(f"{what} failed (rc={rc}): {lib().lnx_last_error().decode()}")


# lnx_aug_pointwise operations (include/lnx.h)
AUG_CLAMP, AUG_POSTERIZE, AUG_SOLARIZE, AUG_SOLARIZE_ADD, AUG_INVERT, AUG_BRIGHTNESS, AUG_CONTRAST = range(7)

# lnx_last_nt_kernel / lnx_nt_kernel_launches kinds (include/lnx.h)
NT_KERNEL_NONE, NT_KERNEL_V1, NT_KERNEL_V2, NT_KERNEL_SKINNY, NT_KERNEL_V4, NT_KERNEL_FP8, NT_KERNEL_V7, NT_KERNEL_MX8, NT_KERNEL_V9, NT_KERNEL_EXPERIMENT = 0, 1, 2, 3, 4, 6, 7, 8, 9, 15

# every symbol include/lnx.h declares (kept in sync by tests/test_abi.py)
EXPORTS = [
    "lnx_last_error", "lnx_version", "lnx_device_cus", "lnx_set_cu_margin",
    "lnx_gemm_nt", "lnx_last_nt_kernel", "lnx_nt_kernel_launches", "lnx_nt_dispatch", "lnx_gemm_tn", "lnx_gemm_tn_flush", "lnx_gemm_tn_discard", "lnx_amax", "lnx_quantize_fp8", "lnx_gemm_nt_fp8", "lnx_quantize_mxfp8", "lnx_gemm_nt_mxfp8", "lnx_dropout_mul", "lnx_dropout_residual", "lnx_plan_dropout_bytes", "lnx_plan_set_dropout", "lnx_plan_attn_dropout_bytes", "lnx_plan_set_attn_dropout",
    "lnx_layernorm_fwd", "lnx_layernorm_bwd", "lnx_layernorm_bwd_flush", "lnx_layernorm_bwd_discard",
    "lnx_dwconv7_fwd", "lnx_dwconv7_wgrad",
    "lnx_gemm_nt_group_ok", "lnx_gemm_nt_group", "lnx_rope_cos_table", "lnx_rope_cos_tables", "lnx_attn_bwd_ws_floats", "lnx_attn_fwd", "lnx_attn_bwd", "lnx_attn_bwd_flush", "lnx_attn_bwd_discard",
    "lnx_im2col_stem", "lnx_scale_cast", "lnx_layerscale_bwd", "lnx_layerscale_apply_wgrad", "lnx_fill_rows", "lnx_colsum_rows",
    "lnx_agg2_fwd", "lnx_agg2_bwd", "lnx_pack_meta", "lnx_meta_heads_supported", "lnx_meta_heads_fwd", "lnx_meta_heads_bwd", "lnx_meta_heads_bwd_part_floats", "lnx_prep_weights", "lnx_prep_blocks", "lnx_softce", "lnx_softce_multi", "lnx_stem_fwd", "lnx_stem_fwd_ok", "lnx_adamw_blocks", "lnx_grad_sumsq", "lnx_adamw_step",
    "lnx_mix_rows", "lnx_mix_meta",
    "lnx_aug_pointwise", "lnx_aug_saturation", "lnx_aug_rowstat", "lnx_aug_rescale", "lnx_aug_affine", "lnx_aug_stencil", "lnx_erase_rects", "lnx_u8hwc_to_f32chw",
    "lnx_convmlp_supported", "lnx_convmlp_fwd", "lnx_convmlp_bwd", "lnx_convmlp_bwd_ws_floats",
    "lnx_plan_create", "lnx_plan_destroy", "lnx_plan_workspace_bytes", "lnx_plan_num_params", "lnx_plan_param_name",
    "lnx_plan_param_numel", "lnx_plan_num_drop_calls", "lnx_plan_logits_numel", "lnx_plan_logits_offset", "lnx_plan_logits_ld",
    "lnx_plan_bind", "lnx_plan_forward", "lnx_plan_backward", "lnx_plan_segment_params", "lnx_plan_profile_begin", "lnx_plan_profile_end", "lnx_plan_profile_begin_spans", "lnx_plan_profile_end_ex", "lnx_plan_set_wgrad_stream", "lnx_plan_set_meta_stream",
]


class MetaHeadArgs(C.Structure):  # == lnx_meta_head_args
    _fields_ = [
        ("B", C.c_int), ("C", C.c_int), ("dim", C.c_int), ("off", C.c_int),
        ("meta", C.c_void_p), ("meta_width", C.c_int), ("eps", C.c_float),
        ("w0", C.c_void_p), ("ldw0", C.c_int), ("b0", C.c_void_p), ("ln0_w", C.c_void_p), ("ln0_b", C.c_void_p),
        ("w1", C.c_void_p), ("ldw1", C.c_int), ("b1", C.c_void_p), ("ln1_w", C.c_void_p), ("ln1_b", C.c_void_p),
        ("w2", C.c_void_p), ("ldw2", C.c_int), ("b2", C.c_void_p), ("ln2_w", C.c_void_p), ("ln2_b", C.c_void_p),
        ("t0", C.c_void_p), ("h0", C.c_void_p), ("x", C.c_void_p), ("h1", C.c_void_p), ("n1", C.c_void_p), ("h2", C.c_void_p),
        ("m0", C.c_void_p), ("r0", C.c_void_p), ("m1", C.c_void_p), ("r1", C.c_void_p), ("m2", C.c_void_p), ("r2", C.c_void_p),
        ("tok", C.c_void_p), ("tok_row_stride", C.c_int64), ("tok_row_offset", C.c_int64),
    ]


class MetaHeadBwdArgs(C.Structure):  # == lnx_meta_head_bwd_args
    _fields_ = [
        ("B", C.c_int), ("C", C.c_int), ("dim", C.c_int),
        ("g", C.c_void_p), ("g_row_stride", C.c_int64), ("g_row_offset", C.c_int64),
        ("w1t", C.c_void_p), ("ldw1t", C.c_int), ("w2t", C.c_void_p), ("ldw2t", C.c_int),
        ("ln0_w", C.c_void_p), ("ln1_w", C.c_void_p), ("ln2_w", C.c_void_p),
        ("t0", C.c_void_p), ("h0", C.c_void_p), ("x", C.c_void_p), ("h1", C.c_void_p), ("n1", C.c_void_p), ("h2", C.c_void_p),
        ("m0", C.c_void_p), ("r0", C.c_void_p), ("m1", C.c_void_p), ("r1", C.c_void_p), ("m2", C.c_void_p), ("r2", C.c_void_p),
        ("dp2", C.c_void_p), ("dp1", C.c_void_p), ("dp0", C.c_void_p), ("part", C.c_void_p),
        ("d_w0", C.c_void_p), ("d_b0", C.c_void_p), ("d_ln0_w", C.c_void_p), ("d_ln0_b", C.c_void_p),
        ("d_w1", C.c_void_p), ("d_b1", C.c_void_p), ("d_ln1_w", C.c_void_p), ("d_ln1_b", C.c_void_p),
        ("d_w2", C.c_void_p), ("d_b2", C.c_void_p), ("d_ln2_w", C.c_void_p), ("d_ln2_b", C.c_void_p),
    ]


class LnArgs(C.Structure):
    _fields_ = [
        ("M", C.c_int), ("C", C.c_int), ("eps", C.c_float),
        ("x", C.c_void_p), ("x_dtype", C.c_int), ("ldx", C.c_int64), ("x_map", RowMap),
        ("w", C.c_void_p), ("b", C.c_void_p),
        ("y", C.c_void_p), ("y_dtype", C.c_int), ("ldy", C.c_int64), ("y_map", RowMap),
        ("add", C.c_void_p), ("ldadd", C.c_int64),
        ("mean", C.c_void_p), ("rstd", C.c_void_p),
        ("y8", C.c_void_p), ("ldy8", C.c_int64), ("y8_scales", C.c_void_p),
    ]


class LnBwdArgs(C.Structure):
    _fields_ = [
        ("M", C.c_int), ("C", C.c_int),
        ("dy", C.c_void_p), ("dy_dtype", C.c_int), ("lddy", C.c_int64), ("dy_map", RowMap),
        ("x", C.c_void_p), ("x_dtype", C.c_int), ("ldx", C.c_int64), ("x_map", RowMap),
        ("w", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p),
        ("gin", C.c_void_p), ("ldgin", C.c_int64),
        ("dx", C.c_void_p), ("dx_dtype", C.c_int), ("lddx", C.c_int64),
        ("dw", C.c_void_p), ("db", C.c_void_p), ("relu_mask", C.c_int),
        ("ws", C.c_void_p), ("ws_floats", C.c_int64),
        ("dx2", C.c_void_p), ("dx2_dtype", C.c_int), ("lddx2", C.c_int64), ("dx2_rowscale", C.c_void_p), ("dx2_rows_per_sample", C.c_int),
        ("dx2_8", C.c_void_p), ("dx2_8_scales", C.c_void_p), ("lddx2_8", C.c_int64),
        ("defer", C.c_int),
    ]


class RopeTable(C.Structure):
    _fields_ = [
        ("freqs", C.c_void_p), ("cos_out", C.c_void_p), ("dsin_out", C.c_void_p),
        ("heads", C.c_int), ("H", C.c_int), ("W", C.c_int), ("pad_", C.c_int),
    ]


class DwconvArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int),
        ("x", C.c_void_p), ("x_dtype", C.c_int), ("w49", C.c_void_p), ("bias", C.c_void_p),
        ("flip", C.c_int), ("res", C.c_void_p), ("y", C.c_void_p), ("y_dtype", C.c_int),
    ]


class DwconvWgradArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int),
        ("x", C.c_void_p), ("x_dtype", C.c_int), ("dy", C.c_void_p), ("dy_dtype", C.c_int),
        ("dw", C.c_void_p), ("db", C.c_void_p),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("B", C.c_int), ("N", C.c_int), ("E", C.c_int), ("heads", C.c_int),
        ("qkv", C.c_void_p), ("cos_tab", C.c_void_p), ("o", C.c_void_p), ("lse", C.c_void_p),
        ("drop_mask", C.c_void_p), ("drop_inv_keep", C.c_float),
    ]


class AttnBwdArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("B", C.c_int), ("N", C.c_int), ("E", C.c_int), ("heads", C.c_int),
        ("qkv", C.c_void_p), ("cos_tab", C.c_void_p), ("o", C.c_void_p), ("lse", C.c_void_p),
        ("d_o", C.c_void_p), ("dqkv", C.c_void_p), ("freq_ws", C.c_void_p), ("delta", C.c_void_p),
        ("drop_mask", C.c_void_p), ("drop_inv_keep", C.c_float),
        ("dsin_tab", C.c_void_p), ("dfreqs", C.c_void_p), ("defer_freqs", C.c_int),
    ]


class PrepDesc(C.Structure):
    _fields_ = [
        ("src", C.c_void_p), ("dst", C.c_void_p), ("dst_t", C.c_void_p),
        ("rows", C.c_int), ("cols", C.c_int), ("ld", C.c_int), ("ld_t", C.c_int),
        ("P", C.c_int), ("mode", C.c_int), ("block_start", C.c_int),
    ]


PREP_CAST, PREP_CONV_PERM, PREP_DW49 = 0, 1, 2


class ConvMlpArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("M", C.c_int), ("C", C.c_int),
        ("ln", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
        ("gamma", C.c_void_p), ("rowscale", C.c_void_p), ("rows_per_sample", C.c_int),
        ("x", C.c_void_p), ("out", C.c_void_p), ("z", C.c_void_p),
        ("y", C.c_void_p), ("ln_w", C.c_void_p), ("ln_b", C.c_void_p), ("ln_eps", C.c_float), ("ln_out", C.c_void_p),
        ("mean", C.c_void_p), ("rstd", C.c_void_p),
    ]


class ConvMlpBwdArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("M", C.c_int), ("C", C.c_int),
        ("g", C.c_void_p), ("ln", C.c_void_p), ("z", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p),
        ("w2t", C.c_void_p), ("w1t", C.c_void_p), ("gamma", C.c_void_p), ("rowscale", C.c_void_p), ("rows_per_sample", C.c_int),
        ("act", C.c_void_p), ("dh", C.c_void_p), ("dz", C.c_void_p), ("dln", C.c_void_p), ("dgamma", C.c_void_p),
        ("y", C.c_void_p), ("ln_w", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("d_ln_w", C.c_void_p), ("d_ln_b", C.c_void_p),
        ("ws", C.c_void_p), ("ws_floats", C.c_int64),
        ("dz_plain", C.c_int), ("ln_defer", C.c_int),
    ]


class StemArgs(C.Structure):  # == lnx_stem_args
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("ln_w", C.c_void_p), ("ln_b", C.c_void_p), ("patches", C.c_void_p),
                ("pre", C.c_void_p), ("y", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("B", C.c_int), ("Cin", C.c_int), ("H", C.c_int),
                ("W", C.c_int), ("Cout", C.c_int), ("eps", C.c_float)]


class MixArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("perm", C.c_void_p), ("valid", C.c_void_p), ("out", C.c_void_p),
        ("B", C.c_int), ("row", C.c_int64), ("H", C.c_int), ("W", C.c_int), ("lam", C.c_float),
        ("h0", C.c_int), ("h1", C.c_int), ("w0", C.c_int), ("w1", C.c_int), ("mode", C.c_int),
    ]
