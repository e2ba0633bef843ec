"""ctypes binding of `libief_hip.so` (C-ABI in `include/ief_hip.h`).

This is the ONLY route from the host code to compute: every function below launches a
hand-written gfx950 kernel on torch's current HIP stream, with torch tensors used purely as
device-memory handles (`data_ptr()`).  There is no CPU or eager-PyTorch fallback: if the library
is missing, or a tensor is not an fp16/fp32 device tensor of the documented layout, the call raises.
"""
import ctypes
import os
from ctypes import POINTER, Structure, byref, c_char_p, c_float, c_int, c_longlong, c_void_p

import torch

# IEF_HIP_LIB / IEF_PLAN_FILE: A/B two builds of the library (and their tuned tables) on one GPU box
ABI_VERSION = 4   # include/ief_hip.h IEF_ABI_VERSION
_LIB_PATH = os.environ.get("IEF_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libief_hip.so")
_lib = None


class HipExtensionMissing(RuntimeError):
    pass


class IefGemmParams(Structure):
    _fields_ = [
        ("A", c_void_p), ("A2", c_void_p), ("W", c_void_p), ("Out", c_void_p),
        ("bias", c_void_p), ("rowvec", c_void_p), ("residual", c_void_p),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("lda", c_int), ("ldw", c_int), ("ldo", c_int), ("ldr", c_int),
        ("strideA", c_longlong), ("strideW", c_longlong), ("strideO", c_longlong), ("strideR", c_longlong),
        ("H", c_int), ("Wd", c_int), ("C1", c_int), ("C2", c_int), ("Ho", c_int), ("Wo", c_int),
        ("stride", c_int), ("ups", c_int), ("batch_images", c_int),
        ("rows_per_batch", c_int), ("out_scale", c_float), ("tile_hint", c_int),
        ("E1", c_void_p), ("E2", c_void_p), ("CE1", c_int), ("CE2", c_int),
        ("splits", c_int), ("ws", c_void_p), ("flags", c_int), ("zeros", c_void_p), ("stages", c_int),
        ("pad_hi_only", c_int),
        ("rstat_out", c_void_p), ("rstat_in", c_void_p), ("rstat_slots", c_int), ("colsum", c_void_p), ("ln_eps", c_float),
        ("cstat_out", c_void_p), ("cnt", c_void_p),
    ]


class IefAttnParams(Structure):
    _fields_ = [
        ("Q", c_void_p), ("K", c_void_p), ("V", c_void_p), ("Out", c_void_p),
        ("B", c_int), ("heads", c_int), ("N", c_int), ("L", c_int), ("d", c_int),
        ("ldq", c_int), ("ldk", c_int), ("ldv", c_int), ("ldo", c_int),
        ("scale", c_float),
        ("q_src", c_void_p), ("k_src", c_void_p), ("v_src", c_void_p), ("lse", c_void_p), ("variant", c_int),
    ]


class IefAttnBwdParams(Structure):
    _fields_ = [
        ("Q", c_void_p), ("K", c_void_p), ("V", c_void_p), ("dO", c_void_p), ("lse", c_void_p), ("delta", c_void_p),
        ("dQ", c_void_p), ("dK", c_void_p), ("dV", c_void_p),
        ("B", c_int), ("heads", c_int), ("N", c_int), ("L", c_int), ("d", c_int),
        ("ldq", c_int), ("ldk", c_int), ("ldv", c_int), ("ldo", c_int), ("lddq", c_int), ("lddk", c_int), ("lddv", c_int),
        ("scale", c_float), ("ds_mul", c_float), ("kv_splits", c_int), ("ws", c_void_p),
    ]


class IefMapLossParams(Structure):
    _fields_ = [
        ("Q", c_void_p), ("K", c_void_p), ("ref", c_void_p), ("dQ", c_void_p), ("loss", c_void_p),
        ("B", c_int), ("heads", c_int), ("N", c_int), ("L", c_int), ("d", c_int),
        ("ldq", c_int), ("ldk", c_int), ("lddq", c_int),
        ("scale", c_float), ("gcoef", c_float), ("loss_coef", c_float), ("accumulate", c_int),
    ]


class IefCrossParams(Structure):
    _fields_ = [
        ("Q", c_void_p), ("K", c_void_p), ("V", c_void_p), ("Out", c_void_p),
        ("B", c_int), ("heads", c_int), ("N", c_int), ("L", c_int), ("d", c_int),
        ("ldq", c_int), ("ldk", c_int), ("ldv", c_int), ("ldo", c_int),
        ("scale", c_float),
        ("edit_src", c_void_p), ("edit_slot", c_void_p), ("MT", c_void_p), ("coef", c_void_p),
    ]


class IefGemmF32Params(Structure):
    _fields_ = [
        ("A", c_void_p), ("A2", c_void_p), ("W", c_void_p), ("Out", c_void_p),
        ("bias", c_void_p), ("rowvec", c_void_p), ("residual", c_void_p),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("lda", c_int), ("ldw", c_int), ("ldo", c_int), ("ldr", c_int),
        ("rows_per_batch", c_int), ("out_scale", c_float),
        ("conv", c_int), ("H", c_int), ("Wd", c_int), ("C1", c_int), ("C2", c_int), ("Ho", c_int), ("Wo", c_int),
        ("stride", c_int), ("ups", c_int), ("batch_images", c_int), ("pad_hi_only", c_int),
        ("E1", c_void_p), ("E2", c_void_p), ("CE1", c_int), ("CE2", c_int),
        ("batch", c_int), ("heads", c_int),
        ("sAb", c_longlong), ("sAh", c_longlong), ("sWb", c_longlong), ("sWh", c_longlong), ("sOb", c_longlong), ("sOh", c_longlong),
        ("a_src", c_void_p), ("w_src", c_void_p), ("transb", c_int), ("a_scalar", c_int),
        ("splits", c_int), ("ws", c_void_p),
        ("x3", c_int), ("sa", c_float), ("sb", c_float), ("vec_out", c_int), ("al32", c_int), ("fast_ok", c_int),
        ("bytesA", ctypes.c_uint), ("bytesW", ctypes.c_uint), ("bytesA2", ctypes.c_uint), ("bytesE1", ctypes.c_uint),
        ("bytesE2", ctypes.c_uint), ("Wp", c_void_p), ("geglu", c_int),
    ]


class IefAttnF32Params(Structure):
    _fields_ = [
        ("Q", c_void_p), ("K", c_void_p), ("V", c_void_p), ("Out", c_void_p),
        ("B", c_int), ("heads", c_int), ("N", c_int), ("L", c_int), ("d", c_int),
        ("ldq", c_int), ("ldk", c_int), ("ldv", c_int), ("ldo", c_int),
        ("sQb", c_longlong), ("sKb", c_longlong), ("sVb", c_longlong), ("sOb", c_longlong),
        ("scale", c_float),
        ("q_src", c_void_p), ("k_src", c_void_p), ("v_src", c_void_p),
        ("x3", c_int),
        ("OutP", c_void_p), ("planeO", c_longlong), ("sOPb", c_longlong), ("ldp", c_int), ("p_scale", c_float),
        ("Qp", c_void_p), ("Kp", c_void_p), ("Vp", c_void_p), ("planeQ", c_longlong), ("planeK", c_longlong), ("planeV", c_longlong),
        ("zeros", c_void_p), ("lse", c_void_p),
    ]


class IefAttnBwdF32Params(Structure):
    _fields_ = [
        ("Q", c_void_p), ("K", c_void_p), ("V", c_void_p), ("dO", c_void_p), ("lse", c_void_p), ("delta", c_void_p),
        ("dQ", c_void_p), ("dK", c_void_p), ("dV", c_void_p),
        ("B", c_int), ("heads", c_int), ("N", c_int), ("L", c_int), ("d", c_int),
        ("ldq", c_int), ("ldk", c_int), ("ldv", c_int), ("ldo", c_int), ("lddq", c_int), ("lddk", c_int), ("lddv", c_int),
        ("scale", c_float), ("ds_mul", c_float),
    ]


class IefGemmX3pParams(Structure):
    _fields_ = [
        ("A", c_void_p), ("planeA", c_longlong), ("A2", c_void_p), ("planeA2", c_longlong),
        ("E1", c_void_p), ("planeE1", c_longlong), ("E2", c_void_p), ("planeE2", c_longlong),
        ("W", c_void_p), ("planeW", c_longlong),
        ("Out", c_void_p), ("OutP", c_void_p), ("planeO", c_longlong),
        ("bias", c_void_p), ("rowvec", c_void_p), ("residual", c_void_p),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("lda", c_int), ("ldw", c_int), ("ldo", c_int), ("ldp", c_int), ("ldr", c_int),
        ("conv", c_int), ("H", c_int), ("Wd", c_int), ("C1", c_int), ("C2", c_int), ("Ho", c_int), ("Wo", c_int),
        ("stride", c_int), ("ups", c_int), ("batch_images", c_int), ("pad_hi_only", c_int), ("CE1", c_int), ("CE2", c_int),
        ("rows_per_batch", c_int), ("out_scale", c_float), ("inv_scale", c_float),
        ("tile", c_int), ("splits", c_int), ("ws", c_void_p), ("geglu", c_int), ("zeros", c_void_p),
        ("rstat_out", c_void_p), ("cstat_out", c_void_p), ("rstat_in", c_void_p), ("colsum", c_void_p),
        ("rstat_slots", c_int), ("rstat_cnt", c_int), ("ln_eps", c_float),
    ]


EXPORTS = [
    "ief_abi_version", "ief_target_arch", "ief_gemm_f16", "ief_conv3x3_f16", "ief_conv_in_f32",
    "ief_conv_out_f32", "ief_gn_splits", "ief_groupnorm_silu_f16", "ief_layernorm_f16", "ief_geglu_f16",
    "ief_attn_flash_f16", "ief_attn_cross_p2p_f16", "ief_attn_probs_f16", "ief_attn_apply_f16",
    "ief_cfg_ddim_step_f32", "ief_timestep_embedding_f16", "ief_silu_f16", "ief_cast_f32_to_f16",
    "ief_cast_f16_to_f32", "ief_select_step", "ief_advance_step", "ief_add_f16", "ief_struct_size",
    "ief_softmax_rows_f16", "ief_transpose_f16", "ief_pointwise_f32",
    "ief_attn_bwd_delta_f32", "ief_attn_bwd_f16", "ief_groupnorm_bwd_f16", "ief_layernorm_bwd_f16", "ief_geglu_il_f16",
    "ief_geglu_il_bwd_f16", "ief_zero_insert2x_f16", "ief_pool2x2_sum_f16", "ief_conv_out_bwd_f32",
    "ief_nti_loss_grad_f32", "ief_nti_adam_f32", "ief_gemm_tile_bn", "ief_gather_rows_f16",
    "ief_attn_map_loss_bwd_f16", "ief_axpy_f32", "ief_map_loss_blocks", "ief_groupnorm_cstat_f16", "ief_gemm_tile_bm",
    # reference-precision (fp32) mode
    "ief_gemm_f32", "ief_softmax_rows_f32", "ief_p2p_cross_edit_f32", "ief_attn_cross_p2p_f32", "ief_groupnorm_silu_f32", "ief_layernorm_f32",
    "ief_add_f32", "ief_silu_f32", "ief_geglu_il_f32", "ief_timestep_embedding_f32", "ief_gather_rows_f32",
    "ief_conv_in_f32act", "ief_conv_out_f32act", "ief_image_u8", "ief_gemm_f32_bn", "ief_attn_flash_f32", "ief_gemm_x3_bn", "ief_gemm_x3_bm", "ief_gemm_x3_bn_k", "ief_gemm_x3_set_variant", "ief_x3_split_weights", "ief_groupnorm_f32_ws_floats", "ief_groupnorm_silu_f32_ws", "ief_groupnorm_bwd_f32_ws_floats", "ief_groupnorm_bwd_f32_ws",
    # activation gradients of the fp32-storage modes (csrc/backward_f32.hip)
    "ief_groupnorm_bwd_f32", "ief_layernorm_bwd_f32", "ief_geglu_il_bwd_f32", "ief_zero_insert2x_f32", "ief_pool2x2_sum_f32",
    "ief_conv_out_bwd_f32w", "ief_softmax_bwd_rows_f32", "ief_transpose_batched_f32", "ief_map_loss_rows_blocks",
    "ief_map_loss_rows_f32", "ief_nti_adam_f32g",
    # ABI 4: split-operand contractions on pre-split planes (csrc/gemm_x3p.hip)
    "ief_gemm_x3p", "ief_gemm_x3p_tile_bm", "ief_gemm_x3p_tile_bn", "ief_gemm_x3p_tile_wn", "ief_x3_split_act", "ief_groupnorm_silu_x3p_ws", "ief_layernorm_x3p", "ief_groupnorm_silu_x3p_small", "ief_groupnorm_silu_reg", "ief_groupnorm_reg_fits", "ief_attn_bwd_x3", "ief_attn_bwd_delta_f32in",
]


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Load the shared library (once).  Raises HipExtensionMissing when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise HipExtensionMissing(
            f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C image-editing-framework_amd/csrc`).  There is no CPU fallback."
        )
    lib = ctypes.CDLL(_LIB_PATH)
    for name in EXPORTS:
        getattr(lib, name)  # AttributeError here = header / library mismatch
    lib.ief_abi_version.restype = c_int
    lib.ief_target_arch.restype = c_char_p
    lib.ief_gn_splits.restype = c_int
    lib.ief_gn_splits.argtypes = [c_int]
    lib.ief_gemm_f16.argtypes = [POINTER(IefGemmParams), c_int, c_void_p]
    lib.ief_conv3x3_f16.argtypes = [POINTER(IefGemmParams), c_void_p]
    lib.ief_conv_in_f32.argtypes = [c_void_p] * 4 + [c_int] * 5 + [c_void_p]
    lib.ief_conv_out_f32.argtypes = [c_void_p] * 4 + [c_int] * 5 + [c_void_p]
    lib.ief_groupnorm_silu_f16.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int, c_int, c_int, c_float, c_int, c_void_p]
    lib.ief_layernorm_f16.argtypes = [c_void_p] * 4 + [c_int, c_int, c_float, c_void_p]
    lib.ief_geglu_f16.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p]
    lib.ief_attn_flash_f16.argtypes = [POINTER(IefAttnParams), c_void_p]
    lib.ief_attn_cross_p2p_f16.argtypes = [POINTER(IefCrossParams), c_void_p]
    lib.ief_attn_probs_f16.argtypes = [POINTER(IefAttnParams), c_void_p, c_void_p]
    lib.ief_attn_apply_f16.argtypes = [POINTER(IefAttnParams), c_void_p, c_void_p]
    lib.ief_cfg_ddim_step_f32.argtypes = [c_void_p] * 6 + [c_longlong, c_void_p]
    lib.ief_timestep_embedding_f16.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p]
    lib.ief_silu_f16.argtypes = [c_void_p, c_void_p, c_longlong, c_void_p]
    lib.ief_cast_f32_to_f16.argtypes = [c_void_p, c_void_p, c_longlong, c_void_p]
    lib.ief_cast_f16_to_f32.argtypes = [c_void_p, c_void_p, c_longlong, c_void_p]
    lib.ief_select_step.argtypes = [c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_void_p]
    lib.ief_advance_step.argtypes = [c_void_p, c_void_p]
    lib.ief_add_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_longlong, c_void_p]
    lib.ief_softmax_rows_f16.argtypes = [c_void_p, c_int, c_int, c_void_p]
    lib.ief_transpose_f16.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p]
    lib.ief_pointwise_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    lib.ief_attn_bwd_delta_f32.argtypes = [c_void_p] * 3 + [c_int] * 6 + [c_void_p]
    lib.ief_attn_bwd_f16.argtypes = [POINTER(IefAttnBwdParams), c_int, c_void_p]
    lib.ief_attn_map_loss_bwd_f16.argtypes = [POINTER(IefMapLossParams), c_void_p]
    lib.ief_axpy_f32.argtypes = [c_void_p, c_void_p, c_float, c_longlong, c_void_p]
    lib.ief_map_loss_blocks.restype = c_int
    lib.ief_map_loss_blocks.argtypes = [c_int, c_int]
    lib.ief_groupnorm_cstat_f16.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                            c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p]
    lib.ief_gemm_tile_bm.restype = c_int
    lib.ief_gemm_tile_bm.argtypes = [c_int]
    lib.ief_groupnorm_bwd_f16.argtypes = [c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 8 + [c_int, c_int, c_int, c_float,
                                                                                              c_int, c_void_p]
    lib.ief_layernorm_bwd_f16.argtypes = [c_void_p] * 5 + [c_int, c_int, c_float, c_void_p]
    lib.ief_geglu_il_f16.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p]
    lib.ief_geglu_il_bwd_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]
    lib.ief_zero_insert2x_f16.argtypes = [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]
    lib.ief_pool2x2_sum_f16.argtypes = [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]
    lib.ief_conv_out_bwd_f32.argtypes = [c_void_p] * 3 + [c_int] * 5 + [c_void_p]
    lib.ief_nti_loss_grad_f32.argtypes = [c_void_p] * 7 + [c_int, c_float, c_void_p]
    lib.ief_nti_adam_f32.argtypes = [c_void_p] * 8 + [c_int, c_void_p]
    lib.ief_gather_rows_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_longlong, c_void_p]
    lib.ief_gemm_f32.argtypes = [POINTER(IefGemmF32Params), c_void_p]
    lib.ief_softmax_rows_f32.argtypes = [c_void_p, c_longlong, c_int, c_void_p]
    lib.ief_attn_flash_f32.argtypes = [POINTER(IefAttnF32Params), c_void_p]
    lib.ief_p2p_cross_edit_f32.argtypes = [c_void_p] * 5 + [c_int] * 4 + [c_void_p]
    lib.ief_attn_cross_p2p_f32.argtypes = [POINTER(IefAttnF32Params)] + [c_void_p] * 5
    lib.ief_groupnorm_silu_f32.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                           c_float, c_int, c_void_p]
    lib.ief_gemm_x3_bm.argtypes = [c_int, c_int]
    lib.ief_gemm_x3_bn_k.argtypes = [c_int, c_int, c_int]
    if os.environ.get("IEF_X3_WIDE"):          # A/B runs: bit 0 clear = keep every launch on the 128 x 80 tile; bit 1 set = 32-key flash tiles
        lib.ief_gemm_x3_set_variant.argtypes = [c_int]
        lib.ief_gemm_x3_set_variant(int(os.environ["IEF_X3_WIDE"]))
    lib.ief_x3_split_weights.argtypes = [c_void_p, c_void_p, c_longlong, c_float, c_void_p]
    lib.ief_groupnorm_bwd_f32.argtypes = [c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 6 + [c_int, c_int, c_int, c_float, c_int, c_void_p]
    lib.ief_layernorm_bwd_f32.argtypes = [c_void_p] * 5 + [c_longlong, c_int, c_float, c_void_p]
    lib.ief_geglu_il_bwd_f32.argtypes = [c_void_p] * 3 + [c_longlong, c_int, c_void_p]
    lib.ief_zero_insert2x_f32.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    lib.ief_pool2x2_sum_f32.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    lib.ief_conv_out_bwd_f32w.argtypes = [c_void_p] * 3 + [c_int] * 5 + [c_void_p]
    lib.ief_softmax_bwd_rows_f32.argtypes = [c_void_p, c_void_p, c_longlong, c_int, c_float, c_void_p]
    lib.ief_transpose_batched_f32.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
    lib.ief_map_loss_rows_blocks.argtypes = [c_longlong]
    lib.ief_map_loss_rows_f32.argtypes = [c_void_p] * 4 + [c_longlong, c_int, c_float, c_float, c_void_p]
    lib.ief_nti_adam_f32g.argtypes = [c_void_p] * 7 + [c_int, c_void_p]
    lib.ief_groupnorm_f32_ws_floats.argtypes = [c_int, c_int, c_int]
    lib.ief_groupnorm_f32_ws_floats.restype = c_longlong
    lib.ief_groupnorm_silu_f32_ws.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                              c_float, c_int, c_void_p, c_longlong, c_void_p]
    lib.ief_groupnorm_bwd_f32_ws_floats.argtypes = [c_int, c_int, c_int]
    lib.ief_groupnorm_bwd_f32_ws_floats.restype = c_longlong
    lib.ief_groupnorm_bwd_f32_ws.argtypes = [c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 6 + [c_int, c_int, c_int, c_float, c_int,
                                             c_void_p, c_longlong, c_void_p]
    lib.ief_layernorm_f32.argtypes = [c_void_p] * 4 + [c_longlong, c_int, c_float, c_void_p]
    lib.ief_add_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_longlong, c_void_p]
    lib.ief_silu_f32.argtypes = [c_void_p, c_void_p, c_longlong, c_void_p]
    lib.ief_geglu_il_f32.argtypes = [c_void_p, c_void_p, c_longlong, c_int, c_void_p]
    lib.ief_timestep_embedding_f32.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p]
    lib.ief_gather_rows_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_longlong, c_void_p]
    lib.ief_conv_in_f32act.argtypes = [c_void_p] * 4 + [c_int] * 5 + [c_void_p]
    lib.ief_conv_out_f32act.argtypes = [c_void_p] * 4 + [c_int] * 5 + [c_void_p]
    lib.ief_image_u8.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    lib.ief_gemm_x3p.argtypes = [POINTER(IefGemmX3pParams), c_void_p]
    lib.ief_gemm_x3p_tile_bm.argtypes = [c_int]
    lib.ief_gemm_x3p_tile_bn.argtypes = [c_int]
    lib.ief_gemm_x3p_tile_wn.argtypes = [c_int]
    lib.ief_x3_split_act.argtypes = [c_void_p, c_void_p, c_longlong, c_longlong, c_int, c_int, c_int, c_float, c_void_p]
    lib.ief_groupnorm_silu_x3p_ws.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_longlong, c_void_p, c_void_p,
                                              c_int, c_int, c_int, c_float, c_int, c_void_p, c_longlong, c_void_p]
    lib.ief_attn_bwd_x3.argtypes = [POINTER(IefAttnBwdF32Params), c_int, c_void_p]
    lib.ief_attn_bwd_delta_f32in.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.ief_groupnorm_reg_fits.argtypes = [c_int] * 4
    lib.ief_groupnorm_silu_reg.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_longlong, c_void_p, c_void_p,
                                           c_int, c_int, c_int, c_float, c_int, c_void_p]
    lib.ief_groupnorm_silu_x3p_small.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_longlong, c_void_p, c_void_p,
                                                 c_int, c_int, c_int, c_float, c_int, c_void_p]
    lib.ief_layernorm_x3p.argtypes = [c_void_p, c_void_p, c_longlong, c_void_p, c_void_p, c_longlong, c_int, c_float, c_void_p]
    if lib.ief_abi_version() != ABI_VERSION:
        raise HipExtensionMissing("libief_hip.so ABI version mismatch; rebuild")
    lib.ief_struct_size.argtypes = [c_int]
    for which, st in ((0, IefGemmParams), (1, IefAttnParams), (2, IefCrossParams), (3, IefAttnBwdParams), (4, IefMapLossParams),
                      (5, IefGemmF32Params), (6, IefAttnF32Params), (7, IefGemmX3pParams)):
        if lib.ief_struct_size(which) != ctypes.sizeof(st):
            raise HipExtensionMissing(f"{st.__name__}: ctypes layout ({ctypes.sizeof(st)} B) != library "
                                      f"({lib.ief_struct_size(which)} B); rebuild libief_hip.so")
    _lib = lib
    return lib


_ERR = {-1: "IEF_EINVAL (null / missing pointer)", -2: "IEF_ESHAPE (unsupported shape)", -3: "IEF_EALIGN (alignment)"}


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {_ERR.get(rc, f'hipError {rc}')}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# ------------------------------------------------------------------------------- live kernel timing
# bench.py brackets individual launches with HIP events ON THE STREAM THE KERNEL RUNS ON to get the
# dominant kernel's average launch duration (roofline.achieved).  Off by default (zero overhead).
_prof = None
PROF_SHAPES = os.environ.get("IEF_PROF_SHAPES", "0") == "1"     # append MxNxK to the timed kernel names (tests/exp_shapes.py)
# tile id -> (BM, BN, WAVES_M, WAVES_N), as in csrc/gemm_conv.hip
_TILES = {1: (128, 128, 2, 2), 2: (64, 128, 2, 2), 3: (64, 64, 2, 2), 4: (128, 64, 2, 2), 5: (64, 160, 2, 2),
          6: (128, 160, 2, 2), 7: (128, 160, 4, 2), 8: (256, 128, 4, 2), 9: (128, 128, 4, 2),
          14: (256, 80, 8, 1), 15: (256, 80, 8, 1),
          16: (128, 160, 4, 2), 17: (128, 128, 4, 2), 18: (256, 128, 4, 2), 19: (64, 160, 2, 2), 20: (64, 64, 2, 2),
          21: (128, 160, 2, 2)}
_TILE_NL = {16: 4, 17: 4, 18: 4, 19: 2, 20: 2, 21: 2}   # loader waves of a tile (csrc/gemm_conv.hip, NL): they stage, the others multiply
_HALO_TILES = (14, 15)  # conv3x3_halo_kernel (csrc/gemm_conv.hip): 3x3 / stride 1 / pad 1 convolutions only, rows of <= 64 pixels; 15: + 4 loader waves


def _kname(tile, conv, stages=2):
    bm, bn, wm, wn = _TILES[tile]
    if tile in _HALO_TILES:
        return f"conv3x3_halo_kernel<{bm}, {bn}, {wm}, {wn}, {4 if tile == 15 else 0}>"
    return f"igemm_f16_kernel<{bm}, {bn}, {wm}, {wn}, {stages or 2}, {_TILE_NL.get(tile, 0)}, {'true' if conv else 'false'}>"


def _ring_bytes(tile, stages):
    bm, bn, wm, wn = _TILES[tile]
    if tile in _HALO_TILES:         # fixed LDS image (two super-tile buffers + a 4-slot weight ring); one "ring depth"
        return 155776 if stages == 4 else 1 << 30
    rp = 64 * (_TILE_NL.get(tile, 0) or wm * wn) // 8
    rows = -(-bm // rp) * rp + -(-bn // rp) * rp
    return 2 * stages * rows * 64


_prof_staged = None      # kernel name -> bytes its launches staged into LDS by LDS-DMA since profile_begin (the planes kernels report them)


def profile_begin():
    global _prof, _prof_staged
    _prof = []
    _prof_staged = {}


def note_staged(name: str, nbytes: float):
    """a planes kernel's launch reports the bytes it moves L2 -> LDS (both operands, both planes, every K tile of every output tile):
    what the per-CU LDS-DMA fill rate prices"""
    if _prof is not None and _prof_staged is not None:
        _prof_staged[name] = _prof_staged.get(name, 0.0) + nbytes


def profile_staged():
    """{kernel name: LDS-staged bytes} of the launches since the last profile_begin"""
    return dict(_prof_staged or {})


def profile_end(with_bytes=False):
    """-> list of (kernel_name, algorithmic_flops, milliseconds[, algorithmic_bytes])"""
    global _prof
    rec, _prof = _prof, None
    torch.cuda.synchronize()
    if with_bytes:
        return [(name, flops, e0.elapsed_time(e1), nbytes) for name, flops, nbytes, e0, e1 in rec]
    return [(name, flops, e0.elapsed_time(e1)) for name, flops, nbytes, e0, e1 in rec]


class _Timed:
    """`nbytes`: the launch's ALGORITHMIC HBM traffic (every operand read once, the output written once)"""

    def __init__(self, name, flops, nbytes=0.0):
        self.name, self.flops, self.nbytes = name, flops, nbytes

    def __enter__(self):
        if _prof is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream())

    def __exit__(self, *a):
        if _prof is not None:
            self.e1.record(torch.cuda.current_stream())
            _prof.append((self.name, self.flops, self.nbytes, self.e0, self.e1))


# GroupNorm statistics from the producer: a conv / 1x1-projection launch whose output is an NHWC activation of a level with
# at least this many pixels per image can leave per-tile column sums next to it (`col_stats=True` -> `(out, ColStats)`);
# the caller hands that object to the `groupnorm` that consumes the output (`cstat=` / `cstat2=`), which then folds the
# sums instead of running its own statistics pass.  The hand-off is an explicit value travelling beside the tensor it
# describes: nothing is attached to tensor objects, so a buffer re-written by any other launch cannot be paired with
# statistics of its earlier contents unless the caller itself keeps passing the old object.
GN_CSTAT = os.environ.get("IEF_GN_CSTAT", "1") == "1"
HALO_HEURISTIC = os.environ.get("IEF_HALO_HEURISTIC", "1") == "1"   # untuned plain 3x3 convolutions take the halo kernel
_CSTAT_MIN_HW = 256


class ColStats:
    """(sum, sum of squares) per M tile and output channel of ONE launch's fp16 output (IefGemmParams.cstat_out)"""
    __slots__ = ("buf", "bm", "hw", "ptr", "numel")

    def __init__(self, buf, bm, hw, out):
        self.buf, self.bm, self.hw = buf, bm, hw
        self.ptr, self.numel = out.data_ptr(), out.numel()

    def describes(self, t) -> bool:
        return t.data_ptr() == self.ptr and t.numel() == self.numel


def _attach_cstat(lib, p, out, M, N, hw):
    """decide whether this launch emits column statistics; returns the ColStats (owning the scratch tensor) or None"""
    if not GN_CSTAT or hw is None or hw < _CSTAT_MIN_HW or (p.splits > 1 and not p.cnt) or (p.flags & 2) or AUTOTUNE:
        return None
    bm = lib.ief_gemm_tile_bm(p.tile_hint)
    if bm <= 0 or hw % bm or M % hw:
        return None
    cs = torch.empty(M // bm, N, 2, dtype=torch.float32, device=out.device)
    p.cstat_out = cs.data_ptr()
    return ColStats(cs, bm, hw, out)


def pick_tile(M: int, N: int, batch: int = 1) -> int:
    """tile choice (mirrors dispatch_igemm in csrc/gemm_conv.hip): fill the 256 CUs"""
    nblk = lambda bm, bn: -(-M // bm) * -(-N // bn) * batch
    if nblk(128, 128) >= 384:
        return 1
    if nblk(64, 128) >= 256:
        return 2
    return 3


_zero_pages = {}   # device zero page: source of padded / out-of-range chunks for the LDS-DMA staging


def _zeros(device):
    z = _zero_pages.get(device)
    if z is None:
        z = _zero_pages[device] = torch.zeros(128, dtype=torch.float16, device=device)
    return z.data_ptr()


# ---- split-K arrival counters (IefGemmParams.cnt): one zeroed int per output tile, zero again when the launch ends.
# Launches that can overlap must not share counters, so:
#   eager launches   take the next slots of a ring of 2^20 ints per device (thousands of launches deep: a launch has ended
#                    long before its slots come round again);
#   captured launches take slots of an ARENA its owner allocated for that graph (`counter_arena`): a captured step graph
#                    keeps its own counters for as long as it lives, whatever other graphs or eager launches do; without an
#                    arena a captured launch falls back to the separate reducer launch.
SPLITK_INLAUNCH = os.environ.get("IEF_SPLITK_INLAUNCH", "1") == "1"
# ... and only where the last arriver's serial read stays short: splits x tile bytes (fp32) at most this many KiB.  Measured
# (same box, SD1.5 batch-4 step): combining EVERY split-K launch in the launch took the step from 7.45 to 7.97 ms — the
# tuned plans cut K 2-16 ways over 64..256-row tiles, 128 KiB - 1.3 MiB per output tile, which one workgroup reads at
# ~100 GB/s while the reducer launch reads with the whole chip (the guide's splitk-seam verdict).
SPLITK_INLAUNCH_MAX_KB = int(os.environ.get("IEF_SPLITK_INLAUNCH_MAX_KB", "64"))
_CNT_RING = 1 << 20
_cnt_ring = {}          # device -> [tensor, position]
_cnt_arena = None       # [tensor, position] while a capture that owns one is running
_cnt_used = 0           # ints handed out so far (owners size their arena by the delta over a warm-up pass)


def counters_used() -> int:
    return _cnt_used


class counter_arena:
    """`with counter_arena(n, device) as arena:` — split-K launches captured inside take their arrival counters from
    `arena` (a zeroed int32 tensor the caller keeps alive with the graph)"""

    def __init__(self, n, device):
        if isinstance(device, int):
            device = torch.device("cuda", device)
        self.t = torch.zeros(max(int(n), 1), dtype=torch.int32, device=device)

    def __enter__(self):
        global _cnt_arena
        self.prev, _cnt_arena = _cnt_arena, [self.t, 0]
        return self.t

    def __exit__(self, *a):
        global _cnt_arena
        _cnt_arena = self.prev


def _splitk_counters(device, n, combine_kb=0):
    """device pointer of n zeroed ints for one split-K launch, or None (-> separate reducer launch)"""
    global _cnt_used
    if not SPLITK_INLAUNCH or combine_kb > SPLITK_INLAUNCH_MAX_KB:
        return None
    if _capturing():
        if _cnt_arena is None or _cnt_arena[1] + n > _cnt_arena[0].numel():
            return None
        t, pos = _cnt_arena
        _cnt_arena[1] = pos + n
        _cnt_used += n
        return t.data_ptr() + 4 * pos
    ring = _cnt_ring.get(device)
    if ring is None:
        ring = _cnt_ring[device] = [torch.zeros(_CNT_RING, dtype=torch.int32, device=device), 0]
    if n > _CNT_RING:
        return None
    if ring[1] + n > _CNT_RING:
        # wrap: the slots about to be re-used may belong to launches of OTHER streams that have not run yet, and a launch
        # that faulted (or an exception between two ticket draws) leaves non-zero counters behind -- wait for the device and
        # zero the ring before handing its start out again (once per ~_CNT_RING counters: not on the hot path, which is the
        # captured graph with its own arena)
        reset_counters(device)
    ptr = ring[0].data_ptr() + 4 * ring[1]
    ring[1] += n
    _cnt_used += n
    return ptr


def reset_counters(device=None):
    """zero the eager split-K arrival-counter ring(s) after a device sync -- on wrap, and for callers recovering from a failed
    launch (a faulted kernel leaves its tickets drawn: every later launch re-using those slots would combine too early or never)"""
    for dev, ring in _cnt_ring.items():
        if device is None or dev == device:
            torch.cuda.synchronize(dev)
            ring[0].zero_()
            ring[1] = 0


def _tile_count(lib, tile_hint, M, N, splits=1):
    """(output tiles, KiB the last arriver of a tile would read: splits x BM x BN fp32)"""
    bm, bn = lib.ief_gemm_tile_bm(tile_hint), lib.ief_gemm_tile_bn(tile_hint)
    if bm <= 0 or bn <= 0:
        return 0, 0
    return -(-M // bm) * -(-N // bn), splits * bm * bn * 4 // 1024


SPLITK = os.environ.get("IEF_SPLITK", "1") != "0"
_SPLIT_TARGET = int(os.environ.get("IEF_SPLIT_TARGET", "512"))   # blocks wanted on the 256 CUs
_SPLIT_MAX = int(os.environ.get("IEF_SPLIT_MAX", "16"))
_SPLIT_MIN_KT = int(os.environ.get("IEF_SPLIT_MIN_KT", "6"))     # K tiles (of 64) each slice keeps at least

# ---- plan selection: tuned table first, heuristic otherwise -----------------------------------------
# `tuned_plans.json` (next to this file) maps "conv|M|N|K" / "gemm|M|N|K" -> [tile, splits]; it is produced
# on the GPU by `autotune_plan` (tests/tune_plans.py) and covers the SD1.5 layer shapes at batch 1 / 2 / 4.
_PLAN_FILE = os.environ.get("IEF_PLAN_FILE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_plans.json")
_plans = None
AUTOTUNE = os.environ.get("IEF_AUTOTUNE", "0") == "1"   # tune unseen shapes on first use (outside graph capture)


def _plan_table():
    global _plans
    if _plans is None:
        _plans = {}
        if os.path.exists(_PLAN_FILE) and os.environ.get("IEF_NO_PLAN_TABLE", "0") != "1":
            import json
            with open(_PLAN_FILE) as f:
                _plans = {k: tuple(v) for k, v in json.load(f).items()}
    return _plans


def save_plans(path=None):
    import json
    with open(path or _PLAN_FILE, "w") as f:
        json.dump({k: list(v) for k, v in sorted(_plan_table().items())}, f, indent=0)


def heuristic_plan(M: int, N: int, K: int):
    """(tile, splits) without measurements.  SD channel widths are multiples of 320 = 2 x 160, so 160-wide
    tiles waste nothing on N; layers whose M x N alone gives too few tiles (the 32x32 .. 8x8 UNet levels, K up to
    23040) are cut along K until ~512 blocks exist, each slice keeping >= 6 K tiles of 64."""
    nk = -(-K // 64)
    if N % 160 == 0 and M >= 128:
        t6 = -(-M // 128) * (N // 160)
        if t6 >= 192 and (nk < 24 or t6 >= 384):
            return 7, 1
        if nk >= 12 and SPLITK:
            splits = max(1, min(-(-_SPLIT_TARGET // t6), nk // _SPLIT_MIN_KT, _SPLIT_MAX))
            if t6 * splits >= 128:
                return 6, splits
    t128 = -(-M // 128) * -(-N // 128)
    if not SPLITK or t128 >= 384 or nk < 24:
        return pick_tile(M, N), 1
    tile, tiles = 1, t128
    if M <= 64 or (tiles < 32 and M < 128):
        tile, tiles = 2, -(-M // 64) * -(-N // 128)
    splits = min(-(-_SPLIT_TARGET // tiles), nk // _SPLIT_MIN_KT, _SPLIT_MAX)
    if splits <= 1:
        return pick_tile(M, N), 1
    return tile, splits


def pick_plan(M: int, N: int, K: int, conv: bool = False, variant: str = ""):
    """(tile, splits, stages); `variant` ("|u": the convolution with the fused nearest-2x upsample) has its own table key
    and falls back to the plain key of the same (M, N, K)"""
    kind = "conv" if conv else "gemm"
    hit = _plan_table().get(f"{kind}|{M}|{N}|{K}{variant}") if variant else None
    if hit is None:
        hit = _plan_table().get(f"{kind}|{M}|{N}|{K}")
    if hit is None:
        hit = heuristic_plan(M, N, K)
    return (hit[0], hit[1], hit[2] if len(hit) > 2 else 2)


def candidate_plans(M: int, N: int, K: int, conv: bool = False):
    nk = -(-K // 64)
    out = []
    for t, (bm, bn, _, _) in _TILES.items():
        if t in _HALO_TILES and not conv:
            continue
        for s in (1, 2, 4, 8, 16):
            blocks = -(-M // bm) * -(-N // bn) * s
            if s > 1 and (nk // s < 4 or blocks > 2048):
                continue
            if blocks < 48 and s < 16:
                continue
            for st in (2, 3, 4):
                if _ring_bytes(t, st) <= 160 * 1024 and (st == 2 or nk // s >= st):
                    out.append((t, s, st))
    return out


def _time_graph(fn, iters=20):
    """us per call of fn(i), i = 0..iters-1, replayed from one hipGraph"""
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    used0 = counters_used()
    with torch.cuda.stream(st):
        fn(0)
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    arena = counter_arena((counters_used() - used0) * iters, torch.cuda.current_device())   # split-K combines in the launch
    g = torch.cuda.CUDAGraph()
    with arena, torch.cuda.graph(g):
        for i in range(iters):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def _cold_copies(w, iters=20, budget=320 << 20):
    """enough clones of a weight tensor that consecutive launches never find it in the 256 MiB Infinity Cache —
    inside the UNet every layer's weights stream from HBM once per step, which is what the tuner must see"""
    k = max(1, min(iters, -(-budget // max(1, w.numel() * w.element_size()))))
    return [w] + [w.clone() for _ in range(k - 1)]


def autotune_plan(kind: str, M: int, N: int, K: int, run, variant: str = ""):
    """time run(tile, splits, stages, i) for every candidate plan (hipGraph of 20 launches each, launch i using
    the i-th cache-cold weight copy); remember the best"""
    best = None
    for t, sp, st in candidate_plans(M, N, K, conv=(kind == "conv")):
        try:
            us = _time_graph(lambda i: run(t, sp, st, i))
        except RuntimeError:
            continue
        if best is None or us < best[0]:
            best = (us, t, sp, st)
    if best is not None:
        _plan_table()[f"{kind}|{M}|{N}|{K}{variant}"] = (best[1], best[2], best[3])
    return best


def _capturing() -> bool:
    return torch.cuda.is_current_stream_capturing()


def _ptr(t):
    return None if t is None else t.data_ptr()


def _dev16(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float16):
        raise TypeError(f"{name}: expected an fp16 device tensor, got "
                        f"{type(t).__name__ if not isinstance(t, torch.Tensor) else (t.dtype, t.device)}")
    if t.stride(-1) != 1:
        raise ValueError(f"{name}: last dimension must be contiguous")
    return t


# How the contractions of the fp32-storage modes run (csrc/exact_f32.hip, csrc/split_x3.hip):
#   "f32"  the fp32-input MFMA (bitwise an fp32 fma chain; precision="f32")
#   "x3"   split operands: hi + lo fp16 halves of every fp32 element, three fp16 MFMAs per product (precision="f16x3")
# A model sets it for the duration of its own calls (`with hip.f32_contraction(mode)`): one Python thread drives the
# library, and a captured graph keeps the kernels that were chosen while it was recorded.
_F32_CONTRACT = "f32"
# powers of two.  Activations: 1 — the fp16 range itself (|x| < 65504, as on the fp16-storage path; a larger scale would narrow it:
# 4 capped activations at 16376) and the scale the pre-split operand planes carry (planes.py); gfx950's MFMA keeps fp16 subnormal
# inputs, so the lo half of a small element only loses ABSOLUTE resolution (2^-25), far below the 2^-22 relative error of the
# large elements that dominate a dot product.  Weights 2^8 (|w| < 255), softmax maps 2^14 (<= 1).
X3_SCALE_ACT, X3_SCALE_W, X3_SCALE_PROB = 1.0, 256.0, 16384.0


class f32_contraction:
    def __init__(self, mode: str):
        if mode not in ("f32", "x3"):
            raise ValueError('f32_contraction: mode must be "f32" or "x3"')
        self.mode = mode

    def __enter__(self):
        global _F32_CONTRACT
        self.saved, _F32_CONTRACT = _F32_CONTRACT, self.mode
        return self

    def __exit__(self, *exc):
        global _F32_CONTRACT
        _F32_CONTRACT = self.saved
        return False


# (data_ptr, shape, scale) -> (weakref to the weight tensor, its _version, planes fp16 [2, N, K]).  The entry dies WITH the weight
# (weakref.finalize): a model that is dropped frees its planes (they are as large as the weights).  A captured graph bakes the
# planes' address in: an in-place update of a weight after capture would replace the entry and free planes the graph still
# reads, so weights are immutable once a graph over them exists (the weight broadcast runs before any forward).
_x3_planes = {}
X3_FUSE_GEGLU = os.environ.get("IEF_X3_FUSE_GEGLU", "1") == "1"   # 0: FF1 writes its pre-activation, ief_geglu_il_f32 follows (A/B)
X3_PRESPLIT = os.environ.get("IEF_X3_PRESPLIT", "1") == "1"       # 0: split the weights in every launch, like the activations


def x3_weight_planes(w, scale=None):
    """the pre-split fp16 planes [2, N, K] of an fp32 WEIGHT tensor (hi = fp16(s w), lo = fp16(s w - hi)), made once per tensor
    and kept: a weight is static, so splitting it in every launch and workgroup would spend vector instructions the GEMM's
    steady-state loop has none to spare of.  None while a graph is being captured and the planes do not exist yet."""
    scale = X3_SCALE_W if scale is None else scale
    key = (w.data_ptr(), tuple(w.shape), float(scale))
    hit = _x3_planes.get(key)
    if hit is not None and hit[0]() is w and hit[1] == w._version:
        return hit[2]
    if _capturing() or w.numel() % 4 or not w.is_contiguous():
        return None
    lib = load()
    n = w.shape[0]
    planes = torch.empty(2, n, w.numel() // n, dtype=torch.float16, device=w.device)
    _check(lib.ief_x3_split_weights(w.data_ptr(), planes.data_ptr(), w.numel(), float(scale), _stream()), "ief_x3_split_weights")
    import weakref
    _x3_planes[key] = (weakref.ref(w), w._version, planes)
    weakref.finalize(w, _x3_planes.pop, key, None)
    return planes


def _set_x3(p, sa, sb, w=None) -> str:
    """fill the split-operand fields of an IefGemmF32Params from the current mode; returns the kernel family's name.
    w: the launch's B operand when it is a weight ([N, K...] contiguous fp32): its cached pre-split planes ride along"""
    if _F32_CONTRACT == "x3":
        p.x3, p.sa, p.sb = 1, sa, sb
        if w is not None and X3_PRESPLIT and (w.numel() // w.shape[0]) % 8 == 0:
            planes = x3_weight_planes(w, sb)
            if planes is not None:
                p._planes = planes          # keep alive until the launch is queued
                p.Wp = planes.data_ptr()
        return "igemm_x3_kernel"
    return "igemm_f32_kernel"


def _is32(t) -> bool:
    """True for the fp32 activations / weights of the reference-precision mode (`csrc/exact_f32.hip`)"""
    return isinstance(t, torch.Tensor) and t.dtype == torch.float32


def _act32(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
        raise TypeError(f"{name}: expected an fp32 device tensor (reference-precision mode), got "
                        f"{type(t).__name__ if not isinstance(t, torch.Tensor) else (t.dtype, t.device)}")
    if t.stride(-1) != 1:
        raise ValueError(f"{name}: last dimension must be contiguous")
    return t


def _rows_ld32(t, name):
    _act32(t, name)
    cols = t.shape[-1]
    ld = t.stride(-2) if t.dim() >= 2 else cols
    rows = 1
    for s in t.shape[:-1]:
        rows *= s
    if t.dim() == 3 and t.shape[0] > 1 and t.stride(0) != t.shape[1] * ld:
        raise ValueError(f"{name}: batch stride must equal rows * ld")
    return rows, cols, ld


def _dev32(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise TypeError(f"{name}: expected a contiguous fp32 device tensor")
    return t


def _devi32(t, name):
    if t is None:
        return None
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
        raise TypeError(f"{name}: expected a contiguous int32 device tensor")
    return t


def _rows_ld(t, name):
    """(rows, cols, ld) of a 2-D/3-D tensor whose leading dims collapse to rows with one stride."""
    _dev16(t, name)
    cols = t.shape[-1]
    ld = t.stride(-2) if t.dim() >= 2 else cols
    rows = 1
    for s in t.shape[:-1]:
        rows *= s
    if t.dim() == 3 and t.shape[0] > 1 and t.stride(0) != t.shape[1] * ld:
        raise ValueError(f"{name}: batch stride must equal rows * ld")
    return rows, cols, ld


# ------------------------------------------------------------------------------- GEMM / conv
def gemm(a, w, bias=None, residual=None, rowvec=None, rows_per_batch=0, out=None, out_scale=1.0, tile_hint=0, splits=1,
         stages=2, geglu=False, ln=None, row_stats=False, col_stats=False):
    """out[..., n] = (a[..., :] . w[n, :] + bias[n] + rowvec[row // rows_per_batch, n] + residual[..., n]) * out_scale

    a: fp16 [..., K] (last dim contiguous, uniform row stride); w: fp16 [N, K]; bias/rowvec fp32.
    LayerNorm folding (see include/ief_hip.h): `row_stats=True` also returns the per-row moments of the output,
    a fp32 [M, tiles_n, 2] tensor; `ln=(stats, colsum, eps)` consumes the moments of `a`'s rows: the product is then
    rstd * (a . w - mean * colsum) + bias, i.e. LayerNorm(a) . W^T when w = W * gamma and bias carries beta . W^T.
    `col_stats=True` (a 1x1 projection over an NHWC activation): returns (out, ColStats | None) for the consuming GroupNorm.
    fp32 `a` / `w` (reference-precision mode) run on the fp32-input MFMA; LayerNorm folding does not exist there.
    """
    if _is32(a):
        if ln is not None or row_stats:
            raise ValueError("gemm: LayerNorm folding / row statistics exist only on the fp16 path")
        if geglu and _F32_CONTRACT == "x3" and residual is None and rowvec is None and w.shape[0] % 16 == 0 and X3_FUSE_GEGLU:
            o = _gemm_f32(a, w, bias=bias, out=out, out_scale=out_scale, geglu=True)       # GEGLU in the GEMM's epilogue
            return (o, None) if col_stats else o
        o = _gemm_f32(a, w, bias=bias, residual=residual, rowvec=rowvec, rows_per_batch=rows_per_batch,
                      out=None if geglu else out, out_scale=out_scale)
        if geglu:
            o = geglu_il(o, out=out)
        return (o, None) if col_stats else o
    lib = load()
    M, K, lda = _rows_ld(a, "a")
    _dev16(w, "w")
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError(f"gemm: K mismatch {w.shape[1]} vs {K}")
    No_expect = N // 2 if geglu else N
    if out is None:
        out = torch.empty(*a.shape[:-1], No_expect, dtype=torch.float16, device=a.device)
    Mo, No, ldo = _rows_ld(out, "out")
    if (Mo, No) != (M, No_expect):
        raise ValueError("gemm: out shape mismatch")
    p = IefGemmParams()
    p.A, p.W, p.Out = a.data_ptr(), w.data_ptr(), out.data_ptr()
    p.bias = _ptr(_dev32(bias, "bias")) if bias is not None else None
    p.rowvec = _ptr(_dev32(rowvec, "rowvec")) if rowvec is not None else None
    if residual is not None:
        Mr, Nr, ldr = _rows_ld(residual, "residual")
        if (Mr, Nr) != (M, N):
            raise ValueError("gemm: residual shape mismatch")
        p.residual, p.ldr = residual.data_ptr(), ldr
    p.M, p.N, p.K = M, N, K
    p.lda, p.ldw, p.ldo = lda, w.stride(0), ldo
    p.rows_per_batch = rows_per_batch
    p.out_scale = out_scale
    if tile_hint == 0:
        if AUTOTUNE and f"gemm|{M}|{N}|{K}" not in _plan_table() and not _capturing() and _prof is None:
            wc = _cold_copies(w)
            autotune_plan("gemm", M, N, K, lambda t, sp, st, i: gemm(a, wc[i % len(wc)], bias=bias, residual=residual,
                                                                      rowvec=rowvec, rows_per_batch=rows_per_batch, out=out,
                                                                      out_scale=out_scale, tile_hint=t, splits=sp, stages=st,
                                                                      geglu=geglu))
            del wc
        p.tile_hint, p.splits, p.stages = pick_plan(M, N, K)
        if (geglu or ln is not None or row_stats) and p.splits > 1:
            p.tile_hint, p.splits, p.stages = heuristic_plan(M, N, K)[0], 1, 2
    else:
        p.tile_hint, p.splits, p.stages = tile_hint, max(1, splits), stages
    if p.splits > 1:
        ws = torch.empty(p.splits * M * N, dtype=torch.float32, device=a.device)
        p.ws = ws.data_ptr()
        nt, kb = _tile_count(lib, p.tile_hint, M, N, p.splits)
        p.cnt = _splitk_counters(a.device, nt, kb) if nt else None
    stats = None
    if row_stats:
        bn = lib.ief_gemm_tile_bn(p.tile_hint)
        stats = torch.empty(M, -(-N // bn), 2, dtype=torch.float32, device=a.device)
        p.rstat_out = stats.data_ptr()
    if ln is not None:
        st_in, colsum, eps = ln
        if _dev32(st_in, "ln stats").dim() != 3 or st_in.shape[0] != M or st_in.shape[2] != 2:
            raise ValueError("gemm: ln stats must be fp32 [M, slots, 2]")
        if _dev32(colsum, "colsum").numel() != N:
            raise ValueError("gemm: colsum must have N entries")
        p.rstat_in, p.rstat_slots, p.colsum, p.ln_eps = st_in.data_ptr(), st_in.shape[1], colsum.data_ptr(), eps
    p.flags, p.zeros = (3 if geglu else 1), _zeros(a.device)
    cst = None
    if col_stats and a.dim() == 4 and out.dim() == 4 and out.is_contiguous():       # a 1x1 projection over an NHWC activation
        cst = _attach_cstat(lib, p, out, M, N, a.shape[1] * a.shape[2])
    nbytes = 2.0 * (M * K + N * K + M * No_expect + (M * N if residual is not None else 0))
    with _Timed(_kname(p.tile_hint, False, p.stages) + (f" {M}x{N}x{K} s{p.splits}" if PROF_SHAPES else ""), 2.0 * M * N * K, nbytes):
        _check(lib.ief_gemm_f16(byref(p), 1, _stream()), "ief_gemm_f16")
    if col_stats:
        return out, cst
    return (out, stats) if row_stats else out


def conv3x3(x, w, bias=None, x2=None, stride=1, upsample=False, rowvec=None, residual=None, out=None, tile_hint=0,
            extra=None, splits=1, stages=2, pad_hi_only=False, col_stats=False):
    """3x3 / pad 1 convolution over NHWC fp16.  x [B,H,W,C1] (+ x2 [B,H,W,C2] channel-concat),
    w [Cout, 3, 3, C1+C2] fp16; `upsample` = nearest-2x of the input fused into the gather.
    extra=(e1, e2|None): fused 1x1 convolution over more NHWC sources sampled at the output pixel; w is
    then [Cout, 9*(C1+C2) + CE1 + CE2].  `col_stats=True`: returns (out, ColStats | None) for the consuming GroupNorm."""
    if _is32(x):
        o = _conv3x3_f32(x, w, bias, x2, stride, upsample, rowvec, residual, out, extra, pad_hi_only)
        return (o, None) if col_stats else o
    lib = load()
    _dev16(x, "x")
    _dev16(w, "w")
    if not x.is_contiguous() or (x2 is not None and not x2.is_contiguous()) or not w.is_contiguous():
        raise ValueError("conv3x3: x, x2, w must be contiguous")
    B, Hp, Wp, C1 = x.shape
    C2 = 0 if x2 is None else x2.shape[-1]
    if x2 is not None and tuple(x2.shape[:3]) != (B, Hp, Wp):
        raise ValueError("conv3x3: x2 spatial shape mismatch")
    Cout = w.shape[0]
    e1 = e2 = None
    CE1 = CE2 = 0
    if extra is not None:
        e1, e2 = extra
        CE1 = e1.shape[-1]
        CE2 = 0 if e2 is None else e2.shape[-1]
        for e in (e1, e2):
            if e is not None and (not _dev16(e, "extra").is_contiguous() or tuple(e.shape[:3]) != (B, Hp, Wp)):
                raise ValueError("conv3x3: extra sources must be contiguous NHWC at the output resolution")
        if tuple(w.shape) != (Cout, 9 * (C1 + C2) + CE1 + CE2):
            raise ValueError("conv3x3: fused weight must be [Cout, 9*(C1+C2)+CE1+CE2]")
    elif tuple(w.shape[1:]) != (3, 3, C1 + C2):
        raise ValueError(f"conv3x3: weight shape {tuple(w.shape)} does not match C1+C2={C1 + C2}")
    H, Wd = (Hp * 2, Wp * 2) if upsample else (Hp, Wp)
    pad_total = 1 if pad_hi_only else 2
    Ho, Wo = (H + pad_total - 3) // stride + 1, (Wd + pad_total - 3) // stride + 1
    if out is None:
        out = torch.empty(B, Ho, Wo, Cout, dtype=torch.float16, device=x.device)
    p = IefGemmParams()
    p.A, p.A2, p.W, p.Out = x.data_ptr(), _ptr(x2), w.data_ptr(), out.data_ptr()
    p.bias = _ptr(_dev32(bias, "bias")) if bias is not None else None
    if rowvec is not None:
        _dev32(rowvec, "rowvec")
        if rowvec.dim() != 2 or rowvec.shape[1] != Cout or rowvec.shape[0] not in (1, B):
            raise ValueError("conv3x3: rowvec must be [B, Cout] or [1, Cout]")
        p.rowvec = rowvec.data_ptr()
        p.rows_per_batch = Ho * Wo if rowvec.shape[0] == B and B > 1 else B * Ho * Wo
    if residual is not None:
        _dev16(residual, "residual")
        if tuple(residual.shape) != tuple(out.shape) or not residual.is_contiguous():
            raise ValueError("conv3x3: residual must match the output and be contiguous")
        p.residual, p.ldr = residual.data_ptr(), Cout
    p.N, p.ldo = Cout, Cout
    p.H, p.Wd, p.C1, p.C2 = H, Wd, C1, C2
    p.stride, p.ups, p.batch_images = stride, 1 if upsample else 0, B
    p.pad_hi_only = 1 if pad_hi_only else 0
    p.out_scale = 1.0
    M, K = B * Ho * Wo, 9 * (C1 + C2) + CE1 + CE2
    if tile_hint == 0:
        variant = "|u" if upsample else ""
        if AUTOTUNE and f"conv|{M}|{Cout}|{K}{variant}" not in _plan_table() and not _capturing() and _prof is None:
            wc = _cold_copies(w)
            autotune_plan("conv", M, Cout, K, lambda t, sp, st, i: conv3x3(x, wc[i % len(wc)], bias, x2=x2, stride=stride,
                                                                            upsample=upsample, rowvec=rowvec,
                                                                            residual=residual, out=out, tile_hint=t,
                                                                            splits=sp, extra=extra, stages=st,
                                                                            pad_hi_only=pad_hi_only), variant=variant)
            del wc
        p.tile_hint, p.splits, p.stages = pick_plan(M, Cout, K, conv=True, variant=variant)
        halo_ok = stride == 1 and not pad_hi_only and extra is None and Hp >= 2 and (
            (not upsample and 2 <= Wd <= 64) or (upsample and Wd <= 128 and (H * Wd) % 256 == 0 and 256 % Wd == 0))
        if halo_ok and HALO_HEURISTIC and f"conv|{M}|{Cout}|{K}{variant}" not in _plan_table() and (
                not upsample or f"conv|{M}|{Cout}|{K}" not in _plan_table()):
            # no measured plan for this shape: the halo kernel (256 x 80 tiles) beat the implicit GEMM on every plain 3x3
            # convolution that was tuned; cut K (whole 64-channel blocks) until ~256 workgroups exist
            tiles = -(-M // 256) * -(-Cout // 80)
            ncb = (C1 + C2) // 64
            sp = 1
            while tiles * sp < 192 and sp * 2 <= ncb and sp < 16:
                sp *= 2
            if tiles * sp >= 96:
                p.tile_hint, p.splits, p.stages = 15, sp, 4
        if p.tile_hint in _HALO_TILES and (not halo_ok or (upsample and p.tile_hint != 15) or p.splits > (C1 + C2) // 64):
            # the table is keyed by (M, N, K) alone: a convolution of another geometry that shares the key
            p.tile_hint, p.splits, p.stages = heuristic_plan(M, Cout, K) + (2,)
    else:
        p.tile_hint, p.splits, p.stages = tile_hint, max(1, splits), stages
    if p.splits > 1:
        ws = torch.empty(p.splits * M * Cout, dtype=torch.float32, device=x.device)
        p.ws = ws.data_ptr()
        nt, kb = _tile_count(lib, p.tile_hint, M, Cout, p.splits)
        p.cnt = _splitk_counters(x.device, nt, kb) if nt and p.tile_hint not in _HALO_TILES else None
    p.E1, p.E2, p.CE1, p.CE2 = _ptr(e1), _ptr(e2), CE1, CE2
    p.flags, p.zeros = 1, _zeros(x.device)
    cst = _attach_cstat(lib, p, out, M, Cout, Ho * Wo) if col_stats and out.is_contiguous() else None
    nbytes = 2.0 * (B * Hp * Wp * (C1 + C2) + M * (CE1 + CE2) + Cout * K + M * Cout * (2 if residual is not None else 1))
    with _Timed(_kname(p.tile_hint, True, p.stages) + (f" {M}x{Cout}x{K} s{p.splits}" if PROF_SHAPES else ""), 2.0 * M * Cout * K, nbytes):
        _check(lib.ief_conv3x3_f16(byref(p), _stream()), "ief_conv3x3_f16")
    return (out, cst) if col_stats else out


def _splits_f32(lib, p, M, N, K, device):
    """split-K for the fp32 GEMM when M x N alone gives too few 128-row tiles for the 256 CUs (the 16x16 / 8x8 levels):
    aim at ~512 workgroups, every slice keeping >= 4 K tiles of 32"""
    x3 = _F32_CONTRACT == "x3"
    nk = -(-K // 32)
    if x3:      # 128 x 80 tiles, two workgroups per CU: split below 1.5 per CU, towards 2 per CU (measured: without the split the
        # 32x32 / 16x16 levels run 1.5-2x slower than the reducer launches cost); 128 x 160 tiles, one 8-wave workgroup per CU:
        # split below one per two CUs, towards one per CU
        bn = lib.ief_gemm_x3_bn_k(1 if p.conv else 0, N, K)
        tiles = -(-M // lib.ief_gemm_x3_bm(M, N)) * -(-N // bn)
        below, target = (X3_SPLIT_BELOW_WIDE, X3_SPLIT_TARGET_WIDE) if bn == 160 else (X3_SPLIT_BELOW, X3_SPLIT_TARGET)
        if tiles >= below or nk < 8:
            return None
        splits = max(1, min(-(-target // tiles), nk // 4, 16))
    else:
        tiles = -(-M // 128) * -(-N // lib.ief_gemm_f32_bn(N))
        if tiles >= 256 or nk < 8:
            return None
        splits = max(1, min(-(-512 // tiles), nk // 4, 16))
    if splits <= 1:
        return None
    ws = torch.empty(splits * M * N, dtype=torch.float32, device=device)
    p.splits, p.ws = splits, ws.data_ptr()
    return ws


def _gemm_f32(a, w, bias=None, residual=None, rowvec=None, rows_per_batch=0, out=None, out_scale=1.0, transb=False, geglu=False):
    """fp32 out[..., n] = (a . w[n] + bias[n] + rowvec[row // rows_per_batch, n] + residual[..., n]) * out_scale;
    transb: w is [K, N] (rows along K) instead of [N, K]"""
    lib = load()
    M, K, lda = _rows_ld32(a, "a")
    _act32(w, "w")
    if w.dim() != 2:
        raise ValueError("gemm: w must be 2-D")
    N = w.shape[1] if transb else w.shape[0]
    if (w.shape[0] if transb else w.shape[1]) != K:
        raise ValueError(f"gemm: K mismatch {tuple(w.shape)} vs {K}")
    No_expect = N // 2 if geglu else N
    if out is None:
        out = torch.empty(*a.shape[:-1], No_expect, dtype=torch.float32, device=a.device)
    Mo, No, ldo = _rows_ld32(out, "out")
    if (Mo, No) != (M, No_expect):
        raise ValueError("gemm: out shape mismatch")
    p = IefGemmF32Params()
    p.geglu = 1 if geglu else 0
    p.A, p.W, p.Out = a.data_ptr(), w.data_ptr(), out.data_ptr()
    p.bias = _ptr(_dev32(bias, "bias")) if bias is not None else None
    if rowvec is not None:
        p.rowvec, p.rows_per_batch = _dev32(rowvec, "rowvec").data_ptr(), rows_per_batch
    if residual is not None:
        Mr, Nr, ldr = _rows_ld32(residual, "residual")
        if (Mr, Nr) != (M, N):
            raise ValueError("gemm: residual shape mismatch")
        p.residual, p.ldr = residual.data_ptr(), ldr
    p.M, p.N, p.K, p.lda, p.ldw, p.ldo = M, N, K, lda, w.stride(0), ldo
    p.out_scale, p.transb = out_scale, 1 if transb else 0
    ws = None if geglu else _splits_f32(lib, p, M, N, K, a.device)     # noqa: F841  (keeps the slabs alive until the launch is queued)
    kn = _set_x3(p, X3_SCALE_ACT, X3_SCALE_ACT if transb else X3_SCALE_W,        # transb: activation x activation
                 w if (not transb and w.is_contiguous()) else None)
    nbytes = 4.0 * (M * K + N * K + M * N * (2 if residual is not None else 1))
    with _Timed(f"{kn}<false, {'true' if transb else 'false'}>" + (f" {M}x{N}x{K} s{p.splits}" if PROF_SHAPES else ""), 2.0 * M * N * K, nbytes):
        _check(lib.ief_gemm_f32(byref(p), _stream()), "ief_gemm_f32")
    return out


def gemm_nt(a, b, out=None):
    """out = a @ b for a [M, K], b [K, N] (no transposed copy of b on the fp32 path)"""
    if _is32(a):
        return _gemm_f32(a, b, out=out, transb=True)
    return gemm(a, transpose(b.contiguous()), out=out)


def _conv3x3_f32(x, w, bias, x2, stride, upsample, rowvec, residual, out, extra, pad_hi_only):
    lib = load()
    _act32(x, "x"), _act32(w, "w")
    if not x.is_contiguous() or (x2 is not None and not _act32(x2, "x2").is_contiguous()) or not w.is_contiguous():
        raise ValueError("conv3x3: x, x2, w must be contiguous")
    B, Hp, Wp, C1 = x.shape
    C2 = 0 if x2 is None else x2.shape[-1]
    Cout = w.shape[0]
    e1 = e2 = None
    CE1 = CE2 = 0
    if extra is not None:
        e1, e2 = extra
        CE1 = _act32(e1, "extra").shape[-1]
        CE2 = 0 if e2 is None else _act32(e2, "extra").shape[-1]
        for e in (e1, e2):
            if e is not None and (not e.is_contiguous() or tuple(e.shape[:3]) != (B, Hp, Wp)):
                raise ValueError("conv3x3: extra sources must be contiguous NHWC at the output resolution")
        if tuple(w.shape) != (Cout, 9 * (C1 + C2) + CE1 + CE2):
            raise ValueError("conv3x3: fused weight must be [Cout, 9*(C1+C2)+CE1+CE2]")
    elif tuple(w.shape[1:]) != (3, 3, C1 + C2):
        raise ValueError(f"conv3x3: weight shape {tuple(w.shape)} does not match C1+C2={C1 + C2}")
    H, Wd = (Hp * 2, Wp * 2) if upsample else (Hp, Wp)
    pad_total = 1 if pad_hi_only else 2
    Ho, Wo = (H + pad_total - 3) // stride + 1, (Wd + pad_total - 3) // stride + 1
    if out is None:
        out = torch.empty(B, Ho, Wo, Cout, dtype=torch.float32, device=x.device)
    p = IefGemmF32Params()
    p.A, p.A2, p.W, p.Out = x.data_ptr(), _ptr(x2), w.data_ptr(), out.data_ptr()
    p.bias = _ptr(_dev32(bias, "bias")) if bias is not None else None
    if rowvec is not None:
        _dev32(rowvec, "rowvec")
        if rowvec.dim() != 2 or rowvec.shape[1] != Cout or rowvec.shape[0] not in (1, B):
            raise ValueError("conv3x3: rowvec must be [B, Cout] or [1, Cout]")
        p.rowvec = rowvec.data_ptr()
        p.rows_per_batch = Ho * Wo if rowvec.shape[0] == B and B > 1 else B * Ho * Wo
    if residual is not None:
        if tuple(_act32(residual, "residual").shape) != tuple(out.shape) or not residual.is_contiguous():
            raise ValueError("conv3x3: residual must match the output and be contiguous")
        p.residual, p.ldr = residual.data_ptr(), Cout
    M, K = B * Ho * Wo, 9 * (C1 + C2) + CE1 + CE2
    p.M, p.N, p.K, p.ldo, p.ldw = M, Cout, K, Cout, K
    p.conv, p.H, p.Wd, p.C1, p.C2, p.Ho, p.Wo = 1, H, Wd, C1, C2, Ho, Wo
    p.stride, p.ups, p.batch_images, p.pad_hi_only = stride, 1 if upsample else 0, B, 1 if pad_hi_only else 0
    p.E1, p.E2, p.CE1, p.CE2 = _ptr(e1), _ptr(e2), CE1, CE2
    p.out_scale = 1.0
    ws = _splits_f32(lib, p, M, Cout, K, x.device)  # noqa: F841
    kn = _set_x3(p, X3_SCALE_ACT, X3_SCALE_W, w)
    nbytes = 4.0 * (B * Hp * Wp * (C1 + C2) + M * (CE1 + CE2) + Cout * K + M * Cout * (2 if residual is not None else 1))
    with _Timed(f"{kn}<true, false>" + (f" {M}x{Cout}x{K} s{p.splits}" if PROF_SHAPES else ""), 2.0 * M * Cout * K, nbytes):
        _check(lib.ief_gemm_f32(byref(p), _stream()), "ief_gemm_f32 (conv)")
    return out


def conv3x3_shortcut(h, w_fused, bias_fused, x, skip=None, col_stats=False):
    """ResnetBlock2D tail with a channel-changing shortcut in ONE launch:
    conv3x3(h) + conv1x1([x | skip]) + (b2 + bs); w_fused [Cout, 9*Cout + Cin]."""
    return conv3x3(h, w_fused, bias_fused, extra=(x, skip), col_stats=col_stats)


def softmax_rows_(x):
    """in-place softmax over the last dim of a contiguous fp16 (or, reference-precision mode, fp32) tensor"""
    lib = load()
    if _is32(x):
        if not _act32(x, "x").is_contiguous():
            raise ValueError("softmax_rows_: x must be contiguous")
        _check(lib.ief_softmax_rows_f32(x.data_ptr(), x.numel() // x.shape[-1], x.shape[-1], _stream()), "ief_softmax_rows_f32")
        return x
    _dev16(x, "x")
    if not x.is_contiguous():
        raise ValueError("softmax_rows_: x must be contiguous")
    L = x.shape[-1]
    _check(lib.ief_softmax_rows_f16(x.data_ptr(), x.numel() // L, L, _stream()), "ief_softmax_rows_f16")
    return x


def transpose(x):
    """[R, C] fp16 -> [C, R]"""
    lib = load()
    _dev16(x, "x")
    if x.dim() != 2 or not x.is_contiguous():
        raise ValueError("transpose: contiguous 2-D tensor expected")
    out = torch.empty(x.shape[1], x.shape[0], dtype=torch.float16, device=x.device)
    _check(lib.ief_transpose_f16(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], _stream()), "ief_transpose_f16")
    return out


def pointwise_f32(x, w, bias=None):
    """1x1 conv on fp32 NCHW [B,Cin,H,W], w fp32 [Cout,Cin], Cin/Cout <= 8"""
    lib = load()
    _dev32(x, "x")
    _dev32(w, "w")
    B, Cin = x.shape[0], x.shape[1]
    Cout = w.shape[0]
    out = torch.empty(B, Cout, *x.shape[2:], dtype=torch.float32, device=x.device)
    _check(lib.ief_pointwise_f32(x.data_ptr(), w.data_ptr(), _ptr(bias), out.data_ptr(), B, Cin, Cout,
                                 x.numel() // (B * Cin), _stream()), "ief_pointwise_f32")
    return out


def add(a, b, out=None):
    lib = load()
    if _is32(a):
        _act32(a, "a"), _act32(b, "b")
        if a.shape != b.shape or not a.is_contiguous() or not b.is_contiguous():
            raise ValueError("add: operands must be contiguous and of equal shape")
        out = torch.empty_like(a) if out is None else out
        _check(lib.ief_add_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "ief_add_f32")
        return out
    _dev16(a, "a")
    _dev16(b, "b")
    if a.shape != b.shape or not a.is_contiguous() or not b.is_contiguous():
        raise ValueError("add: operands must be contiguous and of equal shape")
    if out is None:
        out = torch.empty_like(a)
    _check(lib.ief_add_f16(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "ief_add_f16")
    return out


def gather_rows(x, src):
    """out[b] = x[src[b]] along the batch dimension; src: device int32 [B]"""
    lib = load()
    if _is32(x):
        _devi32(src, "src")
        if not _act32(x, "x").is_contiguous() or src.numel() != x.shape[0]:
            raise ValueError("gather_rows: contiguous x [B, ...] and src [B] expected")
        out = torch.empty_like(x)
        _check(lib.ief_gather_rows_f32(x.data_ptr(), out.data_ptr(), src.data_ptr(), x.shape[0], x.numel() // x.shape[0],
                                       _stream()), "ief_gather_rows_f32")
        return out
    _dev16(x, "x")
    _devi32(src, "src")
    if not x.is_contiguous() or src.numel() != x.shape[0]:
        raise ValueError("gather_rows: contiguous x [B, ...] and src [B] expected")
    out = torch.empty_like(x)
    _check(lib.ief_gather_rows_f16(x.data_ptr(), out.data_ptr(), src.data_ptr(), x.shape[0], x.numel() // x.shape[0], _stream()),
           "ief_gather_rows_f16")
    return out


def conv_in(x, w, bias, out=None):
    """latent fp32 NCHW [B,Cin,H,W] -> fp16 NHWC [B,H,W,Cout]; w fp16 [3,3,Cin,Cout] (k-major).  fp32 w: fp32 NHWC out."""
    lib = load()
    _dev32(x, "x")
    if _is32(w):
        B, Cin, H, Wd = x.shape
        if tuple(w.shape[:3]) != (3, 3, Cin) or not _act32(w, "w").is_contiguous():
            raise ValueError("conv_in: weight must be contiguous [3, 3, Cin, Cout]")
        if out is None:
            out = torch.empty(B, H, Wd, w.shape[3], dtype=torch.float32, device=x.device)
        _check(lib.ief_conv_in_f32act(x.data_ptr(), w.data_ptr(), _ptr(bias), out.data_ptr(), B, Cin, H, Wd, w.shape[3],
                                      _stream()), "ief_conv_in_f32act")
        return out
    _dev16(w, "w")
    B, Cin, H, Wd = x.shape
    if tuple(w.shape[:3]) != (3, 3, Cin) or not w.is_contiguous():
        raise ValueError("conv_in: weight must be contiguous [3, 3, Cin, Cout]")
    Cout = w.shape[3]
    if out is None:
        out = torch.empty(B, H, Wd, Cout, dtype=torch.float16, device=x.device)
    _check(lib.ief_conv_in_f32(x.data_ptr(), w.data_ptr(), _ptr(bias), out.data_ptr(), B, Cin, H, Wd, Cout, _stream()),
           "ief_conv_in_f32")
    return out


def conv_out(x, w, bias, out=None):
    """fp16 NHWC [B,H,W,C] -> fp32 NCHW [B,Cout,H,W]; w fp16 [Cout,3,3,C] (both fp32 in the reference-precision mode)."""
    lib = load()
    if _is32(x):
        B, H, Wd, C = x.shape
        if not _act32(x, "x").is_contiguous() or not _act32(w, "w").is_contiguous() or tuple(w.shape[1:]) != (3, 3, C):
            raise ValueError("conv_out: contiguous NHWC x and [Cout, 3, 3, C] weight expected")
        if out is None:
            out = torch.empty(B, w.shape[0], H, Wd, dtype=torch.float32, device=x.device)
        _check(lib.ief_conv_out_f32act(x.data_ptr(), w.data_ptr(), _ptr(bias), out.data_ptr(), B, C, H, Wd, w.shape[0],
                                       _stream()), "ief_conv_out_f32act")
        return out
    _dev16(x, "x")
    _dev16(w, "w")
    B, H, Wd, C = x.shape
    Cout = w.shape[0]
    if out is None:
        out = torch.empty(B, Cout, H, Wd, dtype=torch.float32, device=x.device)
    _check(lib.ief_conv_out_f32(x.data_ptr(), w.data_ptr(), _ptr(bias), out.data_ptr(), B, C, H, Wd, Cout, _stream()),
           "ief_conv_out_f32")
    return out


# ------------------------------------------------------------------------------- norms
def groupnorm(x, gamma, beta, groups, eps, silu=False, x2=None, out=None, return_stats=False, cstat=None, cstat2=None):
    """GroupNorm over NHWC / tokens-major fp16 [B, ..., C] (+ optional channel-concat x2), optional SiLU.
    return_stats: also return the fp32 (mean, rstd) [B, groups, 2] the backward reuses.
    cstat / cstat2: the ColStats the launches that PRODUCED x / x2 returned (`col_stats=True`), or None."""
    lib = load()
    if _is32(x):
        if return_stats:        # the fp32 backward recomputes its (two-pass) statistics: nothing to hand over
            return groupnorm(x, gamma, beta, groups, eps, silu=silu, x2=x2, out=out), None
        if not _act32(x, "x").is_contiguous() or (x2 is not None and not _act32(x2, "x2").is_contiguous()):
            raise ValueError("groupnorm: inputs must be contiguous")
        B, C1 = x.shape[0], x.shape[-1]
        C2 = 0 if x2 is None else x2.shape[-1]
        HW = x.numel() // (B * C1)
        if out is None:
            out = torch.empty(*x.shape[:-1], C1 + C2, dtype=torch.float32, device=x.device)
        if GN3_F32 and C1 % 4 == 0 and C2 % 4 == 0:
            nws = lib.ief_groupnorm_f32_ws_floats(B, HW, C1 + C2)
            ws = torch.empty(nws, dtype=torch.float32, device=x.device)
            with _Timed("gn3_f32 (stats + finalize + apply)", 0.0, 12.0 * (x.numel() + (0 if x2 is None else x2.numel()))):
                _check(lib.ief_groupnorm_silu_f32_ws(x.data_ptr(), _ptr(x2), C1, C2, out.data_ptr(), _dev32(gamma, "gamma").data_ptr(),
                                                     _dev32(beta, "beta").data_ptr(), B, HW, groups, eps, 1 if silu else 0,
                                                     ws.data_ptr(), nws, _stream()), "ief_groupnorm_silu_f32_ws")
            return out
        with _Timed("groupnorm_f32_kernel", 0.0, 8.0 * (x.numel() + (0 if x2 is None else x2.numel()))):
            _check(lib.ief_groupnorm_silu_f32(x.data_ptr(), _ptr(x2), C1, C2, out.data_ptr(), _dev32(gamma, "gamma").data_ptr(),
                                              _dev32(beta, "beta").data_ptr(), B, HW, groups, eps, 1 if silu else 0, _stream()),
                   "ief_groupnorm_silu_f32")
        return out
    _dev16(x, "x")
    if not x.is_contiguous() or (x2 is not None and not x2.is_contiguous()):
        raise ValueError("groupnorm: inputs must be contiguous")
    B, C1 = x.shape[0], x.shape[-1]
    C2 = 0 if x2 is None else x2.shape[-1]
    HW = x.numel() // (B * C1)
    if out is None:
        out = torch.empty(*x.shape[:-1], C1 + C2, dtype=torch.float16, device=x.device)
    cs1, cs2 = cstat, None if x2 is None else cstat2
    for t, c, nm in ((x, cs1, "cstat"), (x2, cs2, "cstat2")):
        if c is not None and not c.describes(t):
            raise ValueError(f"groupnorm: {nm} was produced for another tensor (address / size mismatch)")
    if (GN_CSTAT and not return_stats and cs1 is not None and cs1.hw == HW
            and (x2 is None or (cs2 is not None and cs2.hw == HW))):
        # statistics were left by the producers' epilogues: fold them and apply — ONE launch while the fold is small
        # (2-8 tiles per image: the 32x32 / 16x16 levels), a fold launch + an apply launch above
        stats = torch.empty(B * groups * 2, dtype=torch.float32, device=x.device)
        with _Timed("groupnorm(stats+apply)", 0.0, 4.0 * x.numel() + (0 if x2 is None else 4.0 * x2.numel())):
            _check(lib.ief_groupnorm_cstat_f16(x.data_ptr(), _ptr(x2), C1, C2, out.data_ptr(), _dev32(gamma, "gamma").data_ptr(),
                                               _dev32(beta, "beta").data_ptr(), cs1.buf.data_ptr(), cs1.bm,
                                               None if cs2 is None else cs2.buf.data_ptr(), 0 if cs2 is None else cs2.bm,
                                               stats.data_ptr(), B, HW, groups, eps, 1 if silu else 0, _stream()),
                   "ief_groupnorm_cstat_f16")
        return out
    splits = lib.ief_gn_splits(HW)
    partial = torch.empty(B * (splits + 1) * groups * 2, dtype=torch.float32, device=x.device)
    with _Timed("groupnorm(stats+apply)", 0.0, 4.0 * x.numel() + (0 if x2 is None else 4.0 * x2.numel())):
        _check(lib.ief_groupnorm_silu_f16(x.data_ptr(), _ptr(x2), C1, C2, out.data_ptr(), _dev32(gamma, "gamma").data_ptr(),
                                          _dev32(beta, "beta").data_ptr(), partial.data_ptr(), B, HW, groups, eps,
                                          1 if silu else 0, _stream()), "ief_groupnorm_silu_f16")
    if return_stats:
        return out, partial[B * splits * groups * 2:].view(B, groups, 2)
    return out


def layernorm(x, gamma, beta, eps=1e-5, out=None):
    lib = load()
    if _is32(x):
        if not _act32(x, "x").is_contiguous():
            raise ValueError("layernorm: x must be contiguous")
        out = torch.empty_like(x) if out is None else out
        _check(lib.ief_layernorm_f32(x.data_ptr(), out.data_ptr(), _dev32(gamma, "gamma").data_ptr(),
                                     _dev32(beta, "beta").data_ptr(), x.numel() // x.shape[-1], x.shape[-1], eps, _stream()),
               "ief_layernorm_f32")
        return out
    _dev16(x, "x")
    if not x.is_contiguous():
        raise ValueError("layernorm: x must be contiguous")
    C = x.shape[-1]
    rows = x.numel() // C
    if out is None:
        out = torch.empty_like(x)
    _check(lib.ief_layernorm_f16(x.data_ptr(), out.data_ptr(), _dev32(gamma, "gamma").data_ptr(),
                                 _dev32(beta, "beta").data_ptr(), rows, C, eps, _stream()), "ief_layernorm_f16")
    return out


def geglu(x, out=None):
    lib = load()
    _dev16(x, "x")
    if not x.is_contiguous():
        raise ValueError("geglu: x must be contiguous")
    Ch = x.shape[-1] // 2
    rows = x.numel() // (2 * Ch)
    if out is None:
        out = torch.empty(*x.shape[:-1], Ch, dtype=torch.float16, device=x.device)
    _check(lib.ief_geglu_f16(x.data_ptr(), out.data_ptr(), rows, Ch, _stream()), "ief_geglu_f16")
    return out


# ------------------------------------------------------------------------------- attention
def _attn_common(p, q, k, v, out, heads):
    B, N, _ = q.shape
    L = k.shape[1]
    d = out.shape[-1] // heads
    for t, nm in ((q, "q"), (k, "k"), (v, "v"), (out, "out")):
        _dev16(t, nm)
        if t.dim() != 3:
            raise ValueError(f"{nm}: expected [B, rows, heads*d]")
        if t.shape[0] > 1 and t.stride(0) != t.shape[1] * t.stride(1):
            raise ValueError(f"{nm}: batch stride must equal rows * ld")
    p.Q, p.K, p.V, p.Out = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    p.B, p.heads, p.N, p.L, p.d = B, heads, N, L, d
    p.ldq, p.ldk, p.ldv, p.ldo = q.stride(1), k.stride(1), v.stride(1), out.stride(1)


_FLASH_VARIANT_ENV = int(os.environ.get("IEF_FLASH_VARIANT", "0"))      # read once: which flash kernel variant 0 means


def _batched32(p, t, heads, d, which):
    """strides of a [B, rows, heads*d] fp32 operand of a batched product: per batch row and per head"""
    if t.dim() != 3 or t.stride(2) != 1 or (t.shape[0] > 1 and t.stride(0) != t.shape[1] * t.stride(1)):
        raise ValueError(f"{which}: expected [B, rows, heads*d] with batch stride rows * ld")
    return t.stride(0), d, t.stride(1)


X3_SPLIT_BELOW = int(os.environ.get("IEF_X3_SPLIT_BELOW", "384"))      # split-K policy of the x3 GEMM (A/B runs)
X3_SPLIT_TARGET = int(os.environ.get("IEF_X3_SPLIT_TARGET", "512"))
X3_SPLIT_BELOW_WIDE = int(os.environ.get("IEF_X3_SPLIT_BELOW_WIDE", "129"))
X3_SPLIT_TARGET_WIDE = int(os.environ.get("IEF_X3_SPLIT_TARGET_WIDE", "256"))
GN3_F32 = os.environ.get("IEF_GN3_F32", "1") == "1"          # 0: the one-launch fp32 GroupNorm (A/B runs)
FLASH_F32 = os.environ.get("IEF_FLASH_F32", "1") == "1"      # 0: always materialise the fp32 maps (A/B runs)


def _attn_flash_f32(q, k, v, heads, scale, q_src=None, k_src=None, v_src=None, out=None, out_planes=False, lse=None):
    """fused fp32 attention (`ief_attn_flash_f32`): no map is written; None when the head dim has no instantiation.
    out_planes (split-operand mode only): the result leaves as operand planes for to_out's GEMM (`planes.Planes`);
    lse (split-operand mode only): fp32 [B, heads, N] receiving the row log-sum-exp (log2 units) for `ief_attn_bwd_x3`"""
    lib = load()
    B, N, C = q.shape
    L, d = k.shape[1], C // heads
    if not FLASH_F32 or d not in (32, 40, 64, 80, 160):
        return None
    p = IefAttnF32Params()
    p.Q, p.K, p.V = _act32(q, "q").data_ptr(), _act32(k, "k").data_ptr(), _act32(v, "v").data_ptr()
    op = None
    if out_planes:
        if _F32_CONTRACT != "x3":
            raise ValueError("attn_flash: operand planes exist in the split-operand mode only")
        from . import planes as _pl
        op, _ = _pl.attn_out_args(p, B, N, C, q.device)
    else:
        if out is None:
            out = torch.empty(B, N, C, dtype=torch.float32, device=q.device)
        p.Out = _act32(out, "out").data_ptr()
        p.sOb, _, p.ldo = _batched32(p, out, heads, d, "out")
    p.B, p.heads, p.N, p.L, p.d, p.scale = B, heads, N, L, d, scale
    p.sQb, _, p.ldq = _batched32(p, q, heads, d, "q")
    p.sKb, _, p.ldk = _batched32(p, k, heads, d, "k")
    p.sVb, _, p.ldv = _batched32(p, v, heads, d, "v")
    p.q_src, p.k_src, p.v_src = _ptr(_devi32(q_src, "q_src")), _ptr(_devi32(k_src, "k_src")), _ptr(_devi32(v_src, "v_src"))
    p.x3 = 1 if _F32_CONTRACT == "x3" else 0
    if lse is not None:
        if not p.x3 or tuple(_dev32(lse, "lse").shape) != (B, heads, N) or not lse.is_contiguous():
            raise ValueError("attn_flash: lse (contiguous fp32 [B, heads, N]) is written by the split-operand kernel only")
        p.lse = lse.data_ptr()
    with _Timed(f"attn_flash_{'x3' if p.x3 else 'f32'}_kernel<{d}>", 4.0 * B * heads * N * L * d, 4.0 * B * heads * d * (2 * N + 2 * L)):
        _check(lib.ief_attn_flash_f32(byref(p), _stream()), "ief_attn_flash_f32")
    return op if out_planes else out


def _attn_scores_f32(q, k, heads, scale, q_src=None, k_src=None, out=None, softmax=True):
    """materialised maps softmax(scale q k^T) as contiguous fp32 [B*heads, N, L] (`register.py:43-47`): one batched
    launch of the fp32 GEMM over (batch row, head) + a row softmax (softmax=False: the scaled scores themselves)"""
    lib = load()
    _act32(q, "q"), _act32(k, "k")
    B, N, C = q.shape
    L = k.shape[1]
    d = C // heads
    if out is None:
        out = torch.empty(B * heads, N, L, dtype=torch.float32, device=q.device)
    elif tuple(_act32(out, "out").shape) != (B * heads, N, L) or not out.is_contiguous():
        raise ValueError("attn_probs: out must be contiguous fp32 [B*heads, N, L]")
    p = IefGemmF32Params()
    p.A, p.W, p.Out = q.data_ptr(), k.data_ptr(), out.data_ptr()
    p.M, p.N, p.K = N, L, d
    p.sAb, p.sAh, p.lda = _batched32(p, q, heads, d, "q")
    p.sWb, p.sWh, p.ldw = _batched32(p, k, heads, d, "k")
    p.sOb, p.sOh, p.ldo = heads * N * L, N * L, L
    p.batch, p.heads, p.out_scale = B, heads, scale
    p.a_src, p.w_src = _ptr(_devi32(q_src, "q_src")), _ptr(_devi32(k_src, "k_src"))
    kn = _set_x3(p, X3_SCALE_ACT, X3_SCALE_ACT)
    with _Timed(f"{kn}<scores {d}>", 2.0 * B * heads * N * L * d, 4.0 * B * heads * (N * d + L * d + N * L)):
        _check(lib.ief_gemm_f32(byref(p), _stream()), "ief_gemm_f32 (attention scores)")
    if softmax:
        with _Timed("softmax_rows_f32_kernel", 0.0, 8.0 * out.numel()):
            _check(lib.ief_softmax_rows_f32(out.data_ptr(), out.numel() // L, L, _stream()), "ief_softmax_rows_f32")
    return out


def attn_scores(q, k, heads, scale):
    """-> (sim, attn): the PRE-softmax scores scale * q k^T and their row softmax, both contiguous [B*heads, N, L] in q's
    dtype — the two tensors MasaCtrl's editor protocol hands to user editors
    (`/root/reference/masactrl/model/register.py:35,44`).  Generic-path only: computed on the fp32 MFMA / fp32 softmax
    whatever the model's storage dtype (any key count, e.g. 77)."""
    half = not _is32(q)
    if half:
        q, k = to_f32(_dev16(q, "q")), to_f32(_dev16(k, "k"))
    sim = _attn_scores_f32(q, k, heads, scale, softmax=False)
    probs = softmax_rows_(sim.clone())
    return (to_f16(sim), to_f16(probs)) if half else (sim, probs)


def _attn_apply_f32(probs, v, heads, v_src=None, out=None, map_scale=None):
    """out [B, N, heads*d] = probs [B*heads, N, L] @ v [B, L, heads*d] per (batch row, head), fp32.
    map_scale: split scale of the maps (default 2^14: entries <= 1; edited maps can exceed 1, `map_split_scale`)"""
    lib = load()
    _act32(probs, "probs"), _act32(v, "v")
    if not probs.is_contiguous():
        raise ValueError("attn_apply: probs must be contiguous")
    B = v.shape[0]
    N, L = probs.shape[1], probs.shape[2]
    d = v.shape[2] // heads
    if probs.shape[0] != B * heads or v.shape[1] != L:
        raise ValueError("attn_apply: shape mismatch")
    if out is None:
        out = torch.empty(B, N, v.shape[2], dtype=torch.float32, device=v.device)
    p = IefGemmF32Params()
    p.A, p.W, p.Out = probs.data_ptr(), v.data_ptr(), out.data_ptr()
    p.M, p.N, p.K = N, d, L
    p.sAb, p.sAh, p.lda = heads * N * L, N * L, L
    p.sWb, p.sWh, p.ldw = _batched32(p, v, heads, d, "v")
    p.sOb, p.sOh, p.ldo = _batched32(p, _act32(out, "out"), heads, d, "out")
    p.batch, p.heads, p.out_scale, p.transb = B, heads, 1.0, 1
    p.w_src = _ptr(_devi32(v_src, "v_src"))
    kn = _set_x3(p, X3_SCALE_PROB if map_scale is None else map_scale, X3_SCALE_ACT)    # maps <= 1: a large scale keeps the lo halves of small entries normal
    with _Timed(f"{kn}<apply {d}>", 2.0 * B * heads * N * L * d, 4.0 * B * heads * (N * d + L * d + N * L)):
        _check(lib.ief_gemm_f32(byref(p), _stream()), "ief_gemm_f32 (attention apply)")
    return out


def p2p_cross_edit_(probs, B, heads, edit_src, edit_slot, mt32, coef):
    """in place on fp32 maps [B*heads, N, L]: P'[w] = c1[w] (P_src M)[w] + c2[w] P_tgt[w] for the rows edit_src marks"""
    lib = load()
    _act32(probs, "probs"), _dev32(mt32, "mt32"), _dev32(coef, "coef")
    if tuple(mt32.shape[-2:]) != (96, 96) or coef.shape[-1] != 96 or not probs.is_contiguous():
        raise ValueError("p2p_cross_edit_: mt32 must be [slots,96,96] fp32, coef [slots,2,96] fp32, probs contiguous")
    _check(lib.ief_p2p_cross_edit_f32(probs.data_ptr(), _devi32(edit_src, "edit_src").data_ptr(),
                                      _devi32(edit_slot, "edit_slot").data_ptr(), mt32.data_ptr(), coef.data_ptr(), B, heads,
                                      probs.shape[1], probs.shape[2], _stream()), "ief_p2p_cross_edit_f32")
    return probs


def attn_flash(q, k, v, heads, scale, q_src=None, k_src=None, v_src=None, out=None, lse=None, variant=0, out_planes=False):
    """out[b] = softmax(q[q_src[b]] k[k_src[b]]^T * scale) v[v_src[b]]; q [B,N,h*d], k/v [B,L,h*d] (strided views ok).
    lse: optional fp32 [B, heads, N] receiving the row log-sum-exp (log2 units) that `attn_bwd` consumes.
    fp32 operands (reference-precision mode): the maps are materialised in HBM, as the reference does."""
    if _is32(q):
        if lse is not None and not x3_fused_bwd_ok(q.shape[2] // heads, k.shape[1]):
            raise ValueError("attn_flash: on fp32 operands lse exists only where ief_attn_bwd_x3 consumes it (x3_fused_bwd_ok); "
                             "elsewhere the fp32 backward recomputes the maps")
        o = _attn_flash_f32(q, k, v, heads, scale, q_src, k_src, v_src, out, out_planes=out_planes, lse=lse)
        if o is not None:
            return o
        if lse is not None:
            raise ValueError("attn_flash: the fused fp32 kernel is switched off (IEF_FLASH_F32=0): no lse")
        o = _attn_apply_f32(_attn_scores_f32(q, k, heads, scale, q_src, k_src), v, heads, v_src, out)
        if out_planes:
            from . import planes as _pl
            return _pl.split(o)
        return o
    lib = load()
    if out is None:
        out = torch.empty(q.shape[0], q.shape[1], q.shape[2], dtype=torch.float16, device=q.device)
    p = IefAttnParams()
    _attn_common(p, q, k, v, out, heads)
    p.scale = scale
    p.variant = variant if variant else _FLASH_VARIANT_ENV                              # env: A/B runs only
    if lse is not None:
        if tuple(_dev32(lse, "lse").shape) != (q.shape[0], heads, q.shape[1]):
            raise ValueError("attn_flash: lse must be fp32 [B, heads, N]")
        p.lse = lse.data_ptr()
    p.q_src, p.k_src, p.v_src = _ptr(_devi32(q_src, "q_src")), _ptr(_devi32(k_src, "k_src")), _ptr(_devi32(v_src, "v_src"))
    kern = {1: "attn_flash_kernel", 2: "attn_flash_pp_kernel"}.get(p.variant, "attn_flash_sp_kernel")
    with _Timed(f"{kern}<{p.d}>", 4.0 * p.B * p.heads * p.N * p.L * p.d, 2.0 * p.B * p.heads * p.d * (2 * p.N + 2 * p.L)):
        _check(lib.ief_attn_flash_f16(byref(p), _stream()), "ief_attn_flash_f16")
    return out


X3_FUSE_CROSS = os.environ.get("IEF_X3_FUSE_CROSS", "1") == "1"      # 0: materialised cross maps (scores, softmax, edit, apply: A/B runs)


def map_split_scale(coef_bound: float) -> float:
    """power-of-two scale for the hi / lo split of EDITED maps P' = c1 T + c2 P whose entries reach `coef_bound` = max_w (|c1| +
    |c2|) (maps <= 1 otherwise): the largest 2^k <= 2^14 with 2^k * coef_bound inside the fp16 range.  AttentionReweight lowers to
    c1 = alpha * equalizer (`/root/reference/p2p/model/attention_control.py:42-46`): with the fixed 2^14 an equalizer of 5 on a
    peaky source map overflowed to inf / NaN"""
    b = max(1.0, float(coef_bound))
    s = X3_SCALE_PROB
    while s > 1.0 and s * b > 32768.0:        # a factor 2 of head-room below 65504
        s *= 0.5
    return s


def _attn_cross_p2p_x3(q, k, v, heads, scale, edit_src, edit_slot, mt32, coef, out=None, out_planes=False, coef_bound=1.0):
    """`ief_attn_cross_p2p_f32`: the edited cross-attention layer of the f16x3 mode in one launch (maps stay in registers).
    coef_bound: max_w (|c1| + |c2|) over the plan's coefficient tables (sizes the split of the edited maps)"""
    lib = load()
    B, N, C = q.shape
    L, d = k.shape[1], C // heads
    _dev32(mt32, "mt32"), _dev32(coef, "coef")
    if tuple(mt32.shape[-2:]) != (96, 96) or coef.shape[-1] != 96 or not mt32.is_contiguous() or not coef.is_contiguous():
        raise ValueError("attn_cross_p2p: mt must be contiguous [slots,96,96] fp32, coef [slots,2,96] fp32")
    p = IefAttnF32Params()
    p.Q, p.K, p.V = _act32(q, "q").data_ptr(), _act32(k, "k").data_ptr(), _act32(v, "v").data_ptr()
    op = None
    if out_planes:
        from . import planes as _pl
        op, _ = _pl.attn_out_args(p, B, N, C, q.device)
    else:
        if out is None:
            out = torch.empty(B, N, C, dtype=torch.float32, device=q.device)
        p.Out = _act32(out, "out").data_ptr()
        p.sOb, _, p.ldo = _batched32(p, out, heads, d, "out")
    p.B, p.heads, p.N, p.L, p.d, p.scale = B, heads, N, L, d, scale
    p.sQb, _, p.ldq = _batched32(p, q, heads, d, "q")
    p.sKb, _, p.ldk = _batched32(p, k, heads, d, "k")
    p.sVb, _, p.ldv = _batched32(p, v, heads, d, "v")
    p.p_scale = map_split_scale(coef_bound)
    p.x3 = 1
    nedit = 2.0 * B * heads * N * 96 * 96
    with _Timed(f"attn_cross_p2p_x3_kernel<{d}>", 4.0 * B * heads * N * L * d + nedit, 4.0 * B * heads * d * (2 * N + 2 * L)):
        _check(lib.ief_attn_cross_p2p_f32(byref(p), _devi32(edit_src, "edit_src").data_ptr(), _devi32(edit_slot, "edit_slot").data_ptr(),
                                          mt32.data_ptr(), coef.data_ptr(), _stream()), "ief_attn_cross_p2p_f32")
    return op if out_planes else out


def attn_cross_p2p(q, k, v, heads, scale, edit_src=None, edit_slot=None, mt=None, coef=None, out=None, out_planes=False,
                   coef_bound=1.0):
    """Cross-attention (<= 96 keys) with the fused Prompt-to-Prompt map edit (see include/ief_hip.h).
    fp32 operands: materialised maps, `mt` must then be the fp32 table.  out_planes (split-operand mode): the result as operand
    planes for to_out's GEMM; coef_bound: max (|c1| + |c2|) of the plan (sizes the split of the edited maps)."""
    if _is32(q):
        if edit_src is None:
            o = _attn_flash_f32(q, k, v, heads, scale, out=out, out_planes=out_planes)
            if o is not None:
                return o
        elif _F32_CONTRACT == "x3" and X3_FUSE_CROSS and k.shape[1] <= 96 and q.shape[2] // heads in (40, 64, 80, 160):
            return _attn_cross_p2p_x3(q, k, v, heads, scale, edit_src, edit_slot, mt, coef, out, out_planes, coef_bound)
        probs = _attn_scores_f32(q, k, heads, scale)
        if edit_src is not None:
            p2p_cross_edit_(probs, q.shape[0], heads, edit_src, edit_slot, mt, coef)
        o = _attn_apply_f32(probs, v, heads, None, out, map_scale=map_split_scale(coef_bound) if edit_src is not None else None)
        if out_planes:
            from . import planes as _pl
            return _pl.split(o)
        return o
    lib = load()
    if out is None:
        out = torch.empty(q.shape[0], q.shape[1], q.shape[2], dtype=torch.float16, device=q.device)
    p = IefCrossParams()
    _attn_common(p, q, k, v, out, heads)
    p.scale = scale
    p.edit_src, p.edit_slot = _ptr(_devi32(edit_src, "edit_src")), _ptr(_devi32(edit_slot, "edit_slot"))
    if edit_src is not None:
        _dev16(mt, "mt")
        _dev32(coef, "coef")
        if tuple(mt.shape[-2:]) != (96, 96) or not mt.is_contiguous() or coef.shape[-1] != 96:
            raise ValueError("attn_cross_p2p: mt must be [slots,96,96] fp16, coef [slots,2,96] fp32")
        p.MT, p.coef = mt.data_ptr(), coef.data_ptr()
    with _Timed(f"attn_cross_p2p_kernel<{p.d}>", 4.0 * p.B * p.heads * p.N * p.L * p.d,
                2.0 * p.B * p.heads * p.d * (2 * p.N + 2 * p.L)):
        _check(lib.ief_attn_cross_p2p_f16(byref(p), _stream()), "ief_attn_cross_p2p_f16")
    return out


def attn_probs(q, k, heads, scale, out=None):
    """materialised maps [B*heads, N, L] fp16 for the generic controller path (out: where to write them)."""
    if _is32(q):
        return _attn_scores_f32(q, k, heads, scale, out=out)
    lib = load()
    B, N, C = q.shape
    L = k.shape[1]
    if out is None:
        probs = torch.empty(B * heads, N, L, dtype=torch.float16, device=q.device)
    else:
        probs = _dev16(out, "out")
        if tuple(probs.shape) != (B * heads, N, L) or not probs.is_contiguous():
            raise ValueError("attn_probs: out must be contiguous fp16 [B*heads, N, L]")
    p = IefAttnParams()
    _attn_common(p, q, k, k, q, heads)
    p.scale = scale
    _check(lib.ief_attn_probs_f16(byref(p), probs.data_ptr(), _stream()), "ief_attn_probs_f16")
    return probs


def attn_apply(probs, v, heads, out=None):
    """out [B,N,h*d] = probs [B*heads,N,L] @ v [B,L,h*d]."""
    if _is32(probs):
        return _attn_apply_f32(probs, v, heads, None, out)
    lib = load()
    _dev16(probs, "probs")
    if not probs.is_contiguous():
        raise ValueError("attn_apply: probs must be contiguous")
    B = v.shape[0]
    N, L = probs.shape[1], probs.shape[2]
    if probs.shape[0] != B * heads or v.shape[1] != L:
        raise ValueError("attn_apply: shape mismatch")
    if out is None:
        out = torch.empty(B, N, v.shape[2], dtype=torch.float16, device=v.device)
    p = IefAttnParams()
    _attn_common(p, out, v, v, out, heads)
    p.N, p.L = N, L
    _check(lib.ief_attn_apply_f16(byref(p), probs.data_ptr(), _stream()), "ief_attn_apply_f16")
    return out


# ------------------------------------------------------------------------------- sampler
def cfg_ddim_step(eps_u, eps_c, x, coef, out=None, x0_out=None):
    """x' from (eps_u, eps_c, x) with coef = device fp32 [a_from, a_to, guidance]; eps_u None => no CFG."""
    lib = load()
    _dev32(eps_c, "eps_c")
    _dev32(x, "x")
    _dev32(coef, "coef")
    if out is None:
        out = torch.empty_like(x)
    _check(lib.ief_cfg_ddim_step_f32(_ptr(eps_u), eps_c.data_ptr(), x.data_ptr(), out.data_ptr(), _ptr(x0_out),
                                     coef.data_ptr(), x.numel(), _stream()), "ief_cfg_ddim_step_f32")
    return out


def ddim_step(eps, x, a_from: float, a_to: float):
    """scheduler.step on device tensors: returns (x_prev, x0)."""
    coef = torch.tensor([a_from, a_to, 1.0], dtype=torch.float32, device=x.device)
    x0 = torch.empty_like(x, dtype=torch.float32)
    xf = x.float().contiguous()
    out = cfg_ddim_step(None, eps.float().contiguous(), xf, coef, x0_out=x0)
    return out, x0


def timestep_embedding(t, dim, dtype=torch.float16):
    lib = load()
    _dev32(t, "t")
    if dtype == torch.float32:
        out = torch.empty(t.shape[0], dim, dtype=torch.float32, device=t.device)
        _check(lib.ief_timestep_embedding_f32(t.data_ptr(), out.data_ptr(), t.shape[0], dim, _stream()),
               "ief_timestep_embedding_f32")
        return out
    out = torch.empty(t.shape[0], dim, dtype=torch.float16, device=t.device)
    _check(lib.ief_timestep_embedding_f16(t.data_ptr(), out.data_ptr(), t.shape[0], dim, _stream()),
           "ief_timestep_embedding_f16")
    return out


def silu(x):
    lib = load()
    if _is32(x):
        out = torch.empty_like(_act32(x, "x"))
        _check(lib.ief_silu_f32(x.contiguous().data_ptr(), out.data_ptr(), x.numel(), _stream()), "ief_silu_f32")
        return out
    _dev16(x, "x")
    out = torch.empty_like(x)
    _check(lib.ief_silu_f16(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "ief_silu_f16")
    return out


def to_f16(x, out=None):
    lib = load()
    _dev32(x, "x")
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    _check(lib.ief_cast_f32_to_f16(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "ief_cast_f32_to_f16")
    return out


def to_f32(x, out=None):
    lib = load()
    _dev16(x, "x")
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _check(lib.ief_cast_f16_to_f32(x.contiguous().data_ptr(), out.data_ptr(), x.numel(), _stream()), "ief_cast_f16_to_f32")
    return out


def image_u8(x):
    """decoder output fp32 NCHW [B,C,H,W] in [-1, 1] -> uint8 NHWC [B,H,W,C] on the device:
    (x / 2 + 0.5).clamp(0, 1) * 255, truncated (`/root/reference/p2p/model/sd_utils.py:85-88`)"""
    lib = load()
    _dev32(x, "x")
    B, C, H, Wd = x.shape
    out = torch.empty(B, H, Wd, C, dtype=torch.uint8, device=x.device)
    _check(lib.ief_image_u8(x.data_ptr(), out.data_ptr(), B, C, H, Wd, _stream()), "ief_image_u8")
    return out


def select_step(table, out, step):
    """out[...] = table[clamp(step[0], 0, rows - 1), ...] inside the stream (graph-capturable); step: device int32 [1]."""
    lib = load()
    nbytes = out.numel() * out.element_size()
    if table[0].numel() * table.element_size() != nbytes or not table.is_contiguous() or not out.is_contiguous():
        raise ValueError("select_step: table[step] and out must have identical contiguous byte size")
    _check(lib.ief_select_step(table.data_ptr(), out.data_ptr(), _devi32(step, "step").data_ptr(), nbytes, table.shape[0],
                               _stream()), "ief_select_step")
    return out


def advance_step(step):
    lib = load()
    _check(lib.ief_advance_step(_devi32(step, "step").data_ptr(), _stream()), "ief_advance_step")


# ------------------------------------------------------------------------------- activation gradients (null-text inversion)
def map_loss_blocks(N, d):
    """loss partials per (batch, head) that `attn_map_loss_bwd` writes"""
    return load().ief_map_loss_blocks(int(N), int(d))


def attn_map_loss_bwd(q, k, ref, dq, heads, scale, gcoef, accumulate=True, loss=None, loss_coef=1.0):
    """Pix2Pix-zero map objective of one cross-attention module (include/ief_hip.h, IefMapLossParams): with
    P = softmax(scale q k^T) and e = P - ref,  dq (+)= gcoef * scale * (P * (e - sum_j e_j P_j)) k.
    q [B,N,h*d], k [B,L,h*d] (strided views ok), ref fp16 [B*heads,N,L];
    loss: optional fp32 [B*heads*map_loss_blocks(N, d)] partials of sum e^2 (each times loss_coef)."""
    lib = load()
    if _is32(q):
        return _attn_map_loss_bwd_f32(q, k, ref, dq, heads, scale, gcoef, accumulate, loss, loss_coef)
    _dev16(q, "q"), _dev16(k, "k"), _dev16(ref, "ref"), _dev16(dq, "dq")
    B, N, C = q.shape
    L = k.shape[1]
    d = C // heads
    if tuple(ref.shape) != (B * heads, N, L) or not ref.is_contiguous():
        raise ValueError("attn_map_loss_bwd: ref must be contiguous fp16 [B*heads, N, L]")
    if tuple(dq.shape) != (B, N, C) or k.shape[0] != B or k.shape[2] != C:
        raise ValueError("attn_map_loss_bwd: shape mismatch")
    for nm, t in (("q", q), ("k", k), ("dq", dq)):
        if t.stride(2) != 1 or (t.shape[0] > 1 and t.stride(0) != t.shape[1] * t.stride(1)):
            raise ValueError(f"attn_map_loss_bwd: {nm} must be [B, rows, h*d] with batch stride rows * ld")
    p = IefMapLossParams()
    p.Q, p.K, p.ref, p.dQ = q.data_ptr(), k.data_ptr(), ref.data_ptr(), dq.data_ptr()
    p.B, p.heads, p.N, p.L, p.d = B, heads, N, L, d
    p.ldq, p.ldk, p.lddq = q.stride(1), k.stride(1), dq.stride(1)
    p.scale, p.gcoef, p.loss_coef, p.accumulate = scale, gcoef, loss_coef, 1 if accumulate else 0
    if loss is not None:
        if _dev32(loss, "loss").numel() < B * heads * lib.ief_map_loss_blocks(N, d):
            raise ValueError("attn_map_loss_bwd: loss needs B*heads*map_loss_blocks(N, d) floats")
        p.loss = loss.data_ptr()
    _check(lib.ief_attn_map_loss_bwd_f16(byref(p), _stream()), "ief_attn_map_loss_bwd_f16")
    return dq



# ------------------------------------------------------------------------------- attention gradients, fp32-storage modes
def _transpose_maps_f32(x):
    """[R, N, L] fp32 -> contiguous [R, L, N]"""
    lib = load()
    R, N, L = x.shape
    out = torch.empty(R, L, N, dtype=torch.float32, device=x.device)
    _check(lib.ief_transpose_batched_f32(_act32(x, "maps").data_ptr(), out.data_ptr(), R, N, L, _stream()), "ief_transpose_batched_f32")
    return out


def _attn_apply_t_f32(maps, x, heads, out=None):
    """out [B, L, heads*d] = maps[b, h]^T [L x N] @ x[b, h] [N x d] per (batch row, head) WITHOUT transposing the maps: the product
    is taken as (x^T maps)^T -- x^T [d x N] is the small A operand, the maps [N x L] are read in place as the [K][N] operand,
    and only the [d x L] result is transposed back (the explicit [R, N, L] -> [R, L, N] copy moved 2 x 537 MB per 64x64
    self-attention layer and gradient)"""
    lib = load()
    _act32(maps, "maps"), _act32(x, "x")
    B, N, Cx = x.shape
    d = Cx // heads
    L = maps.shape[2]
    if tuple(maps.shape) != (B * heads, N, L) or not maps.is_contiguous():
        raise ValueError("attn_apply_t: maps must be contiguous [B*heads, N, L]")
    xt = x.reshape(B, N, heads, d).permute(0, 2, 3, 1).contiguous()                    # [B, heads, d, N]
    tmp = torch.empty(B * heads, d, L, dtype=torch.float32, device=x.device)
    p = IefGemmF32Params()
    p.A, p.W, p.Out = xt.data_ptr(), maps.data_ptr(), tmp.data_ptr()
    p.M, p.N, p.K = d, L, N
    p.sAb, p.sAh, p.lda = heads * d * N, d * N, N
    p.sWb, p.sWh, p.ldw = heads * N * L, N * L, L
    p.sOb, p.sOh, p.ldo = heads * d * L, d * L, L
    p.batch, p.heads, p.out_scale, p.transb = B, heads, 1.0, 1
    kn = _set_x3(p, X3_SCALE_ACT, X3_SCALE_PROB)
    with _Timed(f"{kn}<apply^T {d}>", 2.0 * B * heads * N * L * d, 4.0 * B * heads * (N * d + L * d + N * L)):
        _check(lib.ief_gemm_f32(byref(p), _stream()), "ief_gemm_f32 (attention apply, transposed maps)")
    res = tmp.reshape(B, heads, d, L).permute(0, 3, 1, 2).reshape(B, L, heads * d)
    if out is None:
        return res.contiguous()
    out.copy_(res)
    return out


def _softmax_bwd_f32_(probs, dprobs, scale):
    """in place on dprobs: dS = scale * P o (dP - rowsum(dP o P))"""
    lib = load()
    L = probs.shape[-1]
    _check(lib.ief_softmax_bwd_rows_f32(probs.data_ptr(), dprobs.data_ptr(), probs.numel() // L, L, float(scale), _stream()),
           "ief_softmax_bwd_rows_f32")
    return dprobs


X3_FUSED_BWD = os.environ.get("IEF_X3_FUSED_BWD", "1") == "1"      # 0: the reverse pass keeps the materialised maps everywhere (A/B runs)


def x3_fused_bwd_ok(d: int, L: int) -> bool:
    """the fp32-storage reverse pass differentiates this attention layer with `ief_attn_bwd_x3` (P and dS recomputed per tile from
    the forward's lse, no map in HBM): split-operand mode, head dims 40 / 64 (the levels whose maps are hundreds of MB), at
    least 128 keys (self-attention; the 77-key cross maps are small and keep the materialised path)"""
    return X3_FUSED_BWD and _F32_CONTRACT == "x3" and FLASH_F32 and d in (40, 64) and L >= 128


def _attn_bwd_x3(q, k, v, o, do, lse, heads, scale, dq, dk, dv, want_dq, want_dkv):
    lib = load()
    B, N, C = q.shape
    L, d = k.shape[1], C // heads
    for t, nm in ((q, "q"), (k, "k"), (v, "v"), (o, "o"), (do, "do")):
        _act32(t, nm)
        if t.dim() != 3 or t.stride(2) != 1 or (t.shape[0] > 1 and t.stride(0) != t.shape[1] * t.stride(1)):
            raise ValueError(f"{nm}: expected [B, rows, heads*d] with unit channel stride and batch stride rows * ld")
    delta = torch.empty(B, heads, N, dtype=torch.float32, device=q.device)
    _check(lib.ief_attn_bwd_delta_f32in(o.data_ptr(), do.data_ptr(), delta.data_ptr(), B, heads, N, d, o.stride(1), do.stride(1),
                                        _stream()), "ief_attn_bwd_delta_f32in")
    p = IefAttnBwdF32Params()
    p.Q, p.K, p.V, p.dO, p.lse, p.delta = q.data_ptr(), k.data_ptr(), v.data_ptr(), do.data_ptr(), _dev32(lse, "lse").data_ptr(), delta.data_ptr()
    p.B, p.heads, p.N, p.L, p.d = B, heads, N, L, d
    p.ldq, p.ldk, p.ldv, p.ldo = q.stride(1), k.stride(1), v.stride(1), do.stride(1)
    p.scale, p.ds_mul = scale, X3_SCALE_PROB
    what = 0

    def grad_like(t, g, nm):
        if g is None:
            g = torch.empty(t.shape, dtype=torch.float32, device=t.device)
        _act32(g, nm)
        if tuple(g.shape) != tuple(t.shape) or g.stride(2) != 1 or (g.shape[0] > 1 and g.stride(0) != g.shape[1] * g.stride(1)):
            raise ValueError(f"{nm}: expected the shape of its operand, unit channel stride, batch stride rows * ld")
        return g
    if want_dq:
        dq = grad_like(q, dq, "dq")
        p.dQ, p.lddq = dq.data_ptr(), dq.stride(1)
        what |= 1
    if want_dkv:
        dk, dv = grad_like(k, dk, "dk"), grad_like(v, dv, "dv")
        p.dK, p.dV, p.lddk, p.lddv = dk.data_ptr(), dv.data_ptr(), dk.stride(1), dv.stride(1)
        what |= 2
    fl = (6.0 if want_dq else 0.0) + (8.0 if want_dkv else 0.0)
    with _Timed(f"attn_bwd_x3_kernel<{d}>", fl * B * heads * N * L * d):
        _check(lib.ief_attn_bwd_x3(byref(p), what, _stream()), "ief_attn_bwd_x3")
    return dq, dk, dv


def _attn_bwd_f32(q, k, v, do, heads, scale, dq, dk, dv, want_dq, want_dkv, o=None, lse=None):
    """gradients of softmax(scale q k^T) v on MATERIALISED fp32 maps (`/root/reference/p2p/model/register.py:43-51` is the
    forward being differentiated): P recomputed, dP = dO V^T, dS, then dQ = dS K, dK = dS^T Q, dV = P^T dO -- every
    product an `ief_gemm_f32` launch in the model's contraction mode.  With the forward's `lse` (split-operand mode, head dims
    40 / 64, >= 128 keys: `x3_fused_bwd_ok`) the fused recomputing kernel runs instead and no map is written."""
    if lse is not None and o is not None and x3_fused_bwd_ok(q.shape[2] // heads, k.shape[1]):
        return _attn_bwd_x3(q, k, v, o, do, lse, heads, scale, dq, dk, dv, want_dq, want_dkv)
    for t, nm in ((q, "q"), (k, "k"), (v, "v"), (do, "do")):
        _act32(t, nm)
    probs = _attn_scores_f32(q, k, heads, scale)                          # [B*h, N, L]
    ds = _attn_scores_f32(do, v, heads, 1.0, softmax=False)               # dP = dO V^T
    _softmax_bwd_f32_(probs, ds, scale)
    if want_dq:
        dq = _attn_apply_f32(ds, k, heads, out=dq)                        # dQ = dS K   (out: the caller's column slice, if any)
    if want_dkv:
        if ds.shape[2] % 4 == 0 and ds.shape[1] % 4 == 0:      # 16-byte rows of the [K][N] operand: every self-attention level
            dk = _attn_apply_t_f32(ds, q, heads, out=dk)                                # dK = dS^T Q
            dv = _attn_apply_t_f32(probs, do, heads, out=dv)                            # dV = P^T dO
        else:                                                  # 77 keys: the maps are small, transpose them
            dk = _attn_apply_f32(_transpose_maps_f32(ds), q, heads, out=dk)
            dv = _attn_apply_f32(_transpose_maps_f32(probs), do, heads, out=dv)
    return dq, dk, dv


def map_loss_blocks_f32(rows):
    return load().ief_map_loss_rows_blocks(rows)


def _attn_map_loss_bwd_f32(q, k, ref, dq, heads, scale, gcoef, accumulate, loss, loss_coef):
    """fp32 form of `attn_map_loss_bwd` on materialised maps: e = P - ref, dP = gcoef e, dS = scale P o (dP - sum dP P),
    dq (+)= dS k; loss: fp32 [map_loss_blocks_f32(B*heads*N)] partials"""
    lib = load()
    B, N, C = q.shape
    L = k.shape[1]
    if tuple(_act32(ref, "ref").shape) != (B * heads, N, L) or not ref.is_contiguous():
        raise ValueError("attn_map_loss_bwd: ref must be contiguous fp32 [B*heads, N, L]")
    probs = _attn_scores_f32(q, k, heads, scale)
    dp = torch.empty_like(probs)
    rows = B * heads * N
    if loss is not None and _dev32(loss, "loss").numel() < lib.ief_map_loss_rows_blocks(rows):
        raise ValueError("attn_map_loss_bwd: loss needs map_loss_blocks_f32(B*heads*N) floats")
    _check(lib.ief_map_loss_rows_f32(probs.data_ptr(), ref.data_ptr(), dp.data_ptr(), _ptr(loss), rows, L, float(gcoef),
                                     float(loss_coef), _stream()), "ief_map_loss_rows_f32")
    _softmax_bwd_f32_(probs, dp, scale)
    g = _attn_apply_f32(dp, k, heads)
    if accumulate:
        add(dq, g, out=dq)
    else:
        dq.copy_(g)
    return dq


def axpy(y, x, a):
    """y += a * x (fp32, in place)"""
    lib = load()
    _dev32(y, "y"), _dev32(x, "x")
    if y.shape != x.shape or not y.is_contiguous() or not x.is_contiguous():
        raise ValueError("axpy: operands must be contiguous and of equal shape")
    _check(lib.ief_axpy_f32(y.data_ptr(), x.data_ptr(), float(a), y.numel(), _stream()), "ief_axpy_f32")
    return y


def attn_bwd(q, k, v, o, do, lse, heads, scale, dq=None, dk=None, dv=None, ds_mul=None, want_dq=True, want_dkv=True):
    """Gradients of out = softmax(q k^T scale) v w.r.t. q, k, v given dO (`do`), the forward output `o` and its `lse`.
    q/do/o [B,N,h*d], k/v [B,L,h*d] (strided views ok); dq/dk/dv may be column slices of larger buffers."""
    lib = load()
    if _is32(q):
        return _attn_bwd_f32(q, k, v, do, heads, scale, dq, dk, dv, want_dq, want_dkv, o=o, lse=lse)
    B, N, _ = q.shape
    L = k.shape[1]
    d = o.shape[-1] // heads
    for t, nm in ((q, "q"), (k, "k"), (v, "v"), (o, "o"), (do, "do")):
        _dev16(t, nm)
        if t.dim() != 3 or (t.shape[0] > 1 and t.stride(0) != t.shape[1] * t.stride(1)):
            raise ValueError(f"{nm}: expected [B, rows, heads*d] with batch stride rows * ld")
    _dev32(lse, "lse")
    delta = torch.empty(B, heads, N, dtype=torch.float32, device=q.device)
    _check(lib.ief_attn_bwd_delta_f32(o.data_ptr(), do.data_ptr(), delta.data_ptr(), B, heads, N, d, o.stride(1),
                                      do.stride(1), _stream()), "ief_attn_bwd_delta_f32")
    p = IefAttnBwdParams()
    p.Q, p.K, p.V, p.dO, p.lse, p.delta = q.data_ptr(), k.data_ptr(), v.data_ptr(), do.data_ptr(), lse.data_ptr(), delta.data_ptr()
    p.B, p.heads, p.N, p.L, p.d = B, heads, N, L, d
    p.ldq, p.ldk, p.ldv, p.ldo = q.stride(1), k.stride(1), v.stride(1), do.stride(1)
    p.scale = scale
    p.ds_mul = float(ds_mul) if ds_mul is not None else float(min(L, 1024))
    what = 0
    if want_dq:
        if dq is None:
            dq = torch.empty(B, N, heads * d, dtype=torch.float16, device=q.device)
        _dev16(dq, "dq")
        p.dQ, p.lddq = dq.data_ptr(), dq.stride(1)
        what |= 1
    if want_dkv:
        if dk is None:
            dk = torch.empty(B, L, heads * d, dtype=torch.float16, device=q.device)
        if dv is None:
            dv = torch.empty(B, L, heads * d, dtype=torch.float16, device=q.device)
        _dev16(dk, "dk"), _dev16(dv, "dv")
        p.dK, p.dV, p.lddk, p.lddv = dk.data_ptr(), dv.data_ptr(), dk.stride(1), dv.stride(1)
        what |= 2
        # few keys (cross-attention) => few workgroups: cut the query range so ~256 of them run
        blocks, tiles = ((L + 127) // 128) * heads * B, (N + 63) // 64
        if blocks < 128 and tiles >= 4:
            p.kv_splits = max(1, min(tiles // 2, 256 // blocks))
            if p.kv_splits > 1:
                ws = torch.empty(2 * p.kv_splits * B * L * heads * d, dtype=torch.float32, device=q.device)
                p.ws = ws.data_ptr()
    fl = (6.0 if want_dq else 0.0) + (8.0 if want_dkv else 0.0)
    with _Timed(f"attn_bwd_kernel<{d}>", fl * B * heads * N * L * d):
        _check(lib.ief_attn_bwd_f16(byref(p), what, _stream()), "ief_attn_bwd_f16")
    return dq, dk, dv


def groupnorm_bwd(x, dy, gamma, beta, groups, eps, silu=False, x2=None, add=None, stats=None):
    """dx (and dx2 for a channel-concat input) of groupnorm(x | x2) [+ SiLU]; `add` is summed into the result.
    stats: (mean, rstd) [B, groups, 2] from `groupnorm(..., return_stats=True)` — enables the split two-launch path."""
    lib = load()
    if _is32(x):
        _act32(x, "x"), _act32(dy, "dy")
        B, C1 = x.shape[0], x.shape[-1]
        C2 = 0 if x2 is None else _act32(x2, "x2").shape[-1]
        HW = x.numel() // (B * C1)
        if not x.is_contiguous() or not dy.is_contiguous() or dy.numel() != B * HW * (C1 + C2) or (
                x2 is not None and (not x2.is_contiguous() or x2.numel() != B * HW * C2)) or (
                add is not None and (not _act32(add, "add").is_contiguous() or add.numel() != dy.numel())):
            raise ValueError("groupnorm_bwd: x / x2 / dy / add must be contiguous, dy and add [B, HW, C1+C2]")
        dx = torch.empty_like(x)
        dx2 = torch.empty_like(x2) if x2 is not None else None
        al16 = all(t is None or t.data_ptr() % 16 == 0 for t in (x, x2, dy, add, gamma, beta))
        if GN3_F32 and al16 and C1 % 4 == 0 and C2 % 4 == 0 and (C1 + C2) // groups <= 256:
            nws = lib.ief_groupnorm_bwd_f32_ws_floats(B, HW, C1 + C2)
            ws = torch.empty(nws, dtype=torch.float32, device=x.device)
            with _Timed("gn3_bwd_f32 (5 launches)", 0.0):
                _check(lib.ief_groupnorm_bwd_f32_ws(x.data_ptr(), _ptr(x2), C1, C2, dy.data_ptr(), _ptr(add), dx.data_ptr(), _ptr(dx2),
                                                    _dev32(gamma, "gamma").data_ptr(), _dev32(beta, "beta").data_ptr(), B, HW, groups,
                                                    eps, 1 if silu else 0, ws.data_ptr(), nws, _stream()), "ief_groupnorm_bwd_f32_ws")
            return (dx, dx2) if x2 is not None else dx
        with _Timed("gn_bwd_f32_kernel", 0.0):
            _check(lib.ief_groupnorm_bwd_f32(x.data_ptr(), _ptr(x2), C1, C2, dy.data_ptr(), _ptr(add), dx.data_ptr(), _ptr(dx2),
                                             _dev32(gamma, "gamma").data_ptr(), _dev32(beta, "beta").data_ptr(), B, HW, groups,
                                             eps, 1 if silu else 0, _stream()), "ief_groupnorm_bwd_f32")
        return (dx, dx2) if x2 is not None else dx
    _dev16(x, "x"), _dev16(dy, "dy")
    B, C1 = x.shape[0], x.shape[-1]
    C2 = 0 if x2 is None else x2.shape[-1]
    HW = x.numel() // (B * C1)
    if not x.is_contiguous() or not dy.is_contiguous() or dy.numel() != B * HW * (C1 + C2):
        raise ValueError("groupnorm_bwd: x / dy must be contiguous, dy [B, HW, C1+C2]")
    if x2 is not None and (not _dev16(x2, "x2").is_contiguous() or x2.numel() != B * HW * C2):
        raise ValueError("groupnorm_bwd: x2 shape")
    if add is not None and (not _dev16(add, "add").is_contiguous() or add.numel() != dy.numel()):
        raise ValueError("groupnorm_bwd: add must match dy")
    dx = torch.empty_like(x)
    dx2 = torch.empty_like(x2) if x2 is not None else None
    partial = None
    if stats is not None:
        if tuple(_dev32(stats, "stats").shape) != (B, groups, 2):
            raise ValueError("groupnorm_bwd: stats must be fp32 [B, groups, 2]")
        partial = torch.empty(B * lib.ief_gn_splits(HW) * groups * 2, dtype=torch.float32, device=x.device)
    with _Timed("groupnorm_bwd", 0.0):
        _check(lib.ief_groupnorm_bwd_f16(x.data_ptr(), _ptr(x2), C1, C2, dy.data_ptr(), _ptr(add), dx.data_ptr(), _ptr(dx2),
                                         _dev32(gamma, "gamma").data_ptr(), _dev32(beta, "beta").data_ptr(), _ptr(stats),
                                         _ptr(partial), B, HW, groups, eps, 1 if silu else 0, _stream()),
               "ief_groupnorm_bwd_f16")
    return (dx, dx2) if x2 is not None else dx


def layernorm_bwd(x, dy, gamma, eps=1e-5, add=None):
    lib = load()
    if _is32(x):
        _act32(x, "x"), _act32(dy, "dy")
        if not x.is_contiguous() or not dy.is_contiguous() or dy.shape != x.shape or (
                add is not None and (not _act32(add, "add").is_contiguous() or add.shape != x.shape)):
            raise ValueError("layernorm_bwd: x / dy / add must be contiguous and of equal shape")
        C = x.shape[-1]
        dx = torch.empty_like(x)
        _check(lib.ief_layernorm_bwd_f32(x.data_ptr(), dy.data_ptr(), _ptr(add), dx.data_ptr(), _dev32(gamma, "gamma").data_ptr(),
                                         x.numel() // C, C, eps, _stream()), "ief_layernorm_bwd_f32")
        return dx
    _dev16(x, "x"), _dev16(dy, "dy")
    if not x.is_contiguous() or not dy.is_contiguous() or dy.shape != x.shape:
        raise ValueError("layernorm_bwd: x / dy must be contiguous and of equal shape")
    if add is not None and (not _dev16(add, "add").is_contiguous() or add.shape != x.shape):
        raise ValueError("layernorm_bwd: add must match x")
    C = x.shape[-1]
    dx = torch.empty_like(x)
    _check(lib.ief_layernorm_bwd_f16(x.data_ptr(), dy.data_ptr(), _ptr(add), dx.data_ptr(), _dev32(gamma, "gamma").data_ptr(),
                                     x.numel() // C, C, eps, _stream()), "ief_layernorm_bwd_f16")
    return dx


def geglu_il(pre, out=None):
    """pre [..., 2*Ch] in the interleaved FF1 layout -> hidden * gelu(gate) [..., Ch]"""
    lib = load()
    if _is32(pre):
        if not _act32(pre, "pre").is_contiguous():
            raise ValueError("geglu_il: pre must be contiguous")
        Ch = pre.shape[-1] // 2
        if out is None:
            out = torch.empty(*pre.shape[:-1], Ch, dtype=torch.float32, device=pre.device)
        _check(lib.ief_geglu_il_f32(pre.data_ptr(), out.data_ptr(), pre.numel() // (2 * Ch), Ch, _stream()), "ief_geglu_il_f32")
        return out
    _dev16(pre, "pre")
    if not pre.is_contiguous():
        raise ValueError("geglu_il: pre must be contiguous")
    Ch = pre.shape[-1] // 2
    out = torch.empty(*pre.shape[:-1], Ch, dtype=torch.float16, device=pre.device)
    _check(lib.ief_geglu_il_f16(pre.data_ptr(), out.data_ptr(), pre.numel() // (2 * Ch), Ch, _stream()), "ief_geglu_il_f16")
    return out


def geglu_il_bwd(pre, dy):
    lib = load()
    if _is32(pre):
        _act32(pre, "pre"), _act32(dy, "dy")
        Ch = pre.shape[-1] // 2
        if not pre.is_contiguous() or not dy.is_contiguous() or dy.shape[-1] != Ch or dy.numel() * 2 != pre.numel():
            raise ValueError("geglu_il_bwd: pre [..., 2*Ch], dy [..., Ch], both contiguous")
        dpre = torch.empty_like(pre)
        _check(lib.ief_geglu_il_bwd_f32(pre.data_ptr(), dy.data_ptr(), dpre.data_ptr(), pre.numel() // (2 * Ch), Ch, _stream()),
               "ief_geglu_il_bwd_f32")
        return dpre
    _dev16(pre, "pre"), _dev16(dy, "dy")
    Ch = pre.shape[-1] // 2
    if not pre.is_contiguous() or not dy.is_contiguous() or dy.shape[-1] != Ch or dy.numel() * 2 != pre.numel():
        raise ValueError("geglu_il_bwd: pre [..., 2*Ch], dy [..., Ch], both contiguous")
    dpre = torch.empty_like(pre)
    _check(lib.ief_geglu_il_bwd_f16(pre.data_ptr(), dy.data_ptr(), dpre.data_ptr(), pre.numel() // (2 * Ch), Ch, _stream()),
           "ief_geglu_il_bwd_f16")
    return dpre


def zero_insert2x(x):
    """[B,H,W,C] -> [B,2H,2W,C] with x at the even positions, zeros elsewhere (stride-2 conv data gradient)"""
    lib = load()
    if _is32(x):
        if _act32(x, "x").dim() != 4 or not x.is_contiguous():
            raise ValueError("zero_insert2x: contiguous NHWC expected")
        B, H, W, C = x.shape
        out = torch.empty(B, 2 * H, 2 * W, C, dtype=torch.float32, device=x.device)
        _check(lib.ief_zero_insert2x_f32(x.data_ptr(), out.data_ptr(), B, H, W, C, _stream()), "ief_zero_insert2x_f32")
        return out
    _dev16(x, "x")
    if x.dim() != 4 or not x.is_contiguous():
        raise ValueError("zero_insert2x: contiguous NHWC expected")
    B, H, W, C = x.shape
    out = torch.empty(B, 2 * H, 2 * W, C, dtype=torch.float16, device=x.device)
    _check(lib.ief_zero_insert2x_f16(x.data_ptr(), out.data_ptr(), B, H, W, C, _stream()), "ief_zero_insert2x_f16")
    return out


def pool2x2_sum(x):
    """[B,2H,2W,C] -> [B,H,W,C] summing each 2x2 block (nearest-2x upsample backward)"""
    lib = load()
    if _is32(x):
        if _act32(x, "x").dim() != 4 or not x.is_contiguous() or (x.shape[1] & 1) or (x.shape[2] & 1):
            raise ValueError("pool2x2_sum: contiguous NHWC with even H, W expected")
        B, H2, W2, C = x.shape
        out = torch.empty(B, H2 // 2, W2 // 2, C, dtype=torch.float32, device=x.device)
        _check(lib.ief_pool2x2_sum_f32(x.data_ptr(), out.data_ptr(), B, H2 // 2, W2 // 2, C, _stream()), "ief_pool2x2_sum_f32")
        return out
    _dev16(x, "x")
    if x.dim() != 4 or not x.is_contiguous() or (x.shape[1] & 1) or (x.shape[2] & 1):
        raise ValueError("pool2x2_sum: contiguous NHWC with even H, W expected")
    B, H2, W2, C = x.shape
    out = torch.empty(B, H2 // 2, W2 // 2, C, dtype=torch.float16, device=x.device)
    _check(lib.ief_pool2x2_sum_f16(x.data_ptr(), out.data_ptr(), B, H2 // 2, W2 // 2, C, _stream()), "ief_pool2x2_sum_f16")
    return out


def conv_out_bwd(d_eps, w):
    """d_eps fp32 NCHW [B,Cout,H,W], w fp16 [Cout,3,3,C] (conv_out's own weight) -> fp16 NHWC [B,H,W,C]"""
    lib = load()
    if _is32(w):
        _dev32(d_eps, "d_eps")
        B, Cout, H, W = d_eps.shape
        C = w.shape[-1]
        if tuple(_act32(w, "w").shape) != (Cout, 3, 3, C) or not w.is_contiguous():
            raise ValueError("conv_out_bwd: weight must be contiguous [Cout, 3, 3, C]")
        out = torch.empty(B, H, W, C, dtype=torch.float32, device=w.device)
        _check(lib.ief_conv_out_bwd_f32w(d_eps.data_ptr(), w.data_ptr(), out.data_ptr(), B, C, H, W, Cout, _stream()),
               "ief_conv_out_bwd_f32w")
        return out
    _dev32(d_eps, "d_eps"), _dev16(w, "w")
    B, Cout, H, W = d_eps.shape
    C = w.shape[-1]
    if tuple(w.shape) != (Cout, 3, 3, C) or not w.is_contiguous():
        raise ValueError("conv_out_bwd: weight must be contiguous [Cout, 3, 3, C]")
    out = torch.empty(B, H, W, C, dtype=torch.float16, device=w.device)
    _check(lib.ief_conv_out_bwd_f32(d_eps.data_ptr(), w.data_ptr(), out.data_ptr(), B, C, H, W, Cout, _stream()),
           "ief_conv_out_bwd_f32")
    return out


def nti_loss_grad(eps_u, eps_c, x, target, coef, d_eps, stats, grad_scale=1.0):
    """NTI objective + normalised gradient w.r.t. eps_u (see include/ief_hip.h); stats fp32 [2] <- (loss, factor)."""
    lib = load()
    for t, nm in ((eps_u, "eps_u"), (eps_c, "eps_c"), (x, "x"), (target, "target"), (coef, "coef"), (d_eps, "d_eps"),
                  (stats, "stats")):
        _dev32(t, nm)
    n = eps_u.numel()
    if any(t.numel() != n for t in (eps_c, x, target, d_eps)) or coef.numel() < 3 or stats.numel() < 2:
        raise ValueError("nti_loss_grad: shape mismatch")
    _check(lib.ief_nti_loss_grad_f32(eps_u.data_ptr(), eps_c.data_ptr(), x.data_ptr(), target.data_ptr(), coef.data_ptr(),
                                     d_eps.data_ptr(), stats.data_ptr(), n, grad_scale, _stream()), "ief_nti_loss_grad_f32")


def nti_adam(param, m, v, grad16, stats, hyper, step, param16):
    """One torch.optim.Adam step on fp32 `param` with g = grad16 * stats[1]; hyper fp32 {lr, beta1, beta2, eps}."""
    lib = load()
    for t, nm in ((param, "param"), (m, "m"), (v, "v"), (stats, "stats"), (hyper, "hyper")):
        _dev32(t, nm)
    if _is32(grad16):       # fp32-storage modes: the context IS the fp32 parameter (param16 unused)
        if grad16.numel() != param.numel() or not _act32(grad16, "grad").is_contiguous():
            raise ValueError("nti_adam: gradient shape mismatch")
        _check(lib.ief_nti_adam_f32g(param.data_ptr(), m.data_ptr(), v.data_ptr(), grad16.data_ptr(), stats.data_ptr(),
                                     hyper.data_ptr(), _devi32(step, "step").data_ptr(), param.numel(), _stream()),
               "ief_nti_adam_f32g")
        return
    _dev16(grad16, "grad16"), _dev16(param16, "param16")
    _devi32(step, "step")
    n = param.numel()
    if any(t.numel() != n for t in (m, v, grad16, param16)) or not grad16.is_contiguous() or not param16.is_contiguous():
        raise ValueError("nti_adam: shape mismatch")
    _check(lib.ief_nti_adam_f32(param.data_ptr(), m.data_ptr(), v.data_ptr(), grad16.data_ptr(), stats.data_ptr(),
                                hyper.data_ptr(), step.data_ptr(), param16.data_ptr(), n, _stream()), "ief_nti_adam_f32")
