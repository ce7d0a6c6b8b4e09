"""ctypes binding of libtoad_hip.so (the C ABI declared in include/toad_hip.h).

There is no CPU fallback: if the shared library is missing or a symbol is absent the import
fails loudly, and every op refuses non-CUDA tensors.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtoad_hip.so")      # the one library the product loads (A/B builds: tools/ab/select_lib.py rebinds this in the TOOL's process)
ABI_VERSION = 13

P, I64, I, F, SZ, U64 = c_void_p, c_int64, c_int, c_float, c_size_t, c_uint64

# name -> (restype, argtypes); mirrors include/toad_hip.h one to one
SIGNATURES = {
    "toad_abi_version": (I, []),
    "toad_last_error": (c_char_p, []),
    "toad_fallback_launches": (I64, []),
    "toad_amax_floats": (SZ, [I64]),
    "toad_absmax_rows256_f32": (I, [P, I64, I64, P, P]),
    "toad_linear_h2_ok": (I, [I64, I64, I64]),
    "toad_linear_ws_bytes": (SZ, [I64, I64, I64]),
    "toad_relu_bits_bytes": (SZ, [I64, I64]),
    "toad_linear_act_fwd_f32": (I, [P, P, P, P, I64, I64, I64, I, F, U64, P, P, P, P, SZ, P]),
    "toad_linear_dgrad_f32": (I, [P, P, P, P, F, P, I64, I64, I64, P, P, P, I, P, P, P, P, SZ, P]),
    "toad_dropout_mask_f32": (I, [P, I64, F, U64, P]),
    "toad_linear_wgrad_ws_bytes": (SZ, [I64, I64, I64]),
    "toad_linear_wgrad_f32": (I, [P, P, P, P, I64, I64, I64, F, P, P, P, SZ, P]),
    "toad_transpose_f32": (I, [P, P, I64, I64, P]),
    "toad_gated_pool_ws_bytes": (SZ, [I64, I, I, I]),
    "toad_gated_pool_fwd_f32": (I, [P, P, I64, P, P, P, P, P, P, P, SZ, I64, I, I, I, F, U64, U64, P]),
    "toad_gated_pool_bwd_ws_bytes": (SZ, [I64, I, I, I]),
    "toad_gated_pool_bwd_f32": (I, [P, P, I64, P, P, P, P, P, P, P, P, P, I64, P, P, P, F, P, P, SZ, I64, I, I, I, F, U64, U64, P]),
    "toad_heads_fwd_f32": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, P]),
    "toad_heads_bwd_f32": (I, [P, P, P, P, P, P, P, P, P, P, P, P, F, I, I, P]),
    "toad_heads_ce_fused_f32": (I, [P, P, P, P, P, P, P, P, F, F, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, F, I, I, P]),
    "toad_sgd_step_f32": (I, [P, P, P, I64, F, F, F, I64, P]),
    "toad_mtl_ce_fwd_bwd_f32": (I, [P, P, P, P, F, F, P, P, P, I, P]),
    "toad_adam_step_f32": (I, [P, P, P, P, I64, F, F, F, F, F, I64, P]),
    "toad_linear_act_res_fwd_f32": (I, [P, P, P, P, P, I64, I64, I64, I, P, SZ, P]),
    "toad_conv_nhwc_f32": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, P, SZ, P]),
    "toad_im2col_nhwc_f32": (I, [P, P, I, I, I, I, I, I, I, I, P]),
    "toad_im2col_stem_nchw_f32": (I, [P, P, I, I, I, P]),
    "toad_stem_s2d_nchw_f32": (I, [P, P, I, I, I, P]),
    "toad_stem_conv_s2d_f32": (I, [P, P, P, P, I, I, I, I, P, SZ, P]),
    "toad_stem_conv_pool_s2d_f32": (I, [P, P, P, P, I, I, I, P, SZ, P]),
    "toad_stem_pool_nchw_f32": (I, [P, P, P, P, I, I, I, P, SZ, P]),
    "toad_maxpool3x3s2_nhwc_f32": (I, [P, P, I, I, I, I, P]),
    "toad_avgpool_nhwc_f32": (I, [P, P, I, I, I, P]),
    "toad_resnet50_trunc_ws_bytes": (SZ, [I, I, I]),
    "toad_resnet50_trunc_fwd_f32": (I, [P, P, P, P, I, I, I, P, SZ, P]),
    "toad_mil_buffer_align": (SZ, [I64]),
    "toad_mil_arena_bytes": (SZ, [I64, I, I]),
    "toad_mil_arena_layout": (I, [I64, I, I, P]),
    "toad_mil_scratch_bytes": (SZ, [I64, I, I]),
    "toad_mil_fwd_f32": (I, [P, P, P, I64, I, I, F, U64, P, I, P, SZ, P, SZ, P]),
    "toad_mil_bwd_f32": (I, [P, P, F, P, I64, I, I, F, U64, P, SZ, P, P, P, P, P, P, P, SZ, P]),
    "toad_mil_step_ws_bytes": (SZ, [I64, I, I]),
    "toad_mil_step_f32": (I, [P, P, F, P, P, P, P, F, F, I64, I, I, F, U64, P, P, P, P, P, SZ, P, P]),
    "toad_mil_x16_ok": (I, [I64]),
    "toad_mil_fwd_x16_f32": (I, [P, P, P, I64, I, I, F, U64, I, P, SZ, P, SZ, P]),
    "toad_mil_bwd_x16_f32": (I, [P, P, F, P, I64, I, I, F, U64, P, SZ, P, P, P, P, P, P, SZ, P]),
    "toad_mil_step_x16_f32": (I, [P, P, F, P, P, P, P, F, F, I64, I, I, F, U64, P, P, P, P, SZ, P, P]),
    "toad_bag_planes_bytes": (SZ, [I64, I64]),
    "toad_bag_prepare_f32": (I, [P, I64, I64, P, P, P]),
    "toad_linear_wgrad_xp_f32": (I, [P, P, P, P, P, I64, I64, I64, F, P, P, SZ, P]),
    "toad_mil_fwd_xp_f32": (I, [P, P, P, P, I64, I, I, F, U64, I, P, SZ, P, SZ, P]),
    "toad_mil_bwd_xp_f32": (I, [P, P, F, P, P, I64, I, I, F, U64, P, SZ, P, P, P, P, P, P, SZ, P]),
    "toad_mil_multi_ws_bytes": (SZ, [I64, I, I, I]),
    "toad_mil_multi_step_f32": (I, [P, P, F, P, P, I, P, P, P, F, F, I, I, F, U64, P, P, P, P, SZ, P, P]),
    "toad_mil_step_xp_f32": (I, [P, P, F, P, P, P, P, P, F, F, I64, I, I, F, U64, P, P, P, P, SZ, P, P]),
}

_lib = None


def load() -> ctypes.CDLL:
    """dlopen libtoad_hip.so and bind every declared symbol (raises if anything is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP extension first (python -m toad_amd.build). "
            "toad_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.toad_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"libtoad_hip.so ABI version {got} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().toad_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
