"""ctypes binding of libmmamd.so (the C-ABI declared in include/mmamd.h).

The library is the product: there is NO fallback.  If the shared object is missing, cannot be loaded, or
lacks a symbol, `lib()` raises — loudly — instead of routing anything through PyTorch eager ops.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("MMAMD_LIB", _PKG / "lib" / "libmmamd.so"))

ABI_VERSION = 1

F32, BF16 = 0, 1
ACT_NONE, ACT_QUICKGELU, ACT_GELU_ERF = 0, 1, 2
ACT_MUL_QUICKGELU_GRAD, ACT_MUL_GELU_GRAD = 3, 4  # backward of the MLP: (A.W^T) * act'(residual)
REDUCE_MEAN, REDUCE_SUM = 0, 1

_vp, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64

# name -> (restype, argtypes); must list every prototype of include/mmamd.h (tests/test_capi_symbols.py checks)
PROTOTYPES = {
    "mmamd_abi_version": (_i, []),
    "mmamd_last_error": (C.c_char_p, []),
    "mmamd_clear_last_hip_error": (_i, []),
    "mmamd_set_gemm_variant": (_i, [_i]),
    "mmamd_get_gemm_variant": (_i, []),
    "mmamd_debug_set_gemm_stagger": (_i, [_i]),
    "mmamd_debug_set_gemm_trace": (_i, [_vp]),
    "mmamd_debug_set_gemm_knob": (_i, [_i, _i]),
    "mmamd_debug_tile_order": (_i, [_i, _i, _i, _i, _i, _vp]),
    "mmamd_pack_w_frag": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "mmamd_debug_set_gemm_wp": (_i, [_vp]),
    "mmamd_debug_set_attn_variant": (_i, [_i]),
    "mmamd_debug_set_colsum_wide": (_i, [_i]),
    "mmamd_debug_launch_count": (C.c_ulonglong, [C.c_char_p]),
    "mmamd_layernorm": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "mmamd_add_layernorm_grouped": (_i, [_vp, _i, _vp]),
    "mmamd_patch_embed_gemm": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mmamd_vit_cls_lnpre_ln": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _f, _vp, _i, _i, _i, _vp]),
    "mmamd_gemm_bf16": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "mmamd_gemm_bf16_grouped": (_i, [_vp, _i, _i, _i, _vp]),
    "mmamd_gemm_bf16_residual_ln_grouped": (_i, [_vp, _i, _vp]),
    "mmamd_gemm_bf16_residual_ln_supported": (_i, [_i, _i, _i]),
    "mmamd_pack_w_ksteps": (_i, [_vp, _i, _i, _vp, _vp]),
    "mmamd_pack_weights": (_i, [_vp, _i, _vp]),
    "mmamd_attention_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "mmamd_attention_fwd_lse": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "mmamd_attention_fwd_grouped": (_i, [_vp, _i, _f, _vp]),
    "mmamd_gemm_bf16_splitk": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mmamd_gemm_bf16_tn_splitk": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mmamd_gemm_bf16_tn_splitk_colsum": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mmamd_attention_probs_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "mmamd_attention_probs_from_lse": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "mmamd_contrastive_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                    _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "mmamd_layernorm_bwd_groups": (_i, [_i, _i]),
    "mmamd_layernorm_bwd": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "mmamd_colsum_stage2_batched": (_i, [_vp, _i, _vp]),
    "mmamd_gemm_bf16_tn_splitk_group_ws": (C.c_longlong, [_vp, _i, _i]),
    "mmamd_gemm_bf16_tn_splitk_group": (_i, [_vp, _i, _i, _vp, _vp]),
    "mmamd_colsum": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "mmamd_gemm_bf16_dual": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "mmamd_bicubic_pos_embed": (_i, [_vp, _i, _i, _vp, _i, _i, _f, _f, _vp]),
    "mmamd_offset_position_ids": (_i, [_vp, _i64, _vp, _i, _i, _vp]),
    "mmamd_image_resample": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "mmamd_group_mean_normalize": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "mmamd_scale_normalize": (_i, [_vp, _vp, _i, _i, _f, _vp]),
    "mmamd_target_rank": (_i, [_vp, _i64, _vp, _i, _i, _vp, _vp]),
    "mmamd_mask_labels": (_i, [_vp, _vp, _i64, _i64, _vp]),
    "mmamd_relu_bwd": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "mmamd_conv_gemm_bf16": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "mmamd_dalle_stem_im2col": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "mmamd_dalle_maxpool2": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mmamd_dalle_argmax": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "mmamd_row_softmax_": (_i, [_vp, _i64, _i, _vp]),
    "mmamd_dalle_pack": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "mmamd_stream_create_cu_mask": (_i, [_vp, _i, _vp]),
    "mmamd_stream_destroy": (_i, [_vp]),
    "mmamd_stream_cus": (_i, [_vp]),
    "mmamd_stream_set_cus": (_i, [_vp, _i]),
    "mmamd_debug_cu_census": (_i, [_vp, _i, C.c_longlong, _vp]),
    "mmamd_dropout": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i64, _i64, _f, C.c_uint64, C.c_uint32, _vp]),
    "mmamd_act_fwd": (_i, [_vp, _vp, _i64, _i, _vp]),
    "mmamd_act_bwd": (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
    "mmamd_activation": (_i, [_vp, _vp, _vp, _i, _i64, _i, _vp]),
    "mmamd_transpose_to_bf16": (_i, [_vp, _i, _i64, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "mmamd_l2_normalize_bwd": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp]),
    "mmamd_scatter_add_rows": (_i, [_vp, _vp, _i, _i, _vp, _i64, _vp]),
    "mmamd_f32_gemm_strided": (_i, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "mmamd_select_tokens": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mmamd_gather_rows": (_i, [_vp, _i64, _vp, _i, _i, _vp, _i, _vp, _vp]),
    "mmamd_cross_entropy_bwd": (_i, [_vp, _i64, _vp, _i, _i, _i64, _vp, _vp, _i, _i64, _vp, _vp]),
    "mmamd_cross_entropy": (_i, [_vp, _i64, _vp, _i, _i, _i64, _vp, _vp, _vp]),
    "mmamd_attention_x_fwd": (_i, [_vp, _i, _i64, _vp, _vp, _i, _i, _i64, _vp, _vp, _i64, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "mmamd_attention_x_fwd_head_mask": (_i, [_vp, _i, _i64, _vp, _vp, _i, _i, _i64, _vp, _vp, _i64, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _f,
                                             _vp, _i64, _i64, _i64, _i64, _vp]),
    "mmamd_attention_x_fwd_dropout": (_i, [_vp, _i, _i64, _vp, _vp, _i, _i, _i64, _vp, _vp, _i64, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _f,
                                            C.c_uint64, C.c_uint32, _vp]),
    "mmamd_attention_x_bwd_dropout": (_i, [_vp, _i, _i64, _vp, _vp, _i, _i, _i64, _vp, _vp, _i64, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i,
                                            _i, _i, _i, _f, _f, C.c_uint64, C.c_uint32, _vp]),
    "mmamd_attention_x_fwd_dropout_head_mask": (_i, [_vp, _i, _i64, _vp, _vp, _i, _i, _i64, _vp, _vp, _i64, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _f,
                                                      C.c_uint64, C.c_uint32, _vp, _i64, _i64, _i64, _i64, _vp]),
    "mmamd_attention_x_bwd_dropout_head_mask": (_i, [_vp, _i, _i64, _vp, _vp, _i, _i, _i64, _vp, _vp, _i64, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i,
                                                      _i, _i, _i, _i, _f, _f, C.c_uint64, C.c_uint32, _vp, _i64, _i64, _i64, _i64, _vp]),
    "mmamd_attention_x_bwd": (_i, [_vp, _i, _i64, _vp, _vp, _i, _i, _i64, _vp, _vp, _i64, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i,
                                    _i, _i, _i, _f, _vp]),
    "mmamd_attention_x_bwd_head_mask": (_i, [_vp, _i, _i64, _vp, _vp, _i, _i, _i64, _vp, _vp, _i64, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i,
                                              _i, _i, _i, _f, _vp, _i64, _i64, _i64, _i64, _vp]),
    "mmamd_attention_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "mmamd_coca_text_embed": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mmamd_coca_text_mask": (_i, [_vp, _i, _i64, _vp, _i, _i, _vp]),
    "mmamd_token_mean": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "mmamd_key_mask": (_i, [_vp, _i, _i64, _vp, _i64, _vp]),
    "mmamd_bert_embed_ln": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "mmamd_flava_image_embed": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "mmamd_rows_linear_f32": (_i, [_vp, _i64, _vp, _vp, _i, _vp, _i, _i, _i, _vp]),
    "mmamd_patchify": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "mmamd_vit_assemble_ln": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _vp]),
    "mmamd_embed_tokens": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mmamd_pool_ln_proj": (_i, [_vp, _i, _i, _vp, _vp, _vp, _f, _vp, _i, _i, _vp, _i, _i, _i, _vp, _vp]),
    "mmamd_l2_normalize": (_i, [_vp, _i, _vp, _i, _i, _i, _f, _vp]),
    "mmamd_l2_normalize_ld": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _f, _vp]),
    "mmamd_clamp_scalar": (_i, [_vp, _i, _f, _i, _f, _vp]),
    "mmamd_contrastive_fwd": (
        _i,
        [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp],
    ),
    "mmamd_contrastive_fwd_ld": (
        _i,
        [_vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp],
    ),
    "mmamd_convert": (_i, [_vp, _i, _vp, _i, _i64, _vp]),
    "mmamd_timer_create": (_vp, []),
    "mmamd_timer_destroy": (None, [_vp]),
    "mmamd_timer_start": (_i, [_vp, _vp]),
    "mmamd_timer_stop": (_i, [_vp, _vp]),
    "mmamd_timer_elapsed_ms": (_i, [_vp, C.POINTER(C.c_float)]),
}


class MmamdError(RuntimeError):
    pass


_LIB: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load (once) and return the bound library; raise MmamdError if it is unusable."""
    global _LIB
    if _LIB is not None:
        return _LIB
    try:
        # torch's ROCm wheels bundle their own libamdhip64.so; libmmamd.so depends on the same SONAME.  Whichever copy is loaded
        # first serves both, and if it is /opt/rocm's (because this library was dlopen'ed before `import torch`) torch then finds
        # "No HIP GPUs": make sure torch's runtime is the one in the process.  (A host without torch simply uses /opt/rocm's.)
        import torch  # noqa: F401
    except ImportError:
        pass
    if not LIB_PATH.exists():
        raise MmamdError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m multimodal_amd.build` "
            "(or __graft_entry__.build()). There is no eager/CPU fallback for this path."
        )
    try:
        handle = C.CDLL(str(LIB_PATH))
    except OSError as e:  # missing libamdhip64 etc.
        raise MmamdError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            if os.environ.get("MMAMD_LIB_ALLOW_MISSING") == "1":  # A/B tools against an OLDER build (MMAMD_LIB=...): entries it lacks stay unbound
                continue
            raise MmamdError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    got = handle.mmamd_abi_version()
    if got != ABI_VERSION:
        raise MmamdError(f"{LIB_PATH} has ABI version {got}, host code expects {ABI_VERSION}; rebuild it")
    _LIB = handle
    return handle


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().mmamd_last_error()
        raise MmamdError(f"{what} failed (status {status}): {msg.decode() if msg else '?'}")
