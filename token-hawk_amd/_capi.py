"""ctypes declarations for every entry point of include/thk.h (libthk.so).

The library is the product; this file only describes its C-ABI to Python so that
tests/ and bench.py can call through it.  There is NO CPU fallback: if libthk.so is
missing or does not load, importing fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("THK_LIB") or os.path.join(_HERE, "libthk.so")   # THK_LIB: development builds (libthk_trace.so)

THK_OK = 0
THK_F32, THK_F16 = 0, 1
THK_LMHEAD_CORRECT, THK_LMHEAD_FAITHFUL = 0, 1
THK_STAGE_EMBED, THK_STAGE_HEAD = 1, 2


class HParams(C.Structure):
    _fields_ = [("n_vocab", C.c_int32), ("n_embd", C.c_int32), ("n_mult", C.c_int32),
                ("n_head", C.c_int32), ("n_layer", C.c_int32), ("n_ctx", C.c_int32)]


vp, i64, i32, u64, u32 = C.c_void_p, C.c_int64, C.c_int32, C.c_uint64, C.c_uint32
pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); must list every symbol declared in include/thk.h
SIGNATURES = {
    "thk_abi_version": (C.c_int, []),
    "thk_ctx_create": (C.c_int, [C.c_int, pp]),
    "thk_ctx_create_on_stream": (C.c_int, [C.c_int, vp, pp]),
    "thk_ctx_destroy": (C.c_int, [vp]),
    "thk_sync": (C.c_int, [vp]),
    "thk_last_error": (C.c_char_p, [vp]),
    "thk_ctx_stream": (vp, [vp]),
    "thk_ctx_device_info": (C.c_int, [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "thk_buf_alloc": (C.c_int, [vp, C.c_size_t, pp]),
    "thk_buf_free": (C.c_int, [vp, vp]),
    "thk_buf_ptr": (vp, [vp]),
    "thk_buf_size": (C.c_size_t, [vp]),
    "thk_buf_upload": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_size_t]),
    "thk_buf_download": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_size_t]),
    "thk_buf_copy": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_size_t]),
    "thk_matvec_f16": (C.c_int, [vp, vp, i64, i64, vp, vp]),
    "thk_rms_norm": (C.c_int, [vp, vp, i64, i64]),
    "thk_row_element_multiply": (C.c_int, [vp, vp, vp, i64, i64]),
    "thk_rope": (C.c_int, [vp, vp, i64, i64, i64, i64]),
    "thk_kv_append": (C.c_int, [vp, vp, vp, vp, vp, i64, i64, i64]),
    "thk_attn_decode": (C.c_int, [vp, vp, vp, vp, i64, i64, i64, vp]),
    "thk_attn_prefill": (C.c_int, [vp, vp, vp, vp, i64, i64, i64, i64, vp]),
    "thk_row_softmax": (C.c_int, [vp, vp, i64, i64]),
    "thk_add": (C.c_int, [vp, vp, vp, vp, i64]),
    "thk_silu": (C.c_int, [vp, vp, i64]),
    "thk_mul_inplace": (C.c_int, [vp, vp, vp, i64]),
    "thk_lmhead_f16": (C.c_int, [vp, vp, i64, i64, vp, vp, C.c_int]),
    "thk_argmax": (C.c_int, [vp, vp, i64, vp]),
    "thk_topk_f32": (C.c_int, [vp, vp, i64, i32, vp, vp]),
    "thk_embed_f16": (C.c_int, [vp, vp, i64, i32, vp]),
    "thk_gemm_f16_prefill": (C.c_int, [vp, vp, i64, i64, vp, i64, vp]),
    "thk_synth_f16": (C.c_int, [vp, C.c_char_p, u64, C.c_float, i64, vp]),
    "thk_synth_gain_f32": (C.c_int, [vp, C.c_char_p, u64, C.c_float, i64, vp]),
    "thk_model_create": (C.c_int, [vp, C.POINTER(HParams), i32, i32, u32, i32, pp]),
    "thk_model_destroy": (C.c_int, [vp]),
    "thk_model_n_ff": (i32, [vp]),
    "thk_model_set_tensor": (C.c_int, [vp, C.c_char_p, C.c_int, i64, i64, vp]),
    "thk_model_set_tensor_dev": (C.c_int, [vp, C.c_char_p, C.c_int, i64, i64, vp]),
    "thk_model_get_tensor": (C.c_int, [vp, C.c_char_p, i64, i64, vp]),
    "thk_model_fill_synthetic": (C.c_int, [vp, u64, C.c_float]),
    "thk_model_finalize": (C.c_int, [vp]),
    "thk_model_reset_kv": (C.c_int, [vp, i32]),
    "thk_model_set_lmhead_mode": (C.c_int, [vp, C.c_int]),
    "thk_model_eval": (C.c_int, [vp, i32, vp, i32, i32, vp, vp]),
    "thk_model_prefill": (C.c_int, [vp, i32, vp, i32, i32, vp]),
    "thk_model_prefill_stage": (C.c_int, [vp, i32, vp, vp, i32, i32, vp]),
    "thk_model_prepare_prefill": (C.c_int, [vp]),
    "thk_model_prefill_uses_tile_images": (C.c_int, [vp]),
    "thk_model_seq_set": (C.c_int, [vp, i32, i32, i32]),
    "thk_model_seq_set_token": (C.c_int, [vp, i32, i32]),
    "thk_model_decode_step": (C.c_int, [vp, i32, C.c_int]),
    "thk_model_decode_steps": (C.c_int, [vp, i32, i32, C.c_int]),
    "thk_model_prepare_steps": (C.c_int, [vp, i32, i32]),
    "thk_model_uses_engine": (C.c_int, [vp]),
    "thk_model_debug_buffer": (C.c_int, [vp, C.c_char_p, vp, i64, C.POINTER(i64)]),
    "thk_model_engine_trace": (C.c_int, [vp, vp, i64, C.POINTER(i32), C.POINTER(i32)]),
    "thk_model_hidden_in": (vp, [vp, i32]),
    "thk_model_hidden_out": (vp, [vp, i32]),
    "thk_model_token_dev": (vp, [vp, i32]),
    "thk_model_logits_dev": (vp, [vp, i32]),
    "thk_model_seq_get": (C.c_int, [vp, i32, vp, i32, C.POINTER(i32), C.POINTER(i32)]),
    "thk_model_seq_last_token": (C.c_int, [vp, i32, C.POINTER(i32)]),
    "thk_model_seq_clock": (C.c_int, [vp, i32, vp, i32, C.POINTER(i32)]),
    "thk_model_logits_topk": (C.c_int, [vp, i32, i32, vp, vp]),
    "thk_model_eval_topk": (C.c_int, [vp, i32, vp, i32, i32, i32, vp, vp]),
    "thk_model_read_logits": (C.c_int, [vp, i32, vp]),
    "thk_model_bytes_per_token": (i64, [vp, i32]),
    "thk_model_profile_step": (C.c_int, [vp, i32, i32, vp, vp, C.POINTER(i32)]),
    "thk_model_step_trace": (C.c_int, [vp, i32, vp, i64, i32, vp, C.POINTER(i32), C.POINTER(i32)]),
    "thk_model_n_embd": (i32, [vp]),
    "thk_model_n_ctx": (i32, [vp]),
    "thk_pp_get_unique_id": (C.c_int, [vp]),
    "thk_pp_create": (C.c_int, [vp, C.c_int, C.c_int, vp, pp]),
    "thk_pp_destroy": (C.c_int, [vp]),
    "thk_pp_rank": (C.c_int, [vp]),
    "thk_pp_size": (C.c_int, [vp]),
    "thk_pp_group_begin": (C.c_int, [vp]),
    "thk_pp_group_end": (C.c_int, [vp]),
    "thk_pp_send": (C.c_int, [vp, vp, C.c_size_t, C.c_int]),
    "thk_pp_recv": (C.c_int, [vp, vp, C.c_size_t, C.c_int]),
    "thk_pp_send_hidden": (C.c_int, [vp, vp, i32, C.c_int]),
    "thk_pp_recv_hidden": (C.c_int, [vp, vp, i32, C.c_int]),
    "thk_pp_send_token": (C.c_int, [vp, vp, i32, C.c_int]),
    "thk_pp_recv_token": (C.c_int, [vp, vp, i32, C.c_int]),
    "thk_peer_create": (C.c_int, [vp, vp, i32, pp]),
    "thk_peer_export": (C.c_int, [vp, vp]),
    "thk_peer_connect": (C.c_int, [vp, vp]),
    "thk_peer_send": (C.c_int, [vp, i32, C.c_int]),
    "thk_peer_recv": (C.c_int, [vp, i32, C.c_int]),
    "thk_peer_send_bulk": (C.c_int, [vp, i32, vp, C.c_size_t]),
    "thk_peer_recv_bulk": (C.c_int, [vp, i32, vp, C.c_size_t]),
    "thk_peer_check": (C.c_int, [vp]),
    "thk_peer_memory_kind": (C.c_int, [vp]),
    "thk_peer_destroy": (C.c_int, [vp]),
    "thk_set_tunable": (C.c_int, [vp, C.c_char_p, i64]),
    "thk_get_tunable": (C.c_int, [vp, C.c_char_p, C.POINTER(i64)]),
}

_lib = None


def load() -> C.CDLL:
    """dlopen libthk.so and bind every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                "There is no CPU fallback for the product path.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib
