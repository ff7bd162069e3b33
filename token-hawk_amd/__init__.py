"""token-hawk_amd — Python face of libthk (MI355X-native TokenHawk decode path).

Thin object wrappers over the C-ABI in include/thk.h for tests/, bench.py and the
pipeline driver.  All compute happens in libthk.so (hand-written HIP for gfx950); nothing
here computes on the CPU and there is no fallback — constructing a Context without a
working HIP device raises ThkError.

The directory name has a hyphen (layout contract), so import it through
`__graft_entry__.load_package()` which registers it as module `token_hawk_amd`.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _capi
from ._capi import (HParams, THK_F16, THK_F32, THK_LMHEAD_CORRECT, THK_LMHEAD_FAITHFUL, THK_OK, THK_STAGE_EMBED,
                    THK_STAGE_HEAD)

__all__ = ["Context", "Buffer", "Model", "ModelShape", "ThkError", "LLAMA_7B", "LLAMA_13B", "TINY", "TINY_Q1",
           "THK_LMHEAD_CORRECT", "THK_LMHEAD_FAITHFUL", "THK_STAGE_EMBED", "THK_STAGE_HEAD", "TENSOR_SEED", "TENSOR_SIGMA"]

TENSOR_SEED = 20230517   # synthetic-weight key
TENSOR_SIGMA = 0.02


class ThkError(RuntimeError):
    pass


@dataclass(frozen=True)
class ModelShape:
    """LlamaModel hyper-parameters (th-llama.hpp:103-112)."""
    n_vocab: int = 32000
    n_embd: int = 4096
    n_mult: int = 256
    n_head: int = 32
    n_layer: int = 32
    n_ctx: int = 512

    @property
    def n_ff(self) -> int:   # th-llama-loader.cpp:349
        return ((2 * (4 * self.n_embd) // 3 + self.n_mult - 1) // self.n_mult) * self.n_mult

    @property
    def head_dim(self) -> int:
        return self.n_embd // self.n_head

    def weight_bytes(self, l0: int = 0, l1: int | None = None, head: bool = True) -> int:
        l1 = self.n_layer if l1 is None else l1
        E, F, V = self.n_embd, self.n_ff, self.n_vocab
        return (l1 - l0) * (4 * E * E + 3 * E * F) * 2 + (V * E * 2 if head else 0)

    def bytes_per_token(self, T: int, l0: int = 0, l1: int | None = None, head: bool = True, kv_bytes: int = 4) -> int:
        """Algorithmic HBM bytes of one decode step: W + KVr(T) + KVw + G; kv_bytes = s_kv (4 = f32 cache
        as the reference, 2 = the optional binary16 cache)."""
        l1 = self.n_layer if l1 is None else l1
        E = self.n_embd
        nl = l1 - l0
        return self.weight_bytes(l0, l1, head) + nl * (2 * T * E * kv_bytes + 2 * E * kv_bytes + 2 * E * 4) + (E * 4 if head else 0)


LLAMA_7B = ModelShape()
LLAMA_13B = ModelShape(n_embd=5120, n_head=40, n_layer=40)
TINY = ModelShape(n_vocab=2048, n_embd=512, n_mult=256, n_head=8, n_layer=2, n_ctx=64)
TINY_Q1 = ModelShape(n_vocab=32000, n_embd=512, n_mult=256, n_head=8, n_layer=2, n_ctx=64)


def _ptr(x) -> int:
    if x is None:
        return 0
    if isinstance(x, Buffer):
        return x.ptr
    if isinstance(x, np.ndarray):
        raise TypeError("device pointer expected, got a host numpy array")
    return int(x)


class Context:
    """thk_ctx: one HIP device + one stream (replaces WGPUDevice/WGPUQueue)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self.lib = _capi.load()
        h = C.c_void_p()
        rc = self.lib.thk_ctx_create(device, C.byref(h)) if stream is None else \
            self.lib.thk_ctx_create_on_stream(device, C.c_void_p(stream), C.byref(h))
        if rc != THK_OK:
            raise ThkError(f"thk_ctx_create(device={device}) failed with status {rc}: no usable HIP device "
                           "(libthk has no CPU fallback)")
        self.h = h
        self.device = device

    def check(self, rc: int, what: str = ""):
        if rc != THK_OK:
            raise ThkError(f"{what} failed ({rc}): {self.lib.thk_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.thk_ctx_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def sync(self):
        self.check(self.lib.thk_sync(self.h), "thk_sync")

    @property
    def stream(self) -> int:
        return int(self.lib.thk_ctx_stream(self.h) or 0)

    def device_info(self):
        name = C.create_string_buffer(256)
        ncu, hbm = C.c_int(), C.c_size_t()
        self.check(self.lib.thk_ctx_device_info(self.h, name, 256, C.byref(ncu), C.byref(hbm)), "thk_ctx_device_info")
        return {"name": name.value.decode(), "n_cu": ncu.value, "hbm_bytes": hbm.value}

    def set_tunable(self, name: str, value: int):
        self.check(self.lib.thk_set_tunable(self.h, name.encode(), int(value)), f"thk_set_tunable({name})")

    def get_tunable(self, name: str) -> int:
        v = C.c_int64()
        self.check(self.lib.thk_get_tunable(self.h, name.encode(), C.byref(v)), f"thk_get_tunable({name})")
        return v.value

    # ---- buffers
    def alloc(self, nbytes: int) -> "Buffer":
        return Buffer(self, nbytes)

    def from_numpy(self, a: np.ndarray) -> "Buffer":
        a = np.ascontiguousarray(a)
        b = Buffer(self, a.nbytes)
        b.upload(a)
        return b

    # ---- operators (one per cmdbuf_* of th.hpp; pointers are device pointers / Buffers)
    def matvec_f16(self, W, R, Cc, x, y):
        self.check(self.lib.thk_matvec_f16(self.h, _ptr(W), R, Cc, _ptr(x), _ptr(y)), "thk_matvec_f16")

    def rms_norm(self, x, rows, N):
        self.check(self.lib.thk_rms_norm(self.h, _ptr(x), rows, N), "thk_rms_norm")

    def row_element_multiply(self, x, gain, rows, N):
        self.check(self.lib.thk_row_element_multiply(self.h, _ptr(x), _ptr(gain), rows, N), "thk_row_element_multiply")

    def rope(self, x, n_tok, H, D, n_past):
        self.check(self.lib.thk_rope(self.h, _ptr(x), n_tok, H, D, n_past), "thk_rope")

    def kv_append(self, kc, vc, k, v, pos, H, D):
        self.check(self.lib.thk_kv_append(self.h, _ptr(kc), _ptr(vc), _ptr(k), _ptr(v), pos, H, D), "thk_kv_append")

    def attn_decode(self, q, kc, vc, T, H, D, out):
        self.check(self.lib.thk_attn_decode(self.h, _ptr(q), _ptr(kc), _ptr(vc), T, H, D, _ptr(out)), "thk_attn_decode")

    def attn_prefill(self, q, kc, vc, n_past, M, H, D, out):
        self.check(self.lib.thk_attn_prefill(self.h, _ptr(q), _ptr(kc), _ptr(vc), n_past, M, H, D, _ptr(out)), "thk_attn_prefill")

    def row_softmax(self, x, rows, N):
        self.check(self.lib.thk_row_softmax(self.h, _ptr(x), rows, N), "thk_row_softmax")

    def add(self, a, b, c, n):
        self.check(self.lib.thk_add(self.h, _ptr(a), _ptr(b), _ptr(c), n), "thk_add")

    def silu(self, x, n):
        self.check(self.lib.thk_silu(self.h, _ptr(x), n), "thk_silu")

    def mul_inplace(self, a, b, n):
        self.check(self.lib.thk_mul_inplace(self.h, _ptr(a), _ptr(b), n), "thk_mul_inplace")

    def lmhead_f16(self, W, V, E, x, logits, mode=THK_LMHEAD_CORRECT):
        self.check(self.lib.thk_lmhead_f16(self.h, _ptr(W), V, E, _ptr(x), _ptr(logits), mode), "thk_lmhead_f16")

    def argmax(self, logits, V, id_out):
        self.check(self.lib.thk_argmax(self.h, _ptr(logits), V, _ptr(id_out)), "thk_argmax")

    def topk_f32(self, logits, V, k):
        """The k largest of V device logits as (values f32[k], ids int32[k]); value descending, ties by ascending id."""
        vals, ids = np.empty(k, np.float32), np.empty(k, np.int32)
        self.check(self.lib.thk_topk_f32(self.h, _ptr(logits), V, k, vals.ctypes.data, ids.ctypes.data), "thk_topk_f32")
        return vals, ids

    def embed_f16(self, table, E, token, x):
        self.check(self.lib.thk_embed_f16(self.h, _ptr(table), E, token, _ptr(x)), "thk_embed_f16")

    def gemm_f16_prefill(self, W, R, Cc, X, M, Y):
        self.check(self.lib.thk_gemm_f16_prefill(self.h, _ptr(W), R, Cc, _ptr(X), M, _ptr(Y)), "thk_gemm_f16_prefill")

    def synth_f16(self, name: str, n: int, out, seed=TENSOR_SEED, sigma=TENSOR_SIGMA):
        self.check(self.lib.thk_synth_f16(self.h, name.encode(), seed, sigma, n, _ptr(out)), "thk_synth_f16")

    def synth_gain_f32(self, name: str, n: int, out, seed=TENSOR_SEED, sigma=TENSOR_SIGMA):
        self.check(self.lib.thk_synth_gain_f32(self.h, name.encode(), seed, sigma, n, _ptr(out)), "thk_synth_gain_f32")


class Buffer:
    """thk_buf: device allocation (TensorBuffer's GPU half, th.hpp:83-148)."""

    def __init__(self, ctx: Context, nbytes: int):
        self.ctx = ctx
        h = C.c_void_p()
        ctx.check(ctx.lib.thk_buf_alloc(ctx.h, nbytes, C.byref(h)), f"thk_buf_alloc({nbytes})")
        self.h = h
        self.nbytes = nbytes

    @property
    def ptr(self) -> int:
        return int(self.ctx.lib.thk_buf_ptr(self.h) or 0)

    def upload(self, a: np.ndarray, offset: int = 0):
        a = np.ascontiguousarray(a)
        self.ctx.check(self.ctx.lib.thk_buf_upload(self.ctx.h, self.h, offset, a.ctypes.data, a.nbytes), "thk_buf_upload")

    def download(self, dtype, shape, offset: int = 0) -> np.ndarray:
        out = np.empty(shape, dtype)
        self.ctx.check(self.ctx.lib.thk_buf_download(self.ctx.h, self.h, offset, out.ctypes.data, out.nbytes), "thk_buf_download")
        return out

    def free(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.ctx.lib.thk_buf_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Model:
    """thk_model: device state of a contiguous layer range (a pipeline stage)."""

    def __init__(self, ctx: Context, shape: ModelShape, l0: int = 0, l1: int | None = None,
                 flags: int = THK_STAGE_EMBED | THK_STAGE_HEAD, n_seq: int = 1):
        self.ctx, self.shape = ctx, shape
        self.l0, self.l1 = l0, shape.n_layer if l1 is None else l1
        self.flags, self.n_seq = flags, n_seq
        hp = HParams(shape.n_vocab, shape.n_embd, shape.n_mult, shape.n_head, shape.n_layer, shape.n_ctx)
        h = C.c_void_p()
        ctx.check(ctx.lib.thk_model_create(ctx.h, C.byref(hp), self.l0, self.l1, flags, n_seq, C.byref(h)), "thk_model_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.ctx.lib.thk_model_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_tensor(self, name: str, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        dtype = THK_F16 if arr.dtype in (np.uint16, np.float16) else THK_F32
        ne0 = arr.shape[-1]
        ne1 = arr.shape[0] if arr.ndim == 2 else 1
        self.ctx.check(self.ctx.lib.thk_model_set_tensor(self.h, name.encode(), dtype, ne0, ne1, arr.ctypes.data),
                       f"thk_model_set_tensor({name})")

    def get_tensor(self, name: str, dtype, count: int, offset: int = 0) -> np.ndarray:
        """`count` elements of a tensor starting at element `offset`, read back from the device (thk_model_get_tensor)."""
        out = np.empty(count, dtype)
        self.ctx.check(self.ctx.lib.thk_model_get_tensor(self.h, name.encode(), offset * out.itemsize, out.nbytes, out.ctypes.data), "thk_model_get_tensor")
        return out

    def fill_synthetic(self, seed: int = TENSOR_SEED, sigma: float = TENSOR_SIGMA):
        self.ctx.check(self.ctx.lib.thk_model_fill_synthetic(self.h, seed, sigma), "thk_model_fill_synthetic")

    def set_lmhead_mode(self, mode: int):
        self.ctx.check(self.ctx.lib.thk_model_set_lmhead_mode(self.h, mode), "thk_model_set_lmhead_mode")

    def finalize(self):
        self.ctx.check(self.ctx.lib.thk_model_finalize(self.h), "thk_model_finalize")

    def reset_kv(self, seq: int = 0):
        self.ctx.check(self.ctx.lib.thk_model_reset_kv(self.h, seq), "thk_model_reset_kv")

    def eval(self, tokens, n_past: int, *, seq: int = 0, hidden: np.ndarray | None = None, want_logits: bool = True,
             want_hidden: bool = False):
        """th_eval_gpu (th-llama.cpp:464): returns (logits|None, hidden|None)."""
        s = self.shape
        toks = None if tokens is None else np.ascontiguousarray(tokens, np.int32)
        n = 1 if toks is None else toks.size
        hid = None
        if hidden is not None:
            hid = np.array(hidden, np.float32, copy=True)
        elif want_hidden:
            hid = np.zeros(s.n_embd, np.float32)
        logits = np.empty(s.n_vocab, np.float32) if want_logits else None
        self.ctx.check(self.ctx.lib.thk_model_eval(self.h, seq, None if toks is None else toks.ctypes.data, n, n_past,
                                                   None if hid is None else hid.ctypes.data,
                                                   None if logits is None else logits.ctypes.data), "thk_model_eval")
        return logits, hid

    def prefill(self, tokens, n_past: int = 0, *, seq: int = 0, want_logits: bool = True):
        """Batched prompt prefill on the MFMA GEMM path (config C3); returns the last token's logits."""
        toks = np.ascontiguousarray(tokens, np.int32)
        logits = np.empty(self.shape.n_vocab, np.float32) if want_logits else None
        self.ctx.check(self.ctx.lib.thk_model_prefill(self.h, seq, toks.ctypes.data, toks.size, n_past,
                                                      None if logits is None else logits.ctypes.data), "thk_model_prefill")
        return logits

    def prefill_stage(self, tokens, hidden, n_tokens: int, n_past: int = 0, *, seq: int = 0, want_logits: bool = False):
        """thk_model_prefill_stage: this stage's layers over the prompt rows; hidden = device buffer f32 [n_tokens, n_embd] (Buffer or address),
        used in place (input unless this is the embedding stage, output unless this is the head stage)."""
        toks = None if tokens is None else np.ascontiguousarray(tokens, np.int32)
        logits = np.empty(self.shape.n_vocab, np.float32) if want_logits else None
        self.ctx.check(self.ctx.lib.thk_model_prefill_stage(self.h, seq, None if toks is None else toks.ctypes.data, _ptr(hidden) or None, n_tokens, n_past,
                                                            None if logits is None else logits.ctypes.data), "thk_model_prefill_stage")
        return logits

    def eval_topk(self, tokens, n_past: int, k: int, seq: int = 0):
        """thk_model_eval_topk: the step(s) and the k largest logits of the last token in one stream round trip -> (values, ids), value descending."""
        toks = np.ascontiguousarray(tokens, np.int32)
        vals, ids = np.empty(k, np.float32), np.empty(k, np.int32)
        self.ctx.check(self.ctx.lib.thk_model_eval_topk(self.h, seq, toks.ctypes.data, toks.size, n_past, k, vals.ctypes.data, ids.ctypes.data), "thk_model_eval_topk")
        return vals, ids

    def logits_topk(self, k: int, seq: int = 0):
        vals, ids = np.empty(k, np.float32), np.empty(k, np.int32)
        self.ctx.check(self.ctx.lib.thk_model_logits_topk(self.h, seq, k, vals.ctypes.data, ids.ctypes.data), "thk_model_logits_topk")
        return vals, ids

    def read_logits(self, seq: int = 0):
        out = np.empty(self.shape.n_vocab, np.float32)
        self.ctx.check(self.ctx.lib.thk_model_read_logits(self.h, seq, out.ctypes.data), "thk_model_read_logits")
        return out

    def prepare_prefill(self):
        """Build the prefill workspace and the tile images of the layer matrices now (otherwise the first prefill() does)."""
        self.ctx.check(self.ctx.lib.thk_model_prepare_prefill(self.h), "thk_model_prepare_prefill")

    def prefill_uses_tile_images(self) -> bool:
        return bool(self.ctx.lib.thk_model_prefill_uses_tile_images(self.h))

    def seq_set(self, seq: int, token: int, pos: int):
        self.ctx.check(self.ctx.lib.thk_model_seq_set(self.h, seq, token, pos), "thk_model_seq_set")

    def seq_set_token(self, seq: int, token: int):
        self.ctx.check(self.ctx.lib.thk_model_seq_set_token(self.h, seq, token), "thk_model_seq_set_token")

    def decode_step(self, seq: int = 0, advance: bool = True):
        self.ctx.check(self.ctx.lib.thk_model_decode_step(self.h, seq, int(advance)), "thk_model_decode_step")

    def decode_steps(self, n_steps: int, seq: int = 0, advance: bool = True):
        self.ctx.check(self.ctx.lib.thk_model_decode_steps(self.h, seq, n_steps, int(advance)), "thk_model_decode_steps")

    def prepare_steps(self, n_steps: int, seq: int = 0):
        """Capture the multi-step graphs decode_steps(n_steps) replays, without running anything."""
        self.ctx.check(self.ctx.lib.thk_model_prepare_steps(self.h, seq, n_steps), "thk_model_prepare_steps")

    def engine_trace(self):
        """Development timeline of the last engine step: uint64 array [n_cu, n_ops, 8] (tunable engine_trace=1)."""
        cap = 256 * 512 * 8 * 2
        out = np.zeros(cap, np.uint64)
        ncu, nops = C.c_int32(), C.c_int32()
        self.ctx.check(self.ctx.lib.thk_model_engine_trace(self.h, out.ctypes.data, cap, C.byref(ncu), C.byref(nops)), "thk_model_engine_trace")
        return out[:ncu.value * nops.value * 8].reshape(ncu.value, nops.value, 8)

    def step_trace(self, seq: int = 0):
        """Development timeline of one eager decode step (libthk_trace.so): (names, uint64 [n_kernels, blocks, 8 waves, 4] of 100 MHz stamps)."""
        cap = (6 * (self.l1 - self.l0) + 8) * 2048 * 32
        out = np.zeros(cap, np.uint64)
        names = ((C.c_char * 48) * 512)()
        nk, nb = C.c_int32(), C.c_int32()
        self.ctx.check(self.ctx.lib.thk_model_step_trace(self.h, seq, out.ctypes.data, cap, 512, names, C.byref(nk), C.byref(nb)), "thk_model_step_trace")
        return [names[i].value.decode() for i in range(nk.value)], out[:nk.value * nb.value * 32].reshape(nk.value, nb.value, 8, 4)

    def uses_engine(self) -> bool:
        """True when decode steps run as one persistent loader/consumer launch (thk_engine.hip)."""
        return bool(self.ctx.lib.thk_model_uses_engine(self.h))

    def seq_last_token(self, seq: int = 0) -> int:
        """The sequence's current token (the last greedy pick): a 4-byte read-back after a stream synchronisation."""
        t = C.c_int32()
        self.ctx.check(self.ctx.lib.thk_model_seq_last_token(self.h, seq, C.byref(t)), "thk_model_seq_last_token")
        return int(t.value)

    def debug_buffer(self, name: str):
        """Development aid: a working buffer the last decode step left behind ("x", "q", "u", "part_o", "part_ml")."""
        out = np.empty(1 << 20, np.float32)
        n = C.c_int64()
        self.ctx.check(self.ctx.lib.thk_model_debug_buffer(self.h, name.encode(), out.ctypes.data, out.size, C.byref(n)), "thk_model_debug_buffer")
        return out[:n.value].copy()

    def seq_get(self, seq: int = 0, cap: int = 4096):
        out = np.empty(cap, np.int32)
        n, pos = C.c_int32(), C.c_int32()
        self.ctx.check(self.ctx.lib.thk_model_seq_get(self.h, seq, out.ctypes.data, cap, C.byref(n), C.byref(pos)), "thk_model_seq_get")
        return out[:min(n.value, cap)].copy(), n.value, pos.value

    def seq_clock(self, seq: int = 0, cap: int = 4096):
        """uint64 array: the 100 MHz device counter at the end of each logged step (see thk_model_seq_clock)."""
        out = np.zeros(cap, np.uint64)
        n = C.c_int32()
        self.ctx.check(self.ctx.lib.thk_model_seq_clock(self.h, seq, out.ctypes.data, cap, C.byref(n)), "thk_model_seq_clock")
        return out[:n.value].copy()

    def hidden_in_ptr(self, seq: int = 0) -> int:
        return int(self.ctx.lib.thk_model_hidden_in(self.h, seq) or 0)

    def hidden_out_ptr(self, seq: int = 0) -> int:
        return int(self.ctx.lib.thk_model_hidden_out(self.h, seq) or 0)

    def token_dev_ptr(self, seq: int = 0) -> int:
        return int(self.ctx.lib.thk_model_token_dev(self.h, seq) or 0)

    def logits_dev_ptr(self, seq: int = 0) -> int:
        return int(self.ctx.lib.thk_model_logits_dev(self.h, seq) or 0)

    def bytes_per_token(self, T: int) -> int:
        return int(self.ctx.lib.thk_model_bytes_per_token(self.h, T))

    def profile_step(self, seq: int = 0, max_entries: int = 512):
        names = ((C.c_char * 48) * max_entries)()
        ms = (C.c_float * max_entries)()
        n = C.c_int32()
        self.ctx.check(self.ctx.lib.thk_model_profile_step(self.h, seq, max_entries, names, ms, C.byref(n)), "thk_model_profile_step")
        return [(names[i].value.decode(), float(ms[i])) for i in range(n.value)]
