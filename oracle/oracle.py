"""ctypes binding of the CPU oracle (oracle/thk_oracle.c).

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package never imports it.
Parity is UNPINNED against a real Dawn/WebGPU run (see thk_oracle.c header).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libthk_oracle.so")

FAITHFUL_ORDER = 1  # reference strip+tree summation order, transposes, K9 tiles
LM_FAITHFUL = 2     # reproduce lm-head combine defect Q1
KV_F16 = 4          # K/V rounded to binary16 as they are appended (checker for the build's optional f16 KV cache)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (Makefile in this directory)."""
    src = os.path.join(_HERE, "thk_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libthk_oracle.so"])
    return _LIB_PATH


class HParams(C.Structure):
    _fields_ = [("n_vocab", C.c_int32), ("n_embd", C.c_int32), ("n_mult", C.c_int32),
                ("n_head", C.c_int32), ("n_layer", C.c_int32), ("n_ctx", C.c_int32)]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        f32p, u16p, vp = C.POINTER(C.c_float), C.POINTER(C.c_uint16), C.c_void_p
        i64, i32 = C.c_int64, C.c_int32
        sig = {
            "orc_fp16_to_fp32": (C.c_float, [C.c_uint16]),
            "orc_fp32_to_fp16": (C.c_uint16, [C.c_float]),
            "orc_fp16_to_fp32_n": (None, [vp, vp, i64]),
            "orc_fp32_to_fp16_n": (None, [vp, vp, i64]),
            "orc_synth_key": (C.c_uint64, [C.c_char_p, C.c_uint64]),
            "orc_synth_scale": (C.c_float, [C.c_float]),
            "orc_synth_f16": (None, [C.c_char_p, C.c_uint64, C.c_float, i64, vp]),
            "orc_synth_gain_f32": (None, [C.c_char_p, C.c_uint64, C.c_float, i64, vp]),
            "orc_vector_mat_mul_trans": (C.c_int, [vp, vp, vp, i64, i64]),
            "orc_matvec_f16_fast": (None, [vp, vp, vp, i64, i64]),
            "orc_rms_norm": (C.c_int, [vp, i64, i64]),
            "orc_row_element_multiply": (None, [vp, vp, i64, i64]),
            "orc_rope_angles": (None, [i64, i64, vp, vp]),
            "orc_rope": (None, [vp, i64, i64, i64, i64]),
            "orc_transpose_zy": (None, [vp, vp, i64, i64, i64]),
            "orc_mat_mul": (None, [vp, vp, vp, i64, i64, i64, i64, C.c_int, C.c_int, C.c_float]),
            "orc_row_softmax": (None, [vp, i64, i64]),
            "orc_addition": (None, [vp, vp, vp, i64]),
            "orc_silu": (None, [vp, i64]),
            "orc_element_mult_in_place": (None, [vp, vp, i64]),
            "orc_lmhead_split": (C.c_int, [vp, vp, vp, vp, i64, i64]),
            "orc_q1_covered": (i64, [i64, i64]),
            "orc_vector_reduce": (None, [vp, vp, i64, C.c_int]),
            "orc_greedy": (i32, [vp, i64]),
            "orc_n_ff": (i32, [i32, i32]),
            "orc_model_create": (vp, [C.POINTER(HParams), i32]),
            "orc_model_destroy": (None, [vp]),
            "orc_model_set_tensor": (C.c_int, [vp, C.c_char_p, vp, i64]),
            "orc_model_get_tensor": (C.c_int, [vp, C.c_char_p, vp, i64]),
            "orc_model_fill_synthetic": (None, [vp, C.c_uint64, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]),
            "orc_model_reset_kv": (None, [vp, C.c_int]),
            "orc_model_kv_ptr": (f32p, [vp, C.c_int, C.c_int, C.c_int]),
            "orc_embed": (None, [vp, i32, vp]),
            "orc_model_eval": (C.c_int, [vp, C.c_int, i32, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int]),
            "orc_model_time_decode": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
            "orc_num_threads": (C.c_int, []),
            "orc_set_num_threads": (None, [C.c_int]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
        # Small models make thousands of tiny OpenMP regions; with one thread per visible CPU on a
        # 128/256-thread host (or under a cgroup CPU quota) fork/join dominates.  Default to a few
        # threads; bench.py's cpu_baseline raises it to the usable core count explicitly.
        L.orc_set_num_threads(min(usable_cpus(), 8))
    return _lib


def _p(a: np.ndarray) -> int:
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


# ---------------------------------------------------------------- scalar/array helpers
def fp16_to_fp32(h: np.ndarray) -> np.ndarray:
    h = np.ascontiguousarray(h, dtype=np.uint16)
    out = np.empty(h.shape, np.float32)
    lib().orc_fp16_to_fp32_n(_p(h), _p(out), h.size)
    return out


def fp32_to_fp16(f: np.ndarray) -> np.ndarray:
    f = np.ascontiguousarray(f, dtype=np.float32)
    out = np.empty(f.shape, np.uint16)
    lib().orc_fp32_to_fp16_n(_p(f), _p(out), f.size)
    return out


def synth_f16(name: str, seed: int, sigma: float, n: int) -> np.ndarray:
    out = np.empty(n, np.uint16)
    lib().orc_synth_f16(name.encode(), seed, sigma, n, _p(out))
    return out


def synth_gain(name: str, seed: int, sigma: float, n: int) -> np.ndarray:
    out = np.empty(n, np.float32)
    lib().orc_synth_gain_f32(name.encode(), seed, sigma, n, _p(out))
    return out


# ---------------------------------------------------------------- per-kernel restatements
def vector_mat_mul_trans(a: np.ndarray, b_f16: np.ndarray, faithful: bool = True) -> np.ndarray:
    R, Cc = b_f16.shape
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b_f16, np.uint16)
    out = np.empty(R, np.float32)
    if faithful:
        rc = lib().orc_vector_mat_mul_trans(_p(a), _p(b), _p(out), R, Cc)
        if rc != 0:
            raise ValueError("C must be a multiple of 256 and >= 256 (th.cpp:2996-3006)")
    else:
        lib().orc_matvec_f16_fast(_p(a), _p(b), _p(out), R, Cc)
    return out


def rms_norm(x: np.ndarray) -> np.ndarray:
    x = np.array(x, np.float32, ndmin=2, copy=True)
    if lib().orc_rms_norm(_p(x), x.shape[0], x.shape[1]) != 0:
        raise ValueError("N must be a multiple of 256 (th.cpp:1155)")
    return x


def row_element_multiply(x: np.ndarray, w: np.ndarray) -> np.ndarray:
    x = np.array(x, np.float32, ndmin=2, copy=True); w = np.ascontiguousarray(w, np.float32)
    lib().orc_row_element_multiply(_p(x), _p(w), x.shape[0], x.shape[1])
    return x


def rope(x: np.ndarray, n_past: int) -> np.ndarray:
    """x: [n_tok, H, D] f32."""
    x = np.array(x, np.float32, copy=True)
    n_tok, H, D = x.shape
    lib().orc_rope(_p(x), n_tok, H, D, n_past)
    return x


def rope_angles(D: int, pos: int):
    c = np.empty(D // 2, np.float32); s = np.empty(D // 2, np.float32)
    lib().orc_rope_angles(D, pos, _p(c), _p(s))
    return c, s


def transpose_zy(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, np.float32)
    B, M, N = a.shape
    out = np.empty((M, B, N), np.float32)
    lib().orc_transpose_zy(_p(a), _p(out), B, M, N)
    return out


def mat_mul(A: np.ndarray, B: np.ndarray, transpose_b: bool, scale: float | None = None) -> np.ndarray:
    A = np.ascontiguousarray(A, np.float32); B = np.ascontiguousarray(B, np.float32)
    Bz, M, K = A.shape
    N = B.shape[1] if transpose_b else B.shape[2]
    out = np.empty((Bz, M, N), np.float32)
    lib().orc_mat_mul(_p(A), _p(B), _p(out), Bz, M, K, N, int(transpose_b), int(scale is not None), float(scale or 0.0))
    return out


def row_softmax(a: np.ndarray) -> np.ndarray:
    a = np.array(a, np.float32, ndmin=2, copy=True)
    lib().orc_row_softmax(_p(a), a.shape[0], a.shape[1])
    return a


def addition(a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    out = np.empty_like(a); lib().orc_addition(_p(a), _p(b), _p(out), a.size); return out


def silu(a):
    a = np.array(a, np.float32, copy=True); lib().orc_silu(_p(a), a.size); return a


def element_mult(a, b):
    a = np.array(a, np.float32, copy=True); b = np.ascontiguousarray(b, np.float32)
    lib().orc_element_mult_in_place(_p(a), _p(b), a.size); return a


def lmhead(x: np.ndarray, W_f16: np.ndarray, lm_faithful: bool) -> np.ndarray:
    """K2 + K3 on the unsplit [V,E] matrix (reference summation order)."""
    V, E = W_f16.shape
    x = np.ascontiguousarray(x, np.float32); W = np.ascontiguousarray(W_f16, np.uint16)
    out = np.empty(V, np.float32); scratch = np.empty(V, np.float32)
    if lib().orc_lmhead_split(_p(x), _p(W), _p(out), _p(scratch), V, E) != 0:
        raise ValueError("E/2 must be a multiple of 256 (th.cpp:3728-3739)")
    lib().orc_vector_reduce(_p(out), _p(scratch), V, int(lm_faithful))
    return out


def q1_skipped_indices(V: int) -> np.ndarray:
    L = lib()
    return np.array([i for i in range(V) if not L.orc_q1_covered(V, i)], np.int64)


def greedy(logits: np.ndarray) -> int:
    logits = np.ascontiguousarray(logits, np.float32)
    return int(lib().orc_greedy(_p(logits), logits.size))


# ---------------------------------------------------------------- model
TENSOR_SEED = 20230517
TENSOR_SIGMA = 0.02


@dataclass
class ModelShape:
    n_vocab: int = 32000
    n_embd: int = 4096
    n_mult: int = 256
    n_head: int = 32
    n_layer: int = 32
    n_ctx: int = 512

    @property
    def n_ff(self) -> int:
        return ((2 * (4 * self.n_embd) // 3 + self.n_mult - 1) // self.n_mult) * self.n_mult

    def tensor_specs(self, l0: int = 0, l1: int | None = None):
        """(name, dtype, (ne1, ne0)) in ggjt file order (loader :410-426)."""
        E, F, V = self.n_embd, self.n_ff, self.n_vocab
        l1 = self.n_layer if l1 is None else l1
        yield ("tok_embeddings.weight", "f16", (V, E))
        yield ("norm.weight", "f32", (E,))
        yield ("output.weight", "f16", (V, E))
        for l in range(l0, l1):
            p = f"layers.{l}."
            yield (p + "attention.wq.weight", "f16", (E, E))
            yield (p + "attention.wk.weight", "f16", (E, E))
            yield (p + "attention.wv.weight", "f16", (E, E))
            yield (p + "attention.wo.weight", "f16", (E, E))
            yield (p + "attention_norm.weight", "f32", (E,))
            yield (p + "feed_forward.w1.weight", "f16", (F, E))
            yield (p + "feed_forward.w2.weight", "f16", (E, F))
            yield (p + "feed_forward.w3.weight", "f16", (F, E))
            yield (p + "ffn_norm.weight", "f32", (E,))


TINY = ModelShape(n_vocab=2048, n_embd=512, n_mult=256, n_head=8, n_layer=2, n_ctx=64)
TINY_Q1 = ModelShape(n_vocab=32000, n_embd=512, n_mult=256, n_head=8, n_layer=2, n_ctx=64)
LLAMA_7B = ModelShape()
LLAMA_13B = ModelShape(n_embd=5120, n_head=40, n_layer=40)


class OracleModel:
    def __init__(self, shape: ModelShape, n_seq: int = 1):
        self.shape = shape
        hp = HParams(shape.n_vocab, shape.n_embd, shape.n_mult, shape.n_head, shape.n_layer, shape.n_ctx)
        self._h = lib().orc_model_create(C.byref(hp), n_seq)
        self.n_seq = n_seq

    def close(self):
        if self._h:
            lib().orc_model_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def fill_synthetic(self, seed: int = TENSOR_SEED, sigma: float = TENSOR_SIGMA, l0: int = 0, l1: int | None = None,
                       with_embed: bool = True, with_head: bool = True):
        l1 = self.shape.n_layer if l1 is None else l1
        lib().orc_model_fill_synthetic(self._h, seed, sigma, l0, l1, int(with_embed), int(with_head))

    def set_tensor(self, name: str, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        if lib().orc_model_set_tensor(self._h, name.encode(), _p(arr), arr.size) != 0:
            raise KeyError(name)

    def get_tensor(self, name: str, dtype, shape) -> np.ndarray:
        out = np.empty(shape, dtype)
        if lib().orc_model_get_tensor(self._h, name.encode(), _p(out), out.size) != 0:
            raise KeyError(name)
        return out

    def reset_kv(self, seq: int = 0):
        lib().orc_model_reset_kv(self._h, seq)

    def kv(self, layer: int, seq: int, which: int) -> np.ndarray:
        """View of the f32 cache [n_ctx, H, D] (which: 0=key, 1=value)."""
        s = self.shape
        ptr = lib().orc_model_kv_ptr(self._h, layer, seq, which)
        return np.ctypeslib.as_array(ptr, shape=(s.n_ctx, s.n_head, s.n_embd // s.n_head))

    def eval(self, token: int, n_past: int, *, seq: int = 0, l0: int = 0, l1: int | None = None,
             hidden: np.ndarray | None = None, want_logits: bool = True, flags: int = FAITHFUL_ORDER):
        """One token (th_eval_gpu restated). Returns (logits|None, hidden_out)."""
        s = self.shape
        l1 = s.n_layer if l1 is None else l1
        hid = np.zeros(s.n_embd, np.float32) if hidden is None else np.array(hidden, np.float32, copy=True)
        logits = np.empty(s.n_vocab, np.float32) if want_logits else None
        rc = lib().orc_model_eval(self._h, seq, token if hidden is None else -1, n_past, l0, l1, _p(hid),
                                  _p(logits) if want_logits else None, flags)
        if rc != 0:
            raise RuntimeError(f"orc_model_eval rc={rc}")
        return logits, hid

    def time_decode(self, n_past: int, n_layers_sample: int, steps: int):
        a, b = C.c_double(), C.c_double()
        rc = lib().orc_model_time_decode(self._h, n_past, n_layers_sample, steps, C.byref(a), C.byref(b))
        if rc != 0:
            raise RuntimeError("oracle built without OpenMP")
        return a.value, b.value


def usable_cpus() -> int:
    """CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def set_num_threads(n: int):
    lib().orc_set_num_threads(int(n))


def num_threads() -> int:
    return int(lib().orc_num_threads())
