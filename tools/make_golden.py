#!/usr/bin/env python3
"""Generate tests/golden/*.npz.

The reference cannot be compiled or run in this image (needs Dawn's <webgpu/webgpu.h>;
see oracle/thk_oracle.c header), and it ships no golden vectors, so these fixtures are
produced by the oracle restatement itself plus independent numpy implementations.  They
pin the oracle against regressions and travel to the GPU box as plain data; they do NOT
pin it against a real WebGPU run ("parity unpinned").

Run from the repo root:  python tools/make_golden.py
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def crc(a: np.ndarray) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def main():
    rng = np.random.default_rng(20230517)
    g = {}
    # ---- A17 fp16 <-> fp32: independent numpy float16 is the expected side
    h = np.arange(65536, dtype=np.uint16)
    ref = h.view(np.float16).astype(np.float32)
    finite = np.isfinite(ref)
    g["fp16_all_finite_crc"] = np.uint32(crc(ref[finite].view(np.uint32)))
    xs = np.concatenate([rng.standard_normal(4096).astype(np.float32) * s for s in (1e-8, 1e-5, 1e-3, 1.0, 100.0, 7e4)])
    xs = np.concatenate([xs, np.array([0.0, -0.0, 65504.0, 65520.0, 1e9, -1e9, 5.96e-8, 2.98e-8, 6.1e-5], np.float32)])
    with np.errstate(over="ignore"):
        g["fp32_samples"] = xs
        g["fp32_to_fp16_expected"] = xs.astype(np.float16).view(np.uint16)
    # ---- synthetic generator
    for name in ("tok_embeddings.weight", "layers.0.attention.wq.weight", "layers.31.feed_forward.w2.weight", "output.weight"):
        v = O.synth_f16(name, O.TENSOR_SEED, O.TENSOR_SIGMA, 1 << 16)
        g["synth16_head_" + name] = v[:32].copy()
        g["synth16_crc_" + name] = np.uint32(crc(v))
    gn = O.synth_gain("norm.weight", O.TENSOR_SEED, O.TENSOR_SIGMA, 4096)
    g["synth_gain_head"] = gn[:32].copy()
    g["synth_gain_crc"] = np.uint32(crc(gn))
    # ---- per-kernel vectors (faithful summation order)
    a = rng.standard_normal(512).astype(np.float32)
    W = (rng.standard_normal((8, 512)) * 0.05).astype(np.float16).view(np.uint16)
    g["k1_a"], g["k1_W"], g["k1_out"] = a, W, O.vector_mat_mul_trans(a, W, True)
    x = rng.standard_normal((2, 512)).astype(np.float32)
    g["k4_in"], g["k4_out"] = x, O.rms_norm(x)
    gain = (1 + 0.1 * rng.standard_normal(512)).astype(np.float32)
    g["k5_gain"], g["k5_out"] = gain, O.row_element_multiply(x, gain)
    r = rng.standard_normal((3, 8, 64)).astype(np.float32)
    g["k6_in"], g["k6_out_past5"] = r, O.rope(r, 5)
    for T in (1, 5, 300):
        s = (rng.standard_normal((4, T)) * 3).astype(np.float32)
        g[f"k10_in_T{T}"], g[f"k10_out_T{T}"] = s, O.row_softmax(s)
    A = rng.standard_normal((2, 1, 64)).astype(np.float32)
    B = rng.standard_normal((2, 13, 64)).astype(np.float32)
    g["k9_A"], g["k9_Bt"], g["k9_out_t"] = A, B, O.mat_mul(A, B, True, 0.125)
    P = rng.random((2, 1, 13)).astype(np.float32)
    Vv = rng.standard_normal((2, 13, 64)).astype(np.float32)
    g["k9_P"], g["k9_V"], g["k9_out_n"] = P, Vv, O.mat_mul(P, Vv, False, 1.0)
    u = rng.standard_normal(1536).astype(np.float32)
    g["k12_in"], g["k12_out"] = u, O.silu(u)
    # ---- Q1 index set (lm-head combine defect) for V=32000
    g["q1_skipped_V32000"] = O.q1_skipped_indices(32000).astype(np.int32)
    # ---- whole-model logits, tiny models, prompts of 1, 2 and 17 tokens
    prompts = {"p1": [1], "p2": [1, 77], "p17": [1] + [int(t) for t in rng.integers(3, 2048, 16)]}
    for key, toks in prompts.items():
        g["tiny_prompt_" + key] = np.array(toks, np.int32)
        m = O.OracleModel(O.TINY); m.fill_synthetic()
        for i, t in enumerate(toks):
            lg, _ = m.eval(t, i, flags=O.FAITHFUL_ORDER)
        g["tiny_logits_" + key] = lg
        m.close()
    toks = [1, 1234, 31999]
    for mode, flag in (("correct", 0), ("faithful", O.LM_FAITHFUL)):
        m = O.OracleModel(O.TINY_Q1); m.fill_synthetic()
        for i, t in enumerate(toks):
            lg, _ = m.eval(t, i, flags=O.FAITHFUL_ORDER | flag)
        g["tinyq1_logits_" + mode] = lg
        m.close()
    g["tinyq1_prompt"] = np.array(toks, np.int32)
    np.savez_compressed(os.path.join(OUT, "oracle_golden.npz"), **g)
    print("wrote", os.path.join(OUT, "oracle_golden.npz"), len(g), "arrays")


if __name__ == "__main__":
    main()
