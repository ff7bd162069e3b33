#!/usr/bin/env python3
"""Config C3: LLaMA-7B f16, 128-token prompt prefill on one MI355X through the MFMA GEMM path.
Prints one JSON line: prefill tokens/s, TFLOP/s (2*weights flops, hi/lo activation split counted
once), and the same prompt fed token-by-token through the decode path for comparison."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

thk = graft.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "7b"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 128
shape = {"7b": thk.LLAMA_7B, "13b": thk.LLAMA_13B}[name]
rng = np.random.default_rng(128)
toks = np.concatenate([[1], rng.integers(3, shape.n_vocab, M - 1)]).astype(np.int32)
with thk.Context(0) as ctx:
    for kv in sys.argv[4:]:                      # tunables: name=value
        k, v = kv.split("="); ctx.set_tunable(k, int(v))
    m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
    lp = m.prefill(toks, 0)                      # warm-up (allocates the workspace)
    ts = []
    for _ in range(5):
        m.reset_kv(0); ctx.sync()
        t0 = time.perf_counter(); lp = m.prefill(toks, 0); ts.append(time.perf_counter() - t0)
    if len(sys.argv) > 3 and sys.argv[3] == "prefill-only":      # profiling runs: only the prefill kernels
        ld, t_dec = lp, float("nan")
    else:
        m.reset_kv(0); ctx.sync()
        t0 = time.perf_counter(); ld, _ = m.eval(toks, 0); t_dec = time.perf_counter() - t0
    t = float(np.median(ts))
    flops = 2.0 * (shape.weight_bytes(head=False) / 2) * M + 2.0 * shape.n_vocab * shape.n_embd
    print(json.dumps({"workload": f"LLaMA-{name.upper()} f16, {M}-token prompt prefill, 1 GPU", "prefill_ms": round(t * 1e3, 3),
                      "prefill_tokens_per_s": round(M / t, 1), "tflops": round(flops / t / 1e12, 2),
                      "mfma_peak_tflops_f16_dense": 2500, "frac_of_mfma_peak": round(flops / t / 2.5e15, 4),
                      "weight_pass_hbm_ms_at_8TBs": round(shape.weight_bytes() / 8e12 * 1e3, 3),
                      "token_by_token_decode_ms": round(t_dec * 1e3, 2), "speedup_vs_decode_path": round(t_dec / t, 1),
                      "max_abs_logit_diff_vs_decode": float(np.abs(lp - ld).max())}))
    m.close()
