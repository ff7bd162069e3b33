#!/usr/bin/env python3
"""Development probe: a 128-token (or M-token) prefill of a synthetic model of arbitrary width, for per-GEMM timing under rocprofv3
(tools/dev/gemm_split.py).  Usage: gemm_shapes.py E n_mult L [M] [tunable=value ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
thk = graft.load_package()
E, n_mult, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
M = int(sys.argv[4]) if len(sys.argv) > 4 and "=" not in sys.argv[4] else 128
shape = thk.ModelShape(n_vocab=32000, n_embd=E, n_mult=n_mult, n_head=E // 128, n_layer=L, n_ctx=512)
toks = np.concatenate([[1], np.random.default_rng(1).integers(3, 32000, M - 1)]).astype(np.int32)
with thk.Context(0) as ctx:
    for kv in sys.argv[4:]:
        if "=" in kv:
            k, v = kv.split("="); ctx.set_tunable(k, int(v))
    m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
    ts = []
    for _ in range(8):
        m.reset_kv(0); ctx.sync()
        t0 = time.perf_counter(); m.prefill(toks, 0); ts.append(time.perf_counter() - t0)
    print("E %d F %d L %d M %d: prefill median %.3f ms" % (E, shape.n_ff if hasattr(shape, "n_ff") else -1, L, M, float(np.median(ts)) * 1e3))
