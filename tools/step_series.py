#!/usr/bin/env python3
"""Per-step durations of a long run of graph-replayed decode steps, from the device-side step clock (thk_model_seq_clock).
usage: python tools/step_series.py [7b|13b] [n_steps]   -> prints the series in ms (10 per line) and its percentiles"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

thk = graft.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "7b"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 640
shape = {"7b": thk.LLAMA_7B, "13b": thk.LLAMA_13B}[name]
with thk.Context(0) as ctx:
    m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
    m.seq_set(0, 5, shape.n_ctx - 1)
    m.prepare_steps(n)
    m.decode_steps(32, 0, advance=False); ctx.sync()
    for rep in range(2):
        m.seq_set(0, 5, shape.n_ctx - 1)
        t0 = time.perf_counter()
        m.decode_steps(n, 0, advance=False); ctx.sync()
        wall = (time.perf_counter() - t0) / n * 1e3
        ms = np.diff(m.seq_clock(0).astype(np.int64)) * 1e-5
        print(f"rep {rep}: wall {wall:.4f} ms/step; device clock: mean {ms.mean():.4f} p5 {np.percentile(ms, 5):.4f} p50 {np.percentile(ms, 50):.4f} p95 {np.percentile(ms, 95):.4f}")
        for i in range(0, ms.size, 16):
            print(f"{i:4d}: " + " ".join(f"{v:.3f}" for v in ms[i:i + 16]))
        time.sleep(0.5 if rep == 0 else 0)
    m.close()
