#!/usr/bin/env python3
"""Decode at a long context: LLaMA-7B with an n_ctx-row cache, hold-position steps at n_past = T - 1 (the cache contents do not
matter for timing), per config: ms per step (graph replay) and the attention launch's eager average.

    python tools/attn_ctx_sweep.py N_CTX [T] CONFIG [CONFIG ...]      CONFIG = base | name=value[,name=value]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

thk = graft.load_package()
n_ctx = int(sys.argv[1])
rest = sys.argv[2:]
T = n_ctx
if rest and rest[0].isdigit():
    T = int(rest[0]); rest = rest[1:]
shape = thk.ModelShape(n_ctx=n_ctx)
with thk.Context(0) as ctx:
    m = thk.Model(ctx, shape); m.fill_synthetic()
    defaults = {}
    for cfg in rest:
        kv = {} if cfg == "base" else dict(p.split("=") for p in cfg.split(","))
        for k in list(defaults):
            ctx.set_tunable(k, defaults[k])
        for k, v in kv.items():
            defaults.setdefault(k, ctx.get_tunable(k)); ctx.set_tunable(k, int(v))
        m.finalize()
        m.seq_set(0, 5, T - 1)
        m.prepare_steps(64)
        m.decode_steps(16, 0, advance=False); ctx.sync()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); m.decode_steps(64, 0, advance=False); ctx.sync(); ts.append((time.perf_counter() - t0) / 64 * 1e3)
        agg = {}
        for _ in range(4):
            for name, ms in m.profile_step(0):
                a = agg.setdefault(name, [0.0, 0]); a[0] += ms; a[1] += 1
        att = agg["attn_decode"][0] / agg["attn_decode"][1] * 1e3
        kvb = 2 if ctx.get_tunable("kv_f16") else 4
        byt = 2 * T * shape.n_embd * kvb
        print(json.dumps({"n_ctx": n_ctx, "T": T, "config": cfg, "ms_per_step": round(sorted(ts)[2], 4), "tok_s": round(1e3 / sorted(ts)[2], 2),
                          "attn_eager_us": round(att, 2), "attn_eager_gbs": round(byt / att / 1e3, 1), "attn_bytes": byt}), flush=True)
    m.close()
