#!/usr/bin/env python3
"""Coordinate descent on the per-kernel launch geometry, scored by what bench.py measures: graph-replay tokens/s at
n_past = 511.  Usage: python tools/sweep_graph.py [7b|13b]   (writes gpurun_out/sweep_graph_<model>.json)"""
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
shape = {"7b": thk.LLAMA_7B, "13b": thk.LLAMA_13B}[name]
T = 512
KERNS = ["qkv", "wo", "w13", "w2", "head"]
log = []
with thk.Context(0) as ctx:
    m = thk.Model(ctx, shape); m.fill_synthetic()
    rng = np.random.default_rng(0)

    def score(steps=60, warm=8, reps=3):
        m.finalize()
        m.eval(rng.integers(3, 32000, 4).astype(np.int32), 0, want_logits=False)
        m.seq_set(0, 5, T - 1)
        best = 0.0
        for _ in range(reps):
            for _ in range(warm):
                m.decode_step(0, advance=False)
            ctx.sync(); t0 = time.perf_counter()
            for _ in range(steps):
                m.decode_step(0, advance=False)
            ctx.sync()
            best = max(best, steps / (time.perf_counter() - t0))
        return best

    cur = {k: (-1, -1) for k in KERNS}          # (bpc, variant); -1 = library default
    base = score(); log.append({"config": "defaults", "tok_s": round(base, 2)}); print("defaults", round(base, 2), flush=True)
    best_score = base
    for rnd in range(2):
        for k in KERNS:
            for bpc in (2, 3, 4, 6, 8):
                for var in (0, 1, 2, 3):
                    ctx.set_tunable("gemv_bpc_" + k, bpc); ctx.set_tunable("gemv_variant_" + k, var)
                    try:
                        s = score()
                    except Exception as e:
                        print("failed", k, bpc, var, e, flush=True); continue
                    log.append({"round": rnd, "kernel": k, "bpc": bpc, "var": var, "tok_s": round(s, 2)})
                    if s > best_score * 1.002:
                        best_score = s; cur[k] = (bpc, var); print("  better:", k, bpc, var, round(s, 2), flush=True)
            ctx.set_tunable("gemv_bpc_" + k, cur[k][0]); ctx.set_tunable("gemv_variant_" + k, cur[k][1])
    for sp in (2, 4, 8):
        for w in (4, 8):
            ctx.set_tunable("attn_splits", sp); ctx.set_tunable("attn_waves", w)
            s = score(); log.append({"attn_splits": sp, "attn_waves": w, "tok_s": round(s, 2)}); print("attn", sp, w, round(s, 2), flush=True)
    m.close()
out = {"model": name, "defaults_tok_s": round(base, 2), "best_tok_s": round(best_score, 2), "best": {k: {"bpc": v[0], "var": v[1]} for k, v in cur.items()}, "log": log}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"sweep_graph_{name}.json"), "w"), indent=1)
print("best", out["best"], out["best_tok_s"], "vs defaults", out["defaults_tok_s"])
