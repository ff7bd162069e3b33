#!/usr/bin/env python3
"""In-graph cost of every decode kernel: step time of the full model minus step time of a model whose graph omits
the kernel (tunable measure_skip_kernel), per launch.  Usage: python tools/marginal_costs.py [7b|13b]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

thk = graft.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "7b"
shape = {"7b": thk.LLAMA_7B, "13b": thk.LLAMA_13B}[name]
T, STEPS = 512, 200
NAMES = {0: "full step", 1: "norm_qkv_rope_kv", 2: "attn_decode", 3: "attn_wo_resid", 4: "norm_w13_swiglu", 5: "w2_resid", 6: "norm_lmhead"}
res = {}
with thk.Context(0) as ctx:
    for skip in range(7):
        ctx.set_tunable("measure_skip_kernel", skip)
        m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
        m.seq_set(0, 5, T - 1)
        best = 1e9
        for _ in range(3):
            m.decode_steps(20, 0, advance=False); ctx.sync()
            t0 = time.perf_counter(); m.decode_steps(STEPS, 0, advance=False); ctx.sync()
            best = min(best, (time.perf_counter() - t0) / STEPS * 1e6)
        res[skip] = best
        m.close()
full = res[0]
out = {"model": name, "full_step_us": round(full, 1)}
tot = 0.0
for skip in range(1, 7):
    n = 1 if skip == 6 else shape.n_layer
    out[NAMES[skip]] = round((full - res[skip]) / n, 2)
    tot += full - res[skip]
out["sum_of_marginals_us"] = round(tot, 1)
print(json.dumps(out))
