#!/usr/bin/env python3
"""A/B of libthk tunables on the workload bench.py times: graph-replayed 7B (or 13B) decode steps at n_past = T-1.

    python tools/ab.py [--model 7b] [--steps 96] [--reps 5] CONFIG [CONFIG ...]

CONFIG = "base" (library defaults) or "name=value,name=value".  One model instance is filled once; every config is a
re-finalize of it.  Per config: median / best ms per step over REPS timed loops, tokens/s, and the greatest logit
difference against the FIRST config on a short seeded prompt (a quick correctness tripwire; the parity tests are the gate).
Writes one JSON line per config to stdout and to gpurun_out/ab.jsonl.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="7b")
ap.add_argument("--steps", type=int, default=96)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--ctx", type=int, default=512)
ap.add_argument("--out", default="ab.jsonl")
ap.add_argument("configs", nargs="+")
args = ap.parse_args()

thk = graft.load_package()
shape = {"7b": thk.LLAMA_7B, "13b": thk.LLAMA_13B, "tiny": thk.TINY}[args.model]
T = min(args.ctx, shape.n_ctx)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
outf = open(os.path.join(ROOT, "gpurun_out", args.out), "a")
prompt = np.concatenate([[1], np.random.default_rng(7).integers(3, shape.n_vocab, 5)]).astype(np.int32)

with thk.Context(0) as ctx:
    defaults = {}
    m = thk.Model(ctx, shape)
    m.fill_synthetic()
    ref_logits = None
    for cfg in args.configs:
        kv = {} if cfg == "base" else dict(p.split("=") for p in cfg.split(","))
        for k in list(defaults):                      # back to the defaults, then this config's overrides
            ctx.set_tunable(k, defaults[k])
        for k, v in kv.items():
            defaults.setdefault(k, ctx.get_tunable(k))
            ctx.set_tunable(k, int(v))
        rec = {"config": cfg}
        try:
            m.finalize()
            lg, _ = m.eval(prompt, 0)
            if ref_logits is None:
                ref_logits = lg
            rec["max_logit_diff_vs_first"] = float(np.abs(lg - ref_logits).max())
            rec["argmax"] = int(lg.argmax())
            m.seq_set(0, 5, T - 1)
            m.prepare_steps(args.steps)
            m.decode_steps(16, 0, advance=False)
            ctx.sync()
            ts = []
            for _ in range(args.reps):
                t0 = time.perf_counter()
                m.decode_steps(args.steps, 0, advance=False)
                ctx.sync()
                ts.append((time.perf_counter() - t0) / args.steps * 1e3)
            ts.sort()
            rec.update({"ms_median": round(ts[len(ts) // 2], 4), "ms_best": round(ts[0], 4), "tok_s_median": round(1e3 / ts[len(ts) // 2], 2),
                        "tok_s_best": round(1e3 / ts[0], 2)})
        except Exception as e:   # a config that fails must not lose the others
            rec["error"] = str(e)
        line = json.dumps(rec)
        print(line, flush=True)
        outf.write(line + "\n"); outf.flush()
    m.close()
