#!/usr/bin/env python3
"""Which decode dispatch path wins on which BASELINE configuration?  ("decide per path with a number")

    python tools/paths_matrix.py [--steps 96] [--reps 5] [--out r04_paths_matrix.txt]

Rows: LLaMA-7B, LLaMA-13B, LLaMA-7B with the f16 KV cache, and the N = 8 pipeline's stages of four 7B layers (first: embedding +
layers 0-3, last: layers 28-31 + lm-head) - the short stages are where a boundary-free dispatch could matter most.
Columns: hipGraph replays (the default), the one-launch engine (tunable engine=1) and, when this libthk still carries it, the
overlapped AQL dispatch (overlap_dispatch=1).  Per cell: median ms per step of REPS loops of STEPS hold-position steps at
n_past = 511, or why the path refuses the configuration.  Greedy tokens are compared with the graph path's.
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
ap.add_argument("--steps", type=int, default=96)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--out", default="paths_matrix.txt")
args = ap.parse_args()
thk = graft.load_package()
T = 512

CONFIGS = [   # name, shape, layer range, flags, extra tunables
    ("7B full model", "LLAMA_7B", None, None, {}),
    ("13B full model", "LLAMA_13B", None, None, {}),
    ("7B, f16 KV cache", "LLAMA_7B", None, None, {"kv_f16": 1}),
    ("7B stage 0 of 8 (embed + layers 0-3)", "LLAMA_7B", (0, 4), "EMBED", {}),
    ("7B stage 7 of 8 (layers 28-31 + lm-head)", "LLAMA_7B", (28, 32), "HEAD", {}),
]
PATHS = [("graph", {}), ("engine", {"engine": 1}), ("overlap", {"overlap_dispatch": 1})]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


with thk.Context(0) as ctx:
    info = ctx.device_info()
    say(f"# decode dispatch paths x configurations, {info['name']}, {args.steps} hold-position steps at n_past = {T - 1}, median of {args.reps} loops (ms per step)")
    for name, shape_name, rng, flag, extra in CONFIGS:
        shape = getattr(thk, shape_name)
        l0, l1 = rng if rng else (0, shape.n_layer)
        flags = (thk.THK_STAGE_EMBED | thk.THK_STAGE_HEAD) if flag is None else (thk.THK_STAGE_EMBED if flag == "EMBED" else thk.THK_STAGE_HEAD)
        m = thk.Model(ctx, shape, l0, l1, flags=flags)
        m.fill_synthetic()
        row, ref_tokens = {}, None
        for pname, ptun in PATHS:
            tun = dict(extra, **ptun)
            old = {}
            try:
                for k, v in tun.items():
                    old[k] = ctx.get_tunable(k)
                    ctx.set_tunable(k, v)
            except Exception as e:
                row[pname] = f"n/a ({str(e)[:60]})"
                for k, v in old.items():
                    ctx.set_tunable(k, v)
                continue
            try:
                m.finalize()
                if pname == "engine" and not m.uses_engine():
                    row[pname] = "refused (shape / option not eligible)"
                    continue
                if pname == "overlap" and not getattr(m, "uses_overlap", lambda: False)():      # (the path left libthk in round 4: the tunable is unknown there)
                    row[pname] = "refused (not eligible)"
                    continue
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
                toks = m.seq_get(0)[0][-4:].tolist() if (flags & thk.THK_STAGE_HEAD) else None
                if ref_tokens is None:
                    ref_tokens = toks
                row[pname] = f"{ts[len(ts) // 2]:.4f}" + ("" if toks == ref_tokens else "  TOKENS DIFFER")
            except Exception as e:
                row[pname] = f"error: {str(e)[:80]}"
            finally:
                for k, v in old.items():
                    ctx.set_tunable(k, v)
        m.close()
        say(f"{name:44s} " + "  ".join(f"{p}={row.get(p, '-')}" for p, _ in PATHS))
        with open(os.path.join(ROOT, "gpurun_out", args.out + ".jsonl"), "a") as f:
            f.write(json.dumps({"config": name, **row}) + "\n")
open(os.path.join(ROOT, "gpurun_out", args.out), "w").write("\n".join(lines) + "\n")
