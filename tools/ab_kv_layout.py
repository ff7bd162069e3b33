#!/usr/bin/env python3
"""A/B of the model-internal K/V cache layout (tunable kv_head_major: [H][n_ctx][D] instead of the reference's [n_ctx][H][D]) on the
decode step at n_past = n_ctx - 1, for n_ctx in {512, 2048} and f32 / binary16 caches.  Alternates the two layouts REPS times on one
box (each arm: fresh model, KV filled by the MFMA prefill path, graph-replayed steps) and prints one JSON line per (n_ctx, kv) pair.
    python tools/ab_kv_layout.py [7b] [steps] [reps]"""
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
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
base = {"7b": thk.LLAMA_7B, "13b": thk.LLAMA_13B}[name]
import dataclasses
for n_ctx in (512, 2048):
    for kv_f16 in (0, 1):
        shape = dataclasses.replace(base, n_ctx=n_ctx)
        prompt = np.concatenate([[1], np.random.default_rng(511).integers(3, shape.n_vocab, n_ctx - 1)]).astype(np.int32)
        ms = {0: [], 1: []}
        tok = {}
        for rep in range(reps):
            for hm in (0, 1):
                with thk.Context(0) as ctx:
                    ctx.set_tunable("kv_head_major", hm); ctx.set_tunable("kv_f16", kv_f16)
                    m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
                    m.prefill(prompt[:n_ctx - 1], 0, want_logits=False)
                    m.seq_set(0, int(prompt[n_ctx - 1]), n_ctx - 1)
                    m.prepare_steps(steps)
                    m.decode_steps(16, 0, advance=False); ctx.sync()
                    t0 = time.perf_counter(); m.decode_steps(steps, 0, advance=False); ctx.sync()
                    ms[hm].append((time.perf_counter() - t0) / steps * 1e3)
                    tok[hm] = m.seq_get(0)[0][:4].tolist()
                    m.close()
        b_tok = shape.bytes_per_token(n_ctx, kv_bytes=2 if kv_f16 else 4)
        rec = {"model": name, "n_ctx": n_ctx, "kv": "f16" if kv_f16 else "f32", "steps": steps,
               "ms_per_step_ref_layout": [round(x, 4) for x in ms[0]], "ms_per_step_head_major": [round(x, 4) for x in ms[1]],
               "median_ref": round(float(np.median(ms[0])), 4), "median_head_major": round(float(np.median(ms[1])), 4),
               "head_major_over_ref": round(float(np.median(ms[1]) / np.median(ms[0])), 4),
               "frac_of_hbm_peak_ref": round(b_tok / (np.median(ms[0]) * 1e-3) / 8e12, 4), "frac_of_hbm_peak_head_major": round(b_tok / (np.median(ms[1]) * 1e-3) / 8e12, 4),
               "same_tokens": tok[0] == tok[1]}
        print(json.dumps(rec), flush=True)
