#!/usr/bin/env python3
"""What one decode token costs through each host-side path (7B, positions 32..159): n-step graphs, one-step graph replays enqueued back to back, one-step replays
with a synchronisation each, thk_model_eval_topk per token (the stochastic sampler's path).  ms per token, median of 3."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

thk = graft.load_package()
N, P0 = 128, 32
with thk.Context(0) as ctx:
    m = thk.Model(ctx, thk.LLAMA_7B); m.fill_synthetic(); m.finalize()
    prompt = np.concatenate([[1], np.random.default_rng(1).integers(3, 32000, P0 - 1)]).astype(np.int32)
    out = {}

    def timed(name, fn):
        ts = []
        for _ in range(3):
            m.reset_kv(0); m.eval(prompt, 0, want_logits=False); m.seq_set(0, 5, P0); ctx.sync()
            t0 = time.perf_counter(); fn(); ctx.sync(); ts.append((time.perf_counter() - t0) / N * 1e3)
        out[name] = round(float(np.median(ts)), 4)

    m.prepare_steps(8); m.prepare_steps(32)
    timed("graphs_of_32_steps", lambda: m.decode_steps(N, 0, advance=True))
    timed("graphs_of_8_steps", lambda: [m.decode_steps(8, 0, advance=True) for _ in range(N // 8)])
    timed("one_step_graphs_back_to_back", lambda: [m.decode_step(0, advance=True) for _ in range(N)])

    def synced():
        for _ in range(N):
            m.decode_step(0, advance=True); ctx.sync()
    timed("one_step_graph_plus_sync", synced)
    v, ids = np.empty(41, np.float32), np.empty(41, np.int32)

    def topk():
        for i in range(N):
            ctx.check(ctx.lib.thk_model_eval_topk(m.h, 0, (C.c_int32 * 1)(7), 1, P0 + i, 41, v.ctypes.data, ids.ctypes.data), "eval_topk")
    timed("eval_topk_per_token", topk)

    def evalfull():
        for i in range(N):
            m.eval([7], P0 + i)
    timed("eval_with_full_logits_readback", evalfull)
    print(json.dumps(out))
    m.close()
