#!/usr/bin/env python3
"""Overlapped dispatch (tunable overlap_dispatch, thk_ovl.cpp) against the hipGraph path on the same model instance:
 1. 2-layer model at 7B width: N advancing greedy steps on both paths -> same tokens, same final logits (bit for bit)
 2. --full: the whole 7B (or --model 13b) model: ms per step of both paths at n_past = T-1 (what bench.py times), interleaved
"""
import argparse, os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
ap = argparse.ArgumentParser()
ap.add_argument("--full", action="store_true")
ap.add_argument("--model", default="7b")
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--tunable", action="append", default=[], help="name=value, set before finalize")
ap.add_argument("--no-parity", action="store_true")
args = ap.parse_args()
thk = graft.load_package()
import torch

def parity(shape, n, label):
    with thk.Context(0) as ctx:
        for kv in args.tunable:
            k, v = kv.split("="); ctx.set_tunable(k, int(v))
        m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
        out = {}
        for mode in (0, 1, 0, 1):
            ctx.set_tunable("overlap_dispatch", mode)
            m.reset_kv(0); m.seq_set(0, 1, 0)
            m.decode_steps(n, 0, advance=True)
            ctx.sync()
            gen, ng, pos = m.seq_get(0)
            lg = m.read_logits(0)
            key = f"{mode}"
            if key in out:
                assert (out[key][0] == gen[:ng]).all() and (out[key][1] == lg).all(), f"{label}: mode {mode} not deterministic"
            out[key] = (gen[:ng].copy(), lg.copy())
            print(label, "mode", mode, "uses_overlap", m.uses_overlap(), "tokens", gen[:8].tolist(), "pos", pos, flush=True)
        ctx.set_tunable("overlap_dispatch", 0)
        same_t = (out["0"][0] == out["1"][0]).all()
        d = float(np.abs(out["0"][1] - out["1"][1]).max())
        print(f"{label}: tokens equal {bool(same_t)}, max |logit diff| {d:.3e}", flush=True)
        # single steps + interleaving with the stream (seq_set between steps)
        m.reset_kv(0)
        a = []
        for mode in (0, 1):
            ctx.set_tunable("overlap_dispatch", mode)
            m.reset_kv(0); m.seq_set(0, 1, 0); toks = []
            for i in range(6):
                m.decode_step(0, advance=True)
                toks.append(m.seq_last_token(0))
            a.append(toks)
        ctx.set_tunable("overlap_dispatch", 0)
        print(label, "single steps:", a[0], a[1], "equal", a[0] == a[1], flush=True)
        m.close()
        return bool(same_t) and d < 1e-5 and a[0] == a[1]

if not args.no_parity:
    ok = parity(thk.ModelShape(n_embd=4096, n_head=32, n_layer=2), 24, "7B-width x2 layers")
    ok = parity(thk.ModelShape(n_embd=5120, n_head=40, n_layer=1), 12, "13B-width x1 layer") and ok
    print("PARITY", "OK" if ok else "FAILED", flush=True)

if args.full:
    shape = {"7b": thk.LLAMA_7B, "13b": thk.LLAMA_13B}[args.model]
    T = shape.n_ctx
    with thk.Context(0) as ctx:
        for kv in args.tunable:
            k, v = kv.split("="); ctx.set_tunable(k, int(v))
        m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
        res = {0: [], 1: []}
        m.prepare_steps(args.steps)
        for rep in range(args.reps + 1):
            for mode in (0, 1):
                ctx.set_tunable("overlap_dispatch", mode)
                m.seq_set(0, 5, T - 1)
                m.decode_steps(4, 0, advance=False)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                m.decode_steps(args.steps, 0, advance=False)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / args.steps * 1e3
                if rep: res[mode].append(dt)
                tok = m.seq_last_token(0)
                print(f"rep {rep} mode {mode}: {dt:.4f} ms/step  {1e3/dt:.1f} tok/s  token {tok}", flush=True)
        ctx.set_tunable("overlap_dispatch", 0)
        for mode in (0, 1):
            v = sorted(res[mode]); print(f"mode {mode}: median {v[len(v)//2]:.4f} ms  best {v[0]:.4f} ms -> {1e3/v[len(v)//2]:.1f} tok/s")
        m.close()
