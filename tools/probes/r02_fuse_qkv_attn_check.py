"""Bring-up script of the round-2 fused qkv+attention experiment: runs only with tools/probes/r02_fuse_qkv_attn_experiment.patch applied (its
tunables fuse_qkv_* do not exist in the library).  Tiny parity vs oracle (fuse on/off), 7B greedy agreement and timing."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as graft
from oracle import oracle as orc
orc.build()
thk = graft.load_package()
ctx = thk.Context(0)
for shape_name in ("TINY",):
    for fuse in (0, 1):
        for splits in (1, 4):
            ctx.set_tunable("fuse_qkv_attn", fuse); ctx.set_tunable("attn_splits", splits)
            m = thk.Model(ctx, getattr(thk, shape_name)); m.fill_synthetic(); m.finalize()
            om = orc.OracleModel(getattr(orc, shape_name)); om.fill_synthetic()
            rng = np.random.default_rng(1)
            toks = [1] + rng.integers(3, 2048, 40).tolist()
            worst = 0.0
            try:
                for i, t in enumerate(toks):
                    lg, hid = m.eval([t], i, want_hidden=True)
                    lo, ho = om.eval(t, i, flags=orc.FAITHFUL_ORDER)
                    worst = max(worst, float(np.abs(lg - lo).max()))
                m.seq_set(0, 7, len(toks)); m.decode_steps(9, 0, advance=True)
                gen, n, pos = m.seq_get(0)
                tok, exp = 7, []
                for i in range(9):
                    lo, _ = om.eval(tok, len(toks) + i); tok = orc.greedy(lo); exp.append(tok)
                print(f"{shape_name} fuse={fuse} splits={splits} max|dlogit|={worst:.3e} greedy_ok={gen.tolist() == exp}")
            except Exception as e:
                print(f"{shape_name} fuse={fuse}: EXCEPTION {e}")
            m.close(); om.close()
ctx.set_tunable("attn_splits", 4)
if "7b" in sys.argv:
    ref_tail = None
    cfgs = [{"fuse_qkv_attn": 0}, {"fuse_qkv_attn": 1, "fuse_qkv_sleeps": 0}, {"fuse_qkv_attn": 1, "fuse_qkv_sleeps": 0, "fuse_qkv_dbg": 3}, {"fuse_qkv_attn": 1, "fuse_qkv_sleeps": 0, "fuse_qkv_dbg": 7},
            {"fuse_qkv_attn": 1, "fuse_qkv_sleeps": 0, "fuse_qkv_dbg": 7, "fuse_qkv_bph": 32}, {"fuse_qkv_attn": 1, "fuse_qkv_sleeps": 0, "fuse_qkv_dbg": 7, "fuse_qkv_bph": 16}]
    for cfg in cfgs:
        for k, v in {"fuse_qkv_attn": 0, "fuse_qkv_bph": 0, "attn_splits": 4, "fuse_qkv_sleeps": 4, "fuse_qkv_dbg": 0, **cfg}.items():
            ctx.set_tunable(k, v)
        m = thk.Model(ctx, thk.LLAMA_7B); m.fill_synthetic(); m.finalize()
        m.seq_set(0, 5, 0); m.decode_steps(511, 0, advance=True); ctx.sync()
        m.prepare_steps(100); m.decode_steps(10, 0, advance=False); ctx.sync()
        t0 = time.perf_counter(); m.decode_steps(100, 0, advance=False); ctx.sync(); dt = time.perf_counter() - t0
        gen, n, pos = m.seq_get(0)
        tail = gen[-8:].tolist()
        if ref_tail is None: ref_tail = tail
        print(f"7B {cfg}: {100 / dt:.1f} tok/s {dt * 10:.4f} ms/step same_tokens={tail == ref_tail}")
        m.close()
ctx.close()
