#!/usr/bin/env python3
"""First-difference hunt for the overlapped dispatch: one step from the same state on both paths, logits compared."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
thk = graft.load_package()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1
with thk.Context(0) as ctx:
    if len(sys.argv) > 2: ctx.set_tunable("measure_skip_kernel", int(sys.argv[2]))
    m = thk.Model(ctx, thk.ModelShape(n_embd=4096, n_head=32, n_layer=L)); m.fill_synthetic(); m.finalize()
    res = {}
    for mode in (0, 1):
        ctx.set_tunable("overlap_dispatch", mode)
        m.reset_kv(0); m.seq_set(0, 1, 0)
        for step in range(2):
            m.decode_steps(1, 0, advance=True)
            ctx.sync()
            lg = m.read_logits(0)
            res[(mode, step)] = lg.copy()
            for nm in ("q", "part_ml", "part_o", "u", "x"):
                res[(mode, step, nm)] = m.debug_buffer(nm)
            print("mode", mode, "step", step, "nan", int(np.isnan(lg).sum()), "zeros", int((lg == 0).sum()), "min/max", float(np.nanmin(lg)), float(np.nanmax(lg)), "argmax", int(np.nanargmax(lg)), "tok", m.seq_last_token(0), flush=True)
    ctx.set_tunable("overlap_dispatch", 0)
    for step in range(2):
        d = np.abs(res[(0, step)] - res[(1, step)])
        print("step", step, "max diff", float(np.nanmax(d)), "n diff", int((d != 0).sum()))
        for nm in ("q", "part_ml", "part_o", "u", "x"):
            a, b = res[(0, step, nm)], res[(1, step, nm)]
            fin = np.isfinite(a) & np.isfinite(b)
            dd = np.abs(np.where(fin, a - b, 0.0))
            bad = np.nonzero((dd != 0) | (np.isfinite(a) != np.isfinite(b)))[0]
            print("   ", nm, "n", a.size, "max diff", float(dd.max()), "n diff", bad.size, "first", bad[:6].tolist(), "ref", a[bad[:3]].tolist(), "ovl", b[bad[:3]].tolist())
    m.close()
