#!/usr/bin/env python3
"""dev: the top-k kernel on its own (thk_topk_f32 on a 32000-entry vector, k = 41) and behind a decode step (thk_model_eval_topk), for rocprofv3 --stats."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
thk = graft.load_package()
with thk.Context(0) as ctx:
    lg = ctx.from_numpy((np.random.default_rng(0).standard_normal(32000) * 2).astype(np.float32))
    for _ in range(3): ctx.topk_f32(lg, 32000, 41)
    t0 = time.perf_counter()
    for _ in range(200): ctx.topk_f32(lg, 32000, 41)
    print("thk_topk_f32 (launch + copy + sync) us per call:", round((time.perf_counter() - t0) / 200 * 1e6, 1))
    m = thk.Model(ctx, thk.ModelShape(n_layer=1)); m.fill_synthetic(); m.finalize()
    v, ids = np.empty(41, np.float32), np.empty(41, np.int32)
    for i in range(40):
        ctx.check(ctx.lib.thk_model_eval_topk(m.h, 0, (C.c_int32 * 1)(7), 1, i, 41, v.ctypes.data, ids.ctypes.data), "eval_topk")
    m.close()
