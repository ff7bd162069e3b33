#!/usr/bin/env python3
"""dev: where a prefill GEMM wave spends its cycles.  Needs a libthk built with -DTHK_PREFILL_TRACE:
    python -c "import __graft_entry__ as g; g.build_libthk(out='token-hawk_amd/libthk_pftrace.so', defs=('THK_PREFILL_TRACE=1',))"
    THK_LIB=$PWD/token-hawk_amd/libthk_pftrace.so python tools/dev/prefill_trace.py [tunable=value ...]
Runs a 128-token prompt through a ONE-layer 7B-width model (four GEMM launches: wq|wk|wv, wo, w1|w3, w2) and prints, per launch, the per-wave mean of each
phase of the main loop in shader cycles, and when the waves started / ended on the 100 MHz chip clock."""
import ctypes, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
thk = graft.load_package()
lib = ctypes.CDLL(os.environ.get("THK_LIB") or os.path.join(ROOT, "token-hawk_amd", "libthk.so"))
names = ["prologue issue + flush/drain", "step head (MFMAs before sync)", "wait vmcnt", "barrier", "step rest (MFMA+reads+DMA)", "flush stores", "drain", "TOTAL", "step rest: read slots", "-", "-", "-"]
shape = thk.LLAMA_7B
M = 128
rng = np.random.default_rng(0)
toks = np.concatenate([[1], rng.integers(3, shape.n_vocab, M - 1)]).astype(np.int32)
with thk.Context(0) as ctx:
    for kv in sys.argv[1:]:
        k, v = kv.split("="); ctx.set_tunable(k, int(v))
    import dataclasses
    sh = dataclasses.replace(shape, n_layer=1)
    m = thk.Model(ctx, sh); m.fill_synthetic(); m.finalize()
    for _ in range(3): m.reset_kv(0); m.prefill(toks, 0)          # 4 GEMM launches per prefill: launch index mod 4 = qkv, wo, w13, w2
    buf = (ctypes.c_ulonglong * (4 * 256 * 4 * 12))()
    rc = lib.thk_debug_prefill_trace(buf)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4, 256, 4, 12).astype(np.float64)
    for k, kind in enumerate(["qkv", "wo", "w13", "w2"]):
        print(kind, "rc", rc)
        for i, n in enumerate(names):
            if i in (5, 6, 9, 10, 11): continue
            print("   %-32s mean %9.0f   min %9.0f   max %9.0f cycles" % (n, a[k, :, :, i].mean(), a[k, :, :, i].min(), a[k, :, :, i].max()))
        st, en = a[k, :, :, 5], a[k, :, :, 6]
        t0 = st.min()
        print("   wave start (10 ns ticks after the first): median %.0f  p90 %.0f  max %.0f ;  wave end: min %.0f  median %.0f  max %.0f" %
              (np.median(st - t0), np.percentile(st - t0, 90), (st - t0).max(), (en - t0).min(), np.median(en - t0), (en - t0).max()))
    m.close()
