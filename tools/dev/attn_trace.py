#!/usr/bin/env python3
"""dev: where the prefill attention launch spends its time.  Needs a libthk built with -DTHK_ATTN_TRACE (THK_LIB=...).  One 7B-width layer, a 128-token prompt
at n_past = 0: per workgroup class (query tile 0..3) the stamps of every wave, in us after the launch's first stamp."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
thk = graft.load_package()
lib = ctypes.CDLL(os.environ["THK_LIB"])
M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
names = ["entry", "Q split done", "K in LDS", "S^T done", "softmax done", "V in LDS", "PV done", "past barrier", "factors done", "image written"]
import dataclasses
with thk.Context(0) as ctx:
    m = thk.Model(ctx, dataclasses.replace(thk.LLAMA_7B, n_layer=1)); m.fill_synthetic(); m.finalize()
    toks = np.concatenate([[1], np.random.default_rng(0).integers(3, 32000, M - 1)]).astype(np.int32)
    for _ in range(3): m.reset_kv(0); m.prefill(toks, 0)
    buf = (ctypes.c_ulonglong * (1024 * 4 * 12))()
    lib.thk_debug_attn_trace(buf)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 4, 12).astype(np.float64)
    H = 32
    nwg = H * ((M + 31) // 32)
    a = a[:nwg]
    t0 = a[:, :, 0].min()
    for qt in range((M + 31) // 32):
        blk = a[qt * H:(qt + 1) * H]                     # workgroups of this query tile (blockIdx = qtile * H + head)
        print(f"query tile {qt} ({qt + 1} position tile(s)):")
        for w in range(4):
            active = w <= qt
            row = [(blk[:, w, i].mean() - t0) / 100.0 for i in range(10)]
            print(f"   wave {w} {'(has a tile)' if active else '(no tile)  '}: " + "  ".join(f"{names[i]} {row[i]:.2f}" for i in ([0, 1, 2, 3, 4, 5, 6, 7, 8, 9] if active else [0, 1, 6, 7, 8, 9])))
    if len(sys.argv) > 2 and sys.argv[2] == "reducers":     # the three reducer kernels of the LAST layer-slab: when their workgroups start, have their partial sums, end
        rb = (ctypes.c_ulonglong * (3 * 4096 * 4))()
        lib.thk_debug_reduce_trace(rb)
        r = np.frombuffer(rb, dtype=np.uint64).reshape(3, 4096, 4).astype(np.float64)
        for k, nm in enumerate(["reduce_resid_ximg (last launch: w2's)", "reduce_qkv", "reduce_swiglu_ximg"]):
            x = r[k]; x = x[x[:, 0] > 0]
            if not len(x): continue
            t0 = x[:, 0].min()
            def pct(col, q): 
                v = x[:, col]; v = v[v > 0]
                return (np.percentile(v, q) - t0) / 100.0 if len(v) else float("nan")
            print(f"{nm}: {len(x)} workgroups stamped; entry p50 {pct(0, 50):.2f} p100 {pct(0, 100):.2f} | sums in hand p50 {pct(1, 50):.2f} p100 {pct(1, 100):.2f} | "
                  f"stamp2 p50 {pct(2, 50):.2f} p100 {pct(2, 100):.2f} | end p50 {pct(3, 50):.2f} p100 {pct(3, 100):.2f}  (us after the launch's first stamp)")
    m.close()
