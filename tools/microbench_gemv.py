#!/usr/bin/env python3
"""Streaming mat-vec microbenchmark: time vs rows at fixed C to separate the fixed per-launch
overhead (intercept) from the streaming rate (slope).  Uses thk_matvec_f16 (PRO_COPY/EPI_STORE)
back to back on rotating weight buffers (>= 1 GB apart, nothing stays in L2/MALL)."""
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
res = []
with thk.Context(0) as ctx:
    lib = ctx.lib
    for Cc, var_name in ((4096, "gemv_variant_wo"), (11008, "gemv_variant_w2")):
        pool_bytes = 3 << 30
        pool = ctx.alloc(pool_bytes)
        ctx.synth_f16("pool", pool_bytes // 2, pool)
        x = ctx.from_numpy(np.random.default_rng(0).standard_normal(Cc).astype(np.float32))
        y = ctx.alloc(65536 * 4)
        for var in (0, 1, 2, 3):
            ctx.set_tunable(var_name, var)
            for bpc in (2, 4, 8):
                ctx.set_tunable("gemv_blocks_per_cu", bpc)
                pts = []
                for R in (512, 1024, 2048, 4096, 8192, 12288, 16384, 24576, 32768):
                    nbytes = R * Cc * 2
                    nslots = max(2, min(16, pool_bytes // nbytes))
                    reps = 40
                    for warm in (True, False):
                        ctx.sync(); t0 = time.perf_counter()
                        for i in range(reps):
                            lib.thk_matvec_f16(ctx.h, C.c_void_p(pool.ptr + (i % nslots) * nbytes), R, Cc, C.c_void_p(x.ptr), C.c_void_p(y.ptr))
                        ctx.sync(); dt = (time.perf_counter() - t0) / reps
                    pts.append((nbytes / 1e6, dt * 1e6))
                mb = np.array([p[0] for p in pts]); us = np.array([p[1] for p in pts])
                slope, icpt = np.polyfit(mb[3:], us[3:], 1)
                res.append({"C": Cc, "var": var, "bpc": bpc, "intercept_us": round(float(icpt), 2), "tbps_slope": round(1.0 / slope, 3),
                            "points_mb_us": [(round(a, 1), round(b, 2)) for a, b in pts]})
                print(res[-1], flush=True)
        pool.free()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "microbench_gemv.json"), "w"), indent=1)
