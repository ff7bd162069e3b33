#!/usr/bin/env python3
"""Intra-kernel timeline of one decode step on the launch path (development aid).

Needs the -DTHK_TRACE build: python -c "import __graft_entry__ as g; g.build_libthk(trace=True)" -> token-hawk_amd/libthk_trace.so
    THK_LIB=token-hawk_amd/libthk_trace.so python tools/step_trace.py [7b|13b] [name=value ...]

Every wave stamps s_memrealtime (100 MHz, scalar instructions only) at: kernel entry | activation vector staged in LDS | first weight
batch consumed | done.  Per launch kind (averaged over layers 1..), all in microseconds:
  gap       first workgroup entry - last workgroup done of the PREVIOUS launch (the boundary)
  ramp      last workgroup entry - first workgroup entry (dispatch)
  staged    median / max of (vector staged - own entry)
  first     median / max of (first batch consumed - own entry)
  life      median / max workgroup lifetime
  span      last done - first entry (what a profiler calls the kernel duration)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("THK_LIB", os.path.join(ROOT, "token-hawk_amd", "libthk_trace.so"))
import __graft_entry__ as graft  # noqa: E402

thk = graft.load_package()
name = "7b"
tun = {}
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("="); tun[k] = int(v)
    else:
        name = a
shape = {"7b": thk.LLAMA_7B, "13b": thk.LLAMA_13B, "tiny": thk.TINY}[name]
T = shape.n_ctx
TICK_US = 0.01
with thk.Context(0) as ctx:
    for k, v in tun.items():               # use_graph=0 records eager launches; the default records a replayed graph
        ctx.set_tunable(k, v)
    m = thk.Model(ctx, shape)
    m.fill_synthetic()
    m.finalize()
    m.seq_set(0, 5, T - 1)
    for _ in range(3):
        m.decode_step(0, advance=False)
    reps = []
    for _ in range(4):
        names, tr = m.step_trace(0)
        reps.append(tr.astype(np.int64))
    m.close()

agg = {}
for tr in reps:
    prev_done = None
    for k, nm in enumerate(names):
        t = tr[k].astype(np.float64)                    # [blocks, 8 waves, 4]
        live = t[:, 0, 0] > 0
        if not live.any():
            continue
        t = np.where(t > 0, t, np.nan)[live] * TICK_US
        with np.errstate(all="ignore"):
            e, d = np.nanmin(t[:, :, 0], 1), np.nanmax(t[:, :, 3], 1)          # workgroup: first wave in, last wave out
            s1, s2 = np.nan_to_num(np.nanmax(t[:, :, 1], 1)), np.nan_to_num(np.nanmedian(t[:, :, 2], 1))
        first, last_done = e.min(), d.max()
        rec = {"gap": (first - prev_done) if prev_done is not None else np.nan, "ramp": e.max() - first,
               "staged_med": np.median((s1 - e)[s1 > 0]) if (s1 > 0).any() else np.nan, "staged_max": ((s1 - e)[s1 > 0]).max() if (s1 > 0).any() else np.nan,
               "first_med": np.median((s2 - e)[s2 > 0]) if (s2 > 0).any() else np.nan, "first_max": ((s2 - e)[s2 > 0]).max() if (s2 > 0).any() else np.nan,
               "life_med": np.median(d - e), "life_max": (d - e).max(), "span": last_done - first, "blocks": int(live.sum()),
               "first_done": d.min() - first}
        ev = np.concatenate([np.stack([e, np.ones_like(e)], 1), np.stack([d, -np.ones_like(d)], 1)])     # peak number of workgroups alive at once
        ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]
        rec["peak_resident"] = float(np.cumsum(ev[:, 1]).max())
        prev_done = last_done
        if k >= 6:          # skip layer 0 (cold)
            agg.setdefault(nm, []).append(rec)
out = {}
print(f"{'launch':<18}{'n':>5}{'blocks':>7}{'gap':>7}{'ramp':>7}{'staged med/max':>16}{'first med/max':>16}{'life med/max':>15}{'1st done':>9}{'span':>7}{'peak':>6}")
for nm, recs in agg.items():
    f = lambda key: float(np.nanmean([r[key] for r in recs]))
    out[nm] = {k: round(f(k), 2) for k in recs[0] if k != "blocks"}
    out[nm]["blocks"] = recs[0]["blocks"]
    print(f"{nm:<18}{len(recs):>5}{recs[0]['blocks']:>7}{f('gap'):>7.2f}{f('ramp'):>7.2f}{f('staged_med'):>8.2f}/{f('staged_max'):<7.2f}{f('first_med'):>8.2f}/{f('first_max'):<7.2f}"
          f"{f('life_med'):>8.2f}/{f('life_max'):<6.2f}{f('first_done'):>9.2f}{f('span'):>7.2f}{f('peak_resident'):>6.0f}")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"model": name, "tunables": tun, "launches": out}, open(os.path.join(ROOT, "gpurun_out", "step_trace.json"), "w"), indent=1)
