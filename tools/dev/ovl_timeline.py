#!/usr/bin/env python3
"""Raw step timeline of one layer (libthk_trace.so): percentiles of workgroup entry / exit per launch, microseconds from the
layer's first entry.  python tools/dev/ovl_timeline.py [name=value ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("THK_LIB", os.path.join(ROOT, "token-hawk_amd", "libthk_trace.so"))
import __graft_entry__ as graft
thk = graft.load_package()
tun = dict((a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[1:] if "=" in a)
shape = thk.LLAMA_7B
with thk.Context(0) as ctx:
    for k, v in tun.items(): ctx.set_tunable(k, v)
    m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
    m.seq_set(0, 5, shape.n_ctx - 1)
    for _ in range(3): m.decode_step(0, advance=False)
    names, tr = m.step_trace(0)
    names, tr = m.step_trace(0)
    m.close()
tr = tr.astype(np.float64) * 0.01
L = 7
base = None
pct = [0, 1, 10, 50, 90, 99, 100]
print("launch               blocks   entry pct " + str(pct) + "    | exit pct")
allk = [k for k in range(len(names)) if (tr[k][:, 0, 0] > 0).any()]
t0 = np.nanmin(np.where(tr[allk[0]][:, :, 0] > 0, tr[allk[0]][:, :, 0], np.nan)); t1 = np.nanmax(tr[allk[-1]][:, :, 3])
print(f"step: first entry of {names[allk[0]]} -> last exit of {names[allk[-1]]}: {t1 - t0:.2f} us over {len(allk)} launches")
for k in list(range(5 * L, 5 * L + 6)) + allk[-4:]:
    t = tr[k]; live = t[:, 0, 0] > 0
    if not live.any(): continue
    t = np.where(t > 0, t, np.nan)[live]
    e, d = np.nanmin(t[:, :, 0], 1), np.nanmax(t[:, :, 3], 1)
    if base is None: base = e.min()
    print(f"{names[k]:<20}{int(live.sum()):>6}   " + " ".join(f"{x:7.2f}" for x in np.percentile(e - base, pct)) + "   | " + " ".join(f"{x:7.2f}" for x in np.percentile(d - base, pct)))
