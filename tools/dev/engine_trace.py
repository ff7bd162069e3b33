"""Engine timeline: where does a layer's time go?  Prints per-op-kind averages (us) from the s_memtime trace."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as graft
thk = graft.load_package()
ctx = thk.Context(0)
ctx.set_tunable("engine_trace", 1)
for kv in sys.argv[1:]:
    k, v = kv.split("="); ctx.set_tunable(k, int(v))
m = thk.Model(ctx, thk.LLAMA_7B); m.fill_synthetic(); m.finalize()
m.seq_set(0, 5, 0); m.decode_steps(511, 0, advance=True); ctx.sync()
import time
m.decode_steps(4, 0, advance=False); ctx.sync()
t0 = time.perf_counter(); m.decode_steps(20, 0, advance=False); ctx.sync(); dt = (time.perf_counter() - t0) / 20
tr = m.engine_trace().astype(np.int64)          # [cu, op, 8]
ncu, nops, _ = tr.shape
# clock: total kernel span in cycles vs measured step time
span = (tr[:, :, 2].max() - tr[:, 0, 0].min())
mhz = span / (dt * 1e6)
print(f"step {dt*1e3:.3f} ms, engine span {span} cycles -> ~{mhz:.0f} cycles/us (s_memtime)")
cyc = mhz
kinds = ["QKV", "ATTN", "WO", "W13", "W2"]
start0 = tr[:, 0, 0].min()
def us(x): return x / cyc
print("per-op-kind averages over layers 1..30 and all CUs (us):")
print(f"{'op':5s} {'enter->gathered':>16s} {'gathered->done':>15s} {'op span(all CUs)':>17s} {'loader span':>12s} {'loader blocked':>15s} {'cons. enter skew':>17s}")
for ki, name in enumerate(kinds):
    idx = [l * 5 + ki for l in range(1, 31)]
    t = tr[:, idx, :]
    gat = us((t[:, :, 1] - t[:, :, 0]).mean()) if name != "ATTN" else float('nan')
    done = us((t[:, :, 2] - (t[:, :, 1] if name != "ATTN" else t[:, :, 0])).mean())
    opspan = us((t[:, :, 2].max(axis=0) - t[:, :, 0].min(axis=0)).mean())
    lspan = us((t[:, :, 4] - t[:, :, 3]).mean()) if name != "ATTN" else float('nan')
    lblk = us(t[:, :, 5].mean()) if name != "ATTN" else float('nan')
    skew = us((t[:, :, 0].max(axis=0) - t[:, :, 0].min(axis=0)).mean())
    print(f"{name:5s} {gat:16.2f} {done:15.2f} {opspan:17.2f} {lspan:12.2f} {lblk:15.2f} {skew:17.2f}")
# layer period
ent = tr[:, :, 0].min(axis=0)
per = np.diff(ent[0:160:5])
print("layer period (QKV enter to next QKV enter), us: mean %.2f min %.2f max %.2f" % (us(per[1:].mean()), us(per[1:].min()), us(per[1:].max())))
l = 10
print("timeline of layer 10 (us from its first QKV enter; min/mean/max over CUs):")
base = tr[:, l * 5, 0].min()
for ki, name in enumerate(kinds):
    t = tr[:, l * 5 + ki, :]
    row = []
    for s in (0, 1, 2, 3, 4):
        if name == "ATTN" and s in (1, 3, 4): row.append("      -            "); continue
        v = t[:, s] - base
        row.append(f"{us(v.min()):6.1f}/{us(v.mean()):6.1f}/{us(v.max()):6.1f}")
    print(f"  {name:5s} enter {row[0]} gathered {row[1]} done {row[2]} | loader first {row[3]} last {row[4]}")
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gpurun_out", "r02", "engine_trace.npy"), tr)
m.close(); ctx.close()
