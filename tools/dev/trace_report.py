"""Offline report of an engine timeline (gpurun_out/r02/engine_trace*.npy): 100 MHz wall-clock stamps, common to all CUs."""
import sys, numpy as np
tr = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r02/engine_trace.npy").astype(np.int64)
ncu, nops, _ = tr.shape
kinds = ["QKV", "ATTN", "WO", "W13", "W2"]
L = (nops - 1) // 5
us = lambda c: np.asarray(c) / 100.0
tot = us(tr[:, nops - 1, 2].max() - tr[:, 0, 0].min())
print(f"engine span (first op enter -> last op done, all CUs): {tot:.1f} us; per layer ~{tot / (L + 0.5):.1f} us")
print(f"{'op':5s} {'enter->gath':>11s} {'gath->done':>11s} {'ldr span':>9s} {'ldr blkd':>9s} | {'done skew':>9s} {'lastdone->gath(next)':>21s} {'next gath skew':>14s}  (mean over layers 1..{L-2}; us)")
for ki, name in enumerate(kinds):
    idx = np.array([l * 5 + ki for l in range(1, L - 1)])
    t = tr[:, idx, :]
    e2g = us((t[:, :, 1] - t[:, :, 0]).mean()) if name != "ATTN" else float("nan")
    g2d = us((t[:, :, 2] - t[:, :, 1]).mean()) if name != "ATTN" else us((t[:, :, 2] - t[:, :, 0]).mean())
    ls = us((t[:, :, 4] - t[:, :, 3]).mean()) if name != "ATTN" else float("nan")
    lb = us(t[:, :, 5].mean()) if name != "ATTN" else float("nan")
    if name != "ATTN":
        done = np.maximum(np.maximum(t[:, :, 2], t[:, :, 6]), t[:, :, 7])      # per CU: last consumer wave done
    else:
        done = t[:, :, 2]
    dskew = us((done.max(axis=0) - done.min(axis=0)).mean())
    # next op's gather completion relative to the globally last "done" of this op
    nidx = idx + 1
    nk = kinds[(ki + 1) % 5]
    if nk == "ATTN":
        lat = float("nan"); gsk = float("nan")
    else:
        ng = tr[:, nidx, 1]
        lat = us((ng.mean(axis=0) - done.max(axis=0)).mean()); gsk = us((ng.max(axis=0) - ng.min(axis=0)).mean())
    print(f"{name:5s} {e2g:11.2f} {g2d:11.2f} {ls:9.2f} {lb:9.2f} | {dskew:9.2f} {lat:21.2f} {gsk:14.2f}")
l = L // 2
print(f"layer {l} timeline, us from its QKV first enter (min / mean / max over CUs):")
base = tr[:, l * 5, 0].min()
for ki, name in enumerate(kinds):
    t = tr[:, l * 5 + ki, :]
    f = lambda v: f"{us(v.min() - base):6.1f}/{us(v.mean() - base):6.1f}/{us(v.max() - base):6.1f}"
    if name == "ATTN":
        print(f"  {name:5s} enter {f(t[:,0])}                              done {f(t[:,2])}")
    else:
        done = np.maximum(np.maximum(t[:, 2], t[:, 6]), t[:, 7])
        print(f"  {name:5s} enter {f(t[:,0])} gathered {f(t[:,1])} done {f(done)} | loader first {f(t[:,3])} last {f(t[:,4])}")
