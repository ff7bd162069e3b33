#!/usr/bin/env python3
"""profiles/pmc_traffic.json + profiles/kernel_durations.json from this round's rocprofv3 passes, stamped with the binary they describe.

    python tools/make_pmc_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv kernel_stats.csv [ROUND]

FETCH pass / WRITE pass: `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE` (own passes) around `tools/ab.py --steps 4 --reps 1
use_graph=0` (eager decode steps at T = 512; bench.py itself segfaults inside hipLaunchKernel under --pmc once torch is imported).  FETCH_SIZE is in KiB
and on gfx950 counts 64 B per 128-B request of a wide coalesced stream: bytes = FETCH_SIZE * 1024 * 2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is
reported as KiB * 1024, uncalibrated (the guide's wording).  kernel_stats: `rocprofv3 --kernel-trace --stats` around bench.py in the driver's protocol.
The `_binary` entry (size and sha256/16 of token-hawk_amd/libthk.so, git commit) lets bench.py say whether the library it loaded is the one measured."""
import collections
import csv
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fetch_csv, write_csv, stats_csv = sys.argv[1:4]
rnd = sys.argv[4] if len(sys.argv) > 4 else "r06"


def bench_name(kernel: str):
    """rocprofv3 kernel name -> bench.py launch name (gemv_kernel<NR, U, NS, PRO, EPI, ...>: PRO 1 rms / 2 attention partials; EPI 1 residual, 2 RoPE + KV, 3 SwiGLU, 4 lm-head)."""
    m = re.search(r"gemv_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)", kernel)
    if m:
        pro, epi = int(m.group(4)), int(m.group(5))
        return {2: "norm_qkv_rope_kv", 3: "norm_w13_swiglu", 4: "norm_lmhead"}.get(epi, "attn_wo_resid" if (epi == 1 and pro == 2) else None)
    if "gemv_quarter_kernel" in kernel:
        return "w2_resid"
    if "attn_decode_kernel" in kernel:
        return "attn_decode"
    return None


def pmc_means(path, counter, last_n=32):
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            n = bench_name(r["Kernel_Name"])
            if n:
                vals[n].append(float(r["Counter_Value"]))
    return {k: sum(v[-last_n:]) / len(v[-last_n:]) for k, v in vals.items()}, {k: len(v) for k, v in vals.items()}


lib = os.path.join(ROOT, "token-hawk_amd", "libthk.so")
blob = open(lib, "rb").read()
binary = {"libthk_so_bytes": len(blob), "libthk_so_sha256_16": hashlib.sha256(blob).hexdigest()[:16],
          "git_commit": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None,
          "git_dirty": bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "token-hawk_amd", "include"], capture_output=True, text=True).stdout.strip())}
fetch, nf = pmc_means(fetch_csv, "FETCH_SIZE")
write, _ = pmc_means(write_csv, "WRITE_SIZE")
traffic = {"_method": f"round {rnd[1:]}: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (two passes, no other trace domains) around tools/ab.py --steps 4 --reps 1 use_graph=0 "
                      f"on the binary named in _binary (profiles/{rnd}_pmc_fetch_size_decode.csv, {rnd}_pmc_write_size_decode.csv); mean over the last 32 launches (T = 512); "
                      "read bytes = FETCH_SIZE KiB * 1024 * 2 (gfx950 counts 64 B per 128-B request of a wide coalesced stream, MI355X_MICROARCH.md); the values are HBM READ bytes per launch, "
                      "what bench.py's roofline.traffic quotes; _write_bytes = WRITE_SIZE KiB * 1024, uncalibrated",
           "_units": "HBM read bytes per launch", "_binary": binary, "_launches_counted": nf}
for k, v in fetch.items():
    traffic[k] = int(round(v * 1024 * 2))
traffic["_write_bytes"] = {k: int(round(v * 1024)) for k, v in write.items()}
dur, calls = {}, {}
for r in csv.DictReader(open(stats_csv)):
    n = bench_name(r["Name"])
    if n:      # several instantiations can map to one launch name (13B / 2048-ctx extras): keep the one with the most calls = the 7B headline's
        if n not in calls or int(r["Calls"]) > calls[n]:
            calls[n], dur[n] = int(r["Calls"]), round(float(r["AverageNs"]) / 1e3, 2)
durations = {"_method": f"round {rnd[1:]}: rocprofv3 --kernel-trace --stats, python bench.py --steps 20 --warmup 5 --no-cpu-baseline (the driver's protocol; graph replay) on the binary named in _binary "
                        f"(profiles/{rnd}_kernel_stats_driver_protocol.csv).  AverageNs over all launches of the 7B kernels incl. the prompt-fill steps, the marginal-cost model and the host_api "
                        "generation (the attention average is therefore over many T); kernel-only durations, no launch gaps; norm_lmhead includes the folded greedy pick",
             "_units": "us", "_binary": binary, "_calls": calls}
durations.update(dur)
json.dump(traffic, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
json.dump(durations, open(os.path.join(ROOT, "profiles", "kernel_durations.json"), "w"), indent=1)
print(json.dumps({"traffic": {k: v for k, v in traffic.items() if not k.startswith("_")}, "write": traffic["_write_bytes"], "durations_us": dur, "binary": binary}, indent=1))
