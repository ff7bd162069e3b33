#!/usr/bin/env python3
"""Per-kernel means of every counter in a rocprofv3 `--kernel-trace --pmc ...` pass (one row per kernel name, columns = counters,
mean over the launches; FETCH_SIZE / WRITE_SIZE are also given as bytes: KiB x 1024, FETCH_SIZE x 2 on gfx950 - see
tools/pmc_summary.py).  usage: pmc_table.py <counter_collection.csv> [substring filter ...]"""
import collections, csv, sys
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
flt = sys.argv[2:]
names = sorted({c for k in vals for c in vals[k]})
print("kernel,launches," + ",".join(names) + ("".join(f",{c}_bytes" for c in names if c in ("FETCH_SIZE", "WRITE_SIZE"))))
for k, cs in sorted(vals.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
    if flt and not any(f in k for f in flt):
        continue
    n = max(len(v) for v in cs.values())
    mean = {c: (sum(cs[c]) / len(cs[c]) if c in cs else float("nan")) for c in names}
    extra = "".join(",%d" % round(mean[c] * 1024 * (2 if c == "FETCH_SIZE" else 1)) for c in names if c in ("FETCH_SIZE", "WRITE_SIZE"))
    short = k.split("(")[0][-70:]
    print('"%s",%d,' % (short, n) + ",".join("%.1f" % mean[c] for c in names) + extra)
