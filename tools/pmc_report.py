#!/usr/bin/env python3
"""Summarise gpurun_out/pmcsets/*/r01_counter_collection.csv: per kernel (name prefix + grid), mean of every counter."""
import collections, csv, glob, sys
pat = sys.argv[1] if len(sys.argv) > 1 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmcsets/*/r01_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if pat and pat not in r["Kernel_Name"]:
            continue
        agg[(r["Kernel_Name"][:50], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:40s} n={len(v):5d} mean={sum(v)/len(v):.4g}")
