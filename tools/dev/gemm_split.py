#!/usr/bin/env python3
"""Per-GEMM durations of the prefill chain from a rocprofv3 --kernel-trace CSV: the GEMM launches of a layer come in the fixed order
qkv, wo, w13, w2, so launch number i of a prompt is GEMM i % 4.  Usage: gemm_split.py <kernel_trace.csv> [layers]"""
import csv, sys
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'gemm_prefill' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
per = 4 * L
n_prompts = len(rows) // per
out = {}
for p in range(1, n_prompts):                    # skip the first prompt (cold)
    for i, r in enumerate(rows[p * per:(p + 1) * per]):
        out.setdefault(('qkv', 'wo', 'w13', 'w2')[i % 4], []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in out.items():
    v.sort(); print('  %-4s n %4d  median %6.1f us  min %6.1f  p90 %6.1f' % (k, len(v), v[len(v) // 2], v[0], v[int(len(v) * .9)]))
