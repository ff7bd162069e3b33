#!/bin/bash
R=/root/repo; O=gpurun_out/r02/pmcq
rm -rf $R/$O; mkdir -p $R/$O
cd /tmp; export TMPDIR=/tmp
i=0
for grp in \
 "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
 "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD" ; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$R/$O/p$i" -o p -- python "$R/tools/bench_prefill.py" 7b 128 prefill-only ${PMC_TUN} > /dev/null 2> "$R/$O/p$i.err"
  echo "pass $i exit $?"
done
cd $R
python - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/r02/pmcq/p*/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    # order of the 4 GEMM launches per layer: qkv wo w13 w2
    idx = collections.defaultdict(int)
    for r in rows:
        k = r["Kernel_Name"]
        if "gemm_prefill" not in k: continue
        c = r["Counter_Name"]
        kind = ["qkv", "wo", "w13", "w2"][idx[c] % 4]; idx[c] += 1
        tot[kind][c].append(float(r["Counter_Value"]))
with open("gpurun_out/r02/pmcq/summary.txt", "w") as out:
    for k, cs in tot.items():
        out.write(k + "\n")
        for c, v in cs.items():
            t = v[-32:]
            out.write("   %-28s mean(last %d) %14.1f\n" % (c, len(t), sum(t) / len(t)))
print(open("gpurun_out/r02/pmcq/summary.txt").read())
PY
