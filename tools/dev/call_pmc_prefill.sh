#!/bin/bash
# dev: PMC passes over the 128-token prefill (counters only + kernel trace, one group per run)
R=/root/repo; O=gpurun_out/r02/pmcp
mkdir -p $R/$O
cd /tmp; export TMPDIR=/tmp
i=0
for grp in \
 "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC" \
 "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum" \
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
 "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUSY_avr TCC_EA0_RDREQ_sum" \
 "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE" \
 "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$R/$O/p$i" -o p -- python "$R/tools/bench_prefill.py" 7b 128 prefill-only ${PMC_TUN} > /dev/null 2> "$R/$O/p$i.err"
  echo "pass $i exit $?"
done
cd $R
python - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/r02/pmcp/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_prefill" not in k and "reduce" not in k and "attn_prefill" not in k and "ximg" not in k: continue
        short = k.split("(")[0].replace("void ", "").replace("thk::", "")[:60]
        tot[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/r02/pmcp/summary.txt", "w") as out:
    for k, cs in tot.items():
        out.write(k + "\n")
        for c, v in cs.items():
            t = v[-64:]
            out.write("   %-44s launches %4d   mean(last %d) %14.1f\n" % (c, len(v), len(t), sum(t) / len(t)))
print(open("gpurun_out/r02/pmcp/summary.txt").read()[:200])
PY
