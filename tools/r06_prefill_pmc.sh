#!/bin/bash
# Round 6: counter set of the 512-token prompt (two 256-token slabs) for both slab GEMM kernels - prefill_wave_grid=0 (gemm_prefill_v3h_kernel, round 5) and =1
# (gemm_prefill_v3g_kernel, the 2 x 2 wave grid).  One rocprofv3 pass per counter group (no other trace domains), means per launch.
#   gpurun -- 'bash tools/r06_prefill_pmc.sh OUTDIR'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/$1; mkdir -p $O
R=$PWD
export TMPDIR=/tmp
for wg in 0 1; do
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    t=$(echo $pass | cut -d" " -f1)
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$R/$O/pmc_${wg}_$t" -o p -- python "$R/tools/bench_prefill.py" 7b 512 prefill-only prefill_wave_grid=$wg > /dev/null 2> "$R/$O/pmc_${wg}_$t.err")
    echo "wave_grid $wg pass $t exit $?"
    f=$(find $O/pmc_${wg}_$t -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python tools/pmc_table.py "$f" gemm_prefill reduce_ attn_prefill > $O/prefill512_pmc_wave_grid${wg}_$t.csv 2>> "$R/$O/pmc_${wg}_$t.err" && cut -c1-330 $O/prefill512_pmc_wave_grid${wg}_$t.csv
    rm -rf $O/pmc_${wg}_$t
  done
done
