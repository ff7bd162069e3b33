#!/bin/bash
# Round 6: 256-token slab GEMM, 2 x 2 wave grid (prefill_wave_grid=1, gemm_prefill_v3g_kernel) against the round-5 kernel (=0, v3h) on one box, alternating arms;
# then a rocprofv3 kernel-stats pass of each arm.   gpurun -- 'bash tools/r06_prefill_ab.sh OUTDIR'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/$1; mkdir -p $O
R=$PWD
export TMPDIR=/tmp
for rep in 1 2 3; do
  for wg in 0 1; do
    for M in 512 256; do
      timeout 300 python tools/bench_prefill.py 7b $M prefill-only prefill_wave_grid=$wg 2>> $O/err.log | sed "s/^{/{\"wave_grid\": $wg, \"rep\": $rep, /" | tee -a $O/prefill_ab.jsonl
    done
  done
done
timeout 300 python tools/bench_prefill.py 13b 512 prefill-only prefill_wave_grid=0 2>> $O/err.log | sed "s/^{/{\"wave_grid\": 0, /" | tee -a $O/prefill_ab.jsonl
timeout 300 python tools/bench_prefill.py 13b 512 prefill-only prefill_wave_grid=1 2>> $O/err.log | sed "s/^{/{\"wave_grid\": 1, /" | tee -a $O/prefill_ab.jsonl
for wg in 0 1; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_wg$wg" -o p -- python "$R/tools/bench_prefill.py" 7b 512 prefill-only prefill_wave_grid=$wg > "$R/$O/prof_wg$wg.json" 2> "$R/$O/prof_wg$wg.err")
  f=$(find $O/prof_wg$wg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/prefill512_kernel_stats_wave_grid$wg.csv && head -14 "$f" | cut -c1-220
  find $O/prof_wg$wg -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null; find $O/prof_wg$wg -name "*.db" -delete 2>/dev/null
done
