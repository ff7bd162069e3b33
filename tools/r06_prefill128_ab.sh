#!/bin/bash
# Round 6: the 2 x 2 wave grid on the 128-token slab (prefill_wave_grid_128=1, gemm_prefill_v3g_kernel<4,...>) against gemm_prefill_v3_kernel, alternating arms on one box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/$1; mkdir -p $O
timeout 300 python tools/bench_prefill.py 7b 128 x prefill_wave_grid_128=1 2>> $O/err.log | sed "s/^{/{\"wave_grid_128\": 1, \"parity_run\": true, /" | tee -a $O/prefill128_ab.jsonl
timeout 300 python tools/bench_prefill.py 7b 100 x prefill_wave_grid_128=1 2>> $O/err.log | sed "s/^{/{\"wave_grid_128\": 1, \"parity_run\": true, /" | tee -a $O/prefill128_ab.jsonl
for rep in 1 2 3 4; do
  for wg in 0 1; do
    timeout 300 python tools/bench_prefill.py 7b 128 prefill-only prefill_wave_grid_128=$wg 2>> $O/err.log | sed "s/^{/{\"wave_grid_128\": $wg, \"rep\": $rep, /" | tee -a $O/prefill128_ab.jsonl
  done
done
for wg in 0 1; do timeout 300 python tools/bench_prefill.py 13b 128 prefill-only prefill_wave_grid_128=$wg 2>> $O/err.log | sed "s/^{/{\"wave_grid_128\": $wg, /" | tee -a $O/prefill128_ab.jsonl; done
