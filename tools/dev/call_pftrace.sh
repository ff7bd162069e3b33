#!/bin/bash
mkdir -p gpurun_out/r02
cd /root/repo
timeout 120 python tools/dev/prefill_trace.py $PF_ARGS > gpurun_out/r02/pftrace.txt 2>&1
cat gpurun_out/r02/pftrace.txt
timeout 300 python tools/bench_prefill.py 7b 128 x 2>&1 | tail -1
