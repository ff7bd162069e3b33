mkdir -p gpurun_out/r02
timeout 300 python tools/dev/engine_check.py tiny > gpurun_out/r02/engine_check4.txt 2>&1; tail -5 gpurun_out/r02/engine_check4.txt
for P in 0 1; do
  timeout 200 python tools/dev/engine_trace.py engine_park=$P > gpurun_out/r02/engine_trace4_p$P.txt 2>&1; head -2 gpurun_out/r02/engine_trace4_p$P.txt; cp gpurun_out/r02/engine_trace.npy gpurun_out/r02/engine_trace_p$P.npy
done
