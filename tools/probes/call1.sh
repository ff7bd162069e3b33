cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r02
timeout 240 tools/probes/engine_probe3 16 > gpurun_out/r02/engine_probe3.txt 2>&1; echo "probe exit $?"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile > gpurun_out/r02/bench_s20.json 2> gpurun_out/r02/bench_s20.err; echo "bench20 exit $?"
timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-profile > gpurun_out/r02/bench_s100.json 2> gpurun_out/r02/bench_s100.err; echo "bench100 exit $?"
cat gpurun_out/r02/engine_probe3.txt; cat gpurun_out/r02/bench_s20.json gpurun_out/r02/bench_s100.json | cut -c1-400
