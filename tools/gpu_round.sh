#!/bin/bash
# One gpurun call: parity tests, smoke, bench, sweep, rocprof.  Every stage has its own timeout
# and logs into gpurun_out/ so a failing stage does not waste the call.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
STAGES="${1:-tests smoke bench sweep prof}"
echo "stages: $STAGES"
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
nproc > gpurun_out/host.txt; free -g >> gpurun_out/host.txt; grep -m1 "model name" /proc/cpuinfo >> gpurun_out/host.txt
for s in $STAGES; do
  case $s in
    tests) timeout 1500 python -m pytest tests -m gpu -q -x --timeout 300 2>&1 | tail -40 > gpurun_out/tests.log; echo "tests exit $?"; tail -5 gpurun_out/tests.log ;;
    testsall) timeout 1500 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -80 > gpurun_out/tests.log; tail -30 gpurun_out/tests.log ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -3 gpurun_out/smoke.log ;;
    bench) timeout 900 python bench.py --steps 100 --warmup 10 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"; cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err ;;
    sweep) timeout 900 python tools/sweep.py 7b > gpurun_out/sweep.log 2>&1; echo "sweep exit $?"; tail -25 gpurun_out/sweep.log ;;
    prof) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o r01 -- python "$OLDPWD/bench.py" --steps 30 --warmup 5 --no-cpu-baseline > "$OLDPWD/gpurun_out/prof_bench.json" 2> "$OLDPWD/gpurun_out/prof.err"); echo "prof exit $?"; find gpurun_out/prof -name "*stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f" ;;
    prefill) timeout 600 python tools/bench_prefill.py 7b 128 > gpurun_out/prefill.json 2> gpurun_out/prefill.err; echo "prefill exit $?"; cat gpurun_out/prefill.json; tail -3 gpurun_out/prefill.err ;;
    profprefill) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_prefill" -o r01 -- python "$OLDPWD/tools/bench_prefill.py" 7b 128 prefill-only > "$OLDPWD/gpurun_out/prof_prefill.json" 2> "$OLDPWD/gpurun_out/prof_prefill.err"); echo "profprefill exit $?"; f=$(find gpurun_out/prof_prefill -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -24 "$f" ;;
    pmcprefill) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OLDPWD/gpurun_out/pmc_prefill" -o r01 -- python "$OLDPWD/tools/bench_prefill.py" 7b 128 prefill-only > /dev/null 2> "$OLDPWD/gpurun_out/pmc_prefill.err"); echo "pmcprefill exit $?"; find gpurun_out/pmc_prefill -name "*.csv" | head ;;
    pmcsets) i=0; while read -r set; do [ -z "$set" ] && continue; i=$((i+1)); (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OLDPWD/gpurun_out/pmcsets/$i" -o r01 -- python "$OLDPWD/tools/bench_prefill.py" 7b 128 prefill-only > /dev/null 2> "$OLDPWD/gpurun_out/pmcsets_$i.err"); echo "pmc set $i ($set) exit $?"; done < tools/pmc_sets.txt ;;
    pmc) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OLDPWD/gpurun_out/pmc_fetch" -o r01 -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile --tunable use_graph=0 > /dev/null 2> "$OLDPWD/gpurun_out/pmc.err"); echo "pmc exit $?"; find gpurun_out/pmc_fetch -name "*.csv" | head ;;
  esac
done
ls -la gpurun_out | head -30
