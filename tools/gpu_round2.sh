#!/bin/bash
# Round-2 evidence run (one gpurun call): rocprofv3 kernel stats of the driver-protocol bench, FETCH_SIZE PMC pass (own run,
# --kernel-trace only), the same two for the 128-token prefill, 13B bench, f16-KV bench.  Outputs under gpurun_out/r02/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
STAGES="${1:-prof pmc bench13 prefill profprefill pmcprefill}"
for s in $STAGES; do
  case $s in
    prof) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof" -o r02 -- python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$R/$O/prof_bench.json" 2> "$R/$O/prof.err"); echo "prof exit $?"; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" ;;
    pmc) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/$O/pmc_fetch" -o r02 -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile --tunable use_graph=0 > /dev/null 2> "$R/$O/pmc.err"); echo "pmc exit $?"; find $O/pmc_fetch -name "*.csv" | head -3 ;;
    bench13) timeout 900 python bench.py --model 13b --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_13b.json 2> $O/bench_13b.err; echo "bench13 exit $?"; cut -c1-400 $O/bench_13b.json ;;
    prefill) timeout 600 python tools/bench_prefill.py 7b 128 > $O/prefill.json 2> $O/prefill.err; echo "prefill exit $?"; cat $O/prefill.json ;;
    profprefill) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_prefill" -o r02 -- python "$R/tools/bench_prefill.py" 7b 128 prefill-only > "$R/$O/prof_prefill.json" 2> "$R/$O/prof_prefill.err"); echo "profprefill exit $?"; f=$(find $O/prof_prefill -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f" ;;
    pmcprefill) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/$O/pmc_prefill" -o r02 -- python "$R/tools/bench_prefill.py" 7b 128 prefill-only > /dev/null 2> "$R/$O/pmc_prefill.err"); echo "pmcprefill exit $?"; find $O/pmc_prefill -name "*.csv" | head -3 ;;
  esac
done
