#!/bin/bash
# One gpurun call = a list of stages (round 3).  Usage: gpurun -- 'bash tools/gpu_session.sh OUTDIR stage [stage ...]'
# Everything is written under gpurun_out/OUTDIR/ (merged back by gpurun); the summaries worth keeping are copied to profiles/ by hand.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/$1; shift; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for s in "$@"; do
  echo "=== stage $s"
  case $s in
    tests) timeout 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests exit $?"; tail -4 $O/tests.log ;;
    bench) timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; cut -c1-300 $O/bench.json ;;
    bench200) timeout 600 python bench.py > $O/bench200.json 2> $O/bench200.err; echo "bench200 exit $?"; cut -c1-300 $O/bench200.json ;;
    ab:*) THK_MEASURE_HOOKS=1 timeout 900 python tools/ab.py $ABFLAGS --out $(basename $O)_ab.jsonl $(echo "${s#ab:}" | tr '+' ' ') 2> $O/ab.err | tee -a $O/ab.jsonl; echo "ab exit $?" ;;
    abprof:*) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/abprof" -o ab -- python "$R/tools/ab.py" --reps 2 --steps 48 $(echo "${s#abprof:}" | tr '+' ' ') > "$R/$O/abprof.jsonl" 2> "$R/$O/abprof.err"); echo "abprof exit $?"; f=$(find $O/abprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/abprof_kernel_stats.csv && head -24 "$f" | cut -c1-200 ;;
    prof) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof" -o r04 -- python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$R/$O/prof_bench.json" 2> "$R/$O/prof.err"); echo "prof exit $?"; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv && head -16 "$f" | cut -c1-200 ;;
    pmc) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/$O/pmc_fetch" -o r04 -- python "$R/tools/ab.py" --steps 4 --reps 1 use_graph=0 > /dev/null 2> "$R/$O/pmc.err"); echo "pmc exit $?"; f=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/pmc_summary.py "$f" 32 > $O/pmc_fetch_summary.csv 2>> "$R/$O/pmc.err" && head -20 $O/pmc_fetch_summary.csv; rm -rf $O/pmc_fetch ;;   # (bench.py under --pmc segfaults in hipLaunchKernel once torch is loaded; ab.py drives the same steps)
    bench13) timeout 900 python bench.py --model 13b --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_13b.json 2> $O/bench_13b.err; echo "bench13 exit $?"; cut -c1-300 $O/bench_13b.json ;;
    prefill) timeout 600 python tools/bench_prefill.py 7b 128 > $O/prefill.json 2> $O/prefill.err; echo "prefill exit $?"; cat $O/prefill.json ;;
    profprefill) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_prefill" -o r04 -- python "$R/tools/bench_prefill.py" 7b 128 prefill-only > "$R/$O/prof_prefill.json" 2> "$R/$O/prof_prefill.err"); echo "profprefill exit $?"; f=$(find $O/prof_prefill -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/prefill_kernel_stats.csv && head -20 "$f" | cut -c1-200 ;;
    pmcprefill) for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do t=$(echo $pass | cut -d" " -f1); (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$R/$O/pmc_pf_$t" -o p -- python "$R/tools/bench_prefill.py" 7b 128 prefill-only > /dev/null 2> "$R/$O/pmc_pf_$t.err"); echo "pmc pass $t exit $?"; f=$(find $O/pmc_pf_$t -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/pmc_table.py "$f" gemm_prefill reduce_ ximg attn_prefill > $O/pmc_prefill_$t.csv 2>> "$R/$O/pmc_pf_$t.err" && cat $O/pmc_prefill_$t.csv | cut -c1-400; rm -rf $O/pmc_pf_$t; done ;;
    trace:*) n=$(echo "${s#trace:}" | tr -c 'a-zA-Z0-9_=' '_'); THK_LIB=$R/token-hawk_amd/libthk_trace.so timeout 300 python tools/step_trace.py $(echo "${s#trace:}" | tr '+' ' ') > $O/trace_$n.txt 2> $O/trace_$n.err; echo "trace exit $?"; cat $O/trace_$n.txt; cp gpurun_out/step_trace.json $O/trace_$n.json 2>/dev/null ;;
    profcfg:*) n=$(echo "${s#profcfg:}" | tr -c 'a-zA-Z0-9_=' '_'); (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_$n" -o p -- python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile $(for t in $(echo "${s#profcfg:}" | tr '+' ' '); do [ "$t" != base ] && echo --tunable $t; done) > "$R/$O/prof_$n.json" 2> "$R/$O/prof_$n.err"); echo "profcfg exit $?"; f=$(find $O/prof_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_$n.csv && head -12 "$f" | cut -c1-160 ;;
    sh:*) bash -c "${s#sh:}" ;;
  esac
done
# keep the merge-back small: raw rocprof traces are not needed once the stats are copied out
find $O -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
find $O -name "*.db" -delete 2>/dev/null
du -sh $O
