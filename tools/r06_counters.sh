#!/bin/bash
# Round 6: the decode counter set and the driver-protocol kernel stats, re-taken on the binary in the tree; regenerates profiles/pmc_traffic.json and
# profiles/kernel_durations.json (stamped with the binary) on the GPU box and brings them back under gpurun_out/OUTDIR/.
#   gpurun -- 'bash tools/r06_counters.sh OUTDIR'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/$1; mkdir -p $O
R=$PWD
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$R/$O/pmc_$c" -o p -- python "$R/tools/ab.py" --steps 4 --reps 1 use_graph=0 > /dev/null 2> "$R/$O/pmc_$c.err")
  echo "pmc $c exit $?"
  f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/counter_$c.csv
  rm -rf $O/pmc_$c
done
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof" -o p -- python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$R/$O/prof_bench.json" 2> "$R/$O/prof.err"); echo "prof exit $?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_driver_protocol.csv
rm -rf $O/prof
python tools/make_pmc_traffic.py $O/counter_FETCH_SIZE.csv $O/counter_WRITE_SIZE.csv $O/kernel_stats_driver_protocol.csv r06 > $O/make_pmc_traffic.out 2> $O/make_pmc_traffic.err; echo "make exit $?"
cp profiles/pmc_traffic.json profiles/kernel_durations.json $O/
python tools/pmc_summary.py $O/counter_FETCH_SIZE.csv 32 > $O/pmc_fetch_size_decode.csv
sed 's/FETCH_SIZE/WRITE_SIZE/g' tools/pmc_summary.py > /tmp/pmc_w.py; python /tmp/pmc_w.py $O/counter_WRITE_SIZE.csv 32 | sed 's/hbm_read_bytes_x2_corrected/bytes_x2_NOT_applicable_use_kib_x1024/' > $O/pmc_write_size_decode.csv
# the raw per-dispatch counter files are large: keep the summaries
ls -la $O; rm -f $O/counter_FETCH_SIZE.csv $O/counter_WRITE_SIZE.csv
cat $O/make_pmc_traffic.out | head -60
