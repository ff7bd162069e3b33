mkdir -p gpurun_out/r02
timeout 100 python tools/dev/engine_check.py tiny 2>&1 | grep "engine=1"
for cfg in "engine_park=0" "engine_park=1" "engine_park=0 engine_dev_ring=9" "engine_park=1 engine_dev_ring=9" "engine_park=1 engine_dev_ring=8"; do
  echo "== $cfg"; timeout 200 python tools/dev/engine_trace.py $cfg 2>&1 | head -1
done
