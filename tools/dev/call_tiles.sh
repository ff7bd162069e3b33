#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r02
for t in "" "prefill_tile_qkv=128" "prefill_tile_w13=128" "prefill_tile_qkv=128 prefill_tile_w13=128" "prefill_tile_wo=256" "prefill_tile_w2=256" "prefill_blocks_wo=128" "prefill_blocks_qkv=192"; do
  echo "== $t"; timeout 200 python tools/bench_prefill.py 7b 128 prefill-only $t 2>&1 | tail -1 | cut -c1-140
done | tee gpurun_out/r02/tiles.txt
