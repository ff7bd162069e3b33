mkdir -p gpurun_out/r02
for V in A B; do
  cp tools/dev/libs/libthk_$V.so token-hawk_amd/libthk.so
  echo "== variant $V"
  timeout 300 python tools/dev/engine_check.py tiny 7b 2>&1 | grep -v "engine=0" | tail -8
done
