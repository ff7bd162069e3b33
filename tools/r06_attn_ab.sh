#!/bin/bash
# Round 6: the f16-KV attention lane mapping (8 elements = 16 bytes per lane) against the round-5 library on one box, alternating arms.
#   gpurun -- 'bash tools/r06_attn_ab.sh OUTDIR'   (libthk_old.so = round-5 attention, libthk.so = 16-byte lanes with UB = 4, libthk_ub8.so = UB = 8)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/$1; mkdir -p $O
P=$PWD/token-hawk_amd
for rep in 1 2; do
  for lib in old new ub8; do
    f=$P/libthk_$lib.so; [ $lib = new ] && f=$P/libthk.so
    [ -f $f ] || continue
    echo "== $lib rep $rep ctx 2048"
    THK_LIB=$f timeout 300 python tools/attn_ctx_sweep.py 2048 base kv_f16=1 kv_f16=1,attn_splits=4 2>> $O/err.log | sed "s/^{/{\"lib\": \"$lib\", /" | tee -a $O/ctx2048.jsonl
    echo "== $lib rep $rep ctx 512"
    THK_LIB=$f timeout 300 python tools/ab.py --reps 5 --steps 96 --out r06_attn_ab_512.jsonl base kv_f16=1 2>> $O/err.log | sed "s/^{/{\"lib\": \"$lib\", /" | tee -a $O/ctx512.jsonl
  done
done
