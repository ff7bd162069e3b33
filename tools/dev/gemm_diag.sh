#!/bin/bash
# Development: what bounds the prefill GEMM loop.  One gpurun call:
#   python -c "import __graft_entry__ as g; [g.build_libthk(out=f'token-hawk_amd/libthk_{n}.so', defs=d) for n, d in
#              (('nolo', ('THK_PF_NOLO=1',)), ('fakew', ('THK_PF_FAKEW=1',)), ('nolo_fakew', ('THK_PF_NOLO=1', 'THK_PF_FAKEW=1')), ('pftrace', ('THK_PREFILL_TRACE=1',)))]"
#   gpurun -- 'bash tools/dev/gemm_diag.sh > gpurun_out/gemm_diag.txt'
# Per-GEMM kernel durations (rocprofv3 --kernel-trace, 8 layers of 7B width, median over 7 prompts) of the shipped kernels and of three diagnostic builds whose
# RESULTS ARE WRONG by construction, then the in-kernel phase trace at full occupancy and with one workgroup per row-block (whole K, a third of the CUs busy).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=$R/gpurun_out/gemm_diag_tmp; mkdir -p $O; export TMPDIR=/tmp
prof() { # name tokens
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/$1 -o p -- python $R/tools/dev/gemm_shapes.py 4096 256 8 $2 > $O/$1.out 2> $O/$1.err)
  f=$(find $O/$1 -name "*kernel_trace.csv" | head -1)
  echo "== $1: $(tail -1 $O/$1.out)"; python tools/dev/gemm_split.py "$f" 8
  rm -rf $O/$1
}
for M in 128 256; do
  prof shipped_M$M $M
  for v in nolo fakew nolo_fakew; do THK_LIB=$R/token-hawk_amd/libthk_$v.so prof ${v}_M$M $M; done
done
echo "== phase trace, 256 workgroups (cycles per wave; divide by the chunks per workgroup: qkv 32, wo 16, w13 43, w2 43)"
THK_LIB=$R/token-hawk_amd/libthk_pftrace.so timeout 300 python tools/dev/prefill_trace.py prefill_slab_tokens=128 2>&1 | tail -36
echo "== phase trace, one workgroup per row-block (qkv 48, wo 32, w13 86, w2 32 workgroups x whole K: 128 / 128 / 128 / 344 chunks; idle workgroups count as 0 in the means)"
THK_LIB=$R/token-hawk_amd/libthk_pftrace.so timeout 300 python tools/dev/prefill_trace.py prefill_slab_tokens=128 prefill_blocks_w2=32 prefill_blocks_wo=32 prefill_blocks_w13=86 prefill_blocks_qkv=48 2>&1 | tail -36
rm -rf $O
