#!/bin/bash
# Development, second part of tools/dev/gemm_diag.sh: the 128-token GEMM loop without its fragment reads (-DTHK_PF_NOREAD), without its LDS fill (-DTHK_PF_NODMA),
# and the combinations with -DTHK_PF_NOLO.  Build libthk_{noread,nodma,noread_nodma,noread_nolo,nodma_nolo}.so with __graft_entry__.build_libthk(out=..., defs=...) first.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=$R/gpurun_out/gemm_diag_tmp; mkdir -p $O; export TMPDIR=/tmp
prof() { # name tokens
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/$1 -o p -- python $R/tools/dev/gemm_shapes.py 4096 256 8 $2 > $O/$1.out 2> $O/$1.err)
  f=$(find $O/$1 -name "*kernel_trace.csv" | head -1)
  echo "== $1: $(tail -1 $O/$1.out)"; python tools/dev/gemm_split.py "$f" 8
  rm -rf $O/$1
}
prof shipped 128
for v in noread nodma noread_nodma noread_nolo nodma_nolo; do THK_LIB=$R/token-hawk_amd/libthk_$v.so prof $v 128; done
rm -rf $O
