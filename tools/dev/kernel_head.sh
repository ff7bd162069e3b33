#!/bin/bash
# What stands between a decode kernel's entry and its first weight request?  Compiles thk_kernels.hip to gfx950 assembly and, for
# the default 7B instantiations, lists the scalar loads / waits / divisions / barriers ahead of the first `global_load ... nt`.
#   tools/dev/kernel_head.sh [out.s]      (about two minutes; no GPU needed)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
S=${1:-/tmp/thk_kernels.s}
cd "$ROOT/token-hawk_amd"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-value -Wno-int-to-pointer-cast -Wno-int-to-void-pointer-cast \
      -mllvm -amdgpu-kernarg-preload-count=14 -x hip --cuda-device-only -S csrc/thk_kernels.hip -o "$S" 2>/dev/null
# NR,U,NS,PRO,EPI,NT,NSP,PIPE,WPB of the 7B defaults (single-row qkv / wo / w13 / lm-head; w2 is gemv_quarter_kernel<22, 1>)
for k in "qkv:11gemv_kernelILi1ELi8ELi8ELi1ELi2ELb1ELi0ELb1ELi4EE" "wo:11gemv_kernelILi1ELi8ELi8ELi2ELi1ELb1ELi4ELb1ELi4EE" "w13:11gemv_kernelILi1ELi8ELi8ELi1ELi3ELb1ELi0ELb1ELi4EE" "w2:19gemv_quarter_kernelILi22ELi1EE" "head:11gemv_kernelILi1ELi8ELi8ELi1ELi4ELb1ELi0ELb1ELi4EE"; do
  name=${k%%:*}; sym=${k#*:}
  echo "== $name"
  awk "/^_ZN3thk${sym}[^:]*:/,/s_endpgm/" "$S" | grep -v "^\s*;" | grep -v "^\." | awk '
    /\.p2align/ { n = 0; on = 1; next }              # the entry used when the arguments were preloaded starts here (what precedes it is
    !on { next }                                     # the fall-back prologue for firmware without kernarg preload)
    { n++ }
    /global_load.* nt/ { print "  first weight request after", n, "instructions"; exit }
    /s_load|s_waitcnt|v_rcp|s_barrier|s_cbranch/ { printf "  %4d %s\n", n, $0 }
    /global_load/ { if (!p) { printf "  %4d (first activation request)\n", n; p = 1 } }'
done
