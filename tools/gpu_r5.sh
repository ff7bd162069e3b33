#!/bin/bash
# round 5: one gpurun call = a list of shell stages; everything goes to gpurun_out/$1/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/$1; shift; mkdir -p $O
export TMPDIR=/tmp O
for s in "$@"; do
  echo "=== $s"
  bash -c "$s"
done
find $O -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
find $O -name "*.db" -delete 2>/dev/null
du -sh $O
