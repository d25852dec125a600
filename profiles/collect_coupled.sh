#!/bin/bash
# rocprofv3 kernel stats of the coupled drivers (configs 3, 4, 5) -- run on the GPU box
set -e
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_coupled_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in 3 4 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c$C -- python3 $REPO/bench_coupled.py --configs $C > $OUT/c$C.json 2>$OUT/c$C.err
  cat $OUT/c$C.json
  cat $OUT/c$C/*/*_kernel_stats.csv | cut -c1-200
done
