#!/bin/bash
# rocprofv3 kernel stats of one command (run on the GPU box): bash profiles/collect_one.sh <tag> <cmd...>
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- "$@" > $OUT/stdout.txt 2> $OUT/stderr.txt
cat $OUT/stdout.txt | cut -c1-400
cat $OUT/*/*_kernel_stats.csv | cut -c1-180
