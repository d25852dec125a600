#!/bin/bash
# Do HIP-event spans less the empty-span time equal rocprofv3's kernel durations?  One run of
# probe_kernels.py (config 3) under rocprofv3 --kernel-trace --stats: the script's run averages
# (events, null span subtracted) beside the trace's AverageNs of the same launches.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_nullcheck
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/profiles/r05/probe_kernels.py 3 5 > $OUT/probe.log 2>> $OUT/err.txt
cat $OUT/probe.log
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs cat | cut -d, -f1-5 | head -8
