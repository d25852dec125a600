#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: collects the rocprofv3 evidence
# behind bench.py's roofline block.  Usage: bash profiles/collect.sh <tag>
#   1. --kernel-trace --stats of the default bench run        -> per-kernel durations
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) -> HBM-side bytes per launch
#   3. the same two counters on a beyond-cache single-step run (262144 columns), whose true
#      traffic is known, to calibrate the counters for this kernel's 8-B-per-lane accesses
set -e
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --no-cpu-baseline --no-single-step"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace_bench.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch_bench.json
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write_bench.json
CAL="python3 $REPO/bench.py --no-cpu-baseline --no-single-step --columns 262144 --steps 20 --warmup 2 --steps-per-launch 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cal_trace -- $CAL > $OUT/cal_trace_bench.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- $CAL > $OUT/cal_fetch_bench.json
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -- $CAL > $OUT/cal_write_bench.json
find $OUT -name "*.csv" | head -50
du -sh $OUT
