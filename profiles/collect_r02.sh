#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: the rocprofv3 evidence behind
# bench.py's roofline blocks, round 2.
#   1. --kernel-trace --stats of the DEFAULT bench line (config 2 + the coupled configs 3-5)
#   2. --pmc passes (each its own run, never combined with tracing) on the config-2 part:
#      SQ issue counters, FETCH_SIZE, WRITE_SIZE; the same two traffic counters on the
#      memory-bound regime (262144 columns, one step per launch), whose true traffic is known,
#      to calibrate them for this kernel's accesses (gfx950: FETCH_SIZE reads ~1/2)
set -e
# NOTE gpurun MERGES the box's gpurun_out/ into the local one: delete the local output
# directory of an earlier collection before summarising, or the summary averages old and new runs.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_r02
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FULL="python3 $REPO/bench.py --no-cpu-baseline"
K1="python3 $REPO/bench.py --no-cpu-baseline --no-single-step --no-coupled"
CAL="python3 $REPO/bench.py --no-cpu-baseline --no-single-step --no-coupled --columns 262144 --steps 20 --warmup 2 --steps-per-launch 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $FULL > $OUT/bench_traced.json 2> $OUT/err.txt
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- $K1 > $OUT/bench_sq.json 2>> $OUT/err.txt
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_IFETCH --output-format csv -d $OUT/pmc_sq2 -- $K1 > /dev/null 2>> $OUT/err.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $K1 > /dev/null 2>> $OUT/err.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $K1 > /dev/null 2>> $OUT/err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cal_trace -- $CAL > $OUT/bench_cal.json 2>> $OUT/err.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- $CAL > /dev/null 2>> $OUT/err.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -- $CAL > /dev/null 2>> $OUT/err.txt
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_grbm -- $K1 > /dev/null 2>> $OUT/err.txt
find $OUT -name "*kernel_stats.csv" | head; du -sh $OUT; tail -2 $OUT/err.txt
