#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: SQ issue counters of the coupled
# configs' kernels (BASELINE configs 3, 4, 5 as `bench.py --config N`), rocprofv3 --pmc in its
# own passes, never combined with tracing.  Summarised by profiles/summarize_coupled_r02.py into
# profiles/coupled_counters.json, which bench.py replays (labelled as replayed).
# NOTE gpurun MERGES the box's gpurun_out/ into the local one: delete the local output
# directory of an earlier collection before summarising, or the summary averages old and new runs.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_coupled
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in 3 4 5; do
  BENCH="python3 $REPO/bench.py --no-cpu-baseline --config $C --steps 10 --warmup 2"
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/c${C}_sq -- $BENCH > $OUT/bench_c${C}.json 2>> $OUT/err.txt
  rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $OUT/c${C}_grbm -- $BENCH > /dev/null 2>> $OUT/err.txt
  echo "config $C done"
done
du -sh $OUT; tail -2 $OUT/err.txt
