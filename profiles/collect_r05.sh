#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root.  Round 5 evidence, one directory per
# BASELINE config so that no kernel's statistics mix launches of different configs:
#   gpurun_out/r05/c<N>/trace   rocprofv3 --kernel-trace --stats        (average kernel durations)
#   gpurun_out/r05/c<N>/pmc<k>  rocprofv3 --pmc ... in passes of their own (never with tracing)
#   gpurun_out/r05/c<N>/bench.json  the bench line of the traced run
# c2 = bench.py --config 2 (headline kernel + contracted mode; --no-coupled), c2s = the same
# with the one-step / HBM-regime entries (k_column_stream); c3, c4, c5, c6 = bench.py --config N
# --breakdown-only: the FULL-LENGTH run (10 warm-up intervals + 2400 / 2400 / 3600 / 2400 steps)
# whose HIP-event averages the default line's c<N> blocks report, so that every `frac` of the
# line recomputes from <c>_kernel_stats.csv (config 4 / 6: the update's launches one after the
# other = each kernel alone; c4o / c6o: side by side on two streams, as the drivers run them).
# Summarised by profiles/summarize_r05.py into profiles/r05/ (tracked) -- copy, do not cite
# gpurun_out/.  NOTE gpurun MERGES the box's gpurun_out/ into the local one: delete the local
# gpurun_out/r05 of an earlier collection before summarising.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, bench args
  local name=$1; shift
  local D=$OUT/$name; mkdir -p $D
  local CMD="python3 $REPO/bench.py --no-cpu-baseline $*"
  local T="timeout -k 10 300"   # (a pass that dies must not hold the box)
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- $CMD > $D/bench.json 2>> $D/err.txt
  echo "$name trace done"
  $T rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $D/pmc1 -- $CMD > /dev/null 2>> $D/err.txt
  $T rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --output-format csv -d $D/pmc2 -- $CMD > /dev/null 2>> $D/err.txt
  echo "$name pmc1-2 done"
  $T rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $D/pmc3 -- $CMD > /dev/null 2>> $D/err.txt
  # (the two traffic counters do not fit one pass: "exceeds the capabilities of the hardware")
  $T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc4 -- $CMD > /dev/null 2>> $D/err.txt
  $T rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc5 -- $CMD > /dev/null 2>> $D/err.txt
  find $D -name "*agent_info*" -delete
  find $D -name "*kernel_trace.csv" -delete   # (per-dispatch rows: large; the stats file is kept)
  echo "$name done: $(tail -c 300 $D/bench.json | head -c 120)"
}
run c2 --config 2 --no-coupled --no-single-step --steps 20 --warmup 5
run c2s --config 2 --no-coupled --steps 5 --warmup 2
run c3 --config 3 --breakdown-only
run c4 --config 4 --breakdown-only
run c4o --config 4 --breakdown-only --overlap
run c5 --config 5 --breakdown-only
run c6 --config 6 --breakdown-only
du -sh $OUT; tail -2 $OUT/*/err.txt | tail -4
