#!/bin/bash
# SQ counters of the benchmarked k_column_steps launch (run on the GPU box): how busy the
# vector ALU is and what the waves wait for.  Separate pass from kernel-trace, as required.
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_alu_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --no-cpu-baseline --no-single-step --steps 4 --warmup 1"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/sq -- $BENCH > $OUT/bench.json 2> $OUT/err.txt
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/grbm -- $BENCH > /dev/null 2>> $OUT/err.txt
python3 - <<PY
import csv, glob, collections
for sub in ("sq", "grbm"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv" % sub):
        for r in csv.DictReader(open(f)):
            if "k_column_steps" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        v = v[1:]
        print("%-22s mean per launch %.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
