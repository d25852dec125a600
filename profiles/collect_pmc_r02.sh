#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: SQ counters of one kernel of a bench
# configuration, rocprofv3 --pmc in its own passes (never combined with tracing).
#   bash profiles/collect_pmc_r02.sh <tag> <kernel-substring> <bench args...>
TAG=$1; KERNEL=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --no-cpu-baseline --no-single-step --no-coupled $*"
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- $BENCH > $OUT/bench_$i.json 2>> $OUT/err.txt
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "$KERNEL" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as out:
    for k, v in sorted(acc.items()):
        w = v[len(v) // 4:]  # drop warm-up launches
        line = "%-24s mean per launch %.5g  (n=%d of %d)" % (k, sum(w) / len(w), len(w), len(v))
        print(line); out.write(line + "\n")
PY
tail -3 $OUT/err.txt
