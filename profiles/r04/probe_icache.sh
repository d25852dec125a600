#!/bin/bash
# Instruction-cache counters of config 4's kernels (k_psi_so is ~100 KB of code; the SQC's
# instruction cache holds 64 KB): run ON THE GPU BOX from the repo root.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_icache
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/bench.py --no-cpu-baseline --config 4 --steps 10 --warmup 2"
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVES --output-format csv -d $OUT/pmc1 -- $CMD > /dev/null 2>> $OUT/err.txt
timeout -k 10 300 rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT/pmc2 -- $CMD > /dev/null 2>> $OUT/err.txt
find $OUT -name "*agent_info*" -delete
python3 - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True)):
  acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
  for r in csv.DictReader(open(p)):
    k = r["Kernel_Name"].split("(")[0][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
  for k, v in acc.items():
    if "psi_so" in k or "thermwind" in k or "column_steps" in k:
      print(p.split("/")[-3], k, {c: round(x / max(n[k], 1)) for c, x in v.items()}, "launches", n[k])
PY
tail -3 $OUT/err.txt
