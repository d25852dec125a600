#!/bin/bash
# SQ instruction counters of the fused JN2018 kernels (split layout vs round 4's), bench.py's
# config-5 headline run under rocprofv3 --pmc (counters only: never combined with tracing).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_pmc_jn
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in split nosplit; do
  if [ $v = split ]; then export PYMOC_JN_SPLIT=1; else unset PYMOC_JN_SPLIT; fi
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $OUT/$v -- python3 $REPO/bench.py --no-cpu-baseline --config 5 --steps 10 --warmup 2 > $OUT/$v.json 2>> $OUT/err.txt || { echo "$v failed"; tail -5 $OUT/err.txt; exit 1; }
  echo "$v done"
  find $OUT/$v -name "*agent_info*" -delete
done
python3 - <<PY
import csv, glob, collections
for v in ("split", "nosplit"):
  f = glob.glob("$OUT/%s/**/*counter_collection.csv" % v, recursive=True)
  agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
  for fn in f:
    for r in csv.DictReader(open(fn)):
      k = r["Kernel_Name"].split("(")[0]
      agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
      if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
  for k, d in agg.items():
    if "jn2018" in k and cnt[k]:
      w = d["SQ_WAVES"]
      print(v, k[:60], "launches", cnt[k], {c: round(x / w, 1) for c, x in d.items() if c != "SQ_WAVES"})
PY
