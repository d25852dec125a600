#!/bin/bash
# A/B on ONE box: the split-lane fused JN2018 kernel (default) against round 4's kernel
# (default; PYMOC_JN_SPLIT=1 selects the split layout): run-average kernel durations of config 5 and launch time against the
# number of fused steps.
cd "$(dirname "$0")/../.." || exit 1
out=gpurun_out/r05_ab_split.log
: > $out
for v in "" 1; do
  echo "== PYMOC_JN_SPLIT=$v" >> $out
  if [ -n "$v" ]; then export PYMOC_JN_SPLIT=1; else unset PYMOC_JN_SPLIT; fi
  timeout -k 10 200 python profiles/r05/probe_kernels.py 5 >> $out 2>&1 || exit 1
  timeout -k 10 200 python profiles/r05/probe_jn_prologue.py >> $out 2>&1 || exit 1
done
cat $out
