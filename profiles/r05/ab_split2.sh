#!/bin/bash
# same-box A/B, other shapes / modes: nz = 81 (the script's own grid), contracted columns
cd "$(dirname "$0")/../.." || exit 1
out=gpurun_out/r05_ab_split2.log
: > $out
for v in "" 1; do
  echo "== PYMOC_JN_SPLIT=$v" >> $out
  if [ -n "$v" ]; then export PYMOC_JN_SPLIT=1; else unset PYMOC_JN_SPLIT; fi
  timeout -k 10 200 python profiles/r05/probe_kernels.py 5 --nz81 >> $out 2>&1 || exit 1
  timeout -k 10 200 python profiles/r05/probe_kernels.py 5 --contracted >> $out 2>&1 || exit 1
done
cat $out
