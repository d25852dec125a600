#!/bin/bash
# A/B on one box: K1 inside the two-column loops with a member's two columns on consecutive
# wavefronts (PM_OP_PAIRED) against all basin columns first
cd "$(dirname "$0")/../.." || exit 1
out=gpurun_out/r05_ab_paired.log
: > $out
for v in 0 1 0 1; do
  echo "== PYMOC_K1_PAIRED=$v" >> $out
  PYMOC_K1_PAIRED=$v timeout -k 10 200 python profiles/r05/probe_kernels.py 3 4 >> $out 2>&1 || exit 1
done
cat $out
