#!/bin/bash
# A/B on one box: the fused JN2018 loop's lag-based issue priority refreshed every step (tree)
# against every 2nd / 4th / 8th step (variant builds with -DJF_PRIO_EVERY=N under
# profiles/r05/var_prio<N>/, loaded through PYMOC_HIP_LIB)
cd "$(dirname "$0")/../.." || exit 1
out=gpurun_out/r05_ab_prio.log
: > $out
for v in "" 2 4 8 ""; do
  echo "== JF_PRIO_EVERY=${v:-1 (tree)}" >> $out
  if [ -n "$v" ]; then export PYMOC_HIP_LIB=$PWD/profiles/r05/var_prio$v/libpymoc_hip.so; else unset PYMOC_HIP_LIB; fi
  timeout -k 10 200 python profiles/r05/probe_kernels.py 5 >> $out 2>&1 || exit 1
done
cat $out
