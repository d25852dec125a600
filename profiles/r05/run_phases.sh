#!/bin/bash
# phase clocks / wave timelines of the profiling build (profiles/r05/prof_lib/: the tree built with
# EXTRA=-DPM_PHASE_PROFILE in a scratch copy, loaded through PYMOC_HIP_LIB)
cd "$(dirname "$0")/../.." || exit 1
export PYMOC_HIP_LIB=$PWD/profiles/r05/prof_lib/libpymoc_hip.so
out=gpurun_out/r05_phases.log
: > $out
for spec in "3 4096" "4 4096" "4 8192" "5 4096"; do
  set -- $spec
  echo "== thermal wind, CONFIG=$1 N=$2" >> $out
  CONFIG=$1 N=$2 timeout -k 10 200 python profiles/probe_tw_phases.py >> $out 2>&1 || exit 1
done
echo "== fused JN2018 loop (round 4 kernel)" >> $out
timeout -k 10 200 python profiles/r03/probe_jn_phases.py >> $out 2>&1 || exit 1
echo "== Psi_SO adaptive (config 4)" >> $out
timeout -k 10 200 python profiles/probe_so_phases.py >> $out 2>&1 || exit 1
cat $out
