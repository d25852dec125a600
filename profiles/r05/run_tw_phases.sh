#!/bin/bash
# phase clocks of the thermal wind in the profiling build (profiles/r05/prof_lib/, see run_phases.sh)
cd "$(dirname "$0")/../.." || exit 1
export PYMOC_HIP_LIB=$PWD/profiles/r05/prof_lib/libpymoc_hip.so
out=gpurun_out/r05_tw_phases.log
: > $out
for spec in "3 4096" "4 8192" "5 4096"; do
  set -- $spec
  echo "== thermal wind, CONFIG=$1 N=$2" >> $out
  CONFIG=$1 N=$2 timeout -k 10 200 python profiles/probe_tw_phases.py >> $out 2>&1 || exit 1
done
cat $out
