#!/bin/bash
# ring depth x columns per wave of the straight-line streaming kernel (k_column_stream<2,D,...,CPWU>),
# library built with EXTRA=-DPM_STREAM_VARIANTS in a scratch copy (profiles/r05/var_stream/)
cd "$(dirname "$0")/../.." || exit 1
export PYMOC_HIP_LIB=$PWD/profiles/r05/var_stream/libpymoc_hip.so
out=gpurun_out/r05_stream_variants.log
: > $out
run() { python bench.py --config 2 --no-coupled --no-cpu-baseline --steps 5 --warmup 2 | python -c "import sys,json; d=json.loads(sys.stdin.read())['hbm_regime']; print('$1', d['kernel_us'], d['frac'])" >> $out 2>&1; }
for rep in 1 2; do
  unset PYMOC_STREAM_VARIANT; run "default(5,16)"
  for v in 4,16 6,16 3,16 5,32 5,8; do export PYMOC_STREAM_VARIANT=$v; run "variant($v)"; done
done
cat $out
