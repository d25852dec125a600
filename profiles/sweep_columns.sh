#!/bin/bash
# throughput of k_column_steps vs ensemble size and lane geometry (run on the GPU box)
for C in 1024 2048 4096 8192 16384 65536; do
  for G in 64 32 16; do
    python bench.py --no-cpu-baseline --no-single-step --columns $C --lanes $G --steps 4 --warmup 1 2>&1 | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('C=%6d G=%2d  %.3e col-steps/s  %.1f us/launch' % ($C, $G, d['value'], d['roofline']['kernel_ms_per_launch']*1e3))"
  done
done
