#!/bin/bash
# Round 5, VERDICT r4 item 1: the diagnostic exchange off the critical path.  Config 4 (8192
# members, a gather every 240 steps = 10 MOC intervals) and config 5 as bench.py's headline, on
# the one GPU of a box: without a communicator (the pack alone), and with a one-rank RCCL
# communicator (--force-rccl: every gather a real collective on device buffers), exchange on the
# communication stream / in line, all-gather / gather to root.
cd "$(dirname "$0")/../.." || exit 1
out=gpurun_out/r05_gather.log
: > $out
for cfg in 4 5; do
  for extra in "" "--force-rccl" "--force-rccl --gather-inline" "--force-rccl --gather all" "--force-rccl --gather all --gather-inline"; do
    echo "== config $cfg $extra" >> $out
    timeout -k 10 300 python bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline $extra 2>&1 \
      | python -c "
import sys, json
for ln in sys.stdin:
  if ln.startswith('{'):
    d = json.loads(ln)
    print({k: d.get(k) for k in ('value', 'ms_per_step', 'steps_per_s', 'gathers_in_timed_region', 'rccl_collectives_in_timed_region', 'gather_bytes_per_rank')})
  else:
    print(ln.rstrip())
" >> $out || exit 1
  done
done
cat $out
