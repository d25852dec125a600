"""Does splitting a coupled ensemble over K HIP streams (members are independent) raise the
throughput of one GPU?  Each sub-ensemble's kernels leave the machine partly idle (memory-bound
prologues, tails); kernels of other streams can fill those gaps.  CONFIG=3|4|5."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pymoc_amd as gpu
from pymoc_amd import configs
from pymoc_amd.device import Stream

CONFIG = int(os.environ.get("CONFIG", 5))
N = {3: 4096, 4: 8192, 5: 4096}[CONFIG]
INTERVALS = 20

def make(k, K, stream):
  sl = (k * N // K, (k + 1) * N // K)
  if CONFIG == 5:
    cfg = configs.config5(N=N, members=sl)
    cfg["rest_mask"] = np.repeat(cfg["rest_mask"][None], sl[1] - sl[0], axis=0)
    return gpu.JN2018Ensemble(cfg, stream=stream)
  cfg = configs.config3(N=N, members=sl) if CONFIG == 3 else configs.config4(N=N, members=sl)
  return gpu.TwoColEnsemble(cfg, stream=stream)

for K, stagger in ((1, False), (2, False), (2, True), (4, False), (4, True), (8, True)):
  streams = [Stream() for _ in range(K)]
  ens = [make(k, K, streams[k]) for k in range(K)]
  M = ens[0].M
  for k, e in enumerate(ens):
    e.run(2 * M + ((k * M) // K if stagger else 0))
  for s in streams:
    s.sync()
  t0 = time.perf_counter()
  for _ in range(INTERVALS):
    for e in ens:
      e.run(M)
  for s in streams:
    s.sync()
  dt = time.perf_counter() - t0
  print("config %d, %d stream(s)%s: %.3e coupled steps/s (%.1f us per interval)"
        % (CONFIG, K, " staggered" if stagger else "", N * M * INTERVALS / dt, dt / INTERVALS * 1e6), flush=True)
  del ens, streams
