"""config 4: updates side by side (two streams, PM_OP_WA_PSI) against serial updates, same box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pymoc_amd as gpu
from pymoc_amd import configs
N = 8192
cfg = configs.config4(N=N)
for rep in range(2):
  for ov in (False, True):
    e = gpu.TwoColEnsemble(cfg, overlap_updates=ov)
    e.run(241)
    gpu.synchronize()
    t0 = time.perf_counter()
    e.run(2400)
    gpu.synchronize()
    dt = time.perf_counter() - t0
    print("overlap_updates=%s: %.3e coupled steps/s (%.1f us per 24-step interval)" % (ov, N * 2400 / dt, dt / 100 * 1e6), flush=True)
    del e
