"""Launch time of the fused JN2018 kernel against the number of fused steps: the intercept is
what a launch pays before / after its time loop (loads, block tables, PCR multipliers, stores)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pymoc_amd as gpu
from pymoc_amd import configs

N = 4096
e = gpu.JN2018Ensemble(configs.config5(N=N), arith=os.environ.get("ARITH", "exact"))
e.run(361 + 35)
gpu.synchronize()
for k in (1, 2, 4, 9, 18, 36, 72):
  for _ in range(3):
    e._fused_steps(k)
  gpu.synchronize()
  t0 = time.perf_counter()
  R = 20
  for _ in range(R):
    e._fused_steps(k)
  gpu.synchronize()
  dt = (time.perf_counter() - t0) / R
  print("%3d steps per launch: %7.1f us per launch, %6.2f us per step" % (k, dt * 1e6, dt * 1e6 / k))
