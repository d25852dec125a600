"""Time of k_thermwind on the config-3/4/5 state after spin-up (inputs fixed, outputs discarded)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pymoc_amd as gpu
from pymoc_amd import configs
for CONFIG in [int(c) for c in os.environ.get("CONFIGS", "3 5").split()]:
  N = {3: 4096, 4: 8192, 5: 4096}[CONFIG]
  if CONFIG == 5:
    c = configs.config5(N=N)
    c["rest_mask"] = np.repeat(c["rest_mask"][None], N, axis=0)
    e = gpu.JN2018Ensemble(c)
    e.run(361)
    b1, b2 = e.cols.b.ptr, e.cols.b.ptr + e._off
  else:
    e = gpu.TwoColEnsemble(configs.config3(N=N) if CONFIG == 3 else configs.config4(N=N))
    e.run(241)
    b1, b2 = e._b_basin, e._b_north
  gpu.synchronize()
  for _ in range(5):
    e.tw.update(b1, b2, store_psib=False)
  gpu.synchronize()
  t0 = time.perf_counter()
  R = 50
  for _ in range(R):
    e.tw.update(b1, b2, store_psib=False)
  gpu.synchronize()
  print("config %d: k_thermwind %.1f us per update of %d members" % (CONFIG, (time.perf_counter() - t0) / R * 1e6, N))
