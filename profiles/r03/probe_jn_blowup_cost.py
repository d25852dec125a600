"""Does a lost (non-finite) member decide the fused JN2018 launch's duration?  Times config 5 with
the two members the reference itself loses (2, 1268) and with them replaced by copies of member 3."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pymoc_amd as gpu
from pymoc_amd import configs
N = 4096
for label, fix in (("with the lost members", False), ("lost members replaced", True)):
  c = configs.config5(N=N)
  c["rest_mask"] = np.repeat(c["rest_mask"][None], N, axis=0)
  if fix:
    for k, v in c.items():
      if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == N:
        v = v.copy()
        v[2] = v[3]
        v[1268] = v[3]
        c[k] = v
  e = gpu.JN2018Ensemble(c)
  e.run(72)
  gpu.synchronize()
  t0 = time.perf_counter()
  e.run(3600)
  gpu.synchronize()
  el = time.perf_counter() - t0
  print("%-24s %.2f us per 36-step block, %.3g coupled steps/s, lost: %s" % (
      label, el / 100 * 1e6, N * 3600 / el, e.nonfinite_members().tolist()))
