"""Time k_psi_so on config-4 shaped input (8192 members, state after 240 steps): adaptive
mesh (bvp_refine=-1) against the fixed 8-fold mesh."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pymoc_amd as gpu
from pymoc_amd import configs

N = int(os.environ.get("N", 8192))
for refine in (8, -1):
  c = dict(configs.config4(N=N), bvp_refine=refine)
  e = gpu.TwoColEnsemble(c)
  e.run(241)
  gpu.synchronize()
  for _ in range(3):
    e.so.update(e._b_basin, e.bs_SO)
  gpu.synchronize()
  t0 = time.perf_counter()
  K = 50
  for _ in range(K):
    e.so.update(e._b_basin, e.bs_SO)
  gpu.synchronize()
  el = (time.perf_counter() - t0) / K
  print("refine %3d: k_psi_so %.1f us per update of %d members" % (refine, el * 1e6, N))
