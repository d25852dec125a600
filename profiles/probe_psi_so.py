"""k_psi_so time per update of the config-4 ensemble: with / without the GM boundary-value
solve and for different mesh refinements R (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pymoc_amd as gpu
from pymoc_amd import configs
from pymoc_amd.device import DeviceArray

n = 8192
cfg = configs.config4(N=n)
z, y = cfg["z"], cfg["y"]
b = DeviceArray.from_host(np.ascontiguousarray(cfg["b_basin0"]))
bs = DeviceArray.from_host(np.repeat(cfg["bs_SO"][None], n, axis=0) if np.ndim(cfg["bs_SO"]) == 1 else cfg["bs_SO"])
for label, kw in (("no BVP (c=None)", dict(c=None)), ("BVP R=1", dict(c=cfg["c"], bvp_refine=1)),
                  ("BVP R=2", dict(c=cfg["c"], bvp_refine=2)), ("BVP R=4", dict(c=cfg["c"], bvp_refine=4)),
                  ("BVP R=8", dict(c=cfg["c"], bvp_refine=8)), ("BVP R=16", dict(c=cfg["c"], bvp_refine=16))):
  t = gpu.PsiSOBatch(z, y, n, tau=cfg["tau"], KGM=cfg["KGM"], f=cfg["f"], L=cfg["L"],
                     bvp_with_Ek=cfg.get("bvp_with_Ek", False), **kw)
  for _ in range(3):
    t.update(b, bs)
  gpu.synchronize()
  t0 = time.perf_counter()
  for _ in range(20):
    t.update(b, bs)
  gpu.synchronize()
  print("%-18s %.1f us per update of %d members" % (label, (time.perf_counter() - t0) / 20 * 1e6, n))
