"""Psi_SO.solve and Psi_Thermwind of one MOC update on two streams (the thermal-wind kernel needs
Psi_SO only for the forcing it writes at its very end): how much of their summed duration
would running them side by side hide?  Timing only (the thermal wind reads the previous
update's Psi_SO here)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pymoc_amd as gpu
from pymoc_amd import configs
from pymoc_amd.device import Stream

for CONFIG in (4, 5):
  N = {4: 8192, 5: 4096}[CONFIG]
  if CONFIG == 5:
    c = configs.config5(N=N)
    c["rest_mask"] = np.repeat(c["rest_mask"][None], N, axis=0)
    e = gpu.JN2018Ensemble(c)
    e.run(361)
    b1, b2, bs = e.cols.b.ptr, e.cols.b.ptr + e._off, e.ml.bs
  else:
    e = gpu.TwoColEnsemble(configs.config4(N=N))
    e.run(241)
    b1, b2, bs = e._b_basin, e._b_north, e.bs_SO
  gpu.synchronize()
  s2 = Stream()
  def pair(two):
    e.so.stream = s2 if two else None
    e.so.update(b1, bs)
    e.tw.update(b1, b2, store_psib=False)
  for two in (False, True, False, True):
    for _ in range(3):
      pair(two)
    gpu.synchronize(); s2.sync()
    t0 = time.perf_counter()
    R = 30
    for _ in range(R):
      pair(two)
    gpu.synchronize(); s2.sync()
    print("config %d: psi_so + thermwind %s: %.1f us per update" %
          (CONFIG, "on two streams" if two else "in one stream", (time.perf_counter() - t0) / R * 1e6), flush=True)
  e.so.stream = None
