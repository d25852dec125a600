"""Probe: cost of the convective-adjustment path in k_column_steps (1024 columns x 1000 steps).

Round-1 finding recorded in DESIGN.md: a variant that put one column on TWO wavefronts
(halo exchange through LDS every 12 steps) was bit-identical but 1.6x SLOWER than one
wave per column, so it was dropped."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pymoc_amd
from pymoc_amd import configs
from pymoc_amd.device import DeviceArray, Event
c = configs.config2(N=1024)
for conv in ("config2", "none", "all"):
  dc = c["do_conv"] if conv == "config2" else np.full(1024, conv == "all")
  for G in (64,):
    b = pymoc_amd.ColumnBatch(c["z"], c["kappa"], c["Area"], c["b0"], bs=c["bs"], bbot=c["bbot"],
                              N2min=c["N2min"], do_conv=dc)
    wA = DeviceArray.from_host(c["wA"])
    b.steps(wA, c["dt"], 1000, lanes_per_col=G)
    e0, e1 = Event(), Event()
    e0.record(); 
    for _ in range(10): b.steps(wA, c["dt"], 1000, lanes_per_col=G)
    e1.record(); pymoc_amd.synchronize()
    ms = e0.elapsed_ms(e1) / 10
    print("conv=%-8s G=%3d  %.1f us per 1000-step launch  %.3e col-steps/s" % (conv, G, ms * 1e3, 1024 * 1000 / (ms * 1e-3)))
# conv flag on everywhere but convection can never trigger (bs far above any b)
for label, bs in (("never-triggers", np.full(1024, 1.0)), ("always-triggers", np.full(1024, -1.0))):
  b = pymoc_amd.ColumnBatch(c["z"], c["kappa"], c["Area"], c["b0"], bs=bs, bbot=c["bbot"],
                            N2min=c["N2min"], do_conv=True)
  wA = DeviceArray.from_host(c["wA"])
  b.steps(wA, c["dt"], 1000, lanes_per_col=64)
  e0, e1 = Event(), Event()
  e0.record()
  for _ in range(10): b.steps(wA, c["dt"], 1000, lanes_per_col=64)
  e1.record(); pymoc_amd.synchronize()
  print("conv=all, %s: %.1f us per 1000-step launch" % (label, e0.elapsed_ms(e1) * 100))
