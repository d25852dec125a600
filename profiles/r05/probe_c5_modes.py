"""Config 5 (4096 Jansen-Nadeau members, nz = 200): us per interval with the update as ONE launch
(pm_so_tw_update) or as Psi_SO.solve and the thermal wind in two."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pymoc_amd as gpu
from pymoc_amd import configs
from pymoc_amd.device import Stream
cfg = configs.config5(N=4096)
for rep in range(2):
  for one in (True, False):
    s = Stream()
    e = gpu.JN2018Ensemble(cfg, stream=s)
    e._one_update_launch = one and e._one_update_launch
    e.run(10 * e.M)
    s.sync()
    t0 = time.perf_counter()
    e.run(50 * e.M)
    s.sync()
    dt = time.perf_counter() - t0
    print("one_update_launch=%-5s  %.1f us per interval = %.3g coupled steps/s" % (one, dt / 50 * 1e6, 4096 * e.M * 50 / dt), flush=True)
    del e
