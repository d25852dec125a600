"""Config 4 (8192 two-column + SO members): us per interval with Psi_SO.solve and the thermal wind of an
update side by side on two streams or one after the other."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pymoc_amd as gpu
from pymoc_amd import configs
from pymoc_amd.device import Stream
cfg = configs.config4(N=8192)
for rep in range(2):
  for ov in (True, False):
    s = Stream()
    e = gpu.TwoColEnsemble(cfg, stream=s, overlap_updates=ov)
    e.run(1 + 10 * e.M)
    s.sync()
    t0 = time.perf_counter()
    e.run(100 * e.M)
    s.sync()
    dt = time.perf_counter() - t0
    print("two_streams=%-5s  %.1f us per interval = %.3g coupled steps/s" % (ov, dt / 100 * 1e6, 8192 * e.M * 100 / dt), flush=True)
    del e
