"""Two-basin driver (2048 members): us per interval with / without graph replay, the forcing formed by
the column kernel (PM_OP_WA_TWOBASIN) or by pm_twobasin_forcing, update pairs on two streams or one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pymoc_amd as gpu
from pymoc_amd import configs
from pymoc_amd.device import Stream
cfg = configs.config_twobasin(N=2048)
for rep in range(2):
  for graph in (True, False):
    for k1 in (True, False):
      for ov in (True, False):
        s = Stream()
        e = gpu.TwoBasinEnsemble(cfg, stream=s, use_graph=graph, overlap_updates=ov)
        e._forcing_in_k1 = k1
        e.run(1 + 10 * e.M)
        s.sync()
        t0 = time.perf_counter()
        e.run(100 * e.M)
        s.sync()
        dt = time.perf_counter() - t0
        print("graph=%-5s forcing_in_k1=%-5s two_streams=%-5s  %.1f us per interval" % (graph, k1, ov, dt / 100 * 1e6), flush=True)
        del e
