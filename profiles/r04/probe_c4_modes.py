"""Config 4: the two updates side by side on two streams (default) against one after the other."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pymoc_amd
from pymoc_amd import configs
from pymoc_amd.device import Stream
st = Stream()
cfg = configs.config4()
for rep in range(2):
  for ov in (True, False):
    e = pymoc_amd.TwoColEnsemble(cfg, stream=st, overlap_updates=ov)
    e.run(241); st.sync()
    t0 = time.perf_counter(); e.run(2400); st.sync(); pymoc_amd.synchronize(); t = time.perf_counter() - t0
    print("overlap_updates=%s: %.4g coupled steps/s, %.1f us per interval" % (ov, 8192 * 2400 / t, t / 100 * 1e6), flush=True)
    del e
