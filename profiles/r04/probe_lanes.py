"""K1 inside the coupled two-column loop with 64 / 32 / 16 lanes per column (run-average kernel
durations, LaunchTimer) and the whole-loop rate.  usage: python profiles/r04/probe_lanes.py 3 4"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pymoc_amd
from pymoc_amd import configs
from pymoc_amd.device import Stream, LaunchTimer
st = Stream()
for c in (sys.argv[1:] or ["3"]):
  c = int(c)
  cfg = configs.config3() if c == 3 else configs.config4()
  ref = None
  for lanes in (64, 32, 16, 64, 32):
    e = pymoc_amd.TwoColEnsemble(cfg, stream=st, lanes_per_col=lanes)
    e.run(10 * e.M); st.sync()
    t0 = time.perf_counter(); e.run(2400); st.sync(); t = time.perf_counter() - t0
    s = e.state()
    if ref is None:
      ref = s
    same = all(np.array_equal(s[k], ref[k], equal_nan=True) for k in ref)
    e2 = pymoc_amd.TwoColEnsemble(cfg, stream=st, lanes_per_col=lanes, **({"overlap_updates": False} if c == 4 else {}))
    e2.run(10 * e2.M); st.sync()
    e2.timer = LaunchTimer(); e2.run(2400)
    print("config", c, "lanes", lanes, "%.4g coupled steps/s" % (e.n * 2400 / t), "bitwise equal to 64 lanes:", same,
          {k: (n, round(1e3 * tt / n, 2)) for k, (n, tt) in e2.timer.summary().items()}, flush=True)
    del e, e2
