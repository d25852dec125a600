"""Run-average duration of every kernel of a coupled config (LaunchTimer: events around every
launch of a full-length run).  usage: python profiles/r04/probe_kernels.py 3 4 5"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pymoc_amd
from pymoc_amd import configs
from pymoc_amd.device import Stream, LaunchTimer
st = Stream()
for c in (sys.argv[1:] or ["5"]):
  c = int(c)
  for rep in range(2):
    if c == 3:
      e = pymoc_amd.TwoColEnsemble(configs.config3(), stream=st); steps = 2400
    elif c == 4:
      e = pymoc_amd.TwoColEnsemble(configs.config4(), stream=st, overlap_updates=False); steps = 2400
    else:
      cfg = configs.config5(); cfg["rest_mask"] = np.repeat(cfg["rest_mask"][None], 4096, axis=0)
      e = pymoc_amd.JN2018Ensemble(cfg, stream=st); steps = 3600
    e.run(10 * e.M); st.sync()
    e.timer = LaunchTimer(); e.run(steps)
    print("config", c, {k: (n, round(1e3 * t / n, 2)) for k, (n, t) in e.timer.summary().items()}, flush=True)
    del e
