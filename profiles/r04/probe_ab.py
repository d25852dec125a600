"""A/B of two builds of the engine on ONE box: the tree's own (pymoc_amd/) against a copy of
another commit's package in profiles/r04/ab_r03/ (round 3's HEAD, built by hand, git-ignored).
Each build runs in its own child process (a process can load one libpymoc_hip.so).
usage: python profiles/r04/probe_ab.py [configs: 3 4 5]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, time, os
sys.path.insert(0, sys.argv[1])
import numpy as np
import pymoc_amd
from pymoc_amd import configs
from pymoc_amd.device import Stream
st = Stream()
for c in sys.argv[3:]:
  c = int(c)
  for rep in range(2):
    if c == 3:
      cfg = configs.config3(); e = pymoc_amd.TwoColEnsemble(cfg, stream=st); n, steps = 4096, 2400
    elif c == 4:
      cfg = configs.config4(); e = pymoc_amd.TwoColEnsemble(cfg, stream=st); n, steps = 8192, 2400
    else:
      cfg = configs.config5(); cfg["rest_mask"] = np.repeat(cfg["rest_mask"][None], 4096, axis=0)
      e = pymoc_amd.JN2018Ensemble(cfg, stream=st); n, steps = 4096, 3600
    e.run(10 * e.M); st.sync()
    t0 = time.perf_counter(); e.run(steps); st.sync(); t = time.perf_counter() - t0
    print("%s config %d: %.4g coupled steps/s, %.1f us per interval" % (sys.argv[2], c, n * steps / t, t / (steps / e.M) * 1e6), flush=True)
    del e
'''
OTHER = os.path.join(ROOT, "profiles", "r04", os.environ.get("PYMOC_AB_OTHER", "ab_r03"))
OTAG = os.environ.get("PYMOC_AB_OTHER", "r03 ")[-4:]
for tag, path in ((OTAG, OTHER), ("tree", ROOT), (OTAG, OTHER), ("tree", ROOT)):
  subprocess.run([sys.executable, "-c", CHILD, path, tag] + (sys.argv[1:] or ["5"]), check=True)
