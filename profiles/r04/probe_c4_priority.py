"""Config 4's update pair (Psi_SO.solve on the side stream, thermal wind on the main stream) with
the side stream at normal / highest priority (PYMOC_SIDE_PRIORITY); child process per setting."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, time, os
sys.path.insert(0, sys.argv[1])
import pymoc_amd
from pymoc_amd import configs
from pymoc_amd.device import Stream
st = Stream()
for rep in range(3):
  e = pymoc_amd.TwoColEnsemble(configs.config4(), stream=st)
  e.run(10 * e.M); st.sync()
  t0 = time.perf_counter(); e.run(2400); st.sync(); t = time.perf_counter() - t0
  print("side stream priority %s: %.4g coupled steps/s, %.1f us per interval" % (os.environ.get("PYMOC_SIDE_PRIORITY", "0"), 8192 * 2400 / t, t / (2400 / e.M) * 1e6), flush=True)
  del e
'''
for rep in range(2):
  for pr in ("0", "1"):
    subprocess.run([sys.executable, "-c", CHILD, ROOT], env=dict(os.environ, PYMOC_SIDE_PRIORITY=pr), check=True)
