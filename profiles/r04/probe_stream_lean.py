"""The lean streaming form (b and weff in flight, kappa formed, Area a scalar): ring depth and
columns per wave.  Each setting in a child process (the settings are read once per process)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, os
sys.path.insert(0, sys.argv[1])
import numpy as np, pymoc_amd
from pymoc_amd import configs
from pymoc_amd.device import Stream, Event
st = Stream()
c = configs.config2(N=262144)
b = pymoc_amd.ColumnBatch(c["z"], c["kappa"], c["Area"], c["b0"], bs=c["bs"], bbot=c["bbot"], N2min=c["N2min"],
                          do_conv=c["do_conv"], stream=st, kappa_affine=(c["kappa_back"], c["kappa_profile"]))
wA = pymoc_amd.DeviceArray.from_host(c["wA"], stream=st)
w = b.combine_forcing(wA)
for _ in range(3): b.steps(w, c["dt"], 1, precombined=True)
e0, e1 = Event(), Event(); e0.record(st)
for _ in range(20): b.steps(w, c["dt"], 1, precombined=True)
e1.record(st); st.sync()
ms = e0.elapsed_ms(e1) / 20
print("D=%s cpw=%s: %.1f us, %.3e column-steps/s, %.0f GB/s on 24 nz" % (os.environ.get("PYMOC_STREAM_LEAN_D", "6"),
      os.environ.get("PYMOC_STREAM_CPW", "auto"), ms * 1e3, 262144 / (ms * 1e-3), 2400 * 262144 / (ms * 1e-3) / 1e9), flush=True)
'''
for d in ("4", "5", "6", "8"):
  for cpw in ("0", "8", "16", "64"):
    env = dict(os.environ, PYMOC_STREAM_LEAN_D=d, PYMOC_STREAM_CPW=cpw)
    subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, check=True)
