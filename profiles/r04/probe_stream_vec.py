"""The lean streaming form with 16-byte accesses (VEC) against the per-level form, same tree
(PYMOC_STREAM_NO_VEC=1 selects the latter); each in a child process, alternating."""
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
for rep in range(3):
  e0, e1 = Event(), Event(); e0.record(st)
  for _ in range(20): b.steps(w, c["dt"], 1, precombined=True)
  e1.record(st); st.sync()
  ms = e0.elapsed_ms(e1) / 20
  print("%s cpw=%s: %.1f us, %.3e column-steps/s, %.0f GB/s on 24 nz" % ("per-level" if os.environ.get("PYMOC_STREAM_NO_VEC") else "16-byte  ",
        os.environ.get("PYMOC_STREAM_CPW", "auto"), ms * 1e3, 262144 / (ms * 1e-3), 2400 * 262144 / (ms * 1e-3) / 1e9), flush=True)
'''
for rep in range(2):
  for novec in ("1", ""):
    for cpw in (sys.argv[1:] or ["0"]):
      env = dict(os.environ, PYMOC_STREAM_CPW=cpw)
      if novec:
        env["PYMOC_STREAM_NO_VEC"] = "1"
      subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, check=True)
