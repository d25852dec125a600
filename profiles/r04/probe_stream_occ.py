import os, subprocess, sys
ROOT = "/root/repo"
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
  print("D=%s cpw=%s: %.1f us" % (os.environ.get("PYMOC_STREAM_VEC_D", "5"), os.environ.get("PYMOC_STREAM_CPW", "auto"), ms * 1e3), flush=True)
'''
for rep in range(2):
  for d in ("5", "4", "3", "2"):
    for cpw in ("0", "8"):
      subprocess.run([sys.executable, "-c", CHILD, os.path.join(ROOT, "profiles/r04/ab_st5")], env=dict(os.environ, PYMOC_STREAM_VEC_D=d, PYMOC_STREAM_CPW=cpw), check=True)
