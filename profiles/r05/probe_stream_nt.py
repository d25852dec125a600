"""The lean streaming form with 16-byte accesses (k_column_stream<2,5,true,true,true>, 262144
columns, one step per launch): the tree against a build with non-temporal loads / stores
(-DPM_STREAM_NT, profiles/r05/var_nt/), same box, alternating."""
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
res = []
for rep in range(3):
  e0, e1 = Event(), Event(); e0.record(st)
  for _ in range(20): b.steps(w, c["dt"], 1, precombined=True)
  e1.record(st); st.sync()
  res.append(e0.elapsed_ms(e1) / 20 * 1e3)
print("%s: %s us per step; best %.3e column-steps/s = %.0f GB/s on 24 nz B; checksum %.17g" % (
    sys.argv[2], [round(x, 1) for x in res], 262144 / (min(res) * 1e-6), 2400 * 262144 / (min(res) * 1e-6) / 1e9,
    float(np.sum(b.get_b()))), flush=True)
'''
for tag, lib in (("tree", None), ("nt", "profiles/r05/var_nt/libpymoc_hip.so")) * 2:
  env = dict(os.environ)
  if lib: env["PYMOC_HIP_LIB"] = os.path.join(ROOT, lib)
  else: env.pop("PYMOC_HIP_LIB", None)
  subprocess.run([sys.executable, "-c", CHILD, ROOT, tag], env=env, check=True)
