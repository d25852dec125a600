"""How often the step loop of k_jn2018_fast takes its rare paths (profiling build:
make -B lib EXTRA="-DPM_PHASE_PROFILE -DJF_COUNT_RARE"; afterwards make -B lib)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pymoc_amd as gpu
from pymoc_amd import configs, _lib
N = 4096
c = configs.config5(N=N)
c["rest_mask"] = np.repeat(c["rest_mask"][None], N, axis=0)
e = gpu.JN2018Ensemble(c)
e.run(360)
gpu.synchronize()
out = (C.c_ulonglong * 72)()
_lib.lib.pm_debug_jf_rare(out)
e.run(360)
gpu.synchronize()
_lib.lib.pm_debug_jf_rare(out)
v = np.array(list(out), dtype=np.float64)
names = ["member-steps", "convect: new pattern", "coefficient set changed", "interp: a point left its interval (wave)",
         "interp: binary search (wave)", "argmin: general path", "convect: adjusting (column-steps)"]
for n, x in zip(names, v):
  print("%-44s %12.0f  = %.4f per member-step" % (n, x, x / v[0]))
print("lanes whose point left its interval, per member-step:", np.round(v[8:72] / v[0], 3).tolist())
