"""Phase clocks of k_jn2018_fast (config 5 at its BASELINE size).  Needs the profiling build:
make -B lib EXTRA=-DPM_PHASE_PROFILE ; afterwards  make -B lib  restores the product.
ARITH=contracted selects the tolerance-mode columns."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pymoc_amd as gpu
from pymoc_amd import configs, _lib

N = 4096
e = gpu.JN2018Ensemble(configs.config5(N=N), arith=os.environ.get("ARITH", "exact"))
e.run(361 + 35)
gpu.synchronize()
out = (C.c_ulonglong * 16)()
_lib.lib.pm_debug_prof(out)
K = 10
for _ in range(K):
  e._fused_steps(36)
gpu.synchronize()
_lib.lib.pm_debug_prof(out)
v = np.array(list(out), dtype=np.float64)
names = ["BC switch + columns", "row write + interp", "argmin / upwell", "tendencies", "PCR + BC",
         "-", "loop control / priority"]
tot = v[:7].sum()
for n, x in zip(names, v[:7]):
  print("%-26s %8.0f cycles per wave and step  %5.1f %%" % (n, x / v[15] / 36, 100 * x / tot))
print("total %.0f cycles per wave and step (%d reports)" % (tot / v[15] / 36, v[15]))
print("per launch (cycles per wave): block tables %.0f, PCR multipliers / constants (wave 0 only; per reporting wave) %.0f, "
      "column loads %.0f, mixed-layer loads %.0f, barrier %.0f, coefficient loads %.0f; slot 6 holds the rest of the "
      "launch's prologue next to the loop control" % tuple(v[k] / v[15] for k in (7, 8, 9, 10, 11, 12)))
