"""Phase clocks of k_thermwind (CONFIG=3|4|5 ensembles at their BASELINE sizes).  Needs the profiling
build:  make -B lib EXTRA=-DPM_PHASE_PROFILE ; afterwards  make -B lib  restores the product."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pymoc_amd as gpu
from pymoc_amd import configs, _lib
from pymoc_amd.device import DeviceArray

CONFIG = int(os.environ.get("CONFIG", 3))
N = int(os.environ.get("N", {3: 4096, 4: 8192, 5: 4096}[CONFIG]))
if CONFIG == 5:
  e = gpu.JN2018Ensemble(configs.config5(N=N))
  e.run(361)
  e._b_basin, e._b_north = e.cols.b.ptr, e.cols.b.ptr + e._off
else:
  e = gpu.TwoColEnsemble(configs.config3(N=N) if CONFIG == 3 else configs.config4(N=N))
  e.run(241)
gpu.synchronize()
out = (C.c_ulonglong * 16)()
_lib.lib.pm_debug_prof(out)
K = 20
for _ in range(K):
  e.tw.update(e._b_basin, e._b_north, store_psib=False)
gpu.synchronize()
_lib.lib.pm_debug_prof(out)
v = np.array(list(out), dtype=np.float64)
names = ["loads", "dG", "scan 1", "dI", "scan 2", "Psi + store", "min/max, linspace",
         "cell staging", "group ranges", "class passes", "Psibz + store"]
tot = v[:11].sum()
for n, x in zip(names, v[:11]):
  print("%-20s %8.0f cycles per member-update  %5.1f %%" % (n, x / v[15], 100 * x / tot))
print("8-cell groups per member-update after the first of a block: general %.1f, all-ones %.1f, all-zeros %.1f"
      % (v[11] / v[15], v[12] / v[15], v[13] / v[15]))

# timeline of the last launch: wave start / end on the 100 MHz clock
if hasattr(_lib.lib, "pm_debug_wave_times"):
  wt = (C.c_ulonglong * (2 * N))()
  _lib.lib.pm_debug_wave_times(wt, N)
  w = np.array(list(wt), dtype=np.float64).reshape(N, 2) * 1e-2  # us
  t0 = w[:, 0].min()
  st, en = w[:, 0] - t0, w[:, 1] - t0
  print("waves: %d; start min/median/max %.1f / %.1f / %.1f us; end median/max %.1f / %.1f us; lifetime median %.1f us"
        % (N, st.min(), np.median(st), st.max(), np.median(en), en.max(), np.median(en - st)))
  hist, edges = np.histogram(st, bins=10)
  print("start-time histogram:", list(zip(np.round(edges[:-1], 1), hist)))
