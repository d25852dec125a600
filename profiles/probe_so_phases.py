"""Phase clocks of the adaptive GM solve (k_psi_so, so_gm_adaptive_reg).  Needs the profiling
build:  make -B lib EXTRA=-DPM_PHASE_PROFILE ; afterwards  make -B lib  restores the product."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pymoc_amd as gpu
from pymoc_amd import configs, _lib

N = int(os.environ.get("N", 8192))
e = gpu.TwoColEnsemble(dict(configs.config4(N=N), bvp_refine=-1))
e.run(241)
gpu.synchronize()
out = (C.c_ulonglong * 16)()
_lib.lib.pm_debug_prof(out)
K = 20
for _ in range(K):
  e.so.update(e._b_basin, e.bs_SO)
gpu.synchronize()
_lib.lib.pm_debug_prof(out)
v = np.array(list(out), dtype=np.float64)
names = ["pass head", "elements+chunk", "prefix scan", "affine scan", "thomas+store", "residual",
         "mesh", None, "before the BVP", "BVP tail", "epilogue",
         # (only in builds that place PM_TICK(11..13) in so_member: staging, ys, Ekman; slot 8 is then calc_GM's head)
         "staging+argmin", "outcrop latitude", "Ekman"]
tot = sum(c for n, c in zip(names, v) if n)
print("passes per member-update: %.2f" % (v[7] / v[15] + 1))
for n, c in zip(names, v):
  if n:
    print("%-16s %8.0f cycles per member-update  %5.1f %%" % (n, c / v[15], 100 * c / tot))
