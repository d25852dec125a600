"""Per-wave timeline (start / end on the 100 MHz clock) of the coupled kernels: shows whether
a launch ran as one batch of resident waves and how long its slowest member took.
Needs the profiling build:  make -B lib EXTRA=-DPM_PHASE_PROFILE ; then  make -B lib."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pymoc_amd as gpu
from pymoc_amd import configs, _lib


def timeline(label, N):
  wt = (C.c_ulonglong * (2 * N))()
  _lib.lib.pm_debug_wave_times(wt, N)
  w = np.array(list(wt), dtype=np.float64).reshape(N, 2) * 1e-2  # us
  st, en = w[:, 0] - w[:, 0].min(), w[:, 1] - w[:, 0].min()
  life = en - st
  late = int((st > 2.0).sum())
  print("%-16s %5d waves, %4d start later than 2 us; lifetime median %.1f, p99 %.1f, max %.1f us (member %d); "
        "last end %.1f us" % (label, N, late, np.median(life), np.percentile(life, 99), life.max(),
                              int(life.argmax()), en.max()))
  if os.environ.get("DUMP"):
    os.makedirs("gpurun_out", exist_ok=True)
    np.save(os.path.join("gpurun_out", "wave_times_%s.npy" % label.split()[0]), w)
  if os.environ.get("HIST"):
    order = np.argsort(-life)[:12]
    print("   slowest members:", [(int(i), round(float(life[i]), 1)) for i in order])
    print("   start-time histogram (us):", np.histogram(st, bins=8)[0].tolist(), "edges",
          np.round(np.histogram(st, bins=8)[1], 1).tolist())
    print("   end-time histogram (us):  ", np.histogram(en, bins=8)[0].tolist(), "edges",
          np.round(np.histogram(en, bins=8)[1], 1).tolist())


CONFIG = int(os.environ.get("CONFIG", 5))
if CONFIG == 5:
  N = 4096
  e = gpu.JN2018Ensemble(configs.config5(N=N))
  e.run(361)
  gpu.synchronize()
  b_basin, b_north = e.cols.b.ptr, e.cols.b.ptr + e._off
  e.so.update(b_basin, e.ml.bs); gpu.synchronize(); timeline("k_psi_so", N)
  e.tw.update(b_basin, b_north, store_psib=False); gpu.synchronize(); timeline("k_thermwind", N)
  e.run(35); gpu.synchronize()   # to the MOC boundary: the next launch fuses a whole interval
  e.run(36); gpu.synchronize(); timeline("k_jn2018_steps x36", N)
else:
  N = 8192 if CONFIG == 4 else 4096
  e = gpu.TwoColEnsemble(configs.config4(N=N) if CONFIG == 4 else configs.config3(N=N))
  e.run(241)
  gpu.synchronize()
  if e.so is not None:
    e.so.update(e._b_basin, e.bs_SO); gpu.synchronize(); timeline("k_psi_so", N)
  e.tw.update(e._b_basin, e._b_north, store_psib=False); gpu.synchronize(); timeline("k_thermwind", N)
