"""A/B of the persistent run kernels against the launch sequence (same members, same steps).
usage: python profiles/r04/probe_fused_run.py [3] [5s] [5]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pymoc_amd
from pymoc_amd import configs
from pymoc_amd.device import Stream

what = sys.argv[1:] or ["3"]
st = Stream()


def timed(make, nsteps, warm):
  e = make()
  e.run(warm)
  st.sync()
  t0 = time.perf_counter()
  e.run(nsteps)
  st.sync()
  return e, time.perf_counter() - t0


for w in what:
  if w == "3":
    c = configs.config3()
    mk = lambda f: (lambda: pymoc_amd.TwoColEnsemble(c, stream=st, fused_run=f))
    n, steps, warm = 4096, 2400, 241
  elif w == "5s":   # config-5 physics at the script's own nz = 81 (fits the run kernel's LDS)
    c = configs.config5(N=4096, nz=81, ny=51, dt_days=30.)
    c["rest_mask"] = np.repeat(c["rest_mask"][None], 4096, axis=0)
    mk = lambda f: (lambda: pymoc_amd.JN2018Ensemble(c, stream=st, fused_run=f))
    n, steps, warm = 4096, 3600, 360
  else:
    c = configs.config5()
    c["rest_mask"] = np.repeat(c["rest_mask"][None], 4096, axis=0)
    mk = lambda f: (lambda: pymoc_amd.JN2018Ensemble(c, stream=st, fused_run=f))
    n, steps, warm = 4096, 3600, 360
  res = {}
  for f in (False, True, False, True):
    try:
      e, t = timed(mk(f), steps, warm)
    except ValueError as ex:
      print(w, "fused_run", f, "not available:", ex)
      continue
    res.setdefault(f, []).append(n * steps / t)
    s = e.state()
    print("config %s fused_run=%s: %.4g coupled steps/s (%.1f us per interval of %d), checksum %.17g, "
          "nonfinite %d" % (w, f, n * steps / t, t / (steps / e.M) * 1e6, e.M,
                            float(np.nansum(s["b_basin"])), e.nonfinite_members().size), flush=True)
