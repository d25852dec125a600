"""Config 5 with the columns of the fused loop in the contracted (tolerance) mode: speed, and
agreement with the reference's sweep members (golden G17) next to the exact mode's."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pymoc_amd as gpu
from pymoc_amd import configs
from conftest import load_golden
g = load_golden("sweep_full")
N = 4096
res = {}
for arith in ("exact", "contracted"):
  c = configs.config5(N=N)
  c["rest_mask"] = np.repeat(c["rest_mask"][None], N, axis=0)
  e = gpu.JN2018Ensemble(c, arith=arith)
  e.run(72); gpu.synchronize()
  t0 = time.perf_counter(); e.run(3600 - 72); gpu.synchronize(); el = time.perf_counter() - t0
  st = e.state(); res[arith] = st
  print(arith, "%.4g coupled steps/s" % (N * (3600 - 72) / el), "lost:", e.nonfinite_members().tolist()[:8])
for k in ("b_basin", "b_north", "bs_SO", "Psi"):
  a, b = res["contracted"][k], res["exact"][k]
  ok = np.isfinite(a).all(axis=1) & np.isfinite(b).all(axis=1)
  d = np.max(np.abs(a[ok] - b[ok]), axis=1) / np.max(np.abs(b[ok]))
  print(k, "contracted vs exact at step 3600: median %.2e, 90%% %.2e, max %.2e (members > 1e-8: %d of %d)" % (
      np.median(d), np.percentile(d, 90), d.max(), (d > 1e-8).sum(), ok.sum()))
