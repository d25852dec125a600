"""Does the contracted column arithmetic keep the COUPLED configs within their tolerances?
configs 3 and 4 (TwoColEnsemble(arith="contracted")) against the reference's sweep members
(goldens G8 / G17) and against the exact mode."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pymoc_amd as gpu
from pymoc_amd import configs
from conftest import load_golden, relerr
import time
for name, cfgf, keys in (("config 3", lambda: configs.config3(N=4096), ("b_basin", "b_north", "Psi")),
                         ("config 4", lambda: configs.config4(N=8192), ("b_basin", "b_north", "Psi", "Psi_SO"))):
  g = load_golden("sweep_full")
  pre = "c3_" if name == "config 3" else "c4_"
  n = int(g[pre + "nsteps"])
  res = {}
  for arith in ("exact", "contracted"):
    e = gpu.TwoColEnsemble(cfgf(), arith=arith)
    e.run(241); gpu.synchronize()
    t0 = time.perf_counter(); e.run(n - 241); gpu.synchronize(); el = time.perf_counter() - t0
    st = e.state()
    idx = g[pre + "members"]
    errs = {k: relerr(st[k][idx], g[pre + k]) for k in keys if pre + k in g}
    res[arith] = st
    print(name, arith, "vs reference after %d steps:" % n, {k: "%.2e" % v for k, v in errs.items()},
          "| %.3g coupled steps/s" % (e.n * (n - 241) / el), "| lost:", e.nonfinite_members().size)
  print(name, "contracted vs exact, whole ensemble:",
        {k: "%.2e" % relerr(res["contracted"][k], res["exact"][k]) for k in keys})
