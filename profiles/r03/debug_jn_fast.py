"""A/B of the two fused JN2018 kernels (k_jn2018_fast vs k_jn2018_steps), launch by launch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pymoc_amd as gpu
from pymoc_amd import configs
N = int(os.environ.get("N", "256"))
c = configs.config5(N=N)
c["rest_mask"] = np.repeat(c["rest_mask"][None], N, axis=0)
a = gpu.JN2018Ensemble(c, fused=True)
b = gpu.JN2018Ensemble(c, fused=True)
done = 0
for n in [1, 35] + [12] * 30:
  os.environ.pop("PYMOC_JN_GENERAL", None)
  a.run(n)
  os.environ["PYMOC_JN_GENERAL"] = "1"
  b.run(n)
  done += n
  sa, sb = a.state(), b.state()
  bad = False
  for k in sa:
    if not np.array_equal(sa[k], sb[k], equal_nan=True):
      d = np.argwhere(~((sa[k] == sb[k]) | (np.isnan(sa[k]) & np.isnan(sb[k]))))
      print("step", done, k, "differs at", len(d), "entries; first:", d[:6].tolist(),
            [(float(sa[k][tuple(i)]), float(sb[k][tuple(i)])) for i in d[:3]])
      bad = True
  st_a, st_b = a.ml.status.download(), b.ml.status.download()
  if not np.array_equal(st_a, st_b):
    print("step", done, "status differs", np.argwhere(st_a != st_b)[:5].tolist())
  if bad:
    break
else:
  print("all equal through step", done)
