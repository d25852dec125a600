"""Coefficient-set switches (bottom-BC switch, run_JansenNadeau_2018.py:233-254) per member over
one MOC interval of config 5: which members flip, and how often."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pymoc_amd as gpu
from pymoc_amd import configs
N = 4096
c = configs.config5(N=N)
c["rest_mask"] = np.repeat(c["rest_mask"][None], N, axis=0)
e = gpu.JN2018Ensemble(c, fused=True)
e.run(361 + 35)
u = gpu.JN2018Ensemble(c, fused=False)
u.cols.set_b(e.cols.get_b()); u.ml.bs.upload(e.ml.bs.download()); u.cols.bbot.upload(e.cols.bbot.download())
u.cols.set_ksel(e.cols.ksel.download()); u.ii = e.ii
prev = u.cols.ksel.download().copy()
flips = np.zeros(2 * N, dtype=int)
for s in range(36):
  u.run(1)
  k = u.cols.ksel.download()
  flips += (k != prev)
  prev = k.copy()
fb, fn = flips[:N], flips[N:]
print("members with >= 1 switch in 36 steps: basin %d, north %d" % ((fb > 0).sum(), (fn > 0).sum()))
print("switches per step, whole ensemble: %.4f" % (flips.sum() / 36. / N))
order = np.argsort(-(fb + fn))[:16]
print("most switching members:", [(int(i), int(fb[i]), int(fn[i])) for i in order])
