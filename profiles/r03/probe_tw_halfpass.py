"""How many (8-cell group, class pass) tiles of Psib need the general path if the two halves of a
128-class pass (the two classes a lane holds) are classified separately?  NumPy on the state
of config-3/5 members after their spin-up."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pymoc_amd as gpu
from pymoc_amd import configs

CONFIG = int(os.environ.get("CONFIG", 3))
N = 256
if CONFIG == 5:
  c = configs.config5(N=N)
  c["rest_mask"] = np.repeat(c["rest_mask"][None], N, axis=0)
  e = gpu.JN2018Ensemble(c)
  e.run(361)
else:
  e = gpu.TwoColEnsemble(configs.config3(N=N))
  e.run(241)
st = e.state()
b1, b2, Psi = st["b_basin"], st["b_north"], st["Psi"]
nb = 500
full = both = one = 0
for m in range(N):
  if not (np.isfinite(b1[m]).all() and np.isfinite(b2[m]).all()):
    continue
  u = -(Psi[m, 1:] - Psi[m, :-1])
  north = u < 0
  top = np.where(north, b2[m, 1:], b1[m, 1:])
  bot = np.where(north, b2[m, :-1], b1[m, :-1])
  bg = np.linspace(min(b1[m].min(), b2[m].min()), max(b1[m].max(), b2[m].max()), nb)
  nc = top.size
  ng = (nc + 7) // 8
  for i0 in range(0, nb, 128):
    halves = []
    for h in range(2):
      lo, hi = i0 + 64 * h, min(i0 + 64 * h + 63, nb - 1)
      if lo >= nb:
        halves.append(np.zeros(ng, bool))
        continue
      gmin, gmax = bg[lo], bg[hi]
      gen = np.zeros(ng, bool)
      for g in range(ng):
        t, b_ = top[8 * g:8 * g + 8], bot[8 * g:8 * g + 8]
        ones = gmax <= b_.min()
        zero = gmin >= t.max()
        gen[g] = not (ones or zero) or (t - b_ <= 0).any()
      halves.append(gen)
    gmin, gmax = bg[i0], bg[min(i0 + 127, nb - 1)]
    for g in range(ng):
      t, b_ = top[8 * g:8 * g + 8], bot[8 * g:8 * g + 8]
      if not (gmax <= b_.min() or gmin >= t.max()) or (t - b_ <= 0).any():
        full += 1
    both += int((halves[0] & halves[1]).sum())
    one += int((halves[0] ^ halves[1]).sum())
print("config %d: per member-update general tiles now %.1f; classified per half: %.1f general in both halves, "
      "%.1f in one half" % (CONFIG, full / N, both / N, one / N))
