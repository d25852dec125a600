#!/usr/bin/env python3
"""Structure of the Psib sum's cells along BASELINE's coupled runs (CPU, oracle as stepper).

For every thermal-wind update of a few members of configs 3 / 4 / 5 / 6: are the upstream cells
(bot_k, top_k) of psi_thermwind.py:175-181 SORTED (bot and top both non-decreasing in k, all
thicknesses > 0), and how many cells does an isopycnal class cut (bot < g < top)?  This decides
whether a per-class evaluation (search the cut cell, one division, table look-ups for the cells
whose mask is exactly one) can replace the (8 cells x 128 classes) tiles of k_thermwind.
usage: python profiles/r05/probe_psib_structure.py [members per config]"""
import os
import sys
from collections import Counter

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import drivers as D  # noqa: E402
from pymoc_amd import configs  # noqa: E402

STATS = {}


def hook(tag):
  real = D.thermwind_psibz

  def wrapped(b1, b2, Psi, nb):
    st = STATS.setdefault(tag, dict(calls=0, unsorted=0, nonpos=0, cuts=Counter(), maxcut=0,
                                    viol=Counter(), nan=0))
    st['calls'] += 1
    b1 = np.asarray(b1, float)
    b2 = np.asarray(b2, float)
    if not (np.isfinite(b1).all() and np.isfinite(b2).all() and np.isfinite(Psi).all()):
      st['nan'] += 1
      return real(b1, b2, Psi, nb)
    u = -(Psi[1:] - Psi[:-1])
    north = u < 0
    bot = np.where(north, b2[:-1], b1[:-1])
    top = np.where(north, b2[1:], b1[1:])
    d = top - bot
    st['nonpos'] += int((d <= 0).any())
    nv = int((np.diff(bot) < 0).sum() + (np.diff(top) < 0).sum())
    st['viol'][min(nv, 9)] += 1
    st['unsorted'] += int(nv > 0 or (d <= 0).any())
    g = np.linspace(min(b1.min(), b2.min()), max(b1.max(), b2.max()), nb)
    c = ((bot[None, :] < g[:, None]) & (g[:, None] < top[None, :])).sum(axis=1)
    for v, n in Counter(np.minimum(c, 9).tolist()).items():
      st['cuts'][v] += n
    st['maxcut'] = max(st['maxcut'], int(c.max()))
    return real(b1, b2, Psi, nb)
  D.thermwind_psibz = wrapped
  return real


def main():
  nm = int(sys.argv[1]) if len(sys.argv) > 1 else 4
  spec = {3: (configs.config3, 4096, 2400), 4: (configs.config4, 8192, 2400),
          5: (configs.config5, 4096, 3600), 6: (configs.config_twobasin, 2048, 2400)}
  for c, (mk, n, steps) in spec.items():
    pick = np.linspace(0, n - 1, nm).astype(int)
    for j in pick:
      cfg = mk(N=n, members=(int(j), int(j) + 1))
      m = configs.member(cfg, 0, c)
      real = hook(c)
      try:
        if c == 3:
          D.run_twocol(m, steps, set())
        elif c == 4:
          D.run_twocol(m, steps, set(), so=True)
        elif c == 5:
          D.run_jn2018(m, steps, set())
        else:
          D.run_twobasin(m, steps, set())
      except Exception as e:  # a member the reference loses
        print("config", c, "member", j, "raised", type(e).__name__)
      finally:
        D.thermwind_psibz = real
    st = STATS[c]
    tot = sum(st['cuts'].values())
    print(f"config {c}: {st['calls']} updates of {nm} members; non-finite {st['nan']}; "
          f"unsorted {st['unsorted']} ({100. * st['unsorted'] / max(st['calls'], 1):.1f} %), "
          f"with a cell of thickness <= 0: {st['nonpos']}")
    print("   order violations per update:", dict(sorted(st['viol'].items())))
    print("   cells cut per class:", {k: f"{100. * v / tot:.2f} %" for k, v in sorted(st['cuts'].items())},
          "max", st['maxcut'])


if __name__ == "__main__":
  main()
