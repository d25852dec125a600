#!/usr/bin/env python3
"""One-off fuzz of the thermal wind's class sums (not collected by pytest): thousands of random
members -- smooth stratifications, flat runs (convecting columns), inverted and flat bottom cells,
quantised buoyancies (classes landing ON cell bounds), random sign structure of the overturning --
through pm_thermwind_update with all classes stored and with the listed classes only, every member
bitwise against the oracle.   usage (GPU box):  python tests/fuzz_thermwind.py [seed] [members]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402  (test infrastructure: the checker)
import pymoc_amd as gpu  # noqa: E402
from pymoc_amd import _lib  # noqa: E402
from pymoc_amd.device import DeviceArray  # noqa: E402


def members(rng, n, nz):
  z = np.linspace(-4000., 0., nz)
  e = np.exp(z / rng.uniform(200, 600, (n, 1)))
  b1 = 0.03 * e * rng.uniform(0.5, 1.5, (n, 1))
  b2 = b1 * rng.uniform(0.05, 1.0, (n, 1))
  kind = rng.integers(0, 8, n)
  for m in range(n):
    k = kind[m]
    if k == 1:  # flat run on top of the northern column
      t = rng.integers(2, nz // 2)
      b2[m, -t:] = b2[m, -t]
    elif k == 2:  # flat / inverted bottom cells
      b1[m, 0] = b1[m, 1] + rng.choice([0., 1e-9, -1e-12])
      b2[m, 0] = b2[m, 1] + rng.choice([0., 1e-9])
      if rng.random() < 0.5:
        b2[m, 1] = b2[m, 2] + rng.choice([0., 1e-10])
    elif k == 3:  # quantised: many classes land exactly on cell bounds, many flat cells
      q = 0.03 / (499 * rng.integers(1, 4))
      b1[m] = np.round(b1[m] / q) * q
      b2[m] = np.round(b2[m] / q) * q
    elif k == 4:  # noise: some levels out of order
      b1[m] += 2e-5 * rng.standard_normal(nz) * (rng.random(nz) < 0.05)
    elif k == 5:  # spread evenly: many classes asked for
      b1[m] = np.sort(rng.uniform(0, 0.03, nz))
      b2[m] = np.sort(rng.uniform(0, 0.03, nz))
    elif k == 6:  # a flat run in the interior of both
      i = rng.integers(5, nz - 10)
      b1[m, i:i + 4] = b1[m, i]
      b2[m, i:i + 4] = b2[m, i]
  return z, b1, b2


def main():
  seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
  n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
  rng = np.random.default_rng(seed)
  bad = 0
  for nz, nb in ((100, 500), (200, 500), (81, 500), (64, 300)):
    z, b1, b2 = members(rng, n, nz)
    f = rng.uniform(0.8e-4, 1.4e-4, n)
    d1, d2 = DeviceArray.from_host(b1), DeviceArray.from_host(b2)
    for user_psi in (False, True):
      full, lazy = gpu.ThermwindBatch(z, n, f=f, nb=nb), gpu.ThermwindBatch(z, n, f=f, nb=nb)
      ops = _lib.PM_TW_PSIB | _lib.PM_TW_PSIBZ
      if user_psi:
        Psi = np.cumsum(rng.standard_normal((n, nz)), axis=1) * (rng.random((n, 1)) < 0.7)
        Psi[:, 0] = 0.
        full.Psi.upload(Psi)
        lazy.Psi.upload(Psi)
      else:
        ops |= _lib.PM_TW_SOLVE
      full.update(d1, d2, ops=ops)
      lazy.update(d1, d2, ops=ops, store_psib=False)
      P, g, pb = full.Psi.download(), full.bgrid.download(), full.psib.download()
      o1, o2 = full.psibz1.download(), full.psibz2.download()
      l1, l2 = lazy.psibz1.download(), lazy.psibz2.download()
      for m in range(n):
        rP = Psi[m] if user_psi else O.thermwind_solve(z, b1[m], b2[m], f[m])
        rg, rp, r1, r2 = O.thermwind_psibz(b1[m], b2[m], rP, nb)
        ok = (np.array_equal(pb[m], rp, equal_nan=True) and np.array_equal(o1[m], r1, equal_nan=True) and
              np.array_equal(o2[m], r2, equal_nan=True) and np.array_equal(l1[m], r1, equal_nan=True) and
              np.array_equal(l2[m], r2, equal_nan=True) and np.array_equal(g[m], rg, equal_nan=True))
        if not ok:
          bad += 1
          if bad <= 5:
            print("MISMATCH nz=%d nb=%d user_psi=%s member %d" % (nz, nb, user_psi, m))
      print("nz=%d nb=%d user_psi=%s: %d members checked, %d mismatches so far" % (nz, nb, user_psi, n, bad),
            flush=True)
  print("fuzz seed %d: %s" % (seed, "OK" if bad == 0 else "%d MISMATCHES" % bad))
  return 1 if bad else 0


if __name__ == "__main__":
  sys.exit(main())
