#!/usr/bin/env python3
"""The two-column user loop of the reference (examples/example_twocol.py:85-96) with the
drop-in classes: the only change a PyMOC script needs is the import line.

    python examples/twocol_user_loop.py [--years 100]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymoc_amd.modules import Psi_Thermwind, Column  # was: from pymoc.modules import ...


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--years", type=float, default=100.)
  args = ap.parse_args()
  bs, bs_north, bbot = 0.03, 0.005, -0.001
  A_basin, A_north = 8e13, 8e13 / 100.
  # the reference script's dt = 30 d is beyond the explicit scheme's limit on this grid: the
  # reference itself goes to 174 Sv after 100 steps and NaN after 150 (and so does this
  # engine, bit for bit until then); 15 d is stable
  dt = 86400. * 15.
  MOC_up_iters = int(np.floor(2. * 360 * 86400 / dt))
  total_iters = int(np.ceil(args.years * 360 * 86400 / dt))
  kappa = lambda z: 1e-5 + 3e-5 * np.exp(z / 100) + 2.5e-4 * np.exp(-z / 1000 - 4)
  z = np.asarray(np.linspace(-4000, 0, 80))
  b_basin = bs * np.exp(z / 300.) + z / z[0] * bbot
  b_north = np.full_like(z, bbot) + (bs_north - bbot) * np.exp(z / 100.)
  AMOC = Psi_Thermwind(z=z, b1=b_basin, b2=b_north)
  AMOC.solve()
  basin = Column(z=z, kappa=kappa, Area=A_basin, b=b_basin, bs=bs, bbot=bbot)
  north = Column(z=z, kappa=kappa, Area=A_north, b=b_north, bs=bs_north, bbot=bbot)
  for ii in range(total_iters):
    wAb = AMOC.Psi * 1e6
    wAN = -AMOC.Psi * 1e6
    basin.timestep(wA=wAb, dt=dt)
    north.timestep(wA=wAN, dt=dt, do_conv=True)
    if ii % MOC_up_iters == 0:
      AMOC.update(b1=basin.b, b2=north.b)
      AMOC.solve()
  print("after %.0f years: max overturning %.3f Sv at z = %.0f m, basin b(-1000 m) = %.5f" % (
      args.years, AMOC.Psi.max(), z[AMOC.Psi.argmax()], np.interp(-1000., z, basin.b)))


if __name__ == "__main__":
  main()
