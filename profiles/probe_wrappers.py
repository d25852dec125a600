"""Probe: cost of one call through the drop-in single-object wrappers (PCIe + launch bound)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pymoc_amd
from pymoc_amd import configs
m = configs.twocol_member(nz=80)
z = m["z"]
basin = pymoc_amd.Column(z=z, kappa=m["kappa"].copy(), Area=m["A_basin"], b=m["b_basin0"].copy(), bs=m["bs"], bbot=m["bbot"])
AMOC = pymoc_amd.Psi_Thermwind(z=z, b1=m["b_basin0"].copy(), b2=m["b_north0"].copy())
wA = 1e6 * np.sin(z / 1000.)
for name, fn in (("Column.timestep", lambda: basin.timestep(wA=wA, dt=m["dt"])),
                 ("Psi_Thermwind.solve", lambda: AMOC.solve()),
                 ("Psi_Thermwind.Psibz", lambda: AMOC.Psibz())):
  fn(); fn()
  t0 = time.perf_counter()
  for _ in range(200): fn()
  print("%-22s %.1f us per call" % (name, (time.perf_counter() - t0) / 200 * 1e6))
