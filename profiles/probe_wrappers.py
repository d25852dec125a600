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
j = configs.jn2018_member(nz=81)
ch = pymoc_amd.SO_ML(y=j["y"], h=j["h"], L=j["L"], Ks=j["Ks"], surflux=j["surflux"], rest_mask=j["rest_mask"],
                     b_rest=j["b_rest"], v_pist=j["v_pist"], bs=j["bs_SO0"].copy())
SO = pymoc_amd.Psi_SO(z=j["z"], y=j["y"], b=j["b_basin0"].copy(), bs=j["bs_SO0"].copy(), tau=0.12, L=j["L"], KGM=800.)
SO.solve()
for name, fn in (("SO_ML.timestep", lambda: ch.timestep(b_basin=j["b_basin0"], Psi_b=SO.Psi, dt=j["dt"])),
                 ("Psi_SO.solve", lambda: SO.solve()),
                 ("Column.timestep", lambda: basin.timestep(wA=wA, dt=m["dt"])),
                 ("Psi_Thermwind.solve", lambda: AMOC.solve()),
                 ("Psi_Thermwind.Psibz", lambda: AMOC.Psibz())):
  fn(); fn()
  t0 = time.perf_counter()
  for _ in range(200): fn()
  print("%-22s %.1f us per call" % (name, (time.perf_counter() - t0) / 200 * 1e6))
