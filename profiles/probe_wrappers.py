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

# ---- the reference's two-column user loop (examples/example_twocol.py:85-96) with ONLY the
# import changed, lazy stepping (default) vs a launch per call; the reference itself runs this
# loop at 2 880 coupled steps/s at nz = 100 (BASELINE.md, 1 core Xeon 2.1 GHz, measured with
# the reference imported in the build container)
from pymoc_amd.modules import column as colmod


def twocol_loop(nsteps, nz=100):
  mm = configs.twocol_member(nz=nz, kappa_4k=2.5e-4)
  zz = mm["z"]
  A = pymoc_amd.Psi_Thermwind(z=zz, b1=mm["b_basin0"].copy(), b2=mm["b_north0"].copy())
  A.solve()
  [pib, pin] = A.Psibz()
  ba = pymoc_amd.Column(z=zz, kappa=mm["kappa"].copy(), Area=mm["A_basin"], b=mm["b_basin0"].copy(),
                        bs=mm["bs"], bbot=mm["bbot"])
  no = pymoc_amd.Column(z=zz, kappa=mm["kappa"].copy(), Area=mm["A_north"], b=mm["b_north0"].copy(),
                        bs=mm["bs_north"], bbot=mm["bbot"])
  t0 = time.perf_counter()
  for ii in range(nsteps):
    wAb = pib * 1e6
    wAN = -pin * 1e6
    ba.timestep(wA=wAb, dt=mm["dt"])
    no.timestep(wA=wAN, dt=mm["dt"], do_conv=True)
    if ii % mm["MOC_up_iters"] == 0:
      A.update(b1=ba.b, b2=no.b)
      A.solve()
      [pib, pin] = A.Psibz()
  el = time.perf_counter() - t0
  return nsteps / el, ba.b.copy()


for lazy in (True, False):
  colmod.LAZY = lazy
  twocol_loop(48)
  rate, b = twocol_loop(4800)
  print("example_twocol loop, nz=100, %s: %.0f coupled steps/s (reference 2880/s -> x%.1f)" % (
      "lazy (default)" if lazy else "a launch per call", rate, rate / 2880.))
colmod.LAZY = True
