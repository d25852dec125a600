"""the two-basin topology (examples/twobasin_NadeauJansen.py) on the GPU."""
import numpy as np
import pytest

from oracle import drivers
from conftest import load_golden, relerr
from pymoc_amd import configs

pytestmark = pytest.mark.gpu

FIELDS = ("b_Atl", "b_north", "b_Pac", "Psi_AMOC", "Psi_ZOC", "Psi_SO_Atl", "Psi_SO_Pac")


def test_twobasin_trajectory_golden(gpu):
  g = load_golden("twobasin")
  m = configs.twobasin_member(nz=80)
  ens = gpu.TwoBasinEnsemble(m)
  snaps = (1, 24, 25, 26, 1200)
  orc = drivers.run_twobasin(m, 1200, set(snaps))
  done = 0
  for s in snaps:
    ens.run(s - done)
    done = s
    st = ens.state()
    for k in FIELDS:
      assert relerr(st[k][0], g["s%05d_%s" % (s, k)]) <= 1e-10, (s, k)
      assert relerr(st[k][0], orc[s][k]) <= 1e-10, (s, k)


def test_twobasin_sweep_members_vs_reference(gpu):
  g = load_golden("twobasin")
  c = configs.config_twobasin(N=2048)
  ens = gpu.TwoBasinEnsemble(c)
  n = int(g["sweep_nsteps"])
  ens.run(n)
  st = ens.state()
  idx = g["sweep_members"]
  for k in ("b_Atl", "b_north", "b_Pac", "Psi_AMOC", "Psi_ZOC", "Psi_SO_Atl"):
    assert relerr(st[k][idx], g["sweep_" + k]) <= 1e-10, k
  assert ens.nonfinite_members().size == 0
