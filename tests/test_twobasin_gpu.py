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


def test_twobasin_update_pairs_side_by_side_equal_four_launches(gpu):
  """Round 5: an update is two one-launch pairs {Psi_SO.solve, thermal wind} on two streams; the
  result equals four separate launches in the script's order bit for bit, and so do the gathers
  of the seven sampled fields at the driver's cadence."""
  c = configs.config_twobasin(N=96)
  s = gpu.Stream()
  # default: the pairs one after the other, the forcing formed by the column kernel
  a = gpu.TwoBasinEnsemble(c, stream=s, keep_history=True, diag_iters=48)
  assert a._forcing_in_k1 and not a._overlap
  # the pairs on two streams, whole intervals replayed from a hipGraph
  a2 = gpu.TwoBasinEnsemble(c, stream=s, keep_history=True, diag_iters=48, overlap_updates=True)
  # ... eagerly, and the forcing by pm_twobasin_forcing (round 5's first form)
  b = gpu.TwoBasinEnsemble(c, stream=s, keep_history=True, diag_iters=48, overlap_updates=True,
                           use_graph=False)
  b._forcing_in_k1 = False
  d = gpu.TwoBasinEnsemble(c, keep_history=True, diag_iters=48, overlap_updates=False)
  d._pairs = False  # four separate launches (what round 4 ran)
  d._forcing_in_k1 = False
  for e in (a, a2, b, d):
    e.run(130)
    e.gather_diagnostics()
  assert a2._graph is not None and a._graph is None and b._graph is None
  sa = a.state()
  for e in (a2, b, d):
    st = e.state()
    for k in FIELDS:
      assert np.array_equal(sa[k], st[k]), k
    assert [x for x, _ in e.diag.history] == [x for x, _ in a.diag.history] == [0, 48, 96, 130]
    for (_, ha), (_, hb) in zip(a.diag.history, e.diag.history):
      for k in FIELDS:
        assert np.array_equal(ha[k], hb[k]), k
  for k in FIELDS:
    assert np.array_equal(a.diag.history[-1][1][k], sa[k]), k


def test_twobasin_contracted_columns_within_tolerance(gpu):
  c = configs.config_twobasin(N=64)
  a, b = gpu.TwoBasinEnsemble(c), gpu.TwoBasinEnsemble(c, arith="contracted")
  a.run(600)
  b.run(600)
  sa, sb = a.state(), b.state()
  for k in FIELDS:
    assert relerr(sb[k], sa[k]) <= 1e-11, k
