"""The persistent per-member run kernels (pm_twocol_run, pm_jn2018_run: many [refresh the
overturning diagnostics, MOC_up_iters steps] intervals of a coupled driver in ONE launch) against
the launch sequence of the same drivers -- bitwise -- and the IEEE leg of the fused Jansen & Nadeau
step loop against the oracle.

Loops restated: examples/example_twocol.py:85-96, examples/run_JansenNadeau_2018.py:201-261."""
import numpy as np
import pytest

import oracle as O
from pymoc_amd import configs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
  import pymoc_amd
  pymoc_amd._lib.require_device()
  return pymoc_amd


def _same(sa, sb, rows=None, what=""):
  for k in sb:
    a, b = (sa[k], sb[k]) if rows is None else (sa[k][rows], sb[k][rows])
    assert np.array_equal(a, b, equal_nan=True), (what, k, float(np.nanmax(np.abs(a - b))))


@pytest.mark.parametrize("nz,N", [(100, 192), (65, 50), (81, 33)])
def test_twocol_fused_run_equals_launch_sequence_bitwise(gpu, nz, N):
  """example_twocol.py's loop: one launch per run() against one launch per phase; every split of
  the run (a lone first step, whole intervals, a tail that stops between two updates, a run that
  ends exactly on an update) and a ragged last block of 16 members."""
  c = configs.config3(N=N, nz=nz)
  a = gpu.TwoColEnsemble(c, fused_run=True)
  b = gpu.TwoColEnsemble(c, fused_run=False)
  assert a._fused_run and not b._fused_run
  _same(a.state(), b.state(), what="initial update")
  for n in (1, 24, 23, 1, 100, 7, 48, 240):
    a.run(n)
    b.run(n)
    assert a.ii == b.ii
    _same(a.state(), b.state(), what="after %d more steps (ii=%d)" % (n, a.ii))
    assert np.array_equal(a.nonfinite_members(), b.nonfinite_members())
  assert np.all(a.run_status.download() == 0)
  assert np.isfinite(a.state()["b_basin"]).all()


def test_twocol_fused_run_gathers_where_the_launch_sequence_does(gpu):
  """Diag_iters gathers split the persistent launch: same gathers, same steps, same content."""
  c = configs.config3(N=64)
  a = gpu.TwoColEnsemble(c, fused_run=True, keep_history=True, diag_iters=48)
  b = gpu.TwoColEnsemble(c, fused_run=False, keep_history=True, diag_iters=48)
  for n in (200, 137):
    a.run(n)
    b.run(n)
  assert a.diag.ngathers == b.diag.ngathers and a.diag.ngathers >= 6
  ha, hb = a.diag.history, b.diag.history
  assert [s for s, _ in ha] == [s for s, _ in hb]
  for (_, fa), (_, fb) in zip(ha, hb):
    _same(fa, fb, what="gather")
  _same(a.state(), b.state())


def test_twocol_fused_run_full_config3_vs_launch_sequence(gpu):
  """BASELINE config 3 at full size and length: 4096 members x 2400 steps, bitwise."""
  c = configs.config3()
  a = gpu.TwoColEnsemble(c, fused_run=True)
  b = gpu.TwoColEnsemble(c, fused_run=False)
  a.run(2400)
  b.run(2400)
  _same(a.state(), b.state())
  assert a.nonfinite_members().size == 0


def test_twocol_fused_run_ieee_leg(gpu):
  """A member whose column operands lie outside the exact-division window steps in the IEEE form
  inside the persistent kernel, like pm_column_steps does in the launch sequence: bitwise equal,
  and flagged (status bit 5)."""
  c = configs.config3(N=32)
  a = gpu.TwoColEnsemble(c, fused_run=True)
  b = gpu.TwoColEnsemble(c, fused_run=False)
  for e in (a, b):
    e.run(25)
    s = e.cols.get_b()
    s[5] *= 2.0**-1000        # basin column of member 5
    s[32 + 9, 40] = 2.0**300  # one level of the northern column of member 9
    e.cols.set_b(s)
    e.run(24)
  _same(a.state(), b.state())
  st = a.run_status.download()
  assert st[5] & 32 and st[9] & 32
  assert np.all(st[[i for i in range(32) if i not in (5, 9)]] & 32 == 0)


@pytest.mark.parametrize("nz,ny,dt_days", [(81, 51, 30.), (100, 40, 30.), (200, 51, 10.)])
def test_jn2018_fused_run_equals_launch_sequence_bitwise(gpu, nz, ny, dt_days):
  """run_JansenNadeau_2018.py's loop: pm_jn2018_run against [pm_psi_so_update, pm_thermwind_update,
  pm_jn2018_steps] per interval, every split of the run; with the diagnostics recorder attached
  (which ends a launch at every Diag_iters)."""
  from pymoc_amd.diagnostics import JN2018Diagnostics
  N = 80
  c = configs.config5(N=N, nz=nz, ny=ny, dt_days=dt_days)
  c["rest_mask"] = np.repeat(c["rest_mask"][None], N, axis=0)
  a = gpu.JN2018Ensemble(c, fused_run=True)
  b = gpu.JN2018Ensemble(c, fused_run=False)
  assert a._fused_run and not b._fused_run
  M = a.M
  a.recorder = JN2018Diagnostics(a, 2 * M, 20 * M)
  b.recorder = JN2018Diagnostics(b, 2 * M, 20 * M)
  for n in (1, M - 1, M, 5, 3 * M + 7, 2 * M - 10, 4 * M):
    a.run(n)
    b.run(n)
    assert a.ii == b.ii
    sb = b.state()
    ok = np.isfinite(sb["b_basin"]).all(axis=1) & np.isfinite(sb["bs_SO"]).all(axis=1)
    assert ok.sum() >= N - 2
    _same(a.state(), sb, rows=ok, what="ii=%d" % a.ii)
    ok2 = np.concatenate([ok, ok])
    assert np.array_equal(a.cols.bbot.download()[ok2], b.cols.bbot.download()[ok2])
    assert np.array_equal(a.cols.ksel.download()[ok2], b.cols.ksel.download()[ok2])
  for k in ("AMOC", "AMOC_b", "bgrid", "b_basin", "b_north", "bs_SO", "Psi_SO"):
    assert np.array_equal(getattr(a.recorder, k)[ok], getattr(b.recorder, k)[ok], equal_nan=True), k


def test_jn2018_fused_steps_ieee_leg_bitwise_vs_oracle(gpu):
  """The fused step loop (pm_jn2018_steps) on members whose column operands lie outside the
  exact-division window -- a basin column scaled by 2^-1000, a northern level at 2^-300 -- takes
  its IEEE leg: an interval of steps bit-identical to the oracle's plain-division column steps
  (column.py:210-271) and mixed layer, driven with the launch's own forcing; the members are
  flagged (status bit 5) and the others are not.  (A level at 2^+300 is flagged as well; that
  member's forcing is NaN after the next thermal-wind update, and what NaNs do to np.interp's
  search in the mixed layer is not defined by the reference, so it is not compared.)"""
  N = 16
  c = configs.config5(N=N, nz=81, ny=51, dt_days=30.)
  c["rest_mask"] = np.repeat(c["rest_mask"][None], N, axis=0)
  for fused_run in (False, True):
    e = gpu.JN2018Ensemble(c, fused_run=fused_run)
    M = e.M
    e.run(M)
    b = e.cols.get_b()
    b[7] *= 2.0**-1000
    b[N + 9, 50] = 2.0**-300
    b[N + 11, 50] = 2.0**300
    e.cols.set_b(b)
    bs0 = e.ml.bs.download()
    bbot0, ksel0 = e.cols.bbot.download(), e.cols.ksel.download()
    e.run(M)
    st = e.ml.status.download()
    assert st[7] & 32 and st[9] & 32 and st[11] & 32
    s = e.state()
    z, y = c["z"], c["y"]
    for m in (7, 9, 3):
      mm = configs.member(c, m, 5)
      PsiSO, pib, pin = s["Psi_SO"][m], s["Psi_iso_b"][m], s["Psi_iso_n"][m]
      bb, bn, bsSO = b[m].copy(), b[N + m].copy(), bs0[m].copy()
      Ab, An = mm["A_basin"] + 0 * z, mm["A_north"] + 0 * z
      kap, kapeff = mm["kappa"], mm["kappaeff"]
      bbot_b, bbot_n = bbot0[m], bbot0[N + m]
      kap_b = kapeff if ksel0[m] else kap
      wAb, wAN = (pib - PsiSO) * 1e6, -pin * 1e6
      for _ in range(M):  # run_JansenNadeau_2018.py:229-261 (oracle/drivers.py:run_jn2018)
        if PsiSO[1] < 0:
          bbot_b, kap_b = bsSO[0], kapeff
        if pib[1] > 0 and bn[0] < bb[1] and bn[0] < bsSO[0]:
          bbot_b, kap_b = bn[0], kapeff
        elif PsiSO[1] >= 0:
          bbot_b, kap_b = bb[1], kap
        if pin[1] < 0 and bb[0] < bn[1]:
          bbot_n, kap_n = bb[0], kapeff
        else:
          bbot_n, kap_n = bn[1], kap
        bb = O.column_timestep(z, kap_b, Ab, bb, wAb, mm["dt"], do_conv=True, bs=mm["bs"],
                               bbot=bbot_b)
        bn = O.column_timestep(z, kap_n, An, bn, wAN, mm["dt"], do_conv=True,
                               bs=mm["bs_north"], bbot=bbot_n)
        bsSO, _ = O.so_ml_advdiff(y, mm["surflux"], mm["rest_mask"], mm["b_rest"], bsSO, bb,
                                  PsiSO, mm["dt"], Ks=mm["Ks"], h=mm["h"], L=mm["L"],
                                  v_pist=mm["v_pist"])
      assert np.array_equal(s["b_basin"][m], bb, equal_nan=True), (fused_run, m)
      assert np.array_equal(s["b_north"][m], bn, equal_nan=True), (fused_run, m)
      assert np.array_equal(s["bs_SO"][m], bsSO, equal_nan=True), (fused_run, m)
    assert st[3] & 32 == 0


@pytest.mark.parametrize("nz,ny,dt_days", [(200, 51, 10.), (81, 51, 30.), (150, 40, 10.), (46, 51, 30.)])
def test_one_update_launch_equals_two_launches_bitwise(gpu, nz, ny, dt_days):
  """pm_so_tw_update (PsiSO.solve + AMOC.solve / Psibz of a member by one wave, one launch) against
  pm_psi_so_update followed by pm_thermwind_update: every lane shape (P = 1 ... 4)."""
  N = 70
  c = configs.config5(N=N, nz=nz, ny=ny, dt_days=dt_days)
  c["rest_mask"] = np.repeat(c["rest_mask"][None], N, axis=0)
  a = gpu.JN2018Ensemble(dict(c, one_update_launch=True), fused_run=False)
  b = gpu.JN2018Ensemble(dict(c, one_update_launch=False), fused_run=False)
  assert a._one_update_launch and not b._one_update_launch
  for n in (1, 2 * a.M, a.M + 3):
    a.run(n)
    b.run(n)
    sb = b.state()
    ok = np.isfinite(sb["b_basin"]).all(axis=1) & np.isfinite(sb["bs_SO"]).all(axis=1)
    _same(a.state(), sb, rows=ok, what="nz=%d ii=%d" % (nz, a.ii))
    for k in ("Psi_Ek", "Psi_GM"):
      assert np.array_equal(getattr(a.so, k).download()[ok], getattr(b.so, k).download()[ok])
    assert np.array_equal(a.so.status.download()[ok], b.so.status.download()[ok])
