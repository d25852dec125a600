"""GPU parity of K4 (pm_psi_so_update) and of the two-column + SO driver (config 4)."""
import numpy as np
import pytest

import oracle as O
from oracle import drivers
from conftest import load_golden, relerr
from pymoc_amd import configs

pytestmark = pytest.mark.gpu

TOL_DIRECT = 1e-13   # everything but the GM boundary-value problem
TOL_BVP_ORACLE = 1e-9   # same collocation scheme and mesh as the oracle, other elimination order
TOL_BVP_REF = 1e-5   # vs SciPy's adaptive solve_bvp (tol=1e-3) in the reference


def _kwargs(g, p):
  kw = {}
  for name in ("f", "rho", "L", "KGM", "smax"):
    kw[name] = float(g[p + "kw_" + name])
  for name in ("c", "Hsill", "HEk", "Htapertop", "Htaperbot"):
    v = float(g[p + "kw_" + name])
    kw[name] = None if np.isnan(v) else v
  kw["bvp_with_Ek"] = bool(g[p + "kw_bvp_with_Ek"])
  return kw


def test_psi_so_golden(gpu):
  from pymoc_amd.device import DeviceArray
  g = load_golden("psi_so")
  for k in range(int(g["ncases"])):
    p = "c%02d_" % k
    kw = _kwargs(g, p)
    tau = g[p + "tau"]
    tau_in = float(tau) if tau.ndim == 0 else tau[None, :]
    KGM = kw.pop("KGM")
    t = gpu.PsiSOBatch(g[p + "z"], g[p + "y"], 1, tau=tau_in, KGM=KGM, bvp_refine=8,
                       diagnostics=True, **kw)
    t.update(DeviceArray.from_host(g[p + "b"][None]), DeviceArray.from_host(g[p + "bs"][None]))
    Psi, Ek, GM = t.Psi.download()[0], t.Psi_Ek.download()[0], t.Psi_GM.download()[0]
    ys = t.ys.download()[0]
    assert relerr(ys, g[p + "ys"]) <= 1e-14, k          # direct inverse vs brentq
    assert relerr(Ek, g[p + "Psi_Ek"]) <= TOL_DIRECT, k
    tau_o = float(tau) if tau.ndim == 0 else tau
    oPsi, oEk, oGM, _ = O.psi_so_solve(g[p + "z"], g[p + "y"], g[p + "b"], g[p + "bs"], tau_o,
                                       KGM=KGM, bvp_refine=8, **kw)
    if kw["c"] is None:
      assert relerr(GM, g[p + "Psi_GM"]) <= TOL_DIRECT, k
      assert relerr(Psi, g[p + "Psi"]) <= TOL_DIRECT, k
    else:
      assert relerr(GM, oGM) <= TOL_BVP_ORACLE, k
      assert relerr(Psi, oPsi) <= TOL_BVP_ORACLE, k
      assert relerr(GM, g[p + "Psi_GM"]) <= TOL_BVP_REF, k
      assert relerr(Psi, g[p + "Psi"]) <= TOL_BVP_REF, k
    assert t.status.download()[0] & 6 == 0


@pytest.mark.parametrize("nz,ny,R", [(2, 2, 1), (3, 5, 4), (64, 40, 8), (65, 51, 8),
                                      (100, 40, 16), (200, 51, 3), (257, 130, 2), (512, 64, 8)])
def test_psi_so_ragged_sizes_vs_oracle(gpu, nz, ny, R):
  from pymoc_amd.device import DeviceArray
  rng = np.random.default_rng(nz + 31 * ny)
  n = 7
  z = np.sort(rng.uniform(-4000, 0, nz))
  z[-1] = 0.
  y = np.sort(rng.uniform(0, 2e6, ny))
  b = np.sort(0.03 * rng.random((n, nz)), axis=1) - 0.002
  bs = np.sort(0.03 * rng.random((n, ny)), axis=1)
  bs[2, : ny // 3] = bs[2, ny // 3::-1][: ny // 3] if ny > 6 else bs[2, : ny // 3]  # min inside
  tau = rng.uniform(0.05, 0.2, n)
  KGM = rng.uniform(500., 1500., n)
  for kw in (dict(), dict(c=0.1, bvp_with_Ek=True), dict(c=0.3, Hsill=500., HEk=100.,
                                                           Htapertop=200., Htaperbot=300.)):
    t = gpu.PsiSOBatch(z, y, n, tau=tau, KGM=KGM, f=1e-4, L=5e6, bvp_refine=R, **kw)
    t.update(DeviceArray.from_host(b), DeviceArray.from_host(bs))
    Psi, Ek, GM = t.Psi.download(), t.Psi_Ek.download(), t.Psi_GM.download()
    for m in range(n):
      oPsi, oEk, oGM, st = O.psi_so_solve(z, y, b[m], bs[m], float(tau[m]), KGM=KGM[m],
                                          f=1e-4, L=5e6, bvp_refine=R, **kw)
      tol = TOL_BVP_ORACLE if "c" in kw else TOL_DIRECT
      assert relerr(Ek[m], oEk) <= TOL_DIRECT, (nz, ny, kw, m)
      assert relerr(GM[m], oGM) <= tol, (nz, ny, kw, m)
      assert relerr(Psi[m], oPsi) <= tol, (nz, ny, kw, m)


def test_psi_so_tau_array_and_wrapper_api(gpu):
  g = load_golden("psi_so")
  p = "c01_"  # tau array, c None
  kw = _kwargs(g, p)
  S = gpu.Psi_SO(z=g[p + "z"], y=g[p + "y"], b=g[p + "b"].copy(), bs=g[p + "bs"].copy(),
                 tau=g[p + "tau"].copy(), **kw)
  S.solve()
  assert relerr(S.Psi, g[p + "Psi"]) <= TOL_DIRECT
  assert relerr(S.Psi_Ek, g[p + "Psi_Ek"]) <= TOL_DIRECT
  assert relerr(S.Psi_GM, g[p + "Psi_GM"]) <= TOL_DIRECT
  assert S.Psi[0] == 0.
  # calc_Ekman / calc_GM return m^3/s; solve() == their sum in Sv (reference test_solve)
  ek = S.calc_Ekman()
  S.Psi_Ek = ek / 1e6
  gm = S.calc_GM()
  psi = ek / 1e6 + gm / 1e6
  psi[0] = 0.
  assert np.array_equal(psi, S.Psi)
  # ys: inverse of bs, clamps (reference test_ys)
  for i in range(len(S.y)):
    assert np.round(S.ys(S.bs(S.y[i])), 3) == np.round(S.y[i], 3)
  assert S.ys(-1.0) == S.y[0] - 1e3 and S.ys(1.0) == S.y[-1]
  S.update(b=10.0, bs=50.0)
  assert np.all(S.b(S.z) == 10.0) and np.all(S.bs(S.y) == 50.0)
  for bad, msg in ((dict(z=-2000), "z needs to be numpy array providing grid levels"),
                   (dict(z=S.z, y=1e6),
                    "y needs to be numpy array providing horizontal grid (or boundaries) of ACC")):
    with pytest.raises(TypeError) as e:
      gpu.Psi_SO(**bad)
    assert str(e.value) == msg
  with pytest.raises(TypeError) as e:
    gpu.Psi_SO(z=S.z, y=S.y, b=S.b)
  assert str(e.value) == "('bs', 'needs to be either function, numpy array, or float')"


def test_reference_unit_tests_of_calc_ekman(gpu):
  """tests/modules/test_psi_SO.py:144-203 re-expressed against the wrapper: constant wind,
  a wind that varies with latitude (averaged north of each level's outcrop), the sill taper and
  the Ekman-layer taper; and calc_Ekman agrees with the class's own taper helpers."""
  S = gpu.Psi_SO(z=np.linspace(-4000, 0, 81), y=np.linspace(0, 2.0e6, 51),
                 b=np.linspace(0.03, -0.001, 81), bs=np.linspace(0.05, 0.10, 51), tau=0.12)
  full = (S.L * 0.12) / (S.f * S.rho)
  want = np.full(81, full)
  want[-1] = 0
  assert np.all(np.around(want, 3) == np.around(S.calc_Ekman(), 3))
  # wind varying with latitude: every 5th surface point is an outcrop of a level
  b, bs = np.linspace(0.03, 0.01, 21), np.linspace(0.01, 0.02, 51)
  y, z, tau = np.linspace(0, 2e6, 51), np.linspace(-4000, 0, 21), np.linspace(0.2, 0.12, 51)
  V = gpu.Psi_SO(y=y, z=z, b=b, bs=bs, tau=tau)
  tau_ave = np.zeros(21)
  tau_ave[0:11] = tau[-1]
  tau_ave[11:-1] = [np.mean(tau[-i * 5 - 1:]) for i in range(1, 10)]
  assert np.all(np.around(V.calc_Ekman(), 3) == np.around(V.L * tau_ave / (V.f * V.rho), 3))
  # sill taper: quadratic over the bottom 1000 m
  S.Hsill = 1000.0
  want = np.full(81, full)
  want[:20] = [full * (1 - (-3000 - zz)**2 / 1.0e6) for zz in S.z[:20]]
  want[-1] = 0
  got = S.calc_Ekman()
  assert np.all(np.around(want, 3) == np.around(got, 3))
  assert np.allclose(got, np.full(81, full) * S.calc_bottom_taper(1000.0, S.z) *
                     S.calc_top_taper(None, S.z, scalar=False), rtol=1e-13)
  S.Hsill = None
  # Ekman layer of 200 m: quadratic taper to the surface instead of the single zero
  S.HEk = 200
  want = np.full(81, full)
  want[77:] = [full * (1 - (zz + 200.0)**2 / 4.0e4) for zz in S.z[77:]]
  got = S.calc_Ekman()
  assert np.all(np.around(want, 3) == np.around(got, 3))
  assert np.allclose(got, np.full(81, full) * S.calc_top_taper(200, S.z, scalar=False), rtol=1e-13)
  S.HEk = None


def test_reference_unit_tests_of_calc_gm(gpu):
  """tests/modules/test_psi_SO.py:205-294 re-expressed against the wrapper."""
  S = gpu.Psi_SO(z=np.linspace(-4000, 0, 81), y=np.linspace(0, 2.0e6, 51),
                 b=np.linspace(0.03, -0.001, 81), bs=np.linspace(0.05, 0.10, 51), tau=0.12)
  ek = (S.L * 0.12) / (S.f * S.rho)
  ekman = np.full(81, ek)
  ekman[-1] = 0
  assert np.all(np.around(ekman, 3) == np.around(S.calc_Ekman(), 3))
  dy_atz = 2001000.0
  GM = np.array([S.L * S.KGM * z / dy_atz for z in S.z])
  S.Psi_Ek = S.calc_Ekman()
  assert np.all(np.around(GM, 3) == np.around(S.calc_GM(), 3))
  S.b = gpu.utils.make_func(np.linspace(0.3, 0.2, 81), S.z, 'b')   # dy floor -> -smax clip
  GM = np.full(81, -S.L * S.KGM * S.smax)
  GM[-1] = 0
  assert np.all(np.around(GM, 3) == np.around(S.calc_GM(), 3))
  S2 = gpu.Psi_SO(z=np.linspace(-4000, 0, 81), y=np.linspace(0, 2.0e3, 51),
                  b=np.linspace(0.03, -0.001, 81), bs=np.linspace(0.05, 0.10, 51), tau=0.12)
  GM = np.full(81, -S2.L * S2.KGM * 0.01)
  GM[-1] = 0
  psi_ek = np.asarray([gm + np.abs(gm / 2.0) for gm in GM])
  S2.Psi_Ek = psi_ek
  assert np.all(np.around(-psi_ek * 1e6, 3) == np.around(S2.calc_GM(), 3))


def test_twocol_so_trajectory_golden(gpu):
  """example_twocol_plusSO physics, nz=100, 2400 steps: 1e-5 from the reference (SciPy's
  adaptive BVP solver sets that floor), 1e-8 from the oracle on the same mesh."""
  g = load_golden("twocol_so")
  m = configs.twocol_so_member(nz=100, ny=40)
  cfg = dict(m, kappa=m["kappa"][None], b_basin0=m["b_basin0"][None],
             b_north0=m["b_north0"][None], bs_SO=m["bs_SO"][None], bvp_refine=8)
  ens = gpu.TwoColEnsemble(cfg)
  snaps = (1, 24, 25, 26, 2400)
  orc = drivers.run_twocol(m, 2400, set(snaps), so=True, bvp_refine=8)
  done = 0
  for s in snaps:
    ens.run(s - done)
    done = s
    st = ens.state()
    for k in ("b_basin", "b_north", "Psi", "Psi_iso_b", "Psi_iso_n", "Psi_SO"):
      assert relerr(st[k][0], g["s%05d_%s" % (s, k)]) <= TOL_BVP_REF, (s, k)
      assert relerr(st[k][0], orc[s][k]) <= 1e-8, (s, k)


def test_config4_sweep_members_vs_reference(gpu):
  g = load_golden("sweep")
  c = dict(configs.config4(N=8192), bvp_refine=8)
  ens = gpu.TwoColEnsemble(c)
  n = int(g["c4_nsteps"])
  ens.run(n)
  st = ens.state()
  idx = g["c4_members"]
  assert relerr(st["b_basin"][idx], g["c4_b_basin"]) <= TOL_BVP_REF
  assert relerr(st["b_north"][idx], g["c4_b_north"]) <= TOL_BVP_REF
  assert relerr(st["Psi"][idx], g["c4_Psi"]) <= TOL_BVP_REF
  assert relerr(st["Psi_SO"][idx], g["c4_Psi_SO"]) <= TOL_BVP_REF
  assert ens.nonfinite_members().size == 0
  keys = ("A_basin", "A_north", "bs", "bs_north", "bbot", "tau", "KGM", "kappa", "b_basin0",
          "b_north0", "bs_SO")
  for i in range(5, 8192, 512):
    m = dict(c)
    for k in keys:
      m[k] = c[k][i]
    s = drivers.run_twocol(m, n, {n}, so=True, bvp_refine=8)[n]
    for k in ("b_basin", "b_north", "Psi", "Psi_SO"):
      assert relerr(st[k][i], s[k]) <= 1e-8, (i, k)


def test_psi_so_callable_surface_profiles(gpu):
  """Callable bs / tau (hazard H7): the reference evaluates them between grid points (inside
  brentq, and on the 100-point wind average, also SOUTH of the grid for isopycnals that do not
  outcrop).  Only the host can call them: the drop-in class finds the outcrop latitudes with the
  same brentq iteration and averages the wind with the reference's own NumPy expression, and the
  kernel takes both as inputs (pm_psi_so.ys_in / tau_ave_in).  Golden G15: 1e-12 (round 1
  tabulated the callables on 2048 points: 1e-4)."""
  g = load_golden("psi_so_callable")
  m = configs.twocol_so_member(nz=100, ny=40)
  for tag, kw, tol in (("slope", dict(c=None), 1e-12), ("bvp", dict(c=0.1, bvp_with_Ek=True), TOL_BVP_ADAPT)):
    for ttag, tau in (("taufn", configs.so_tau_callable), ("tau", 0.13)):
      so = gpu.Psi_SO(z=m["z"], y=m["y"], b=m["b_basin0"], bs=configs.so_bs_callable, tau=tau,
                      f=m["f"], L=m["L"], KGM=m["KGM"], **kw)
      so.solve()
      p = tag + "_" + ttag + "_"
      assert relerr(so.Psi_Ek, g[p + "Psi_Ek"]) <= 1e-12, (p, relerr(so.Psi_Ek, g[p + "Psi_Ek"]))
      assert relerr(so.Psi, g[p + "Psi"]) <= tol, (p, relerr(so.Psi, g[p + "Psi"]))
      assert relerr(so.Psi_GM, g[p + "Psi_GM"]) <= tol, (p, relerr(so.Psi_GM, g[p + "Psi_GM"]))
  # arrays afterwards: back to the exact path on y itself
  so.update(bs=configs.so_bs_callable(m["y"]))
  so.solve()
  assert so._ny == m["y"].size or so._tau_callable


def test_config4_full_length_sweep_vs_reference(gpu):
  """G17: BASELINE config 4 at its configured length (8192 members x 2400 steps = 100 GM
  boundary-value solves per member) against the 8 members run through the reference."""
  g = load_golden("sweep_full")
  c = dict(configs.config4(N=8192), bvp_refine=8)
  n = int(g["c4_nsteps"])
  assert n == c["nsteps"] == 2400
  ens = gpu.TwoColEnsemble(c)
  ens.run(n)
  st = ens.state()
  idx = g["c4_members"]
  for k in ("b_basin", "b_north", "Psi", "Psi_SO"):
    assert relerr(st[k][idx], g["c4_" + k]) <= TOL_BVP_REF, k
  assert ens.nonfinite_members().size == 0


def test_config4_whole_baseline_ensemble_on_one_gpu(gpu):
  """BASELINE config 4's whole 65536-member ensemble on ONE GPU for its 2400 steps: no
  member goes non-finite, and two 8192-member shards run on their own (what each GPU of the
  8-GPU job holds) are bit-identical to their members in the full run."""
  N = 65536
  ens = gpu.TwoColEnsemble(dict(configs.config4(N=N), bvp_refine=8))
  ens.run(2400)
  assert ens.nonfinite_members().size == 0
  st = ens.state()
  for lo in (0, 5 * 8192):
    part = gpu.TwoColEnsemble(dict(configs.config4(N=N, members=(lo, lo + 8192)), bvp_refine=8))
    part.run(2400)
    sp = part.state()
    for k in ("b_basin", "b_north", "Psi", "Psi_SO", "Psi_iso_b"):
      assert np.array_equal(sp[k], st[k][lo:lo + 8192]), (lo, k)


# ------------------------------------------------ the default: solve_bvp's own adaptive mesh
TOL_BVP_ADAPT = 1e-11  # measured 1e-14: SciPy's mesh is reproduced node for node


def test_psi_so_golden_adaptive_mesh(gpu):
  """G5 with the DEFAULT GM solver (bvp_refine <= 0): the device follows scipy solve_bvp's
  mesh refinement (residual estimate, node insertion) decision for decision, so the BVP cases
  agree with the reference to 1e-11 (the fixed 8-fold mesh: 1e-6), and with the oracle's
  adaptive solver to 1e-11."""
  from pymoc_amd.device import DeviceArray
  g = load_golden("psi_so")
  seen = 0
  for k in range(int(g["ncases"])):
    p = "c%02d_" % k
    kw = _kwargs(g, p)
    if kw["c"] is None:
      continue
    seen += 1
    tau = g[p + "tau"]
    KGM = kw.pop("KGM")
    t = gpu.PsiSOBatch(g[p + "z"], g[p + "y"], 3, tau=float(tau) if tau.ndim == 0 else
                       np.stack([tau] * 3), KGM=KGM, **kw)
    t.update(DeviceArray.from_host(np.stack([g[p + "b"]] * 3)),
             DeviceArray.from_host(np.stack([g[p + "bs"]] * 3)))
    Psi, GM = t.Psi.download(), t.Psi_GM.download()
    assert np.array_equal(GM[0], GM[1]) and np.array_equal(GM[0], GM[2])
    assert relerr(GM[0], g[p + "Psi_GM"]) <= TOL_BVP_ADAPT, k
    assert relerr(Psi[0], g[p + "Psi"]) <= TOL_BVP_ADAPT, k
    oPsi, oEk, oGM, _ = O.psi_so_solve(g[p + "z"], g[p + "y"], g[p + "b"], g[p + "bs"],
                                       float(tau) if tau.ndim == 0 else tau, KGM=KGM, **kw)
    assert relerr(GM[0], oGM) <= TOL_BVP_ADAPT, k
    assert np.all(t.status.download() & 14 == 0)  # incl. bit 3: mesh cap never reached
  assert seen >= 12


def test_twocol_so_trajectory_golden_adaptive_mesh(gpu):
  """example_twocol_plusSO physics (BASELINE config 4's member), nz=100, 2400 steps = 100
  boundary-value solves: 1e-11 from the reference with the default GM solver."""
  g = load_golden("twocol_so")
  m = configs.twocol_so_member(nz=100, ny=40)
  cfg = dict(m, kappa=m["kappa"][None], b_basin0=m["b_basin0"][None],
             b_north0=m["b_north0"][None], bs_SO=m["bs_SO"][None])
  ens = gpu.TwoColEnsemble(cfg)
  done = 0
  for s in (1, 24, 25, 26, 2400):
    ens.run(s - done)
    done = s
    st = ens.state()
    for k in ("b_basin", "b_north", "Psi", "Psi_iso_b", "Psi_iso_n", "Psi_SO"):
      assert relerr(st[k][0], g["s%05d_%s" % (s, k)]) <= TOL_BVP_ADAPT, (s, k)


def test_config4_full_length_sweep_adaptive_mesh(gpu):
  """G17 with the default GM solver: BASELINE config 4 (8192 members x 2400 steps) against the
  8 members run through the reference at 1e-10, zero non-finite, no member hits the mesh cap;
  17 more members against the oracle's adaptive solver."""
  g = load_golden("sweep_full")
  c = configs.config4(N=8192)
  ens = gpu.TwoColEnsemble(c)
  ens.run(2400)
  st = ens.state()
  idx = g["c4_members"]
  for k in ("b_basin", "b_north", "Psi", "Psi_SO"):
    assert relerr(st[k][idx], g["c4_" + k]) <= 1e-10, k
  assert ens.nonfinite_members().size == 0
  assert np.all(ens.so.status.download() & 8 == 0)
  for i in range(5, 8192, 500):
    s = drivers.run_twocol(configs.member(c, i, 4), 2400, {2400}, so=True)[2400]
    for k in ("b_basin", "Psi", "Psi_SO"):
      assert relerr(st[k][i], s[k]) <= 1e-10, (i, k)


def test_gm_adaptive_mesh_tall_grid_vs_oracle(gpu):
  """nz > 128 takes the general adaptive kernel (chunks through LDS scratch, meshes up to
  solve_bvp's 1000 nodes): the golden BVP cases re-sampled on 160 and 257 levels against the
  oracle's solve_bvp restatement."""
  from pymoc_amd.device import DeviceArray
  g = load_golden("psi_so")
  seen = 0
  for k in range(int(g["ncases"])):
    p = "c%02d_" % k
    kw = _kwargs(g, p)
    if kw["c"] is None or g[p + "tau"].ndim != 0:
      continue
    seen += 1
    for nz in (160, 257):
      z0, b0 = g[p + "z"], g[p + "b"]
      z = np.linspace(z0[0], z0[-1], nz)
      b = np.interp(z, z0, b0)
      kw2 = dict(kw)
      KGM = kw2.pop("KGM")
      tau = float(g[p + "tau"])
      t = gpu.PsiSOBatch(z, g[p + "y"], 2, tau=tau, KGM=KGM, **kw2)
      t.update(DeviceArray.from_host(np.stack([b] * 2)), DeviceArray.from_host(np.stack([g[p + "bs"]] * 2)))
      GM = t.Psi_GM.download()
      oPsi, oEk, oGM, _ = O.psi_so_solve(z, g[p + "y"], b, g[p + "bs"], tau, KGM=KGM, **kw2)
      assert np.array_equal(GM[0], GM[1])
      assert relerr(GM[0], oGM) <= TOL_BVP_ADAPT, (k, nz)
      assert np.all(t.status.download() & 10 == 0)
    if seen >= 4:
      break
  assert seen >= 3


def test_gm_adaptive_mesh_beyond_the_register_solver(gpu, monkeypatch):
  """A thin GM boundary layer (c = 0.02) drives solve_bvp beyond the 256 nodes the
  register-resident solver follows: the first launch flags the member (status bit 3), the
  follow-up launch redoes exactly the flagged members with the general solver (meshes up to
  solve_bvp's own 1000 nodes), so the result is the oracle's / SciPy's for any input."""
  from pymoc_amd.device import DeviceArray
  m = configs.twocol_so_member(nz=100, ny=40)
  z, y = m["z"], m["y"]
  kw = dict(KGM=m["KGM"], f=m["f"], L=m["L"], bvp_with_Ek=True)
  cs = np.array([0.1, 0.02, 0.1, 0.01])  # members 1 and 3 overflow, 0 and 2 do not
  ref = [O.psi_so_solve(z, y, m["b_basin0"], m["bs_SO"], m["tau"], c=c, **kw)[2] for c in (0.1, 0.02, 0.01)]

  def run(c):
    t = gpu.PsiSOBatch(z, y, 3, tau=m["tau"], c=c, **kw)
    t.update(DeviceArray.from_host(np.stack([m["b_basin0"]] * 3)),
             DeviceArray.from_host(np.stack([m["bs_SO"]] * 3)))
    return t.Psi_GM.download(), t.status.download()

  monkeypatch.setenv("PYMOC_SO_NO_FIXUP", "1")
  GM, st = run(0.02)
  assert np.all(st & 8 == 8) and relerr(GM[0], ref[1]) > 1e-8  # flagged, last mesh's solution
  monkeypatch.delenv("PYMOC_SO_NO_FIXUP")
  for c, r in zip((0.1, 0.02, 0.01), ref):
    GM, st = run(c)
    assert np.all(st & 10 == 0), c
    assert np.array_equal(GM[0], GM[1]) and np.array_equal(GM[0], GM[2])
    assert relerr(GM[0], r) <= TOL_BVP_ADAPT, c


@pytest.mark.parametrize("nz", [9, 40, 64, 65, 66, 128])
def test_gm_adaptive_mesh_short_grids_vs_oracle(gpu, nz):
  """The register-resident adaptive solver is templated on the intervals per lane (1 for
  nz <= 65 ... 4 near its 256-node capacity): golden BVP cases re-sampled on short grids, whose
  meshes pass through several of these variants while solve_bvp refines them."""
  from pymoc_amd.device import DeviceArray
  g = load_golden("psi_so")
  seen = 0
  for k in range(int(g["ncases"])):
    p = "c%02d_" % k
    kw = _kwargs(g, p)
    if kw["c"] is None or g[p + "tau"].ndim != 0:
      continue
    seen += 1
    z0, b0 = g[p + "z"], g[p + "b"]
    z = np.linspace(z0[0], z0[-1], nz)
    b = np.interp(z, z0, b0)
    kw2 = dict(kw)
    KGM = kw2.pop("KGM")
    tau = float(g[p + "tau"])
    t = gpu.PsiSOBatch(z, g[p + "y"], 2, tau=tau, KGM=KGM, **kw2)
    t.update(DeviceArray.from_host(np.stack([b] * 2)), DeviceArray.from_host(np.stack([g[p + "bs"]] * 2)))
    GM = t.Psi_GM.download()
    oGM = O.psi_so_solve(z, g[p + "y"], b, g[p + "bs"], tau, KGM=KGM, **kw2)[2]
    assert np.array_equal(GM[0], GM[1])
    assert relerr(GM[0], oGM) <= TOL_BVP_ADAPT, (k, nz, relerr(GM[0], oGM))
    assert np.all(t.status.download() & 10 == 0)
    if seen >= 4:
      break
  assert seen >= 3


@pytest.mark.gpu
@pytest.mark.parametrize("arith", ["exact", "contracted"])
def test_twocol_so_updates_side_by_side_equal_serial_updates(gpu, arith):
  """TwoColEnsemble with an SO channel runs Psi_SO.solve and the thermal wind of an update on two
  streams and lets the column kernel form wA from Psi_iso / Psi_SO (PM_OP_WA_PSI): bit-identical
  to the serial update with the thermal-wind launch's wA1 / wA2 epilogue, over whole and split
  intervals (launches of 1-2 steps take the forcing from pm_twocol_forcing)."""
  cfg = configs.config4(N=96)
  a = gpu.TwoColEnsemble(cfg, arith=arith, overlap_updates=True)
  b = gpu.TwoColEnsemble(cfg, arith=arith, overlap_updates=False)
  assert a._overlap and not b._overlap
  for n in (1, 2, 23, 24, 25, 2, 1, 70):
    a.run(n)
    b.run(n)
    sa, sb = a.state(), b.state()
    for k in sb:
      assert np.array_equal(sa[k], sb[k], equal_nan=True), (n, k)
  assert np.array_equal(a.nonfinite_members(), b.nonfinite_members())
