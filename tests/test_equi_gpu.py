"""Column.solve_equi on the GPU: `pm_column_equi_pass` against the C
oracle on identical meshes, the refinement loop against the reference's golden vectors (G12),
and the example_iteration loop -- as user code through the drop-in classes and as an
ensemble."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from oracle import drivers
from conftest import load_golden, relerr
from pymoc_amd import configs

pytestmark = pytest.mark.gpu


def _case(g, name):
  p = name + "_"
  bzbot = float(g[p + "bzbot"])
  kappa = configs.iteration_kappa if name in g["fn_names"] else g[p + "kappa"]
  return (g[p + "z"], kappa, g[p + "Area"], g[p + "wA"], float(g[p + "bs"]),
          float(g[p + "bbot"]), None if np.isnan(bzbot) else bzbot)


def _as_arg(a):
  a = np.asarray(a) if not callable(a) else a
  return float(a) if not callable(a) and a.ndim == 0 else a


def test_pass_matches_oracle_on_ragged_meshes(gpu):
  """One launch, 6 members with meshes of 20 ... 700 nodes (non-uniform, per member), both
  boundary conditions, wA interpolated on the device: b, b' and the rms residuals against
  the oracle's mesh pass."""
  from pymoc_amd import _lib
  from pymoc_amd.device import DeviceArray
  from pymoc_amd.equilibrium import point_sets
  rng = np.random.default_rng(7)
  nz = 20
  z = np.linspace(-3500, 0, nz)
  n, mmax = 6, 704
  sizes = [20, 21, 64, 65, 333, 700]
  m = np.array(sizes, np.int32)
  x = np.zeros((n, mmax))
  Ak, dAk = np.ones((4, n, mmax)), np.zeros((4, n, mmax))
  zidx = np.zeros((n, nz), np.int32)
  wA = 1e7 * rng.standard_normal((n, nz)) * np.linspace(0.2, 1.5, n)[:, None]
  bs = rng.uniform(0.02, 0.04, n)
  bbot = rng.uniform(-1e-3, 1e-3, n)
  bzbot = 10**rng.uniform(-7, -4, n)
  flags = np.array([0, 2, 0, 2, 0, 2], np.int32)
  kap = lambda xx, i: configs.iteration_kappa(xx, kappa_4k=1e-4 * (i + 1))
  ref = []
  for i in range(n):
    extra = np.sort(rng.uniform(-3500, 0, sizes[i] - nz))
    xi = np.sort(np.concatenate([z, extra]))
    assert np.diff(xi).min() > 0
    x[i, :m[i]] = xi
    zidx[i] = np.searchsorted(xi, z)
    ak = lambda p, i=i: 8e13 * kap(p, i)
    cs = []
    for s, pts in enumerate(point_sets(xi)):
      Ak[s, i, :pts.size] = ak(pts)
      dAk[s, i, :pts.size] = np.gradient(ak(pts), pts)
      cs.append((np.interp(pts, z, wA[i]) - dAk[s, i, :pts.size]) / Ak[s, i, :pts.size])
    ref.append(O.column_equi_pass(xi, *cs, bs[i], bbot[i], bzbot[i] if flags[i] else None))
  dev = {k: DeviceArray.from_host(v) for k, v in dict(
      m=m, x=x, Ak=Ak, dAk=dAk, zidx=zidx, wA=wA, z=z, bs=bs, bbot=bbot, bzbot=bzbot,
      flags=flags).items()}
  y, rms = DeviceArray.zeros((n, 2, mmax)), DeviceArray.zeros((n, mmax))
  nadd = DeviceArray.zeros((n,), np.int32)
  b, bz = DeviceArray.zeros((n, nz)), DeviceArray.zeros((n, nz))
  d = _lib.pm_column_equi()
  d.n, d.nz, d.mmax = n, nz, mmax
  d.m, d.active, d.x, d.Ak, d.dAk = dev["m"].ptr, None, dev["x"].ptr, dev["Ak"].ptr, dev["dAk"].ptr
  d.wA, d.wA_z, d.z = None, dev["wA"].ptr, dev["z"].ptr
  d.bs, d.bbot, d.bzbot, d.flags = dev["bs"].ptr, dev["bbot"].ptr, dev["bzbot"].ptr, dev["flags"].ptr
  d.zidx, d.tol = dev["zidx"].ptr, 1e-3
  d.y, d.rms, d.nadd, d.b, d.bz = y.ptr, rms.ptr, nadd.ptr, b.ptr, bz.ptr
  _lib.check(_lib.lib.pm_column_equi_pass(C.byref(d), None))
  yh, rh, nh, bh, bzh = y.download(), rms.download(), nadd.download(), b.download(), bz.download()
  for i in range(n):
    yr, rr = ref[i]
    assert relerr(yh[i, 0, :m[i]], yr[0]) <= 1e-12, i
    assert relerr(yh[i, 1, :m[i]], yr[1]) <= 1e-12, i
    # the residuals only matter relative to tol = 1e-3; far below it they are rounding noise
    assert np.allclose(rh[i, :m[i] - 1], rr, rtol=1e-6, atol=1e-9), i
    want = int(((rr > 1e-3) & (rr < 0.1)).sum() + 2 * (rr >= 0.1).sum())
    assert nh[i] == want, (i, nh[i], want)
    assert np.array_equal(bh[i], yh[i, 0, zidx[i]]) and np.array_equal(bzh[i], yh[i, 1, zidx[i]])


def test_solve_equi_golden_through_the_drop_in_column(gpu):
  """Every G12 case through `Column.solve_equi`: same meshes as solve_bvp, 1e-10."""
  g = load_golden("equi")
  refined = 0
  for name in g["names"]:
    z, kap, A, wA, bs, bbot, bzbot = _case(g, str(name))
    col = gpu.Column(z=z, kappa=_as_arg(kap), Area=_as_arg(A), b=0.0, bs=bs, bbot=bbot,
                     bzbot=bzbot)
    col.solve_equi(wA)
    refined += col.equi_nodes > z.size
    if name == "bz_hit":  # solve_bvp gives up at max_nodes: behaviour pinned, not digits
      assert relerr(col.b, g[name + "_b"]) <= 1e-3
      continue
    assert relerr(col.b, g[name + "_b"]) <= 1e-10, name
    assert relerr(col.bz, g[name + "_bz"]) <= 1e-10, name
    _, _, x, _ = O.column_solve_equi(z, drivers.equi_coef(z, kap, A, wA), bs, bbot, bzbot)
    assert col.equi_nodes == x.size, name
  assert refined >= 5


def test_solve_equi_reference_unit_test(gpu):
  """tests/modules/test_column.py:219-245 of the reference, re-expressed: callable wA."""
  g = load_golden("equi")
  z = np.asarray(np.linspace(-4000, 0, 80))
  column = gpu.Column(Area=6e13, z=z, kappa=2e-5, bs=0.05, bbot=0.02, bzbot=0.01, b=0.03,
                      N2min=2e-7)
  column.solve_equi(np.sin)
  assert all(np.around(column.b, decimals=2) ==
             np.around(np.asarray(np.linspace(-39.95, 0.05, 80)), decimals=2))
  assert relerr(column.b, g["unit_b"]) <= 1e-12 and relerr(column.bz, g["unit_bz"]) <= 1e-11
  # ode / bc helpers (test_column.py:199-217)
  column.wA = np.sin
  assert (column.ode(z, [column.b, column.bz]) == np.vstack(
      (column.bz, (np.sin(z) - column.dAkappa_dz(z)) / column.Akappa(z) * column.bz))).all()
  assert (column.bc([1., 2.], [3., 4.]) == np.array([2. - 0.01, 3. - 0.05])).all()


def test_batch_refines_members_independently(gpu):
  """16 members on one 20-level grid, half of which need refinement (to different meshes):
  each against the oracle run on that member alone."""
  z = np.linspace(-3500, 0, 20)
  n = 16
  rng = np.random.default_rng(3)
  amp = np.where(np.arange(n) % 2 == 0, rng.uniform(0.9, 1.6, n), rng.uniform(0.1, 0.3, n))
  kap = rng.uniform(1.5e-5, 3e-5, n)
  A = rng.uniform(5e13, 1e14, n)
  drv = drivers.run_iteration(configs.iteration_member(nz=20), 1, {1})[1]
  wA = amp[:, None] * drv["Psi"][None, :] * 1e6
  eq = gpu.ColumnEquiBatch.from_profiles(z, kap[:, None] + 0 * z, A[:, None] + 0 * z, 0.03,
                                         -0.0004, bzbot=1e-3, n=n)
  eq.solve(wA)
  b, bz = eq.get_b(), eq.get_bz()
  assert eq.passes >= 3
  for i in range(n):
    bo, bzo, x, st = O.column_solve_equi(z, drivers.equi_coef(z, kap[i], A[i], wA[i]), 0.03,
                                         -0.0004, 1e-3)
    assert eq.nodes[i] == x.size and eq.status[i] == st, i
    assert relerr(b[i], bo) <= 1e-11 and relerr(bz[i], bzo) <= 1e-11, i
  assert (eq.nodes > 20).sum() == 8 and (eq.nodes == 20).sum() == 8 and eq.nodes.max() == 52
  # a second solve starts again from the column grid (solve_bvp always does)
  eq.solve(0.2 * wA)
  assert eq.nodes.max() < 40


@pytest.mark.parametrize("tag", ["fn", "arr"])
def test_example_iteration_user_loop(gpu, tag):
  """examples/example_iteration.py:59-68 written as in the script, with the drop-in
  classes; kappa as the script's lambda and as samples on z."""
  g = load_golden("iteration_" + tag)
  m = configs.iteration_member()
  z = m["z"]
  kappa = configs.iteration_kappa if tag == "fn" else m["kappa"]
  AMOC = gpu.Psi_Thermwind(z=z, b1=m["b_basin0"].copy())
  AMOC.solve()
  basin = gpu.Column(z=z, kappa=kappa, Area=m["A_basin"], b=m["b_basin0"].copy(), bs=m["bs"],
                     bbot=m["bbot"])
  for ii in range(30):
    wA = AMOC.Psi * 1e6
    basin.solve_equi(wA)
    AMOC.update(b1=0.8 * AMOC.b1(z) + 0.2 * basin.b)
    AMOC.solve()
    if ii + 1 in (1, 2, 10, 30):
      p = "s%05d_" % (ii + 1)
      assert relerr(basin.b, g[p + "b"]) <= 1e-11 and relerr(basin.bz, g[p + "bz"]) <= 1e-11
      assert relerr(AMOC.Psi, g[p + "Psi"]) <= 1e-11


def test_iteration_ensemble_golden(gpu):
  """EquiIterationEnsemble (256 members) against the reference on 4 members and the oracle
  loop on 12 more."""
  g = load_golden("iteration_sweep")
  c = configs.config_iteration(N=256)
  ens = gpu.EquiIterationEnsemble(c)
  ens.iterate(30)
  st = ens.state()
  for j, i in enumerate(g["members"]):
    for k in ("b", "Psi", "b1"):
      assert relerr(st[k][i], g[k][j]) <= 1e-11, (i, k)
  for i in range(5, 256, 21):
    mem = dict(c, bs=c["bs"][i], A_basin=c["A_basin"][i], kappa=c["kappa"][i],
               b_basin0=c["b_basin0"][i])
    o = drivers.run_iteration(mem, 30, {30})[30]
    for k in ("b", "bz", "Psi", "b1"):
      assert relerr(st[k][i], o[k]) <= 1e-12, (i, k)
  assert np.isfinite(st["b"]).all() and np.isfinite(st["Psi"]).all()
