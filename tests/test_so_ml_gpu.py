"""GPU parity of K5 (pm_so_ml_step), the JN2018 BC switch and the JN2018 driver (config 5)."""
import numpy as np
import pytest

import oracle as O
from oracle import drivers
from conftest import load_golden, relerr
from pymoc_amd import configs

pytestmark = pytest.mark.gpu


def test_so_ml_golden(gpu):
  from pymoc_amd.device import DeviceArray
  g = load_golden("so_ml")
  Ks, h, L, v_pist = g["par"]
  nz = g["z"].size
  for k in range(int(g["ncases"])):
    p = "c%02d_" % k
    dt = float(g[p + "dt"])
    t = gpu.SOMLBatch(g["y"], nz, g[p + "bs0"], surflux=g["surflux"], rest_mask=g["rest_mask"],
                      b_rest=g["b_rest"], Ks=Ks, h=h, L=L, v_pist=v_pist)
    bb, pb = DeviceArray.from_host(g["b_basin"][None]), DeviceArray.from_host(g[p + "Psi_b"][None])
    t.step(bb, pb, dt)
    obs, ops = O.so_ml_advdiff(g["y"], g["surflux"], g["rest_mask"], g["b_rest"], g[p + "bs0"],
                               g["b_basin"], g[p + "Psi_b"], dt, Ks=Ks, h=h, L=L, v_pist=v_pist)
    bs1, ps1 = t.bs.download()[0], t.Psi_s.download()[0]
    assert np.array_equal(bs1, obs) and np.array_equal(ps1, ops), k  # same Thomas sweep
    assert relerr(bs1, g[p + "bs1"]) <= 1e-14, k  # reference: dense inverse
    assert relerr(ps1, g[p + "Psi_s1"]) <= 1e-14 or np.abs(g[p + "Psi_s1"]).max() == 0, k
    for _ in range(4):
      t.step(bb, pb, dt)
    assert relerr(t.bs.download()[0], g[p + "bs5"]) <= 1e-13, k
    assert t.status.download()[0] == 0


@pytest.mark.parametrize("nz,ny", [(2, 3), (5, 4), (64, 51), (100, 64), (200, 65), (81, 200),
                                    (300, 513)])
def test_so_ml_ragged_sizes_vs_oracle_bitwise(gpu, nz, ny):
  from pymoc_amd.device import DeviceArray
  rng = np.random.default_rng(nz * 3 + ny)
  n = 6
  y = np.linspace(0, 2e6, ny)
  b_basin = np.sort(0.03 * rng.random((n, nz)), axis=1) - 0.002
  Psi_b = 5 * rng.standard_normal((n, nz))
  Psi_b[1, : nz // 2] = 0.
  Psi_b[2] = -np.abs(Psi_b[2])
  bs = np.sort(0.03 * rng.random((n, ny)), axis=1) - 0.001
  bs[3, : ny // 3 + 1] = bs[3, ny // 3::-1][: ny // 3 + 1]
  surflux = -1e-9 * rng.random((n, ny))
  rest = (rng.random((n, ny)) < 0.7).astype(float)
  b_rest = bs + 1e-3 * rng.standard_normal((n, ny))
  dt = 86400. * 10
  t = gpu.SOMLBatch(y, nz, bs, surflux=surflux, rest_mask=rest, b_rest=b_rest, Ks=400., h=50.,
                    L=4e6, v_pist=1.5 / 86400)
  bb, pb = DeviceArray.from_host(b_basin), DeviceArray.from_host(Psi_b)
  for _ in range(3):
    t.step(bb, pb, dt)
  out, ps = t.bs.download(), t.Psi_s.download()
  for m in range(n):
    o = bs[m]
    for _ in range(3):
      o, ops = O.so_ml_advdiff(y, surflux[m], rest[m], b_rest[m], o, b_basin[m], Psi_b[m], dt,
                               Ks=400., h=50., L=4e6, v_pist=1.5 / 86400)
    assert np.array_equal(out[m], o), (nz, ny, m)
    assert np.array_equal(ps[m], ops), (nz, ny, m)


def test_reference_unit_test_of_advdiff(gpu):
  """tests/modules/test_SO_ML.py:129-170 re-expressed against the wrapper (a cosine bs against
  the analytic tendency, 5 %), and the kernel against the class's own host helpers composed in
  advdiff's order (SO_ML.py:198-274)."""
  dt = 60 * 86400
  y = np.linspace(0, 2.0e6, 51)
  Ks, L, h, surflux = 100, 4e6, 50, 5.9e3
  dth = 2.0 * np.pi / 2.0e6
  b_basin = np.asarray([0.02 * (n / 2.0e6)**2 for n in y])
  bs = np.asarray([b_basin[-1] * np.cos(n * dth) for n in y])
  conf = dict(y=y, Ks=Ks, h=h, L=L, surflux=surflux, rest_mask=0.0, b_rest=0.0,
              v_pist=2.0 / 86400.0, bs=bs)
  Psi_b = np.linspace(1e4, 2.0e4, 51)
  ml = gpu.SO_ML(**conf)
  dbs_dy = -dth * b_basin[-1] * np.sin(y * dth)
  d2bs_dy2 = -dth**2 * b_basin[-1] * np.cos(y * dth)
  db = -((Psi_b / (h * L)) * dbs_dy + Ks * d2bs_dy2 + surflux / h) * dt
  b = -(ml.bs.copy() + db)
  # the same step through the host helpers, in advdiff's order
  hm = gpu.SO_ML(**dict(conf, bs=bs.copy()))
  Psi_mod = Psi_b.copy()
  ind = np.nonzero(Psi_mod)[0][0]
  Psi_mod[:ind] = Psi_mod[ind]
  hm.Psi_s = np.interp(hm.bs, b_basin, Psi_mod)
  hm.Psi_s[:np.argmin(hm.bs)] = 0.
  hm.Psi_s[0] = 0.
  hm.set_boundary_conditions(b_basin, Psi_b)
  flux = hm.surflux / hm.h + hm.rest_mask * hm.v_pist / hm.h * (hm.b_rest - hm.bs)
  dy = y[1] - y[0]
  hm.bs = hm.bs + dt * (flux + hm.calc_advective_tendency(dy))
  if hm.Psi_s[1] <= 0:
    hm.bs[0] = hm.bs[1]
  hm.bs = hm.calc_implicit_diffusion(dy, dt)
  hm.set_boundary_conditions(b_basin, Psi_b)
  ml.advdiff(b_basin, Psi_b, dt)
  assert all(np.abs(b[i] - ml.bs[i]) / b[i] < 0.05 for i in range(len(b)))
  # (the scale of this test case is |bs| ~ 1e4: surflux / h x dt)
  assert relerr(ml.bs, hm.bs) <= 1e-13 and relerr(ml.Psi_s, hm.Psi_s) <= 1e-14


def test_so_ml_wrapper_api(gpu):
  g = load_golden("so_ml")
  Ks, h, L, v_pist = g["par"]
  p = "c00_"
  ch = gpu.SO_ML(y=g["y"], h=h, L=L, Ks=Ks, surflux=g["surflux"].copy(),
                 rest_mask=g["rest_mask"].copy(), b_rest=g["b_rest"].copy(), v_pist=v_pist,
                 bs=g[p + "bs0"].copy())
  ch.timestep(b_basin=g["b_basin"], Psi_b=g[p + "Psi_b"], dt=float(g[p + "dt"]))
  assert relerr(ch.bs, g[p + "bs1"]) <= 1e-14
  assert relerr(ch.Psi_s, g[p + "Psi_s1"]) <= 1e-14
  with pytest.raises(TypeError) as e:
    ch.timestep(b_basin=1.0, Psi_b=g[p + "Psi_b"])
  assert str(e.value) == 'b_basin needs to be numpy array providing buoyancy levels in basin'
  with pytest.raises(TypeError) as e:
    ch.timestep(b_basin=g["b_basin"], Psi_b=2.0)
  assert str(e.value) == ('Psi_b needs to be numpy array providing overturning at buoyancy '
                          'levels given by b_basin')
  with pytest.raises(IndexError):  # hazard H10: Psi_b identically zero
    ch.timestep(b_basin=g["b_basin"], Psi_b=0 * g[p + "Psi_b"], dt=86400.)
  with pytest.raises(TypeError) as e:
    gpu.SO_ML(y=1.0)
  assert str(e.value) == 'y needs to be numpy array providing (regular) grid'


@pytest.mark.parametrize("name,nz,dtd,steps,use_graph,fused", [
    ("jn2018_nz81", 81, 30., (1, 12, 13, 14, 240, 1200), False, False),
    ("jn2018_nz200", 200, 10., (1, 36, 37, 38, 360, 1200), True, False),
    ("jn2018_nz81", 81, 30., (1, 12, 13, 14, 240, 1200), False, True),
    ("jn2018_nz200", 200, 10., (1, 36, 37, 38, 360, 1200), False, True),
])
def test_jn2018_trajectory_golden(gpu, name, nz, dtd, steps, use_graph, fused):
  """run_JansenNadeau_2018 physics against the reference's snapshots (1e-10: ys by direct
  inversion instead of brentq) -- eager launches, hipGraph replay of whole MOC blocks, and
  the fused one-launch-per-block kernel."""
  g = load_golden(name)
  m = configs.jn2018_member(nz=nz, dt_days=dtd)
  cfg = dict(m)
  for k in ("b_basin0", "b_north0", "bs_SO0", "surflux", "b_rest"):
    cfg[k] = m[k][None]
  cfg["rest_mask"] = m["rest_mask"][None]
  ens = gpu.JN2018Ensemble(cfg, use_graph=use_graph, fused=fused)
  done = 0
  for s in steps:
    ens.run(s - done)
    done = s
    st = ens.state()
    for k in ("b_basin", "b_north", "bs_SO", "Psi", "Psi_SO", "Psi_iso_b", "Psi_iso_n", "Psi_s"):
      assert relerr(st[k][0], g["s%05d_%s" % (s, k)]) <= 1e-10, (s, k)


@pytest.mark.parametrize("name,kw,fused", [
    ("single_basin", {}, False),
    ("single_basin", {}, True),
    ("single_basin_var", dict(kapfac=1.5, tau=0.16, KGM=800., B=3.0e4), True),
])
def test_single_global_basin_trajectory_golden(gpu, name, kw, fused):
  """examples/run_single_global_basin.py (G11): same loop as JN2018, global-ocean parameters,
  nz=46, MOC update every 24 steps -- through JN2018Ensemble unchanged."""
  g = load_golden(name)
  m = configs.single_basin_member(**kw)
  cfg = dict(m)
  for k in ("b_basin0", "b_north0", "bs_SO0", "surflux", "b_rest", "rest_mask"):
    cfg[k] = m[k][None]
  ens = gpu.JN2018Ensemble(cfg, fused=fused)
  done = 0
  for s in (1, 24, 25, 26, 240, 1200):
    ens.run(s - done)
    done = s
    st = ens.state()
    for k in ("b_basin", "b_north", "bs_SO", "Psi", "Psi_SO", "Psi_iso_b", "Psi_iso_n", "Psi_s"):
      assert relerr(st[k][0], g["s%05d_%s" % (s, k)]) <= 1e-10, (s, k)


def test_jn2018_fused_equals_stepwise_bitwise(gpu):
  """The fused per-block kernel against bc_switch + column_steps + so_ml_step launches: bitwise
  on every member that is still finite, and the same members lost (member 2 of this draw goes
  non-finite at step 37 in the reference too, which then raises: what a lost member's NaNs do
  afterwards is not defined by the reference, and the two paths treat them differently)."""
  c = configs.config5(N=256)
  c["rest_mask"] = np.repeat(c["rest_mask"][None], 256, axis=0)
  a = gpu.JN2018Ensemble(c, fused=True)
  b = gpu.JN2018Ensemble(c, fused=False)
  for n in (1, 35, 36, 37, 200):
    a.run(n)
    b.run(n)
    sa, sb = a.state(), b.state()
    ok = np.ones(256, dtype=bool)
    ok[b.nonfinite_members()] = False
    assert np.array_equal(a.nonfinite_members(), b.nonfinite_members())
    assert ok.sum() >= 254
    for k in sa:
      assert np.array_equal(sa[k][ok], sb[k][ok]), (n, k)
    bb_a, bb_b = a.cols.bbot.download(), b.cols.bbot.download()
    ka, kb = a.cols.ksel.download(), b.cols.ksel.download()
    ok2 = np.concatenate([ok, ok])
    assert np.array_equal(bb_a[ok2], bb_b[ok2])
    assert np.array_equal(ka[ok2], kb[ok2])


@pytest.mark.parametrize("nz,ny", [(81, 51), (100, 80), (200, 129)])
def test_jn2018_fused_equals_stepwise_other_shapes(gpu, nz, ny):
  """Fused == stepwise on shapes other than config 5's: fewer levels per lane, and channels
  longer than a wavefront (ny > 64: mixed layer in LDS with the ordered Thomas sweep).
  Compared on the members that stay finite (a member the explicit scheme loses is lost in both;
  what its NaNs do afterwards is not defined by the reference, which raises)."""
  # (the script's effective diffusivity grows as dt shrinks: nz <= 100 is stable at 30 d, not 10)
  c = configs.config5(N=64, nz=nz, ny=ny, dt_days=30. if nz <= 100 else 10.)
  c["rest_mask"] = np.repeat(c["rest_mask"][None], 64, axis=0)
  a = gpu.JN2018Ensemble(c, fused=True)
  b = gpu.JN2018Ensemble(c, fused=False)
  a.run(100)
  b.run(100)
  sa, sb = a.state(), b.state()
  ok = np.isfinite(sb["b_basin"]).all(axis=1) & np.isfinite(sb["bs_SO"]).all(axis=1)
  assert ok.mean() > 0.9
  assert np.array_equal(np.isfinite(sa["b_basin"]).all(axis=1) & np.isfinite(sa["bs_SO"]).all(axis=1), ok)
  for k in sa:
    assert np.array_equal(sa[k][ok], sb[k][ok]), (nz, ny, k)


@pytest.mark.parametrize("nz,dtd,arith", [(200, 10., "exact"), (199, 10., "exact"), (81, 30., "exact"),
                                          (100, 30., "exact"), (200, 10., "contracted")])
def test_jn2018_split_lane_layout_bitwise(gpu, nz, dtd, arith):
  """PM_JN_SPLIT_LANES (round 5, opt-in): both columns of a member stepping together, one per
  half of the wavefront (ceil(nz/32) levels per lane; rows moved through LDS, 16-byte accesses
  for even nz, 8-byte for odd).  Same arithmetic per level: bit-identical to the default fused
  kernel at every split of a run, lost members included, and so are bbot / ksel / Psi_s."""
  c = configs.config5(N=96, nz=nz, dt_days=dtd)
  c["rest_mask"] = np.repeat(c["rest_mask"][None], 96, axis=0)
  a = gpu.JN2018Ensemble(c, arith=arith, split_lanes=True)
  b = gpu.JN2018Ensemble(c, arith=arith)
  for n in (1, 35, 36, 37, 150):
    a.run(n)
    b.run(n)
    sa, sb = a.state(), b.state()
    assert np.array_equal(a.nonfinite_members(), b.nonfinite_members())
    ok = np.ones(96, dtype=bool)
    ok[b.nonfinite_members()] = False
    for k in sa:
      assert np.array_equal(sa[k][ok], sb[k][ok]), (n, k)
    ok2 = np.concatenate([ok, ok])
    assert np.array_equal(a.cols.bbot.download()[ok2], b.cols.bbot.download()[ok2])
    assert np.array_equal(a.cols.ksel.download()[ok2], b.cols.ksel.download()[ok2])
    assert np.array_equal(a.ml.status.download(), b.ml.status.download())


@pytest.mark.parametrize("arith", ["exact", "contracted"])
def test_jn2018_fused_shared_coefficient_rows(gpu, arith):
  """PM_JN_SHARED_COEF: when every member carries the same kappa / Area profiles the fused loop
  reads one copy per column kind (rows 0 and n) -- bit-identical to reading each member's own
  rows; a sweep whose members differ in kappa must not get the hint and equals the stepwise
  launches as before."""
  c = configs.config5(N=128)
  c["rest_mask"] = np.repeat(c["rest_mask"][None], 128, axis=0)
  a = gpu.JN2018Ensemble(c, arith=arith)                     # hint set by the driver
  b = gpu.JN2018Ensemble(c, arith=arith, shared_coef=False)  # every member's own rows
  assert a.cols.shared_halves and a.cols.uniform_area
  for n in (1, 36, 72):
    a.run(n)
    b.run(n)
    sa, sb = a.state(), b.state()
    assert np.array_equal(a.nonfinite_members(), b.nonfinite_members())
    for k in sa:
      assert np.array_equal(sa[k], sb[k], equal_nan=True), (n, k)
  # members with their own kappa: no hint, fused == stepwise
  rng = np.random.default_rng(5)
  c2 = dict(c)
  f = 1.0 + 0.05 * rng.random(128)
  c2["kappa"] = np.asarray(c["kappa"])[None, :] * f[:, None]
  c2["kappaeff"] = np.asarray(c["kappaeff"])[None, :] * f[:, None]
  d = gpu.JN2018Ensemble(c2, fused=True)
  e = gpu.JN2018Ensemble(c2, fused=False)
  assert not d.cols.shared_halves
  d.run(80)
  e.run(80)
  sd, se = d.state(), e.state()
  ok = np.ones(128, dtype=bool)
  ok[e.nonfinite_members()] = False
  assert np.array_equal(d.nonfinite_members(), e.nonfinite_members()) and ok.sum() >= 120
  for k in sd:
    assert np.array_equal(sd[k][ok], se[k][ok]), k


def test_jn2018_fused_area_variants_and_hint_check(gpu):
  """The fused kernel has a uniform-Area variant (pm_jn2018.hints); with a basin area that
  varies in z the driver must pick the general variant (still bit-identical to the stepwise
  launches), and a WRONG hint is detected on the device: members flagged, state untouched."""
  c = configs.config5(N=64)
  c["rest_mask"] = np.repeat(c["rest_mask"][None], 64, axis=0)
  nz = c["z"].size
  c["A_basin"] = np.broadcast_to(np.asarray(c["A_basin"], dtype=np.float64).reshape(-1, 1),
                                 (64, 1)) * (1. + 0.2 * np.linspace(0., 1., nz)[None])
  a = gpu.JN2018Ensemble(c, fused=True)
  b = gpu.JN2018Ensemble(c, fused=False)
  assert not a.cols.uniform_area
  a.run(80)
  b.run(80)
  sa, sb = a.state(), b.state()
  ok = np.ones(64, dtype=bool)
  ok[b.nonfinite_members()] = False  # (a lost member's NaNs: not defined by the reference)
  assert np.array_equal(a.nonfinite_members(), b.nonfinite_members()) and ok.sum() >= 62
  for k in sa:
    assert np.array_equal(sa[k][ok], sb[k][ok]), k
  before = a.cols.get_b().copy()
  a.cols.uniform_area = True  # lie to the library
  a.run(5)
  assert np.all(a.ml.status.download() & 16 == 16)
  assert np.array_equal(a.cols.get_b(), before, equal_nan=True)
  u = gpu.JN2018Ensemble(configs.config5(N=64), fused=True)
  assert u.cols.uniform_area
  u.run(5)
  assert np.all(u.ml.status.download() & 16 == 0)


@pytest.mark.parametrize("fused", [False, True])
def test_config5_sweep_members_vs_reference(gpu, fused):
  g = load_golden("sweep")
  c = configs.config5(N=4096)
  c["rest_mask"] = np.repeat(c["rest_mask"][None], 4096, axis=0)
  ens = gpu.JN2018Ensemble(c, use_graph=not fused, fused=fused)
  n, nl = int(g["c5_nsteps"]), int(g["c5_long_nsteps"])
  ens.run(n)
  st = ens.state()
  idx = g["c5_members"]
  for k in ("b_basin", "b_north", "bs_SO", "Psi_SO"):
    assert relerr(st[k][idx], g["c5_" + k]) <= 1e-10, k
  ens.run(nl - n)
  st = ens.state()
  idx = g["c5_long_members"]
  for k in ("b_basin", "b_north", "bs_SO", "Psi_SO"):
    assert relerr(st[k][idx], g["c5_long_" + k]) <= 1e-9, k


@pytest.mark.parametrize("arith", ["exact", "contracted"])
def test_config5_full_length_sweep_vs_reference(gpu, arith):
  """G17: BASELINE config 5 (4096 members, nz=200) over its configured 3600 steps, sampled
  every 72 steps against the 8 members run through the reference: 1e-10 up to each member's
  first Psib flip (measured; some members never flip), bounded after it (check_config5_full).
  The members the run loses are exactly the two the REFERENCE loses (2 and 1268, non-finite
  from step 37 on; pinned in the fixture and in test_oracle_golden).
  arith="contracted": the columns of the fused loop in the opt-in tolerance mode
  (PM_JN_CONTRACTED) are held to the SAME statement -- 1e-10 until a member's first flip (which
  another rounding may bring earlier), the same bounds after it."""
  from test_oracle_golden import check_config5_full
  g = load_golden("sweep_full")
  c = configs.config5(N=4096)
  c["rest_mask"] = np.repeat(c["rest_mask"][None], 4096, axis=0)
  ens = gpu.JN2018Ensemble(c, arith=arith)
  idx = g["c5_members"]
  steps = [int(t) for t in g["c5_steps"]]
  traj = [dict() for _ in idx]
  done = 0
  for t in steps:
    ens.run(t - done)
    done = t
    st = ens.state()
    for j, i in enumerate(idx):
      traj[j][t] = {k: st[k][i] for k in ("b_basin", "b_north", "bs_SO", "Psi_SO")}
  clean = [check_config5_full(traj[j], g, j) for j in range(len(idx))]
  print("config 5, %s: followed to 1e-10 until step" % arith, clean)
  # ... and sample by sample no further from the reference than the reference is from ITSELF when
  # an initial profile is moved by one ulp (fixture G19)
  from test_oracle_golden import check_config5_envelope
  gc = load_golden("c5_conditioning")
  ratios = [check_config5_envelope(traj[j], g, gc, j) for j in range(len(idx))]
  print("config 5, %s: largest distance / reference's own one-ulp envelope" % arith, ratios)
  assert sum(t == 3600 for t in clean) >= (2 if arith == "exact" else 1), clean  # no drift where no Psib flip happens
  assert np.median(clean) >= (1000 if arith == "exact" else 144), clean
  assert list(ens.nonfinite_members()) == list(g["c5_blowup_members"]) == [2, 1268]
  ens.run(72)  # the bench's 72 warm-up + 3600 steps
  assert list(ens.nonfinite_members()) == [2, 1268]


@pytest.mark.parametrize("arith", ["exact", "contracted"])
def test_config5_teacher_forced_windows_vs_reference(gpu, arith):
  """The 3600 steps of config 5 pinned WINDOW BY WINDOW: the engine restarted
  from the reference's stored state at each of the 50 sample steps of fixture G17 and run for 72
  steps -- 8 sweep members x 50 windows = 400 members of one ensemble.  Every window whose inner
  MOC update takes the reference's branch for the bottom cell of Psi_Thermwind.Psib
  (psi_thermwind.py:177-183; fixture G19 holds the reference's branch at every update) must
  reproduce the reference's next snapshot to 1e-10; windows taking the other branch ("flips") are
  counted and must stay inside the reference's own one-ulp conditioning.  All 8 members, both
  arithmetic modes."""
  from test_oracle_golden import c5_window_config, check_config5_windows
  g, gc = load_golden("sweep_full"), load_golden("c5_conditioning")
  c = configs.config5(N=4096)
  cfg, rows = c5_window_config(c, g)
  cfg["rest_mask"] = np.repeat(c["rest_mask"][None], len(rows), axis=0)
  ens = gpu.JN2018Ensemble(cfg, arith=arith)
  ens.run(36)
  s36 = ens.state()
  ens.run(36)
  s72 = ens.state()
  assert ens.nonfinite_members().size == 0
  mid = [{k: s36[k][r] for k in ("b_basin", "b_north")} for r in range(len(rows))]
  end = [{k: s72[k][r] for k in ("b_basin", "b_north", "bs_SO", "Psi_SO", "Psi")}
         for r in range(len(rows))]
  flips, wc, wf = check_config5_windows(rows, g, gc, mid, end, arith)
  print("config 5, %s: %d of %d teacher-forced windows take another bottom-cell branch than the "
        "reference %s; worst clean window %.2e, worst flip window %.2e" %
        (arith, len(flips), len(rows), flips[:8], wc, wf))
  # measured (profiles/r05/c5_windows.log): exact 0 of 400 (worst clean window 3.6e-14 -- as close
  # as the oracle's 6e-14), contracted 8 of 400 (worst flip window 1.1e-4); bound = that + margin
  assert len(flips) <= (2 if arith == "exact" else 24), flips


def test_config5_blowup_members_go_at_the_references_step(gpu):
  g = load_golden("sweep_full")
  c = configs.config5(N=4096, members=(0, 1280))
  ens = gpu.JN2018Ensemble(c)
  ens.run(36)
  assert ens.nonfinite_members().size == 0
  st = ens.state()
  for i in (2, 1268):
    for k in ("b_basin", "b_north", "bs_SO"):
      assert relerr(st[k][i], g["c5_blowup_%d_%s" % (i, k)]) <= 1e-10, (i, k)
  ens.run(1)
  assert list(ens.nonfinite_members()) == [2, 1268]


def test_config5_whole_baseline_ensemble_on_one_gpu(gpu):
  """BASELINE config 5's whole 32768-member ensemble on ONE GPU for 3600 steps: a
  4096-member shard run on its own is bit-identical to its members of the full run.  A few
  dozen members (mostly db within 1e-6 of 6.019e-4, like members 2 and 1268 of the
  4096-member draw, which the reference itself loses at step 37) go non-finite; every one of
  them also goes non-finite in the oracle, and oracle-finite neighbours stay finite here."""
  N = 32768
  cfg = configs.config5(N=N)
  ens = gpu.JN2018Ensemble(cfg)
  ens.run(3600)
  bad = ens.nonfinite_members()
  assert 0 < bad.size <= N // 500
  for i in bad:
    s = drivers.run_jn2018(configs.member(cfg, int(i), 5), 3600, {3600})[3600]
    assert not np.isfinite(s["b_basin"]).all() or not np.isfinite(s["b_north"]).all(), i
  st = ens.state()
  for i in (0, 4097, 20000, N - 1):
    assert i not in bad
    s = drivers.run_jn2018(configs.member(cfg, i, 5), 3600, {3600})[3600]
    assert np.isfinite(s["b_basin"]).all() and np.isfinite(st["b_basin"][i]).all(), i
  lo = 3 * 4096
  part = gpu.JN2018Ensemble(configs.config5(N=N, members=(lo, lo + 4096)))
  part.run(3600)
  sp = part.state()
  for k in ("b_basin", "b_north", "bs_SO", "Psi_SO"):
    assert np.array_equal(sp[k], st[k][lo:lo + 4096], equal_nan=True), k


def test_jn2018_fused_flags_operands_outside_the_division_window(gpu):
  """The fused loop divides with the 4-instruction exact sequence only (no IEEE leg): a member
  whose column operands leave the window in which that is IEEE-identical (common.hip.h) is
  stepped but FLAGGED (status bit 5), the others are not."""
  c = configs.config5(N=64)
  e = gpu.JN2018Ensemble(c, fused=True)
  e.run(36)
  assert np.all(e.ml.status.download() & 32 == 0)
  b = e.cols.get_b()
  b[7] *= 2.0**-1000          # basin column of member 7
  b[64 + 9, 50] = 2.0**300    # one level of the northern column of member 9
  e.cols.set_b(b)
  e.run(36)
  st = e.ml.status.download()
  lost = set(int(i) for i in e.nonfinite_members())  # (member 2 of this draw: the reference's own)
  flagged = set(int(i) for i in np.nonzero(st & 32)[0])
  assert {7, 9} <= flagged and flagged <= {7, 9} | lost


@pytest.mark.parametrize("arith", ["exact"])
def test_config5_every_member_vs_reference_digests(gpu, arith):
  """Fixture G21 (round 5): all 4096 config-5 members through the REFERENCE for 72 and 360 steps
  (2 and 10 MOC intervals), {sum, sum of squares} of b_basin / b_north / bs_SO / Psi_SO per
  member.  After 72 steps every finite member agrees to 1e-10 (no bottom-cell flip can have
  acted yet: DESIGN.md section 4 fact 2) and the reference's two non-finite members are the
  engine's; after 360 steps the members that took another bottom-cell branch than the reference
  are counted (a few per cent) and stay inside the reference's own one-ulp conditioning."""
  from conftest import digest_err
  g = load_golden("c5_ensemble_digests")
  c = configs.config5(N=4096)
  c["rest_mask"] = np.repeat(c["rest_mask"][None], 4096, axis=0)
  ens = gpu.JN2018Ensemble(c, arith=arith)
  done = 0
  for a, t in enumerate(int(x) for x in g["steps"]):
    ens.run(t - done)
    done = t
    st = ens.state()
    ref = g["digest"][:, a]  # [member, field, 2]
    lost_ref = np.nonzero(~np.isfinite(ref).all(axis=(1, 2)))[0]
    worst = np.zeros(4096)
    for f, k in enumerate(g["fields"]):
      x = st[str(k)]
      e = digest_err(np.where(np.isfinite(x), x, 0.), np.nan_to_num(ref[:, f]))
      worst = np.maximum(worst, e)
    fin = np.ones(4096, dtype=bool)
    fin[lost_ref] = False
    lost_eng = ens.nonfinite_members()
    assert list(lost_ref) == list(lost_eng) == [2, 1268]
    clean = worst[fin] <= 1e-10
    print("config 5 step %d: %d of %d finite members within 1e-10 of the reference (worst %.1e); "
          "%d beyond (worst %.1e)" % (t, int(clean.sum()), int(fin.sum()), worst[fin][clean].max(),
                                      int((~clean).sum()),
                                      worst[fin][~clean].max() if (~clean).any() else 0.))
    if t == 72:
      assert clean.all()
    else:
      assert (~clean).mean() <= 0.10 and (not (~clean).any() or worst[fin][~clean].max() <= 5e-3)
