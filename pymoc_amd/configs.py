"""Synthetic parameter-sweep ensembles for BASELINE.json's configs.

Pure NumPy parameter generators -- no compute.  A member's parameters depend only on
(seed, member index), never on how the ensemble is sharded: generators draw the full
N-member parameter table and then slice `members`.

Physics constants follow the reference's example scripts
(examples/example_timestepping.py, example_twocol.py, example_twocol_plusSO.py,
run_JansenNadeau_2018.py); sweep ranges follow SURVEY.md section 8(d), narrowed where the
explicit column scheme would be unstable (kappa*dt/dz^2 <= 1/2, column.py:245-249).
"""
import numpy as np

DAY = 86400.


def _logu(rng, lo, hi, n):
  return np.exp(rng.uniform(np.log(lo), np.log(hi), n))


def _slice(members, N):
  if members is None:
    return slice(0, N)
  lo, hi = members
  return slice(int(lo), int(hi))


# keys of each ensemble config that carry one entry per member (first axis = member)
PER_MEMBER = {
    2: ("kappa", "kappa_back", "Area", "wA", "b0", "bs", "bbot", "N2min", "do_conv"),
    3: ("kappa", "A_basin", "A_north", "bs", "bs_north", "bbot", "b_basin0", "b_north0"),
    4: ("kappa", "tau", "KGM", "A_basin", "A_north", "bs", "bs_north", "bbot", "bs_SO",
        "b_basin0", "b_north0"),
    5: ("bs", "bs_north", "KGM", "tau", "surflux", "b_rest", "bs_SO_init", "bs_SO0",
        "b_basin0", "b_north0"),
    6: ("tau", "K", "A_Pac", "A_Atl", "A_north"),  # config_twobasin
}


def member(cfg, j, config):
  """Member j (local index) of an ensemble config as the single-member dict the
  member-by-member drivers take (`twocol_member` / `jn2018_member` layout)."""
  m = dict(cfg)
  for k in PER_MEMBER[config]:
    m[k] = cfg[k][j]
  return m


def config1(nz=100):
  """Single Column + Psi_Thermwind updated every step (examples/example_timestepping.py
  :18-80 physics at nz=100, 1000 steps).  The script's dt=60 d is for its nz=70 grid;
  at nz=100 kappa*dt/dz^2 = 0.91 and the reference itself NaNs, so dt=30 d here."""
  z = np.linspace(-3500., 0., nz)
  bs, bbot = 0.03, -0.0004
  kappa = 1e-5 + 3e-5 * np.exp(z / 100) + 3e-4 * np.exp(-z / 1000 - 4)
  return dict(z=z, kappa=kappa, Area=8e13, bs=bs, bbot=bbot,
              b0=bs * np.exp(z / 300.) + bbot, f=1.2e-4, dt=30 * DAY, nsteps=1000)


def config2(N=1024, nz=100, seed=20240, members=None):
  """N independent Columns with a prescribed, static upwelling profile."""
  rng = np.random.default_rng(seed)
  kappa_back = _logu(rng, 5e-6, 5e-5, N)
  area = _logu(rng, 1e13, 3e14, N)
  bs = rng.uniform(0.01, 0.04, N)
  w0 = rng.uniform(-3e-8, 3e-8, N)
  sl = _slice(members, N)
  kappa_back, area, bs, w0 = kappa_back[sl], area[sl], bs[sl], w0[sl]
  idx = np.arange(N)[sl]
  n = idx.size
  z = np.linspace(-4000., 0., nz)
  bbot = -0.003
  kappa_profile = 2e-4 * np.exp(-z / 1000 - 4)
  kappa = kappa_back[:, None] + kappa_profile[None, :]
  Area = np.repeat(area[:, None], nz, axis=1)
  wA = area[:, None] * w0[:, None] * np.sin(np.pi * z / 4000.)[None, :]
  b0 = (bs[:, None] - bbot) * np.exp(z / 300.)[None, :] + bbot
  return dict(z=z, kappa=kappa, kappa_back=kappa_back, kappa_profile=kappa_profile, Area=Area,
              wA=wA, b0=b0, bs=bs, bbot=np.full(n, bbot), N2min=np.full(n, 1e-7),
              do_conv=(idx % 2 == 1), dt=30 * DAY, nsteps=1000, members=idx)


def twocol_member(nz=100, kappa_back=1e-5, kappa_4k=3e-4, A_basin=8e13, area_ratio=100.,
                  bs=0.03, bs_north=0.0, bbot=-0.003, f=1.2e-4, depth=4000.):
  """One `examples/example_twocol.py` member (physics of :16-83) with ARRAY initial
  profiles (callable profiles change the first thermal-wind solve, SURVEY hazard H7)."""
  z = np.linspace(-depth, 0., nz)
  kappa = kappa_back + kappa_4k * np.exp(-z / 1000 - 4)
  return dict(z=z, kappa=kappa, A_basin=A_basin, A_north=A_basin / area_ratio, bs=bs,
              bs_north=bs_north, bbot=bbot, f=f, b_basin0=bs * np.exp(z / 300.),
              b_north0=1e-3 * bs * np.exp(z / 300.), dt=30 * DAY, MOC_up_iters=24, nb=500)


def config3(N=4096, nz=100, seed=20241, members=None):
  """N two-column members (example_twocol.py physics), parameter sweep.

  kappa_4k is drawn from logU[1e-4, 2.5e-4] (SURVEY proposed up to 1e-3, which violates
  kappa*dt/dz^2 <= 1/2 at nz=100, dt=30 d and blows up in the reference too)."""
  rng = np.random.default_rng(seed)
  kappa_back = _logu(rng, 5e-6, 5e-5, N)
  kappa_4k = _logu(rng, 1e-4, 2.5e-4, N)
  A_basin = _logu(rng, 4e13, 1.6e14, N)
  bs = rng.uniform(0.02, 0.04, N)
  sl = _slice(members, N)
  kappa_back, kappa_4k, A_basin, bs = kappa_back[sl], kappa_4k[sl], A_basin[sl], bs[sl]
  idx = np.arange(N)[sl]
  n = idx.size
  z = np.linspace(-4000., 0., nz)
  kappa = kappa_back[:, None] + kappa_4k[:, None] * np.exp(-z / 1000 - 4)[None, :]
  prof = np.exp(z / 300.)
  return dict(z=z, kappa=kappa, A_basin=A_basin, A_north=A_basin / 100., bs=bs,
              bs_north=np.zeros(n), bbot=np.full(n, -0.003), f=1.2e-4,
              b_basin0=bs[:, None] * prof[None, :],
              b_north0=1e-3 * bs[:, None] * prof[None, :], dt=30 * DAY, MOC_up_iters=24,
              nb=500, nsteps=2400, members=idx,
              scalars=dict(kappa_back=kappa_back, kappa_4k=kappa_4k))


def twocol_so_member(nz=100, ny=40, kappa=2e-5, tau=0.13, KGM=1000., A_basin=6e13,
                     bs=0.03, bs_north=0.004, bmin=0.0, c=0.1):
  """One `examples/example_twocol_plusSO.py` member (physics of :17-96), array ICs."""
  z = np.linspace(-4000., 0., nz)
  y = np.linspace(0., 2.e6, ny)
  return dict(z=z, y=y, kappa=kappa + 0 * z, A_basin=A_basin, A_north=A_basin / 50.,
              bs=bs, bs_north=bs_north, bbot=bmin, f=1e-4, L=5e6, KGM=KGM, c=c,
              bvp_with_Ek=True, tau=tau, bs_SO=(bs - bmin) * (y / y[-1])**2 + bmin,
              b_basin0=bs * np.exp(z / 300.), b_north0=bs_north * np.exp(z / 300.),
              dt=30 * DAY, MOC_up_iters=24, nb=500)


def config4(N=8192, nz=100, ny=40, seed=20242, members=None):
  """N two-column + SO-channel members (example_twocol_plusSO.py physics)."""
  rng = np.random.default_rng(seed)
  kappa = _logu(rng, 1e-5, 5e-5, N)
  tau = rng.uniform(0.05, 0.2, N)
  KGM = rng.uniform(500., 1500., N)
  # SURVEY proposed A_basin down to 3e13; below ~3.7e13 the northern column (A/50)
  # violates the advective CFL limit at nz=100, dt=30 d and the reference itself NaNs.
  A_basin = _logu(rng, 4.5e13, 1.2e14, N)
  sl = _slice(members, N)
  kappa, tau, KGM, A_basin = kappa[sl], tau[sl], KGM[sl], A_basin[sl]
  idx = np.arange(N)[sl]
  n = idx.size
  base = twocol_so_member(nz=nz, ny=ny)
  out = dict(base)
  out.update(kappa=np.repeat(kappa[:, None], nz, axis=1), tau=tau, KGM=KGM,
             A_basin=A_basin, A_north=A_basin / 50., bs=np.full(n, base['bs']),
             bs_north=np.full(n, base['bs_north']), bbot=np.full(n, base['bbot']),
             bs_SO=np.repeat(base['bs_SO'][None, :], n, axis=0),
             b_basin0=np.repeat(base['b_basin0'][None, :], n, axis=0),
             b_north0=np.repeat(base['b_north0'][None, :], n, axis=0), nsteps=2400,
             Diag_iters=240, members=idx)
  return out


# GCM diffusivity profile of examples/run_JansenNadeau_2018.py:97-113 (model input data)
_KAPGCM = np.array([
    1.2e-4, 0.882e-4, 0.544e-4, 0.393e-4, 0.305e-4, 0.235e-4, 0.207e-4, 0.210e-4,
    0.213e-4, 0.216e-4, 0.220e-4, 0.226e-4, 0.247e-4, 0.316e-4, 0.377e-4, 0.407e-4,
    0.389e-4, 0.407e-4, 0.454e-4, 0.517e-4, 0.633e-4, 0.757e-4, 0.899e-4, 1.056e-4,
    1.246e-4, 1.584e-4, 1.884e-4, 2.053e-4, 2.168e-4, 2.332e-4
])
_ZGCM = -1e3 * np.array([
    0.0, 0.0200, 0.045, 0.075, 0.110, 0.150, 0.200, 0.260, 0.330, 0.410, 0.500, 0.600,
    0.720, 0.860, 1.020, 1.200, 1.400, 1.600, 1.800, 2.000, 2.200, 2.400, 2.600, 2.800,
    3.000, 3.200, 3.400, 3.600, 3.800, 4.000
])


def jn2018_kappa(z):
  return np.interp(-z, -_ZGCM, _KAPGCM)


def jn2018_kappaeff(z):
  return np.interp(-z, -_ZGCM, _KAPGCM) * (1. - np.maximum(-4000. - z + 500., 0.) / 500.)**2


def jn2018_member(nz=81, ny=51, dt_days=30., db=0.0, B=5.9e3, kapGM=800., tau=0.12):
  """One `examples/run_JansenNadeau_2018.py` member with default flags (physics of
  :33-190).  nz=200 needs dt <= ~19 d (explicit-scheme limit)."""
  bs = 0.02 + db
  bs_north = -0.001 + db
  bminSO = 0.0 + db
  h, L = 50., 4e6
  Bloss = B / L / 2e5
  l = 2.e6
  y = np.linspace(0, l, ny)
  bs_SO_eq = 0. * y + bminSO
  alpha = (1. - np.cos(np.pi * (l - y[5]) / 7.4e6))
  bs_SO_eq[6:] = (bs - bminSO) * (1. - np.cos(np.pi * (y[6:] - y[5]) / 7.4e6)) / alpha + bminSO
  surflux = 0. * y
  surflux[1:6] = -Bloss
  rest_mask = 0. * y
  rest_mask[6:-1] = 1.
  A_basin = 8e13
  dt = DAY * dt_days
  z = np.linspace(-4000., 0., nz)
  b_basin = bs * np.exp(z / 300.) + bs_north * z / z[0]
  b_north = bs_north * (z / z[0])**2.
  bs_SO = bs_SO_eq.copy()
  bs_SO[-1] = bs  # run_JansenNadeau_2018.py:152 (after the initial PsiSO.solve())
  return dict(z=z, y=y, kappa=jn2018_kappa(z), kappaeff=jn2018_kappaeff(z),
              A_basin=A_basin, A_north=A_basin / 50., bs=bs, bs_north=bs_north, h=h, L=L,
              Ks=400., KGM=kapGM, v_pist=1.5 / DAY, tau=tau, f=1.2e-4, surflux=surflux,
              rest_mask=rest_mask, b_rest=bs_SO_eq, bs_SO_init=bs_SO_eq.copy(),
              bs_SO0=bs_SO, b_basin0=b_basin, b_north0=b_north, dt=dt,
              MOC_up_iters=int(np.floor(1. * 360. * DAY / dt)), nb=500)


def single_basin_kappa(z, kapfac=1.0):
  """`kappa` of examples/run_single_global_basin.py:98-99."""
  return kapfac * 9e-6 * np.exp(-z / 1200) + 7e-5 * np.exp(z / 50)


def single_basin_kappaeff(z, kapfac=1.0):
  """`kappaeff` (BBL taper) of examples/run_single_global_basin.py:101-103."""
  return (kapfac * 9e-6 * np.exp(-z / 1200) + 7e-5 * np.exp(z / 50)) * (
      1. - np.maximum(-4500. - z + 500., 0.) / 500.)**2


def single_basin_member(nz=46, ny=51, dt_days=30., bs=0.025, B=5.0e4, Ks=1.0e3, KGM=1.0e3,
                        tau=0.12, kapfac=1.0):
  """One `examples/run_single_global_basin.py` member (physics of :38-170, default flags):
  the JN2018 loop with global-ocean parameters.  Same keys as `jn2018_member`, so
  `JN2018Ensemble` and `oracle.drivers.run_jn2018` run it unchanged."""
  bs_north, bminSO = 0.0, 0.0
  h, L = 50., 2e7
  Bloss = B / L / 2e5
  l = 2.e6
  y = np.linspace(0, l, ny)
  bs_SO_eq = 0. * y + bminSO
  alpha = (1. - np.cos(np.pi * (l - y[5]) / 7.4e6))
  bs_SO_eq[6:] = (bs - bminSO) * (1. - np.cos(np.pi * (y[6:] - y[5]) / 7.4e6)) / alpha + bminSO
  surflux = 0. * y
  surflux[1:6] = -Bloss
  rest_mask = 0. * y
  rest_mask[6:-1] = 1.
  A_basin = 3.2e14
  dt = DAY * dt_days
  z = np.linspace(-4500., 0., nz)
  b_basin = bs * np.exp(z / 400.) - 0.0001 * z / z[0]
  b_north = bs_north - 0.0001 * (z / z[0])**2.
  bs_SO_init = bs_SO_eq.copy()
  bs_SO_init[:6] = -0.0001
  bs_SO = bs_SO_init.copy()
  bs_SO[-1] = bs  # run_single_global_basin.py:137 (after the initial PsiSO.solve())
  return dict(z=z, y=y, kappa=single_basin_kappa(z, kapfac),
              kappaeff=single_basin_kappaeff(z, kapfac), kapfac=kapfac,
              A_basin=A_basin, A_north=A_basin / 100., bs=bs, bs_north=bs_north, h=h, L=L,
              Ks=Ks, KGM=KGM, v_pist=1.5 / DAY, tau=tau, f=1.2e-4, surflux=surflux,
              rest_mask=rest_mask, b_rest=bs_SO_eq, bs_SO_init=bs_SO_init,
              bs_SO0=bs_SO, b_basin0=b_basin, b_north0=b_north, dt=dt,
              MOC_up_iters=int(np.floor(2. * 360. * DAY / dt)), nb=500)


def config5(N=4096, nz=200, ny=51, dt_days=10., seed=20243, members=None):
  """N run_JansenNadeau_2018 members at nz=200, dt=10 d; sweep db, B, kapGM, tau.

  db is drawn from U[0, 8e-4] (SURVEY proposed up to 4e-3): from db = 1e-3 on the
  script's cold-start profiles have bs_north >= 0 and the reference itself blows up
  (checked with the oracle, which tracks the reference to 1e-15 on this config)."""
  rng = np.random.default_rng(seed)
  db = rng.uniform(0., 8e-4, N)
  B = rng.uniform(3e3, 9e3, N)
  kapGM = rng.uniform(600., 1000., N)
  tau = rng.uniform(0.08, 0.16, N)
  sl = _slice(members, N)
  idx = np.arange(N)[sl]
  mem = [jn2018_member(nz, ny, dt_days, db[i], B[i], kapGM[i], tau[i]) for i in idx]
  out = dict(mem[0]) if mem else jn2018_member(nz, ny, dt_days)
  for key in ('bs', 'bs_north', 'KGM', 'tau'):
    out[key] = np.array([m[key] for m in mem])
  for key in ('surflux', 'b_rest', 'bs_SO_init', 'bs_SO0', 'b_basin0', 'b_north0'):
    out[key] = np.stack([m[key] for m in mem]) if mem else np.zeros((0,))
  out.update(nsteps=3600, members=idx, scalars=dict(db=db[sl], B=B[sl]))
  return out


def twobasin_kappaeff(z):
  """Effective diffusivity of examples/twobasin_NadeauJansen.py:36-38."""
  return 1.0 * (1e-4 * (1.1 - np.tanh(np.maximum(z + 2000., 0) / 1000. +
                                      np.minimum(z + 2000., 0) / 1300.)) *
                (1. - np.maximum(-4000. - z + 600., 0.) / 600.)**2)


def twobasin_member(nz=80, ny=51, tau=0.16, K=1800., A_Atl=7e13, A_north=5.5e12, A_Pac=1.7e14):
  """One `examples/twobasin_NadeauJansen.py` member (physics of :21-97): Atlantic, northern
  sinking region and Pacific columns, AMOC and zonal thermal-wind overturnings, one SO
  overturning per basin sector.  Array initial profiles (SURVEY hazard H7)."""
  bs, bs_north, bAABW = 0.02, 0.00036, -0.0011
  bbot = min(bAABW, bs_north)
  y = np.asarray(np.linspace(0, 3.e6, ny))
  offset = 0.0345 * (1 - np.cos(np.pi * (5.55e5 - 1.e5) / 8e6))
  bs_SO = (0.0345 * (1 - np.cos(np.pi * (y - 1.e5) / 8e6)) * (y > 5.55e5) +
           (bAABW - offset) / 5.55e5 * np.maximum(0, 5.55e5 - y) + offset * (y < 5.55e5))
  Lx = 1.3e+07
  z = np.asarray(np.linspace(-4000, 0, nz))
  b_Atl = bs * np.exp(z / 300.) + z / z[0] * bbot
  return dict(z=z, y=y, kappa=twobasin_kappaeff(z), A_Atl=A_Atl, A_north=A_north,
              A_Pac=A_Pac, bs=bs, bs_north=bs_north, bbot=bbot, N2min=2e-7, tau=tau, K=K,
              L_Atl=6. / 21. * Lx, L_Pac=15. / 21. * Lx, f_AMOC=1.2e-4, f_ZOC=1e-4,
              f_SO=1.2e-4, bs_SO=bs_SO, b_Atl0=b_Atl, b_north0=b_Atl.copy(),
              b_Pac0=b_Atl.copy(), b2_init=0.01 * b_Atl, dt=DAY * 30., MOC_up_iters=24, nb=500)


def config_twobasin(N=2048, nz=80, ny=51, seed=20244, members=None):
  """N two-basin members; sweep tau, K, A_Pac."""
  rng = np.random.default_rng(seed)
  tau = rng.uniform(0.1, 0.2, N)
  K = rng.uniform(1200., 2400., N)
  A_Pac = _logu(rng, 1.2e14, 2.2e14, N)
  sl = _slice(members, N)
  idx = np.arange(N)[sl]
  n = idx.size
  out = twobasin_member(nz=nz, ny=ny)
  out.update(tau=tau[sl], K=K[sl], A_Pac=A_Pac[sl], A_Atl=np.full(n, out['A_Atl']),
             A_north=np.full(n, out['A_north']), nsteps=2400, members=idx)
  return out


# ----------------------------------------------------- equilibrium iteration
def iteration_kappa(z, kappa_back=1e-5, kappa_s=3e-5, kappa_4k=3e-4):
  """`kappa` of examples/example_iteration.py:24-26."""
  return kappa_back + kappa_s * np.exp(z / 100) + kappa_4k * np.exp(-z / 1000 - 4)


def iteration_b_basin(z, bs=0.03, bbot=-0.0004):
  """`b_basin` of examples/example_iteration.py:33-34 (a callable initial profile)."""
  return bs * np.exp(z / 300.) + bbot


def iteration_member(nz=100, bs=0.03, bbot=-0.0004, A_basin=8e13, kappa_4k=3e-4, f=1.2e-4):
  """examples/example_iteration.py:13-59: one basin column whose equilibrium profile
  (`Column.solve_equi`) and thermal-wind overturning against b_N = 0 are iterated with
  under-relaxation 0.2."""
  z = np.linspace(-3500, 0, nz)
  return dict(z=z, bs=bs, bbot=bbot, A_basin=A_basin, kappa_4k=kappa_4k, f=f,
              kappa=iteration_kappa(z, kappa_4k=kappa_4k),
              b_basin0=bs * np.exp(z / 300.) + bbot, keep=0.8, relax=0.2, niter=30)


def config_iteration(N=256, nz=100, seed=20245, members=None):
  """Ensemble of `iteration_member`s: kappa_4k in logU[1e-4, 6e-4], A_basin in
  U[5e13, 1.2e14], bs in U[0.02, 0.04]; kappa is SAMPLED on z (array profile)."""
  rng = np.random.default_rng(seed)
  k4 = _logu(rng, 1e-4, 6e-4, N)
  A = rng.uniform(5e13, 1.2e14, N)
  bs = rng.uniform(0.02, 0.04, N)
  sl = _slice(members, N)
  z = np.linspace(-3500, 0, nz)
  bbot = -0.0004
  return dict(z=z, bs=bs[sl], bbot=bbot, A_basin=A[sl], kappa_4k=k4[sl], f=1.2e-4,
              kappa=np.array([iteration_kappa(z, kappa_4k=k) for k in k4[sl]]),
              b_basin0=bs[sl][:, None] * np.exp(z / 300.)[None, :] + bbot, keep=0.8, relax=0.2, niter=30)


# ------------------------------------------------------------- Equi_Column
def equi_column_cases():
  """Named `Equi_Column` problems (profiles as numbers / arrays on z, the forms the device
  solves): the two example scripts (examples/example_Equi_diffusive_thermocline.py:11-21,
  examples/example_Equi_Bint.py:16-60 with its callables sampled on a 161-level grid) and the
  constructor configurations of the reference's tests (tests/modules/test_equi_column.py:13-112)."""
  z80 = np.linspace(-4000, 0, 80)
  z161 = np.linspace(-4000, 0, 161)
  a = 6.37e6
  A = 2 * np.pi * a**2 * 59 / 360 * (np.sin(np.radians(69)) - np.sin(np.radians(-48)))
  kap = iteration_kappa(z161)
  cases = {
      "thermocline": dict(A=1.0e14, b_bot=0.0, b_s=0.02, f=1e-4, kappa=5.0e-5, H=4000.0, nz=200),
      "Hfree_const": dict(B_int=3e3, A=2.0e14, kappa=3e-5),
      "H500": dict(z=z80, B_int=3e3, A=2.0e14, kappa=3e-5, H=500.0),
      "H500_kappa_arr": dict(z=z80, B_int=3e3, A=2.0e14, kappa=np.linspace(3e-5, 1e-5, 80),
                             H=500.0),
      "H500_bs": dict(z=z80, A=2.0e14, kappa=3e-5, H=500.0, b_s=0.05),
      "bbot_fixedH": dict(z=z80, A=1.0e14, kappa=5e-5, H=3000.0, B_int=None, b_bot=0.002,
                          b_s=0.02, f=1e-4),
  }
  for i, (Hm, B) in enumerate(((2000, 3e3), (2000, 1.2e4), (1500, 3e3), (1500, 1.2e4))):
    pso = 4e6 * np.sin(-np.pi * np.maximum(z161, -Hm) / Hm)**2
    cases["Bint%d" % i] = dict(B_int=B, A=A, kappa=kap, psi_so=pso, z=z161)
  return cases


def equi_bint_callable_case(i):
  """Case i of examples/example_Equi_Bint.py:16-60 with its CALLABLE kappa, dkappa_dz and
  psi_so (the drop-in class tabulates them; the oracle calls them like the reference)."""
  Hm, B = ((2000, 3e3), (2000, 1.2e4), (1500, 3e3), (1500, 1.2e4))[i]
  a = 6.37e6
  A = 2 * np.pi * a**2 * 59 / 360 * (np.sin(np.radians(69)) - np.sin(np.radians(-48)))
  kappa = lambda z: 1e-5 + 3e-5 * np.exp(z / 100) + 3e-4 * np.exp(-z / 1000 - 4)
  dkappa_dz = lambda z: 3e-5 / 100 * np.exp(z / 100) - 3e-4 / 1000 * np.exp(-z / 1000 - 4)
  psi_so = lambda z: 4. * 1e6 * np.sin(-np.pi * np.maximum(z, -Hm) / Hm)**2
  return dict(B_int=B, A=A, kappa=kappa, dkappa_dz=dkappa_dz, psi_so=psi_so)


# ----------------------------------------- callable surface profiles for Psi_SO (hazard H7)
def so_bs_callable(y, bs=0.03, bmin=0.0, l=2.e6):
  """Surface buoyancy across the channel as a FUNCTION of y (example_twocol_plusSO.py:57-58)."""
  return (bs - bmin) * (y / l)**2 + bmin


def so_tau_callable(y, tau0=0.13, l=2.e6):
  """A wind-stress profile peaking mid-channel."""
  return tau0 * (0.3 + 0.7 * np.sin(np.pi * y / l)**2)
