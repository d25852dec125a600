"""CPU ORACLE drivers -- TEST INFRASTRUCTURE, NOT PRODUCT.

Member-by-member coupled time loops built from the C oracle functions, following the
loop structure of the reference's example scripts (which are the reference's only
"driver"):
  run_config1    examples/example_timestepping.py:73-80
  run_twocol     examples/example_twocol.py:85-96
  run_twocol_so  examples/example_twocol_plusSO.py:99-115
  run_jn2018     examples/run_JansenNadeau_2018.py:201-261 (default flags)
  run_iteration  examples/example_iteration.py:59-68
Each returns {step: {field: array}} snapshots taken AFTER the given 1-based step count.
"""
import numpy as np

from . import (column_timestep, thermwind_solve, thermwind_psibz, psi_so_solve,
               so_ml_advdiff, column_solve_equi)


def _snap(store, step, snaps, **fields):
  if step in snaps:
    store[step] = {k: np.array(v, copy=True) for k, v in fields.items()}


def run_config1(cfg, nsteps, snaps):
  z, kap = cfg['z'], cfg['kappa']
  area = cfg['Area'] + 0 * z
  b = cfg['b0'].copy()
  b2 = 0. * z  # Psi_Thermwind default b2=0. (psi_thermwind.py:36)
  Psi = thermwind_solve(z, b, b2, cfg['f'])
  out = {}
  snaps = set(snaps)
  for ii in range(nsteps):
    b = column_timestep(z, kap, area, b, Psi * 1e6, cfg['dt'], bs=cfg['bs'],
                        bbot=cfg['bbot'])
    Psi = thermwind_solve(z, b, b2, cfg['f'])
    _snap(out, ii + 1, snaps, b=b, Psi=Psi)
  return out


def run_twocol(m, nsteps, snaps, so=False, bvp_refine=0):
  """m: a `configs.twocol_member` / `twocol_so_member` style dict of one member."""
  z = m['z']
  nz = z.size
  kap = m['kappa'] + 0 * z
  Ab, An = m['A_basin'] + 0 * z, m['A_north'] + 0 * z
  bb, bn = m['b_basin0'].copy(), m['b_north0'].copy()
  nb, dt, M = m['nb'], m['dt'], m['MOC_up_iters']
  Psi = thermwind_solve(z, bb, bn, m['f'])
  bgrid, psib, pib, pin = thermwind_psibz(bb, bn, Psi, nb)
  if so:
    sokw = dict(f=m['f'], L=m['L'], KGM=m['KGM'], c=m['c'], bvp_with_Ek=m['bvp_with_Ek'],
                bvp_refine=bvp_refine)
    PsiSO, Ek, GM, _ = psi_so_solve(z, m['y'], bb, m['bs_SO'], m['tau'], **sokw)
  else:
    PsiSO = np.zeros(nz)
  out = {}
  snaps = set(snaps)
  for ii in range(nsteps):
    wAb = (pib - PsiSO) * 1e6 if so else pib * 1e6
    wAN = -pin * 1e6
    bb = column_timestep(z, kap, Ab, bb, wAb, dt, bs=m['bs'], bbot=m['bbot'])
    bn = column_timestep(z, kap, An, bn, wAN, dt, do_conv=True, bs=m['bs_north'],
                         bbot=m['bbot'])
    if ii % M == 0:
      Psi = thermwind_solve(z, bb, bn, m['f'])
      bgrid, psib, pib, pin = thermwind_psibz(bb, bn, Psi, nb)
      if so:
        PsiSO, Ek, GM, _ = psi_so_solve(z, m['y'], bb, m['bs_SO'], m['tau'], **sokw)
    _snap(out, ii + 1, snaps, b_basin=bb, b_north=bn, Psi=Psi, Psi_iso_b=pib,
          Psi_iso_n=pin, Psi_SO=PsiSO)
  return out


def run_jn2018(m, nsteps, snaps, dense_inverse=False):
  z, y = m['z'], m['y']
  Ab, An = m['A_basin'] + 0 * z, m['A_north'] + 0 * z
  kap, kapeff = m['kappa'], m['kappaeff']
  bb, bn, bsSO = m['b_basin0'].copy(), m['b_north0'].copy(), m['bs_SO0'].copy()
  nb, dt, M = m['nb'], m['dt'], m['MOC_up_iters']
  sokw = dict(f=m['f'], L=m['L'], KGM=m['KGM'])
  # Column(..., kappa=kappaeff, bbot=b[0]) (run_JansenNadeau_2018.py:159-171)
  bbot_b, bbot_n = bb[0], bn[0]
  kap_b, kap_n = kapeff, kapeff
  out = {}
  snaps = set(snaps)
  Psi = pib = pin = PsiSO = None
  for ii in range(nsteps):
    if ii % M == 0:
      Psi = thermwind_solve(z, bb, bn, m['f'])
      bgrid, psib, pib, pin = thermwind_psibz(bb, bn, Psi, nb)
      PsiSO, Ek, GM, _ = psi_so_solve(z, y, bb, bsSO, m['tau'], **sokw)
    wAb = (pib - PsiSO) * 1e6
    wAN = -pin * 1e6
    # bottom BC / BBL diffusivity switching, run_JansenNadeau_2018.py:233-254
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
    bb = column_timestep(z, kap_b, Ab, bb, wAb, dt, do_conv=True, bs=m['bs'], bbot=bbot_b)
    bn = column_timestep(z, kap_n, An, bn, wAN, dt, do_conv=True, bs=m['bs_north'],
                         bbot=bbot_n)
    bsSO, Psi_s = so_ml_advdiff(y, m['surflux'], m['rest_mask'], m['b_rest'], bsSO, bb,
                                PsiSO, dt, Ks=m['Ks'], h=m['h'], L=m['L'],
                                v_pist=m['v_pist'], dense_inverse=dense_inverse)
    _snap(out, ii + 1, snaps, b_basin=bb, b_north=bn, bs_SO=bsSO, Psi=Psi, Psi_SO=PsiSO,
          Psi_iso_b=pib, Psi_iso_n=pin, Psi_s=Psi_s)
  return out


def run_twobasin(m, nsteps, snaps):
  """examples/twobasin_NadeauJansen.py:99-122: three columns, AMOC + zonal thermal-wind
  overturnings, one SO overturning per sector.  m: `configs.twobasin_member` dict."""
  z, y = m['z'], m['y']
  kap = m['kappa']
  A = {k: m['A_' + k] + 0 * z for k in ('Atl', 'north', 'Pac')}
  bA, bN, bP = m['b_Atl0'].copy(), m['b_north0'].copy(), m['b_Pac0'].copy()
  nb, dt, M = m['nb'], m['dt'], m['MOC_up_iters']
  so = dict(f=m['f_SO'], KGM=m['K'])
  Psi_A = thermwind_solve(z, bA, m['b2_init'], m['f_AMOC'])
  _, _, iso_A, iso_N = thermwind_psibz(bA, m['b2_init'], Psi_A, nb)
  Psi_Z = thermwind_solve(z, bA, bP, m['f_ZOC'])
  _, _, zon_A, zon_P = thermwind_psibz(bA, bP, Psi_Z, nb)
  SO_A = psi_so_solve(z, y, bA, m['bs_SO'], m['tau'], L=m['L_Atl'], **so)[0]
  SO_P = psi_so_solve(z, y, bP, m['bs_SO'], m['tau'], L=m['L_Pac'], **so)[0]
  out = {}
  snaps = set(snaps)
  kw = dict(bbot=m['bbot'], N2min=m['N2min'])
  for ii in range(nsteps):
    wA_Atl = (iso_A + zon_A - SO_A) * 1e6
    wAN = -iso_N * 1e6
    wA_Pac = (-zon_P - SO_P) * 1e6
    bA = column_timestep(z, kap, A['Atl'], bA, wA_Atl, dt, bs=m['bs'], **kw)
    bN = column_timestep(z, kap, A['north'], bN, wAN, dt, do_conv=True, bs=m['bs_north'], **kw)
    bP = column_timestep(z, kap, A['Pac'], bP, wA_Pac, dt, bs=m['bs'], **kw)
    if ii % M == 0:
      Psi_A = thermwind_solve(z, bA, bN, m['f_AMOC'])
      _, _, iso_A, iso_N = thermwind_psibz(bA, bN, Psi_A, nb)
      Psi_Z = thermwind_solve(z, bA, bP, m['f_ZOC'])
      _, _, zon_A, zon_P = thermwind_psibz(bA, bP, Psi_Z, nb)
      SO_A = psi_so_solve(z, y, bA, m['bs_SO'], m['tau'], L=m['L_Atl'], **so)[0]
      SO_P = psi_so_solve(z, y, bP, m['bs_SO'], m['tau'], L=m['L_Pac'], **so)[0]
    _snap(out, ii + 1, snaps, b_Atl=bA, b_north=bN, b_Pac=bP, Psi_AMOC=Psi_A, Psi_ZOC=Psi_Z,
          Psi_SO_Atl=SO_A, Psi_SO_Pac=SO_P)
  return out


def equi_coef(z, kappa, Area, wA):
  """c(x) of Column.ode (column.py:161-164) for array / scalar / callable profiles, built the
  way the reference builds its callables (make_func: np.interp for arrays)."""
  def fn(a):
    if callable(a):
      return a
    a = np.asarray(a, dtype=float)
    return (lambda x: a + 0. * x) if a.ndim == 0 else (lambda x: np.interp(x, z, a))
  kf, af, wf = fn(kappa), fn(Area), fn(wA)
  ak = lambda x: af(x) * kf(x)
  return lambda x: (wf(x) - np.gradient(ak(x), x)) / ak(x)


def run_iteration(m, niter, snaps, kappa=None):
  """m: `configs.iteration_member` dict; kappa: callable profile (default: m['kappa'])."""
  z = m['z']
  kap = m['kappa'] if kappa is None else kappa
  b1 = m['b_basin0'].copy()
  b2 = 0. * z
  Psi = thermwind_solve(z, b1, b2, m['f'])
  out = {}
  snaps = set(snaps)
  for ii in range(niter):
    b, bz, _, _ = column_solve_equi(z, equi_coef(z, kap, m['A_basin'], Psi * 1e6), m['bs'],
                                    m['bbot'])
    b1 = m['keep'] * b1 + m['relax'] * b  # example_iteration.py:67
    Psi = thermwind_solve(z, b1, b2, m['f'])
    _snap(out, ii + 1, snaps, b=b, bz=bz, Psi=Psi, b1=b1)
  return out
