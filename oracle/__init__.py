"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT.

ctypes front-end of ``oracle/libpymoc_oracle.so`` (built from ``pymoc_oracle.c`` by
``make oracle`` / ``__graft_entry__.build()``): a plain-C restatement of the PyMOC
``timestep()`` path and of the NumPy/SciPy primitives under it.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this package; nothing under ``pymoc_amd/`` does.

Parity status: pinned -- see ``tests/test_oracle_golden.py`` and
``tests/golden/make_golden.py``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpymoc_oracle.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force=False):
  src = os.path.join(_HERE, "pymoc_oracle.c")
  hdr = os.path.join(_HERE, "pymoc_oracle.h")
  if (not force and os.path.exists(_SO) and
      os.path.getmtime(_SO) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
    return _SO
  subprocess.check_call([
      "gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-std=c99", "-o", _SO,
      src, "-lm"
  ])
  return _SO


class PsiSOPar(C.Structure):
  _fields_ = [("f", C.c_double), ("rho", C.c_double), ("L", C.c_double),
              ("KGM", C.c_double), ("smax", C.c_double), ("c", C.c_double),
              ("Hsill", C.c_double), ("HEk", C.c_double), ("Htapertop", C.c_double),
              ("Htaperbot", C.c_double), ("tau_scalar", C.c_double),
              ("has_c", C.c_int), ("bvp_with_Ek", C.c_int), ("has_Hsill", C.c_int),
              ("has_HEk", C.c_int), ("has_Htapertop", C.c_int),
              ("has_Htaperbot", C.c_int), ("bvp_refine", C.c_int)]


class SOMLPar(C.Structure):
  _fields_ = [("Ks", C.c_double), ("h", C.c_double), ("L", C.c_double),
              ("v_pist", C.c_double)]


_lib = None


def lib():
  global _lib
  if _lib is None:
    # PYMOC_ORACLE_LIB: another build of pymoc_oracle.c, e.g. `make oracle-asan`'s sanitizer build
    alt = os.environ.get("PYMOC_ORACLE_LIB")
    if alt:
      _lib = C.CDLL(os.path.abspath(alt))
    else:
      build()
      _lib = C.CDLL(_SO)
    _lib.orc_np_sum.restype = C.c_double
    _lib.orc_brentq_interp.restype = C.c_double
    _lib.orc_psi_so_ys.restype = C.c_double
    _lib.orc_so_ml_advdiff.restype = C.c_int
  return _lib


def _a(x):
  return np.ascontiguousarray(x, dtype=np.float64)


def _p(x):
  return x.ctypes.data_as(_dp)


# ------------------------------------------------------------------ primitives
def np_interp(x, xp, fp):
  x, xp, fp = _a(np.atleast_1d(x)), _a(xp), _a(fp)
  out = np.empty_like(x)
  lib().orc_np_interp(_p(x), C.c_int(x.size), _p(xp), _p(fp), C.c_int(xp.size), _p(out))
  return out


def np_gradient(f, x):
  f, x = _a(f), _a(x)
  out = np.empty_like(f)
  lib().orc_np_gradient(_p(f), _p(x), C.c_int(f.size), _p(out))
  return out


def np_sum(a):
  a = _a(a)
  return lib().orc_np_sum(_p(a), C.c_int(a.size))


def np_linspace(start, stop, num):
  out = np.empty(num)
  lib().orc_np_linspace(C.c_double(start), C.c_double(stop), C.c_int(num), _p(out))
  return out


def brentq_interp(xp, fp, target, xa, xb):
  xp, fp = _a(xp), _a(fp)
  st = C.c_int(0)
  r = lib().orc_brentq_interp(_p(xp), _p(fp), C.c_int(xp.size), C.c_double(target),
                              C.c_double(xa), C.c_double(xb), C.byref(st))
  return r, st.value


# ---------------------------------------------------------------------- Column
def column_convect(z, b, bs, N2min):
  z, b = _a(z), _a(b).copy()
  lib().orc_column_convect(_p(z), _p(b), C.c_int(z.size), C.c_double(bs),
                           C.c_double(N2min))
  return b


def column_vertadvdiff(z, kappa, area, b, wA, dt, do_conv=False, bs=0.025, bbot=0.0,
                       bzbot=None):
  z, kappa, area, b, wA = _a(z), _a(kappa), _a(area), _a(b).copy(), _a(wA)
  lib().orc_column_vertadvdiff(
      _p(z), _p(kappa), _p(area), _p(b), C.c_int(z.size), _p(wA), C.c_double(dt),
      C.c_int(bool(do_conv)), C.c_double(bs), C.c_double(bbot),
      C.c_int(bzbot is not None), C.c_double(0.0 if bzbot is None else bzbot))
  return b


def column_horadv(area, b, vdx_in, b_in, dt):
  area, b, vdx_in, b_in = _a(area), _a(b).copy(), _a(vdx_in), _a(b_in)
  lib().orc_column_horadv(_p(area), _p(b), C.c_int(b.size), _p(vdx_in), _p(b_in),
                          C.c_double(dt))
  return b


def column_timestep(z, kappa, area, b, wA, dt, do_conv=False, bs=0.025, bbot=0.0,
                    bzbot=None, N2min=1e-7, vdx_in=None, b_in=None):
  z, kappa, area, b, wA = _a(z), _a(kappa), _a(area), _a(b).copy(), _a(wA)
  if vdx_in is not None:
    vdx_in, b_in = _a(vdx_in), _a(b_in)
  lib().orc_column_timestep(
      _p(z), _p(kappa), _p(area), _p(b), C.c_int(z.size), _p(wA), C.c_double(dt),
      C.c_int(bool(do_conv)), C.c_double(bs), C.c_double(bbot),
      C.c_int(bzbot is not None), C.c_double(0.0 if bzbot is None else bzbot),
      C.c_double(N2min), _p(vdx_in) if vdx_in is not None else None,
      _p(b_in) if vdx_in is not None else None)
  return b


def column_ensemble_steps(z, kappa, area, b, wA, dt, do_conv, bs, bbot, N2min, nsteps):
  """ncols x nsteps of Column.timestep with wA fixed; returns the new b [ncols, nz]."""
  z, kappa, area, b, wA = _a(z), _a(kappa), _a(area), _a(b).copy(), _a(wA)
  ncols, nz = b.shape
  do_conv = np.ascontiguousarray(do_conv, dtype=np.int32)
  bs, bbot, N2min = _a(bs), _a(bbot), _a(N2min)
  lib().orc_column_ensemble_steps(
      _p(z), _p(kappa), _p(area), _p(b), C.c_int(ncols), C.c_int(nz), _p(wA),
      C.c_double(dt), do_conv.ctypes.data_as(_ip), _p(bs), _p(bbot), _p(N2min),
      C.c_int(nsteps))
  return b


# --------------------------------------------------------------- Psi_Thermwind
def thermwind_solve(z, b1, b2, f, b1_mid=None, b2_mid=None):
  """b1_mid / b2_mid: the profiles at the interval midpoints (for callable profiles)."""
  z, b1, b2 = _a(z), _a(b1), _a(b2)
  Psi = np.empty_like(z)
  m1 = None if b1_mid is None else _a(b1_mid)
  m2 = None if b2_mid is None else _a(b2_mid)
  lib().orc_thermwind_solve_mid(_p(z), _p(b1), _p(b2), C.c_int(z.size), C.c_double(f),
                                _p(m1) if m1 is not None else None,
                                _p(m2) if m2 is not None else None, _p(Psi))
  return Psi


def thermwind_psib(b1, b2, Psi, nb=500):
  b1, b2, Psi = _a(b1), _a(b2), _a(Psi)
  bgrid, psib = np.empty(nb), np.empty(nb)
  lib().orc_thermwind_psib(_p(b1), _p(b2), _p(Psi), C.c_int(b1.size), C.c_int(nb),
                           _p(bgrid), _p(psib))
  return bgrid, psib


def thermwind_psibz(b1, b2, Psi, nb=500):
  b1, b2, Psi = _a(b1), _a(b2), _a(Psi)
  bgrid, psib = np.empty(nb), np.empty(nb)
  o1, o2 = np.empty_like(b1), np.empty_like(b1)
  lib().orc_thermwind_psibz(_p(b1), _p(b2), _p(Psi), C.c_int(b1.size), C.c_int(nb),
                            _p(bgrid), _p(psib), _p(o1), _p(o2))
  return bgrid, psib, o1, o2


# ---------------------------------------------------------------------- Psi_SO
def psi_so_par(f=1.2e-4, rho=1030, L=1e7, KGM=1e3, c=None, bvp_with_Ek=False, Hsill=None,
               HEk=None, Htapertop=None, Htaperbot=None, smax=0.01, tau=None,
               bvp_refine=0):
  p = PsiSOPar()
  p.f, p.rho, p.L, p.KGM, p.smax = float(f), float(rho), float(L), float(KGM), float(smax)
  p.has_c, p.c = int(c is not None), float(c or 0.0)
  p.bvp_with_Ek = int(bool(bvp_with_Ek))
  for name, v in (("Hsill", Hsill), ("HEk", HEk), ("Htapertop", Htapertop),
                  ("Htaperbot", Htaperbot)):
    setattr(p, "has_" + name, int(v is not None))
    setattr(p, name, float(v or 0.0))
  p.tau_scalar = float(tau) if (tau is not None and np.isscalar(tau)) else 0.0
  p.bvp_refine = int(bvp_refine)
  return p


def psi_so_ys(y, bs, b):
  y, bs = _a(y), _a(bs)
  st = C.c_int(0)
  r = lib().orc_psi_so_ys(_p(y), _p(bs), C.c_int(y.size), C.c_double(b), C.byref(st))
  return r, st.value


def psi_so_solve(z, y, b, bs, tau, **kw):
  z, y, b, bs = _a(z), _a(y), _a(b), _a(bs)
  tau_arr = None if np.isscalar(tau) else _a(tau)
  par = psi_so_par(tau=tau if np.isscalar(tau) else None, **kw)
  Psi, Ek, GM = np.empty_like(z), np.empty_like(z), np.empty_like(z)
  st = C.c_int(0)
  lib().orc_psi_so_solve(_p(z), C.c_int(z.size), _p(y), C.c_int(y.size), _p(b), _p(bs),
                         _p(tau_arr) if tau_arr is not None else None, C.byref(par),
                         _p(Psi), _p(Ek), _p(GM), C.byref(st))
  return Psi, Ek, GM, st.value


# ----------------------------------------------------------------------- SO_ML
def so_ml_advdiff(y, surflux, rest_mask, b_rest, bs, b_basin, Psi_b, dt, Ks=0., h=50.,
                  L=4e6, v_pist=1.5 / 86400., dense_inverse=False, diffusion=None):
  """diffusion: 'thomas', 'dense' (the reference's inv(U) @ V @ bs), 'apply' (propagator
  U^-1 V built by Thomas, applied by fma accumulation) or 'pcr' (parallel cyclic reduction);
  default: 'dense' when dense_inverse, else what the HIP kernel uses -- 'pcr' for ny <= 64,
  'thomas' beyond."""
  if diffusion is None:
    diffusion = 'dense' if dense_inverse else ('pcr' if np.size(y) <= 64 else 'thomas')
  mode = dict(thomas=0, dense=1, apply=2, pcr=3)[diffusion]
  y, surflux, rest_mask, b_rest = _a(y), _a(surflux), _a(rest_mask), _a(b_rest)
  bs, b_basin, Psi_b = _a(bs).copy(), _a(b_basin), _a(Psi_b)
  par = SOMLPar(float(Ks), float(h), float(L), float(v_pist))
  Psi_s = np.empty_like(y)
  rc = lib().orc_so_ml_advdiff(_p(y), C.c_int(y.size), _p(surflux), _p(rest_mask),
                               _p(b_rest), C.byref(par), _p(bs), _p(Psi_s), _p(b_basin),
                               _p(Psi_b), C.c_int(b_basin.size), C.c_double(dt),
                               C.c_int(mode))
  if rc != 0:
    raise IndexError("index 0 is out of bounds for axis 0 with size 0")
  return bs, Psi_s


# ------------------------------------------------------------------ Column.solve_equi
def column_equi_pass(x, c_n, c_m, c_1, c_2, bs, bbot, bzbot=None):
  """One solve_bvp mesh pass (exact collocation solution + rms residuals) on mesh x."""
  x, c_n, c_m, c_1, c_2 = _a(x), _a(c_n), _a(c_m), _a(c_1), _a(c_2)
  y = np.empty((2, x.size))
  rms = np.empty(x.size - 1)
  lib().orc_column_equi_pass(_p(x), C.c_int(x.size), _p(c_n), _p(c_m), _p(c_1), _p(c_2),
                             C.c_double(bs), C.c_double(bbot), C.c_int(bzbot is not None),
                             C.c_double(0. if bzbot is None else bzbot), _p(y), _p(rms))
  return y, rms


def column_solve_equi(z, coef, bs, bbot, bzbot=None, tol=1e-3, max_nodes=1000):
  """Column.solve_equi (column.py:187-208) = scipy.integrate.solve_bvp's mesh loop
  (scipy 1.15.3 _bvp.py:solve_bvp) around `column_equi_pass`.  `coef(x)` returns
  c(x) = (wA(x) - dAkappa_dz(x)) / Akappa(x) for one array of points, evaluated the way the
  reference's `ode` does (column.py:161-164).  Returns (b, bz, mesh, status)."""
  x = _a(z).copy()
  status, it = 0, 0
  while True:
    h = np.diff(x)
    xm = x[:-1] + 0.5 * h
    s = 0.5 * h * (3 / 7)**0.5
    y, rms = column_equi_pass(x, coef(x), coef(xm), coef(xm + s), coef(xm - s), bs, bbot,
                              bzbot)
    it += 1
    ins1, = np.nonzero((rms > tol) & (rms < 100 * tol))
    ins2, = np.nonzero(rms >= 100 * tol)
    added = ins1.size + 2 * ins2.size
    if x.size + added > max_nodes:
      status = 1
      break
    if added > 0:  # _bvp.py:modify_mesh
      x = np.sort(np.hstack((x, 0.5 * (x[ins1] + x[ins1 + 1]),
                             (2 * x[ins2] + x[ins2 + 1]) / 3,
                             (x[ins2] + 2 * x[ins2 + 1]) / 3)))
    else:
      break  # the linear solve satisfies the boundary conditions exactly
  idx = np.searchsorted(x, _a(z))
  return y[0, idx], y[1, idx], x, status
