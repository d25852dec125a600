"""CPU ORACLE for Equi_Column.solve -- TEST INFRASTRUCTURE, NOT PRODUCT.

The reference (src/pymoc/modules/equi_column.py) states a boundary-value problem and hands
it to `scipy.integrate.solve_bvp`, a third-party dependency (scipy 1.15.3 in this image) that
is not part of the reference checkout.  This oracle restates the PROBLEM -- the
non-dimensional profile closures (:116-185), `alpha` (:231-249), `bz` (:251-284), `bc`
(:286-347), `ode` (:349-406), the default initial guess (:187-213) and the output scaling
(:424-435) -- as plain functions of a parameter dict, and calls the same SciPy routine on it.
Pinned against the reference's own outputs (tests/golden/equi_column.npz, set G13).
"""
import numpy as np
from scipy import integrate


def problem(f=1.2e-4, b_s=0.025, b_bot=None, B_int=3e3, A=7e13, nz=100, H_guess=1500.,
            kappa=6e-5, psi_so=None, z=None, H=None, dkappa_dz=None):
  """Parameter dict of one equilibrium column (profiles: numbers, arrays on z or callables)."""
  return dict(f=f, b_s=b_s, b_bot=b_bot, B_int=B_int, A=A, nz=nz, H_guess=H_guess,
              kappa=kappa, psi_so=psi_so, z=z, H=H, dkappa_dz=dkappa_dz)


def _profiles(q):
  f, z = q['f'], q['z']
  kap, pso, dkz = q['kappa'], q['psi_so'], q.get('dkappa_dz')
  if callable(kap):  # equi_column.py:125-128, :146-149
    kappa = lambda x, H: kap(x * H) / (H**2 * f)
    if callable(dkz):
      dkappa = lambda x, H: dkz(x * H) / (H * f)
    else:
      dkappa = lambda x, H: np.gradient(kap(x * H), x * H) / (H * f)
  elif isinstance(kap, np.ndarray):
    dk = np.gradient(kap, z)
    kappa = lambda x, H: np.interp(x * H, z, kap) / (H**2 * f)
    dkappa = lambda x, H: np.interp(x * H, z, dk) / (H * f)
  else:
    kappa = lambda x, H: kap / (H**2 * f)
    dkappa = lambda x, H: 0
  if callable(pso):
    psi = lambda x, H: pso(x * H) / (f * H**3)
  elif isinstance(pso, np.ndarray):
    psi = lambda x, H: np.interp(x * H, z, pso) / (f * H**3)
  else:
    psi = lambda x, H: 0
  return kappa, dkappa, psi


def solve(q, tol=1e-3, max_nodes=1000):
  """-> dict(x, y, H, status, niter, sol): what Equi_Column.solve obtains from solve_bvp."""
  f, A = q['f'], q['A']
  kappa, dkappa, psi = _profiles(q)
  bs = -q['b_s'] / f**2
  has_bbot = q['b_bot'] is not None
  bbot = -q['b_bot'] / f**2 if has_bbot else None
  bz = lambda H: q['B_int'] / (f**3 * H**2 * A * kappa(-1, H))
  hfree = q['H'] is None

  def depth(p):
    return p[0] if hfree else q['H']

  def ode(x, y, p=None):
    H = depth(p)
    alpha = H**2 / (A * kappa(x, H))
    return np.vstack((y[1], y[2], y[3],
                      alpha * y[3] * (y[0] - psi(x, H) - A * dkappa(x, H) / (H**2))))

  def bc(ya, yb, p=None):
    H = depth(p)
    r = [ya[0], yb[0]] + ([ya[1]] if hfree else [])
    r.append(ya[2] - bbot / H if has_bbot else ya[3] + bz(H))
    r.append(yb[2] - bs / H)
    return np.array(r)

  nz = q['nz']
  zi = np.linspace(-1, 0, nz)
  y0 = np.zeros((4, nz))
  y0[0] = 1.0
  y0[3] = -100.0 if has_bbot else -bz(1500.)
  res = integrate.solve_bvp(ode, bc, zi, y0, p=[q['H_guess']] if hfree else None, tol=tol,
                            max_nodes=max_nodes)
  H = res.p[0] if hfree else q['H']
  return dict(x=res.x, y=res.y, yp=res.yp, H=H, status=res.status, niter=res.niter,
              sol=res.sol)


def outputs(q, r):
  """(z, psi, b) as Equi_Column.solve leaves them (equi_column.py:424-435)."""
  f, H = q['f'], r['H']
  if q['z'] is None:
    return r['x'] * H, r['y'][0] * f * H**3 / 1e6, -r['y'][2] * f**2 * H
  z = q['z']
  s = r['sol'](z / H)
  psi = s[0] * f * H**3 / 1e6
  b = -s[2] * f**2 * H
  psi[z < -H] = np.nan
  b[z < -H] = np.nan
  return z, psi, b
