"""EquiColumnBatch: `Equi_Column.solve` for an ensemble, on the GPU.

Arithmetic contract: src/pymoc/modules/equi_column.py:408-435 -- the non-dimensional
equilibrium overturning problem (ode :349-406, bc :286-347) solved by
`scipy.integrate.solve_bvp(ode, bc, zi, sol_init, p=[H_guess] | None)` with SciPy's defaults
(tol = bc_tol = 1e-3, max_nodes = 1000).

The device runs one mesh iteration of solve_bvp for all members at once
(`pm_equi_column_newton`: SciPy's damped Newton with forward-difference Jacobians, the rms
residual estimate and the node-insertion count).  This module is solve_bvp's outer loop
(scipy 1.15.3 `_bvp.py:solve_bvp`): it owns the per-member meshes, inserts nodes, carries
the solution to a refined mesh through the C1 cubic spline of `create_spline`, and stops
with solve_bvp's status codes (0 converged, 1 max_nodes, 2 singular Jacobian, 3 boundary
residual not met after 10 iterations).

Profiles (`kappa`, `psi_so`) are scalars or samples on a grid `z`, exactly the cases the
reference turns into np.interp closures; callables cannot run on the device (they would have
to be called with the unknown depth H inside Newton's iteration): the drop-in class tabulates
them on a fine grid first (`pymoc_amd.modules.Equi_Column`).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib, pm_equi_column
from .device import DeviceArray, _sh
from .equilibrium import insert_nodes, _roundup

MAX_ITERATION = 10  # _bvp.py:solve_bvp


def spline_coefficients(x, y, yp):
  """_bvp.py:create_spline: cubic Hermite pieces (c0 s^3 + c1 s^2 + c2 s + c3), s = x - x_i."""
  h = np.diff(x)
  slope = (y[:, 1:] - y[:, :-1]) / h
  t = (yp[:, :-1] + yp[:, 1:] - 2 * slope) / h
  return t / h, (slope - yp[:, :-1]) / h - t, yp[:, :-1], y[:, :-1]


def spline_eval(x, y, yp, xq):
  """`create_spline(y, yp, x, h)(xq)` with PPoly's evaluation order and extrapolation."""
  c0, c1, c2, c3 = spline_coefficients(x, y, yp)
  idx = np.clip(np.searchsorted(x, xq, side='right') - 1, 0, x.size - 2)
  s = xq - x[idx]
  s2 = s * s
  return c3[:, idx] + c2[:, idx] * s + c1[:, idx] * s2 + c0[:, idx] * (s2 * s)


class EquiColumnBatch(object):
  """n independent `Equi_Column` problems.

  f, A, b_s, b_bot / B_int, H / H_guess, kappa: scalars or [n]; `kappa` may also be
  [nzg] / [n, nzg] samples on `z` (then `dkappa_dz = np.gradient(kappa, z)`, as
  equi_column.py:150-152); `psi_so`: None or [nzg] / [n, nzg] samples on `z`."""

  def __init__(self, n, f=1.2e-4, b_s=0.025, b_bot=None, B_int=3e3, A=7e13, nz=100,
               sol_init=None, H_guess=1500., kappa=6e-5, psi_so=None, z=None, H=None,
               tol=1e-3, max_nodes=1000, stream=None, dkappa_dz=None):
    _lib.require_device()
    if b_bot is None and B_int is None:
      raise Exception('You need to specify either b_bot or B_int for bottom boundary condition')
    self.n, self.nz, self.tol, self.max_nodes = int(n), int(nz), float(tol), int(max_nodes)
    self.stream = stream
    vec = lambda v: np.broadcast_to(np.asarray(v, np.float64), (n,)).copy()
    self.f, self.A = vec(f), vec(A)
    self.bs = -vec(b_s) / self.f**2  # equi_column.py:227
    self.has_bbot = b_bot is not None
    self.bb = -vec(b_bot) / self.f**2 if self.has_bbot else vec(B_int)
    self.hfree = H is None
    self.H0 = vec(H_guess if self.hfree else H)
    self.zg = None if z is None else np.ascontiguousarray(z, np.float64)
    kap = np.asarray(kappa, np.float64)
    # [n, nzg] or [nzg] samples on z are profiles; a scalar or [n] values are constants
    self.kappa_arr = kap.ndim == 2 or (kap.ndim == 1 and self.zg is not None and
                                       kap.shape[0] == self.zg.size)
    if self.kappa_arr:
      if self.zg is None:
        raise ValueError('array kappa needs the grid z')
      self.kappa_z = np.broadcast_to(kap, (n, self.zg.size)).copy()
      if dkappa_dz is not None:  # samples of a given derivative (callable profiles, tabulated)
        self.dkappa_z = np.broadcast_to(np.asarray(dkappa_dz, np.float64),
                                        (n, self.zg.size)).copy()
      else:
        self.dkappa_z = np.array([np.gradient(k, self.zg) for k in self.kappa_z])
      self.kappa = np.zeros(n)
    else:
      self.kappa = vec(kap)
      self.kappa_z = self.dkappa_z = None
    self.psi_arr = isinstance(psi_so, np.ndarray)
    if self.psi_arr:
      if self.zg is None:
        raise ValueError('array psi_so needs the grid z')
      self.psi_z = np.broadcast_to(np.asarray(psi_so, np.float64), (n, self.zg.size)).copy()
    else:
      self.psi_z = None
    self.zi = np.linspace(-1, 0, nz)  # equi_column.py:89-91
    if sol_init is None:
      b_init = np.full(n, -100.0) if self.has_bbot else -self._bz(np.full(n, 1500.))
      y0 = np.zeros((n, 4, nz))
      y0[:, 0, :] = 1.0
      y0[:, 3, :] = b_init[:, None]
    else:
      y0 = np.broadcast_to(np.asarray(sol_init, np.float64), (n, 4, nz)).copy()
    self.y0 = y0
    self.flags = np.full(n, (_lib.PM_EQ_HFREE if self.hfree else 0) |
                         (_lib.PM_EQ_HAS_BBOT if self.has_bbot else 0) |
                         (_lib.PM_EQ_KAPPA_ARRAY if self.kappa_arr else 0) |
                         (_lib.PM_EQ_PSI_ARRAY if self.psi_arr else 0), np.int32)
    self.x = self.y = self.yp = None
    self.H = self.H0.copy()
    self.status = np.zeros(n, np.int32)
    self.niter = np.zeros(n, np.int32)
    self.newton_iters = np.zeros(n, np.int32)

  def _kappa_nd(self, zstar, H):
    """kappa(z, H) of equi_column.py:116-132 for every member (z* scalar)."""
    if self.kappa_arr:
      k = np.array([np.interp(zstar * H[i], self.zg, self.kappa_z[i]) for i in range(self.n)])
    else:
      k = self.kappa
    return k / (H**2 * self.f)

  def _bz(self, H):
    """equi_column.py:251-284."""
    return self.bb / (self.f**3 * H**2 * self.A * self._kappa_nd(-1, H))

  # ---- solve_bvp's outer loop
  def solve(self):
    n = self.n
    m = np.full(n, self.nz, np.int32)
    xs = [self.zi.copy() for _ in range(n)]
    ys = [self.y0[i].copy() for i in range(n)]
    yps = [None] * n
    H = self.H0.copy()
    active = np.ones(n, bool)
    self.status[:] = 0
    self.niter[:] = 0
    self.newton_iters[:] = 0
    const = {k: DeviceArray.from_host(v, stream=self.stream) for k, v in dict(
        f=self.f, A=self.A, bs=self.bs, bb=self.bb, kappa=self.kappa, flags=self.flags).items()}
    if self.zg is not None:
      const["zg"] = DeviceArray.from_host(self.zg, stream=self.stream)
      for k in ("kappa_z", "dkappa_z", "psi_z"):
        if getattr(self, k) is not None:
          const[k] = DeviceArray.from_host(getattr(self, k), stream=self.stream)
    mmax, bufs = 0, None
    iteration = 0
    while active.any():
      need = _roundup(int(m[active].max()))
      if need > mmax:
        mmax = need
        per = int(lib.pm_equi_column_scratch_doubles(mmax))
        bufs = dict(x=DeviceArray((n, mmax)), y=DeviceArray((n, 4, mmax)),
                    yp=DeviceArray((n, 4, mmax)), rms=DeviceArray((n, mmax)),
                    scratch=DeviceArray((n, per)), p=DeviceArray((n,)),
                    m=DeviceArray((n,), np.int32), active=DeviceArray((n,), np.int32),
                    nadd=DeviceArray((n,), np.int32), status=DeviceArray((n,), np.int32),
                    niter=DeviceArray((n,), np.int32), info=DeviceArray((n, 2)))
      xh, yh = np.zeros((n, mmax)), np.zeros((n, 4, mmax))
      for i in np.nonzero(active)[0]:
        xh[i, :m[i]] = xs[i]
        yh[i, :, :m[i]] = ys[i]
      bufs["x"].upload(xh, self.stream)
      bufs["y"].upload(yh, self.stream)
      bufs["p"].upload(H, self.stream)
      bufs["m"].upload(m, self.stream)
      bufs["active"].upload(active.astype(np.int32), self.stream)
      d = pm_equi_column()
      d.n, d.nzg, d.mmax, d.reserved = n, 0 if self.zg is None else self.zg.size, mmax, 0
      d.m, d.active, d.x = bufs["m"].ptr, bufs["active"].ptr, bufs["x"].ptr
      d.y, d.yp, d.p = bufs["y"].ptr, bufs["yp"].ptr, bufs["p"].ptr
      d.f, d.A, d.bs, d.bb = const["f"].ptr, const["A"].ptr, const["bs"].ptr, const["bb"].ptr
      d.kappa, d.flags = const["kappa"].ptr, const["flags"].ptr
      d.zg = const["zg"].ptr if "zg" in const else None
      d.kappa_z = const["kappa_z"].ptr if "kappa_z" in const else None
      d.dkappa_z = const["dkappa_z"].ptr if "dkappa_z" in const else None
      d.psi_z = const["psi_z"].ptr if "psi_z" in const else None
      d.tol = self.tol
      d.rms, d.nadd, d.status = bufs["rms"].ptr, bufs["nadd"].ptr, bufs["status"].ptr
      d.niter, d.info, d.scratch = bufs["niter"].ptr, bufs["info"].ptr, bufs["scratch"].ptr
      check(lib.pm_equi_column_newton(C.byref(d), _sh(self.stream)))
      iteration += 1
      nadd = bufs["nadd"].download(stream=self.stream)
      sing = bufs["status"].download(stream=self.stream)
      info = bufs["info"].download(stream=self.stream)
      H = bufs["p"].download(stream=self.stream)
      yh = bufs["y"].download(stream=self.stream)
      yph = bufs["yp"].download(stream=self.stream)
      rms = bufs["rms"].download(stream=self.stream)
      self.newton_iters[active] += bufs["niter"].download(stream=self.stream)[active]
      for i in np.nonzero(active)[0]:
        mi = m[i]
        ys[i], yps[i] = yh[i, :, :mi].copy(), yph[i, :, :mi].copy()
        self.niter[i] = iteration
        if sing[i] == 2:  # _bvp.py: status 2
          self.status[i], active[i] = 2, False
        elif mi + nadd[i] > self.max_nodes:
          self.status[i], active[i] = 1, False
        elif nadd[i] > 0:
          xn = insert_nodes(xs[i], rms[i, :mi - 1], self.tol)
          ys[i] = spline_eval(xs[i], ys[i], yps[i], xn)  # y = sol(x)
          xs[i], m[i] = xn, xn.size
        elif info[i, 1] <= self.tol:
          self.status[i], active[i] = 0, False
        elif iteration >= MAX_ITERATION:
          self.status[i], active[i] = 3, False
    self.x, self.y, self.yp, self.H = xs, ys, yps, H
    return self

  def sol(self, i, xq):
    """`res.sol(xq)` of member i."""
    return spline_eval(self.x[i], self.y[i], self.yp[i], np.asarray(xq, np.float64))
