"""ColumnEquiBatch: `Column.solve_equi` for an ensemble, on the GPU.

Arithmetic contract: src/pymoc/modules/column.py:187-208 -- the steady state of the column's
advection-diffusion equation, `y1' = y2, y2' = (wA - dAkappa_dz)/Akappa * y2` (:161-164)
with `b(-H) = bbot` or `b'(-H) = bzbot`, `b(0) = bs` (:124-159), handed to
`scipy.integrate.solve_bvp(ode, bc, z, [b, bz])` with default `tol=1e-3, max_nodes=1000`.

solve_bvp (scipy 1.15.3, integrate/_bvp.py) = [solve the collocation system on the mesh;
estimate the rms residual of every interval; insert 1 node where tol < rms < 100 tol and 2
where rms >= 100 tol] until nothing is inserted.  The device does the first two for every
member at once (`pm_column_equi_pass`, exact because the ODE is linear); this module is
solve_bvp's outer loop: it owns the meshes, inserts the nodes, and evaluates the column's
coefficient functions on new meshes with the reference's own NumPy expressions
(`Akappa(x)`, `np.gradient(Akappa(x), x)`), the same way the static `d(A kappa)/dz` of the
time-stepping path is prepared.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib, pm_column_equi
from .device import DeviceArray, _sh

_S37 = (3 / 7)**0.5
MAX_MESH = 1024  # rows of the device mesh arrays (solve_bvp's default max_nodes is 1000)


def point_sets(x):
  """The four point sets solve_bvp evaluates the ODE on for mesh x: nodes, interval
  middles, and the two interior Lobatto points of the residual estimate."""
  h = np.diff(x)
  xm = x[:-1] + 0.5 * h
  s = 0.5 * h * _S37
  return x, xm, xm + s, xm - s


def insert_nodes(x, rms, tol):
  """_bvp.py:solve_bvp / modify_mesh: the refined mesh for residuals `rms`."""
  ins1, = np.nonzero((rms > tol) & (rms < 100 * tol))
  ins2, = np.nonzero(rms >= 100 * tol)
  return np.sort(np.hstack((x, 0.5 * (x[ins1] + x[ins1 + 1]),
                            (2 * x[ins2] + x[ins2 + 1]) / 3,
                            (x[ins2] + 2 * x[ins2 + 1]) / 3)))


def _roundup(v, q=64):
  return int(-(-int(v) // q) * q)


class _Tables(object):
  """Host + device copies of per-member meshes and coefficient tables, rows of `mmax`."""

  def __init__(self, n, nz, mmax, with_wA):
    self.n, self.nz, self.mmax = n, nz, mmax
    self.m = np.zeros(n, np.int32)
    self.x = np.zeros((n, mmax))
    self.Ak = np.ones((4, n, mmax))
    self.dAk = np.zeros((4, n, mmax))
    self.wA = np.zeros((4, n, mmax)) if with_wA else None
    self.zidx = np.zeros((n, nz), np.int32)
    self.dev = None

  def grown(self, mmax):
    t = _Tables(self.n, self.nz, mmax, self.wA is not None)
    t.m[:] = self.m
    t.zidx[:] = self.zidx
    for name in ("x", "Ak", "dAk", "wA"):
      src = getattr(self, name)
      if src is not None:
        getattr(t, name)[..., :self.mmax] = src
    return t

  def upload(self, stream=None):
    self.dev = {k: DeviceArray.from_host(getattr(self, k), stream=stream)
                for k in ("m", "x", "Ak", "dAk", "zidx")}
    if self.wA is not None:
      self.dev["wA"] = DeviceArray.from_host(self.wA, stream=stream)


class ColumnEquiBatch(object):
  """n columns on one grid z.

  akappa(i, x), dakappa_dz(i, x): `Column.Akappa` / `Column.dAkappa_dz` of member i at the
  points x (NumPy in, NumPy out).  `from_profiles` builds them for array / scalar profiles
  the way `make_func` does (np.interp on z)."""

  def __init__(self, z, n, akappa, dakappa_dz, bs, bbot=0.0, bzbot=None, tol=1e-3,
               max_nodes=1000, stream=None, z_dev=None):
    _lib.require_device()
    self.z_host = np.ascontiguousarray(z, dtype=np.float64)
    self.nz = nz = self.z_host.size
    if nz < 3:
      raise ValueError('solve_equi needs at least 3 levels')
    if max(max_nodes, nz) > MAX_MESH:
      raise ValueError('meshes are limited to %d nodes' % MAX_MESH)
    self.n, self.tol, self.max_nodes = int(n), float(tol), int(max_nodes)
    self.akappa, self.dakappa_dz = akappa, dakappa_dz
    self.stream = stream
    self.z = z_dev if z_dev is not None else DeviceArray.from_host(self.z_host)
    bz_given = bzbot is not None
    self.bs = DeviceArray.from_host(np.broadcast_to(np.asarray(bs, np.float64), (n,)))
    self.bbot = DeviceArray.from_host(np.broadcast_to(np.asarray(bbot, np.float64), (n,)))
    self.bzbot = DeviceArray.from_host(np.broadcast_to(
        np.asarray(bzbot if bz_given else 0.0, np.float64), (n,)))
    self.flags = DeviceArray.from_host(np.full(n, _lib.PM_COL_BZBOT if bz_given else 0,
                                               np.int32))
    self.b = DeviceArray.zeros((n, nz))
    self.bz = DeviceArray.zeros((n, nz))
    self.nadd = DeviceArray.zeros((n,), np.int32)
    self.nodes = np.full(n, nz, np.int32)
    self.status = np.zeros(n, np.int32)  # solve_bvp's status: 0 converged, 1 max_nodes hit
    self.passes = 0
    self._base = {}  # with_wA -> _Tables on the column grid itself (every solve starts here)

  @classmethod
  def from_profiles(cls, z, kappa, Area, bs, bbot=0.0, bzbot=None, **kw):
    """kappa, Area: scalar, [nz] or [n][nz] samples on z (interpolated linearly between
    levels, as `make_func` does for array arguments, utils/make_func.py:30-45)."""
    z = np.asarray(z, dtype=np.float64)
    kap, are = np.asarray(kappa, np.float64), np.asarray(Area, np.float64)
    n = max([a.shape[0] for a in (kap, are) if a.ndim == 2] +
            [np.size(v) for v in (bs, bbot) if np.ndim(v) == 1] + [kw.pop('n', 1)])

    def prof(a):
      if a.ndim == 0:
        return lambda i, x: a + 0. * x
      if a.ndim == 1:
        return lambda i, x: np.interp(x, z, a)
      return lambda i, x: np.interp(x, z, a[i])

    kf, af = prof(kap), prof(are)
    ak = lambda i, x: af(i, x) * kf(i, x)  # column.py:94
    dak = lambda i, x: np.gradient(ak(i, x), x)  # column.py:122
    return cls(z, n, ak, dak, bs, bbot, bzbot, **kw)

  # ---- tables
  def _fill(self, t, i, x, wA_fn=None):
    m = x.size
    t.m[i] = m
    t.x[i, :m] = x
    t.zidx[i] = np.searchsorted(x, self.z_host)
    for s, pts in enumerate(point_sets(x)):
      t.Ak[s, i, :pts.size] = self.akappa(i, pts)
      t.dAk[s, i, :pts.size] = self.dakappa_dz(i, pts)
      if wA_fn is not None:
        t.wA[s, i, :pts.size] = wA_fn(pts)

  def _base_tables(self, with_wA):
    t = self._base.get(with_wA)
    if t is None:
      t = _Tables(self.n, self.nz, _roundup(self.nz), with_wA)
      for i in range(self.n):
        self._fill(t, i, self.z_host)
      t.upload(self.stream)
      self._base[with_wA] = t
    return t

  # ---- solve_bvp's loop
  def solve(self, wA, keep_mesh=False):
    """wA: [n, nz] DeviceArray / device pointer / ndarray on the column grid (interpolated
    linearly, `make_func(wA, z, 'w')`), or a sequence of n callables evaluated wherever
    solve_bvp asks (the reference accepts both, column.py:201)."""
    n, nz = self.n, self.nz
    fns = None
    if isinstance(wA, (list, tuple)) and len(wA) == n and all(callable(f) for f in wA):
      fns = list(wA)
    elif callable(wA):
      fns = [wA] * n
    wA_z = None
    if fns is None:
      if isinstance(wA, np.ndarray) or np.isscalar(wA):
        self._wA_up = DeviceArray.from_host(np.broadcast_to(np.asarray(wA, np.float64), (n, nz)))
        wA_z = self._wA_up.ptr
      else:
        wA_z = wA.ptr if isinstance(wA, DeviceArray) else int(wA)
    t = self._base_tables(fns is not None)
    if fns is not None:  # the base mesh's wA tables depend on this call's functions
      for i in range(n):
        for s, pts in enumerate(point_sets(self.z_host)):
          t.wA[s, i, :pts.size] = fns[i](pts)
      t.dev["wA"].upload(t.wA, self.stream)
    active = np.ones(n, np.int32)
    self.status[:] = 0
    self.nodes[:] = nz
    self.passes = 0
    self.mesh = None
    while True:
      act_dev = DeviceArray.from_host(active, stream=self.stream)
      rms = DeviceArray((n, t.mmax))
      ymesh = DeviceArray((n, 2, t.mmax)) if keep_mesh else None
      d = pm_column_equi()
      d.n, d.nz, d.mmax, d.reserved = n, nz, t.mmax, 0
      d.m, d.active, d.x = t.dev["m"].ptr, act_dev.ptr, t.dev["x"].ptr
      d.Ak, d.dAk = t.dev["Ak"].ptr, t.dev["dAk"].ptr
      d.wA = t.dev["wA"].ptr if fns is not None else None
      d.wA_z, d.z = wA_z, self.z.ptr
      d.bs, d.bbot, d.bzbot, d.flags = self.bs.ptr, self.bbot.ptr, self.bzbot.ptr, self.flags.ptr
      d.zidx, d.tol = t.dev["zidx"].ptr, self.tol
      d.y = ymesh.ptr if keep_mesh else None
      d.rms, d.nadd, d.b, d.bz = rms.ptr, self.nadd.ptr, self.b.ptr, self.bz.ptr
      check(lib.pm_column_equi_pass(C.byref(d), _sh(self.stream)))
      self.passes += 1
      nadd = self.nadd.download(stream=self.stream)
      want = (active != 0) & (nadd > 0)
      over = want & (t.m + nadd > self.max_nodes)  # _bvp.py: status 1, keep this solution
      self.status[over] = 1
      refine = want & ~over
      if keep_mesh:
        self.mesh = (t.m.copy(), t.x.copy(), ymesh.download(stream=self.stream))
      if not refine.any():
        break
      rms_h = rms.download(stream=self.stream)
      mmax = _roundup(int((t.m + np.where(refine, nadd, 0)).max()))
      if t is self._base.get(fns is not None) or mmax > t.mmax:
        t = t.grown(max(mmax, t.mmax))  # never edit the cached tables of the column grid
      for i in np.nonzero(refine)[0]:
        x = insert_nodes(t.x[i, :t.m[i]], rms_h[i, :t.m[i] - 1], self.tol)
        self._fill(t, i, x, fns[i] if fns is not None else None)
        self.nodes[i] = x.size
      t.upload(self.stream)
      active = refine.astype(np.int32)
    return self

  def get_b(self):
    return self.b.download(stream=self.stream)

  def get_bz(self):
    return self.bz.download(stream=self.stream)
