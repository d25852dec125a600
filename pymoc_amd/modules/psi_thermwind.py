"""Drop-in `Psi_Thermwind` with the reference's Python API, computing on the GPU.

Same constructor, attributes (`Psi`, `bgrid`, `b1`, `b2`, `f`, `z`, `sol_init`), methods
and error text as `pymoc.modules.Psi_Thermwind` (src/pymoc/modules/psi_thermwind.py:7-232).
Array and float profiles behave exactly like the reference.  Profiles given as CALLABLES are
evaluated where SciPy's solve_bvp evaluates them -- on the mesh, at the collocation midpoints
and at the Lobatto points of its residual estimate -- and `solve()` follows solve_bvp's own mesh
refinement (host: mesh and callables; device: collocation solve and residuals of every mesh,
`_solve_callable`), so it agrees with the reference to 1e-13 also where solve_bvp inserts
nodes (SURVEY hazard H7, golden G14); in `Psib` they are sampled on `z`, exactly as the
reference does.
"""
import numpy as np

from .. import _lib
from ..device import DeviceArray
from ..utils import make_func, make_array


class Psi_Thermwind(object):
  def __init__(
      self,
      f=1.2e-4,
      z=None,
      sol_init=None,
      b1=None,
      b2=0.,
  ):
    self.f = f
    if isinstance(z, np.ndarray):
      self.z = z
      nz = np.size(z)
    else:
      raise TypeError('z needs to be numpy array providing grid levels')
    self.b1 = make_func(b1, self.z, 'b1')
    self.b2 = make_func(b2, self.z, 'b2')
    self._b1_callable, self._b2_callable = callable(b1), callable(b2)
    self.sol_init = np.zeros((2, nz)) if sol_init is None else sol_init
    self._batch = None
    self._nb = 0
    self._nz = 0

  # ---- device plumbing: one arena, one H2D and one D2H per call
  # arena (float64 slots): in  [b1 | b2 | Psi_in | f | b1_mid | b2_mid]
  #                        out [Psi | psibz1 | psibz2 | bgrid | psib]
  def _run(self, ops, nb, need_psi):
    import ctypes as C
    from .. import _lib
    from .column import flush_all
    flush_all()  # b1 / b2 may alias the array of a Column with queued steps
    nz = np.size(self.z)
    if self._batch is None or self._nb < nb or self._nz != nz:
      self._nb, self._nz = max(int(nb), 1), nz
      self._zd = DeviceArray.from_host(np.ascontiguousarray(self.z, dtype=np.float64))
      self._nin = 5 * nz + 1
      self._arena = DeviceArray((self._nin + 3 * nz + 2 * self._nb,))
      self._host = np.zeros(self._nin)
      self._batch = True
    h, p = self._host, self._arena.ptr
    h[0:nz] = np.asarray(make_array(self.b1, self.z, 'b1'), dtype=np.float64) + 0 * self.z
    h[nz:2 * nz] = np.asarray(make_array(self.b2, self.z, 'b2'), dtype=np.float64) + 0 * self.z
    if need_psi:
      h[2 * nz:3 * nz] = self.Psi
    h[3 * nz] = self.f
    mid = (self._b1_callable or self._b2_callable) and bool(ops & _lib.PM_TW_SOLVE)
    if mid:  # x_middle of scipy's collocation_fun
      zm = self.z[:-1] + 0.5 * (self.z[1:] - self.z[:-1])
      h[3 * nz + 1:4 * nz] = self.b1(zm)
      h[4 * nz + 1:5 * nz] = self.b2(zm)
    _lib.check(_lib.lib.pm_memcpy_h2d(p, h.ctypes.data, h.nbytes, None))
    o = p + self._nin * 8
    d = _lib.pm_thermwind()
    d.n, d.nz, d.nb, d.reserved = 1, nz, int(nb), 0
    d.z, d.b1, d.b2, d.f = self._zd.ptr, p, p + nz * 8, p + 3 * nz * 8
    # without PM_TW_SOLVE the kernel reads Psi: point it at the uploaded copy
    d.Psi = o if (ops & _lib.PM_TW_SOLVE) else p + 2 * nz * 8
    d.psibz1, d.psibz2 = o + nz * 8, o + 2 * nz * 8
    d.bgrid, d.psib = o + 3 * nz * 8, o + (3 * nz + self._nb) * 8
    d.Psi_SO, d.wA1, d.wA2 = None, None, None
    d.b1_mid = p + (3 * nz + 1) * 8 if mid else None
    d.b2_mid = p + (4 * nz + 1) * 8 if mid else None
    _lib.check(_lib.lib.pm_thermwind_update(C.byref(d), int(ops), None))
    out = np.empty(3 * nz + 2 * self._nb)
    _lib.check(_lib.lib.pm_memcpy_d2h(out.ctypes.data, o, out.nbytes, None))
    return out

  # ---- the boundary-value problem the reference hands to solve_bvp (psi_thermwind.py:72-123);
  # host NumPy, kept for callers that inspect or reuse them (the device solve does not call them)
  def bc(self, ya, yb):
    """Boundary residuals: Psi vanishes at the bottom (ya) and at the surface (yb)."""
    return np.array([ya[0], yb[0]])

  def ode(self, z, y):
    """First-order form of Psi'' = (b2 - b1)/f with y = (Psi, Psi'): rows (y[1], rhs)."""
    from .column import flush_all
    flush_all()  # b1 / b2 may alias the array of a Column with queued steps
    inv_f = 1. / self.f
    return np.vstack((y[1], inv_f * (self.b2(z) - self.b1(z))))

  def solve(self):
    from .. import _lib
    if self._b1_callable or self._b2_callable:
      self.Psi = self._solve_callable()
      return
    out = self._run(_lib.PM_TW_SOLVE, 1, False)
    self.Psi = out[:np.size(self.z)].copy()

  def _solve_callable(self, tol=1e-3, max_nodes=1000):
    """solve() for CALLABLE profiles: scipy.integrate.solve_bvp's own loop (scipy 1.15.3
    _bvp.py, defaults tol = 1e-3, max_nodes = 1000, as psi_thermwind.py:132 calls it).  The
    host owns the mesh and samples the callables where solve_bvp does (nodes, mid-points, the
    inner Lobatto points of the residual estimate); the collocation solve and the rms
    residuals of every mesh run on the device (pm_thermwind_update on the mesh,
    pm_thermwind_residuals); nodes are inserted by solve_bvp's rule until no interval asks
    for one.  The column levels stay nodes, so Psi on z is the nodal solution there."""
    import ctypes as C
    from .. import _lib
    from .column import flush_all
    flush_all()
    z = np.ascontiguousarray(self.z, dtype=np.float64)
    x = z.copy()
    s37 = 0.5 * (3. / 7.) ** 0.5
    rf = 1. / self.f
    cap = 1024
    if z.size > cap:
      raise ValueError("Psi_Thermwind.solve with callable profiles follows solve_bvp's mesh "
                       "refinement on meshes of up to %d nodes; z has %d levels" % (cap, z.size))
    if getattr(self, "_mesh_arena", None) is None:
      # in : x | b1 | b2 | b1_mid | b2_mid | g | g_lob[2] | f      out: Psi | dPsi | rms
      self._mesh_arena = DeviceArray((12 * cap + 8,))
    p = self._mesh_arena.ptr
    while True:
      m = x.size
      h = x[1:] - x[:-1]
      xm = x[:-1] + 0.5 * h
      b1x, b2x = self.b1(x) + 0 * x, self.b2(x) + 0 * x
      host = np.zeros(9 * cap + 8)
      host[0:m] = x
      host[cap:cap + m] = b1x
      host[2 * cap:2 * cap + m] = b2x
      host[3 * cap:3 * cap + m - 1] = self.b1(xm)
      host[4 * cap:4 * cap + m - 1] = self.b2(xm)
      host[5 * cap:5 * cap + m] = rf * (b2x - b1x)  # the ode's right-hand side, :123
      for side, sg in ((0, 1.), (1, -1.)):
        xl = xm + sg * s37 * h
        host[6 * cap + side * (m - 1):6 * cap + (side + 1) * (m - 1)] = rf * (self.b2(xl) - self.b1(xl))
      host[8 * cap] = self.f
      _lib.check(_lib.lib.pm_memcpy_h2d(p, host.ctypes.data, host.nbytes, None))
      off = lambda k: p + 8 * k
      d = _lib.pm_thermwind()
      d.n, d.nz, d.nb, d.reserved = 1, m, 1, 0
      d.z, d.b1, d.b2, d.f = off(0), off(cap), off(2 * cap), off(8 * cap)
      d.b1_mid, d.b2_mid = off(3 * cap), off(4 * cap)
      d.Psi, d.dPsi = off(9 * cap + 8), off(10 * cap + 8)
      _lib.check(_lib.lib.pm_thermwind_update(C.byref(d), _lib.PM_TW_SOLVE, None))
      _lib.check(_lib.lib.pm_thermwind_residuals(m, off(0), d.Psi, d.dPsi, off(5 * cap),
                                                 off(6 * cap), off(11 * cap + 8), None))
      out = np.empty(3 * cap)
      _lib.check(_lib.lib.pm_memcpy_d2h(out.ctypes.data, off(9 * cap + 8), out.nbytes, None))
      psi_x, rms = out[:m], out[2 * cap:2 * cap + m - 1]
      insert_1, = np.nonzero((rms > tol) & (rms < 100 * tol))
      insert_2, = np.nonzero(rms >= 100 * tol)
      nodes_added = insert_1.shape[0] + 2 * insert_2.shape[0]
      if nodes_added == 0 or m + nodes_added > max_nodes or m + nodes_added > cap:
        break
      # modify_mesh (_bvp.py): original nodes are kept
      x = np.sort(np.hstack((x, 0.5 * (x[insert_1] + x[insert_1 + 1]),
                             (2 * x[insert_2] + x[insert_2 + 1]) / 3,
                             (x[insert_2] + 2 * x[insert_2 + 1]) / 3)))
    self.mesh_nodes = int(x.size)
    return psi_x[np.searchsorted(x, z)].copy()

  def Psib(self, nb=500):
    from .. import _lib
    nz = np.size(self.z)
    out = self._run(_lib.PM_TW_PSIB, nb, True)
    self.bgrid = out[3 * nz:3 * nz + nb].copy()
    return out[3 * nz + self._nb:3 * nz + self._nb + nb].copy()

  def Psibz(self, nb=500):
    from .. import _lib
    nz = np.size(self.z)
    out = self._run(_lib.PM_TW_PSIB | _lib.PM_TW_PSIBZ, nb, True)
    self.bgrid = out[3 * nz:3 * nz + nb].copy()
    return [out[nz:2 * nz].copy(), out[2 * nz:3 * nz].copy()]

  def update(self, b1=None, b2=None):
    if b1 is not None:
      self.b1 = make_func(b1, self.z, 'b1')
      self._b1_callable = callable(b1)
    if b2 is not None:
      self.b2 = make_func(b2, self.z, 'b2')
      self._b2_callable = callable(b2)

