"""Drop-in `Column` with the reference's Python API, computing on the GPU.

Same constructor, attributes, method names, argument meaning and error text as
`pymoc.modules.Column` (src/pymoc/modules/column.py:6-348); the arithmetic of
convect / vertadvdiff / horadv / timestep runs in the HIP kernel `pm_column_steps`
on a one-column batch.  `self.b` stays a host NumPy array that is updated IN PLACE
after every call (the reference's aliasing behaviour, column.py:249), so user loops
that read `basin.b` or poke `basin.bbot` / `basin.kappa` between steps run unchanged.
For throughput use `pymoc_amd.ColumnBatch` / the ensemble drivers instead: this
wrapper pays two small PCIe copies per call.
"""
import numpy as np

from .. import _lib
from ..utils import make_func, make_array


class Column(object):
  def __init__(
      self,
      z=None,
      kappa=None,
      bs=0.025,
      bbot=0.0,
      bzbot=None,
      b=0.0,
      Area=None,
      N2min=1e-7
  ):
    if isinstance(z, np.ndarray) and len(z) > 0:
      self.z = z
    else:
      raise TypeError('z needs to be numpy array providing grid levels')
    self.kappa = make_func(kappa, self.z, 'kappa')
    self.Area = make_func(Area, self.z, 'Area')
    self.bs = bs
    self.bbot = bbot
    self.bzbot = bzbot
    self.N2min = N2min
    self.b = make_array(b, self.z, 'b')
    self.bz = np.gradient(self.b, z)
    self._arena = None
    self._kap_cached = None
    self._area_cached = None

  # ---- host-side helpers of the reference API (column.py:74-122)
  def Akappa(self, z):
    return self.Area(z) * self.kappa(z)

  def dAkappa_dz(self, z):
    return np.gradient(self.Akappa(z), z)

  def bc(self, ya, yb):
    """Boundary-condition residuals of the equilibrium problem (column.py:124-159)."""
    if self.bzbot is None:
      return np.array([ya[0] - self.bbot, yb[0] - self.bs])
    else:
      return np.array([ya[1] - self.bzbot, yb[0] - self.bs])

  def ode(self, z, y):
    """Right-hand side of the equilibrium problem (column.py:161-185)."""
    return np.vstack(
        (y[1], (self.wA(z) - self.dAkappa_dz(z)) / self.Akappa(z) * y[1])
    )

  def solve_equi(self, wA):
    """Equilibrium profile for a given wA (column.py:187-208).  The reference calls
    scipy.integrate.solve_bvp; here solve_bvp's collocation solve and residual estimate
    run on the GPU (`pm_column_equi_pass`) inside the same mesh-refinement loop, the
    coefficient functions being sampled wherever that loop asks, as SciPy would."""
    from ..equilibrium import ColumnEquiBatch
    self.wA = make_func(wA, self.z, 'w')
    eq = ColumnEquiBatch(
        self.z, 1, lambda i, x: self.Akappa(x), lambda i, x: self.dAkappa_dz(x), self.bs,
        self.bbot, self.bzbot
    )
    eq.solve(self.wA)
    # like the reference, solve_equi REBINDS b and bz (column.py:207-208)
    self.b = eq.get_b()[0]
    self.bz = eq.get_bz()[0]
    self.equi_nodes = int(eq.nodes[0])

  # ---- device plumbing: one arena, one H2D and one D2H per call
  # arena (float64 slots): [b | wA | vdx_in | b_in | bs bbot bzbot N2min | flags(int32)]
  def _alloc(self):
    import ctypes as C
    from ..device import DeviceArray
    nz = self.z.size
    self._nz = nz
    self._host = np.zeros(4 * nz + 5)
    self._arena = DeviceArray((4 * nz + 5,))
    self._zd = DeviceArray.from_host(np.ascontiguousarray(self.z, dtype=np.float64))
    self._kap = DeviceArray((nz,))
    self._area = DeviceArray((nz,))
    self._dAk = DeviceArray((nz,))
    p, d = self._arena.ptr, _lib.pm_columns()
    d.ncols, d.nz, d.nsel, d.reserved = 1, nz, 1, 0
    d.z, d.b = self._zd.ptr, p
    d.kappa, d.area, d.dAkappa = self._kap.ptr, self._area.ptr, self._dAk.ptr
    sc = p + 4 * nz * 8
    d.bs, d.bbot, d.bzbot, d.N2min, d.flags = sc, sc + 8, sc + 16, sc + 24, sc + 32
    d.ksel, d.nonfinite = None, None
    self._desc = d
    self._wA_ptr, self._vdx_ptr, self._bin_ptr = p + nz * 8, p + 2 * nz * 8, p + 3 * nz * 8
    self._C = C

  def _run(self, ops, do_conv, wA=None, dt=1., vdx_in=None, b_in=None):
    if getattr(self, "_arena", None) is None or self._nz != self.z.size:
      self._alloc()
      self._kap_cached = None
    z, nz, h = self.z, self._nz, self._host
    kap = np.asarray(self.kappa(z), dtype=np.float64) + 0 * z
    area = np.asarray(self.Area(z), dtype=np.float64) + 0 * z
    if (self._kap_cached is None or not np.array_equal(kap, self._kap_cached) or
        not np.array_equal(area, self._area_cached)):
      # static coefficients changed (first call, or the user re-assigned kappa / Area)
      self._kap.upload(kap)
      self._area.upload(area)
      self._dAk.upload(np.gradient(area * kap, z))  # dAkappa_dz, column.py:96-122
      self._kap_cached, self._area_cached = kap.copy(), area.copy()
    h[0:nz] = self.b
    h[nz:2 * nz] = 0. if wA is None else wA
    if vdx_in is not None:
      h[2 * nz:3 * nz] = vdx_in
      h[3 * nz:4 * nz] = b_in
    h[4 * nz:4 * nz + 4] = (self.bs, self.bbot, 0. if self.bzbot is None else self.bzbot,
                            self.N2min)
    flags = (_lib.PM_COL_DO_CONV if do_conv else 0) | (
        _lib.PM_COL_BZBOT if self.bzbot is not None else 0)
    h[4 * nz + 4:].view(np.int32)[0] = flags
    self._arena.upload(h)
    _lib.check(_lib.lib.pm_column_steps(
        self._C.byref(self._desc), self._wA_ptr,
        self._vdx_ptr if vdx_in is not None else None,
        self._bin_ptr if vdx_in is not None else None, float(dt), 1, int(ops), 0, None))
    out = np.empty(nz)
    _lib.check(_lib.lib.pm_memcpy_d2h(out.ctypes.data, self._arena.ptr, nz * 8, None))
    self.b[...] = out

  # ---- the time-stepping API (column.py:210-348)
  def vertadvdiff(self, wA, dt, do_conv=False):
    wA = make_array(wA, self.z, 'wA')
    self._run(_lib.PM_OP_VERTADVDIFF, do_conv, wA=wA, dt=dt)

  def convect(self):
    self._run(_lib.PM_OP_CONVECT, True)

  def horadv(self, vdx_in, b_in, dt):
    vdx_in = make_array(vdx_in, self.z, 'vdx_in')
    b_in = make_array(b_in, self.z, 'b_in')
    self._run(_lib.PM_OP_HORADV, False, dt=dt, vdx_in=vdx_in, b_in=b_in)

  def timestep(self, wA=0., dt=1., do_conv=False, vdx_in=None, b_in=None):
    wA = make_array(wA, self.z, 'wA')
    ops = _lib.PM_OP_CONVECT | _lib.PM_OP_VERTADVDIFF
    if vdx_in is not None and b_in is None:
      # the reference has already convected and stepped when it raises (column.py:336-348)
      self._run(ops, do_conv, wA=wA, dt=dt)
      raise TypeError('b_in is needed if vdx_in is provided')
    if vdx_in is not None:
      vdx_in = make_array(vdx_in, self.z, 'vdx_in')
      b_in = make_array(b_in, self.z, 'b_in')
      ops |= _lib.PM_OP_HORADV
    self._run(ops, do_conv, wA=wA, dt=dt, vdx_in=vdx_in, b_in=b_in)
