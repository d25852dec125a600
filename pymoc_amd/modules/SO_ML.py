"""Drop-in `SO_ML` with the reference's Python API, computing on the GPU.

Same constructor, attributes, methods and error text as `pymoc.modules.SO_ML`
(src/pymoc/modules/SO_ML.py:5-303).  `self.bs` and `self.Psi_s` are REBOUND to new arrays
every step, as in the reference (SO_ML.py:232,259,269).  The Crank-Nicolson solve is a
Thomas sweep instead of the reference's dense np.linalg.inv (difference ~1e-16).
"""
import numpy as np

from ..device import DeviceArray
from ..utils import make_array


class SO_ML(object):
  def __init__(
      self,
      y=None,
      Ks=0.,
      h=50.,
      L=4e6,
      surflux=0.,
      rest_mask=0.,
      b_rest=0.,
      v_pist=1.5 / 86400.,
      bs=0.0,
      Psi_s=None
  ):
    if isinstance(y, np.ndarray):
      self.y = y
    else:
      raise TypeError('y needs to be numpy array providing (regular) grid')
    self.Ks = Ks
    self.h = h
    self.L = L
    self.surflux = make_array(surflux, self.y, 'surflux')
    self.rest_mask = make_array(rest_mask, self.y, 'rest_mask')
    self.b_rest = make_array(b_rest, self.y, 'b_rest')
    self.v_pist = v_pist
    self.Psi_s = Psi_s
    self.bs = make_array(bs, self.y, 'bs')
    self._batch = None
    self._shape = None

  # ---- the building blocks of advdiff as separate host functions, as in the reference
  # (SO_ML.py:77-196): NumPy on self.bs / self.Psi_s, for callers that use them one by one.
  # `advdiff` / `timestep` do the whole update in one kernel launch and do not call them.
  def set_boundary_conditions(self, b_basin, Psi_b):
    """Southern end of the channel (SO_ML.py:77-98): with upwelling next to the boundary
    (Psi_s[1] > 0) it takes the buoyancy of the densest upwelling basin water, otherwise the
    no-flux condition copies the neighbouring point."""
    if self.Psi_s[1] > 0:
      first_up = np.argwhere(Psi_b > 0)[0][0]
      self.bs[0] = b_basin[first_up]
    else:
      self.bs[0] = self.bs[1]

  def calc_advective_tendency(self, dy):
    """Upwind meridional advection by Psi_s on the (uniform) grid (SO_ML.py:100-134):
    -Psi_s 1e6 (one-sided difference against the flow) / h / L / dy at interior points."""
    tend = 0. * self.y
    psi, bs = self.Psi_s[1:-1], self.bs
    south, north = psi < 0., psi > 0.   # flow direction decides which neighbour is upstream
    inner = tend[1:-1]                  # a view: the boundary points keep their zeros
    inner[south] = -psi[south] * 1e6 * (bs[2:][south] - bs[1:-1][south]) / self.h / self.L / dy
    inner[north] = -psi[north] * 1e6 * (bs[1:-1][north] - bs[:-2][north]) / self.h / self.L / dy
    return tend

  def calc_diffusion_matrix(self, s):
    """tridiag(-s/2, 1+s, -s/2) with identity rows at both ends (SO_ML.py:136-165): U for +s,
    V for -s of the Crank-Nicolson step U bs_new = V bs."""
    n = len(self.y)
    M = (np.diag(np.full(n - 1, -s / 2.), -1) + np.diag(np.full(n, 1 + s), 0) +
         np.diag(np.full(n - 1, -s / 2.), 1))
    M[0, :2] = (1, 0)
    M[-1, -2:] = (0, 1)
    return M

  def calc_implicit_diffusion(self, dy, dt):
    """One Crank-Nicolson diffusion step of bs (SO_ML.py:167-196), the reference's way: dense
    inverse of U.  (The kernel solves the same system by cyclic reduction / a Thomas sweep.)"""
    s = self.Ks * dt / dy**2
    U, V = self.calc_diffusion_matrix(s), self.calc_diffusion_matrix(-s)
    return np.dot(np.dot(np.linalg.inv(U), V), self.bs)

  # one arena: in [bs | surflux | rest_mask | b_rest | b_basin | Psi_b]  out [bs | Psi_s | status]
  def advdiff(self, b_basin, Psi_b, dt):
    import ctypes as C
    from .. import _lib
    from .column import flush_all
    flush_all()  # b_basin may alias the array of a Column with queued steps
    ny, nz = np.size(self.y), np.size(b_basin)
    if self._batch is None or self._shape != (ny, nz):
      self._shape = (ny, nz)
      self._yd = DeviceArray.from_host(np.ascontiguousarray(self.y, dtype=np.float64))
      self._nin = 4 * ny + 2 * nz
      self._arena = DeviceArray((self._nin + ny + 1,))
      self._host = np.zeros(self._nin)
      self._batch = True
    h, p, y0 = self._host, self._arena.ptr, 0 * self.y
    h[0:ny] = np.asarray(self.bs, dtype=np.float64) + y0
    h[ny:2 * ny] = np.asarray(self.surflux, dtype=np.float64) + y0
    h[2 * ny:3 * ny] = np.asarray(self.rest_mask, dtype=np.float64) + y0
    h[3 * ny:4 * ny] = np.asarray(self.b_rest, dtype=np.float64) + y0
    h[4 * ny:4 * ny + nz] = b_basin
    h[4 * ny + nz:] = Psi_b
    _lib.check(_lib.lib.pm_memcpy_h2d(p, h.ctypes.data, h.nbytes, None))
    d = _lib.pm_so_ml()
    d.n, d.nz, d.ny, d.reserved = 1, nz, ny, 0
    d.y, d.bs = self._yd.ptr, p
    d.surflux, d.rest_mask, d.b_rest = p + ny * 8, p + 2 * ny * 8, p + 3 * ny * 8
    d.b_basin, d.Psi_b = p + 4 * ny * 8, p + (4 * ny + nz) * 8
    o = p + self._nin * 8
    d.Psi_s, d.status = o, o + ny * 8
    d.Ks, d.h, d.L, d.v_pist = float(self.Ks), float(self.h), float(self.L), float(self.v_pist)
    _lib.check(_lib.lib.pm_memset(d.status, 0, 8, None))
    _lib.check(_lib.lib.pm_so_ml_step(C.byref(d), float(dt), None))
    # bs sits at the head of the arena, Psi_s and the status word behind the inputs
    out = np.empty(self._nin + ny + 1)
    _lib.check(_lib.lib.pm_memcpy_d2h(out.ctypes.data, p, out.nbytes, None))
    if out[self._nin + ny:].view(np.int32)[0] == 1:
      # np.nonzero(Psi_mod)[0][0] / np.argwhere(Psi_b > 0)[0][0] on an empty result
      raise IndexError('index 0 is out of bounds for axis 0 with size 0')
    self.Psi_s = out[self._nin:self._nin + ny].copy()
    self.bs = out[0:ny].copy()

  def timestep(self, b_basin=None, Psi_b=None, dt=1.):
    if not isinstance(b_basin, np.ndarray):
      raise TypeError('b_basin needs to be numpy array providing buoyancy levels in basin')
    if not isinstance(Psi_b, np.ndarray):
      raise TypeError(
          'Psi_b needs to be numpy array providing overturning at buoyancy levels given by b_basin'
      )
    self.advdiff(b_basin=b_basin, Psi_b=Psi_b, dt=dt)
