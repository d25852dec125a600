"""Drop-in `Psi_SO` with the reference's Python API, computing on the GPU.

Same constructor, attributes, methods and error text as `pymoc.modules.Psi_SO`
(src/pymoc/modules/psi_SO.py:8-375).  Array and float inputs behave exactly like the
reference; a callable `b` is only ever evaluated on `z` (exact).  CALLABLE `bs` / `tau`, which
the reference evaluates between grid points (inside brentq and the 100-point wind average,
psi_SO.py:106-140, :238-240), can only be evaluated on the host: this class then root-finds
the outcrop latitudes with the same brentq iteration (`pymoc_amd.utils.brentq`, a restatement
of SciPy's, bit-identical to it) and averages the wind exactly like the reference, and hands
both to the kernel (`pm_psi_so.ys_in / tau_ave_in`); everything downstream runs on the device
(agreement with the reference 1e-13).  For array `bs`, `ys` is the direct inverse of the
piecewise-linear bs(y) where
that is unique (the reference root-finds it with brentq to xtol=2e-12) and brentq's own
iteration where bs is not monotone; the GM boundary-value problem
(`c` not None) is solved by the same 4th-order collocation on the mesh SciPy's solve_bvp
itself ends on, its residual control followed on the device (agreement 1e-14).
"""
import numpy as np


from .. import _lib
from ..device import DeviceArray
from ..utils import make_func, make_array


class Psi_SO(object):
  def __init__(
      self,
      z=None,
      y=None,
      b=None,
      bs=None,
      tau=None,
      f=1.2e-4,
      rho=1030,
      L=1e7,
      KGM=1e3,
      c=None,
      bvp_with_Ek=False,
      Hsill=None,
      HEk=None,
      Htapertop=None,
      Htaperbot=None,
      smax=0.01,
  ):
    if isinstance(z, np.ndarray):
      self.z = z
    else:
      raise TypeError('z needs to be numpy array providing grid levels')
    if isinstance(y, np.ndarray):
      self.y = y
    else:
      raise TypeError(
          'y needs to be numpy array providing horizontal grid (or boundaries) of ACC')
    self.b = make_func(b, self.z, 'b')
    self.bs = make_func(bs, self.y, 'bs')
    self.tau = make_func(tau, self.y, 'tau')
    self._bs_callable, self._tau_callable = callable(bs), callable(tau)
    self._tau_is_float = isinstance(tau, float)
    self._tau_float = tau if self._tau_is_float else None
    self.f = f
    self.rho = rho
    self.L = L
    self.KGM = KGM
    self.c = c
    self.bvp_with_Ek = bvp_with_Ek
    self.Hsill = Hsill
    self.HEk = HEk
    self.Htapertop = Htapertop
    self.Htaperbot = Htaperbot
    self.smax = smax
    self._arena = None

  # ---- device plumbing: one arena, one H2D and one D2H per call
  # arena (float64 slots): [b | bs | tau | KGM | Psi_Ek | Psi | Psi_GM | Ek_raw | GM_raw | ys]
  def _kernel_y(self):
    """The meridional grid handed to the kernel."""
    return self.y

  def _alloc(self):
    self._yk = np.ascontiguousarray(self._kernel_y(), dtype=np.float64)
    nz, ny = np.size(self.z), np.size(self._yk)
    self._nz, self._ny = nz, ny
    self._nin = nz + 2 * ny + 1 + nz
    self._host = np.zeros(self._nin)
    self._out = np.zeros((6, nz))
    self._arena = DeviceArray((self._nin + 5 * nz,))
    self._zd = DeviceArray.from_host(np.ascontiguousarray(self.z, dtype=np.float64))
    self._yd = DeviceArray.from_host(self._yk)
    self._status = DeviceArray.zeros((1,), np.int32)
    p, d = self._arena.ptr, _lib.pm_psi_so()
    d.n, d.nz, d.ny, d.reserved = 1, nz, ny, 0
    d.z, d.y = self._zd.ptr, self._yd.ptr
    off = lambda k: p + 8 * k
    d.b, d.bs, d.tau, d.KGM = off(0), off(nz), off(nz + ny), off(nz + 2 * ny)
    o = nz + 2 * ny + 1
    d.Psi_Ek, d.Psi, d.Psi_GM = off(o), off(o + nz), off(o + 2 * nz)
    d.Ek_raw, d.GM_raw, d.ys = off(o + 3 * nz), off(o + 4 * nz), off(o + 5 * nz)
    d.status = self._status.ptr
    self._extra = DeviceArray((2 * nz,))  # ys_in | tau_ave_in (callable bs / tau only)
    self._desc, self._out_ptr = d, off(o)

  def _run(self, ops, b=None):
    import ctypes as C
    from .column import flush_all
    flush_all()  # self.b may alias the array of a Column with queued steps
    if getattr(self, "_arena", None) is None or self._nz != np.size(self.z) or \
        self._ny != np.size(self._kernel_y()):
      self._alloc()
    nz, ny, h, d = self._nz, self._ny, self._host, self._desc
    fl = 0
    for name, bit in (("c", _lib.PM_SO_HAS_C), ("Hsill", _lib.PM_SO_HAS_HSILL),
                      ("HEk", _lib.PM_SO_HAS_HEK), ("Htapertop", _lib.PM_SO_HAS_HTAPERTOP),
                      ("Htaperbot", _lib.PM_SO_HAS_HTAPERBOT)):
      v = getattr(self, name)
      if v is not None:
        fl |= bit
      setattr(d, name, float(v) if v is not None else 0.0)
    if self.bvp_with_Ek:
      fl |= _lib.PM_SO_BVP_WITH_EK
    h[0:nz] = make_array(self.b, self.z, 'b') if b is None else b
    h[nz:nz + ny] = make_array(self.bs, self._yk, 'bs')
    if self._tau_is_float:
      h[nz + ny] = self._tau_float
    else:
      fl |= _lib.PM_SO_TAU_ARRAY
      h[nz + ny:nz + 2 * ny] = make_array(self.tau, self._yk, 'tau')
    h[nz + 2 * ny] = self.KGM
    if ops == _lib.PM_SO_OP_GM:
      h[nz + 2 * ny + 1:] = self.Psi_Ek
    d.flags, d.bvp_refine = fl, 0
    d.f, d.rho, d.L, d.smax = float(self.f), float(self.rho), float(self.L), float(self.smax)
    d.ys_in, d.tau_ave_in = None, None
    if self._bs_callable or self._tau_callable:
      # only the host can evaluate the callables between grid points: the reference's own
      # root-finding and wind average (psi_SO.py:106-140, :238-240), handed to the kernel
      ys_h = np.array([self._ys_host(bv) for bv in h[0:nz]])
      yN = self.y[-1]
      tau_h = np.array([np.mean(self.tau(np.linspace(y0, yN, 100))) for y0 in ys_h])
      ex = np.concatenate([ys_h, tau_h])
      _lib.check(_lib.lib.pm_memcpy_h2d(self._extra.ptr, ex.ctypes.data, ex.nbytes, None))
      d.ys_in, d.tau_ave_in = self._extra.ptr, self._extra.ptr + 8 * nz
    self._arena_upload(h)
    _lib.check(_lib.lib.pm_psi_so_update(C.byref(d), int(ops), None))
    _lib.check(_lib.lib.pm_memcpy_d2h(self._out.ctypes.data, self._out_ptr, 6 * nz * 8, None))
    return self._out

  def _arena_upload(self, h):
    _lib.check(_lib.lib.pm_memcpy_h2d(self._arena.ptr, h.ctypes.data, h.nbytes, None))

  def _ys_host(self, b):
    """Psi_SO.ys of the reference, line by line (psi_SO.py:106-140), for callable profiles."""
    from ..utils.brentq import brentq
    bsy = self.bs(self.y)
    if b < np.min(bsy):
      return self.y[0] - 1e3
    if b > self.bs(self.y[-1]):
      return self.y[-1]
    minind = np.argmin(bsy)
    return brentq(lambda yy: self.bs(yy) - b, self.y[minind], self.y[-1])

  # ---- API (psi_SO.py:106-375)
  def ys(self, b):
    return self._run(_lib.PM_SO_OP_EKMAN, b=float(b) + 0 * self.z)[5, 0]

  def calc_N2(self):
    """N^2 = db/dz on the model levels as an np.interp closure: the centred quotient
    (b[i+1] - b[i-1]) / ((z[i+1] - z[i]) + (z[i] - z[i-1])) inside the column, the one-sided
    quotient of the first / last interval at its two ends -- the arithmetic of psi_SO.py:142-162
    (the kernel's s_N2 does the same per level, csrc/psi_so.hip.h)."""
    from .column import flush_all
    flush_all()
    z = self.z
    b = self.b(z)
    h, db = np.diff(z), np.diff(b)
    first, last = db[0] / h[0], db[-1] / h[-1]
    inner = (b[2:] - b[:-2]) / (h[1:] + h[:-1])
    return make_func(np.concatenate(([first], inner, [last])), z, 'N2')

  # The two quadratic tapers of psi_SO.py:164-216 (host NumPy; the kernel applies the same
  # expressions per level).  `H is None` means "no taper".
  def calc_bottom_taper(self, H, z):
    """Weight 1 - (depth into the layer of thickness H above the bottom z[0])^2 / H^2."""
    if H is None:
      return 1.
    into = np.maximum(z[0] + H - z, 0.)
    return 1. - into**2. / H**2.

  def calc_top_taper(self, H, z, scalar=True):
    """Weight 1 - (height into the layer of thickness H below the surface)^2 / H^2.  Without H:
    1, or (scalar=False, the Ekman taper) ones with a 0 at the surface level."""
    if H is not None:
      into = np.maximum(z + H, 0)
      return 1 - into**2. / H**2.
    if scalar:
      return 1.
    taper = np.ones(np.size(z))
    taper[-1] = 0.
    return taper

  def bc_GM(self, ya, yb):
    """Boundary residuals of the GM boundary-value problem (psi_SO.py:245-275): the eddy
    transport vanishes at bottom and surface, or (bvp_with_Ek) cancels the Ekman transport
    there -- Psi_Ek is in Sv, the unknown in m^3/s."""
    if not self.bvp_with_Ek:
      return np.array([ya[0], yb[0]])
    ek = self.Psi_Ek
    return np.array([ya[0] + ek[0] * 1e6, yb[0] + ek[-1] * 1e6])

  def calc_Ekman(self):
    return self._run(_lib.PM_SO_OP_EKMAN)[3].copy()

  def calc_GM(self):
    return self._run(_lib.PM_SO_OP_GM)[4].copy()

  def solve(self):
    out = self._run(_lib.PM_SO_OP_SOLVE)
    self.Psi_Ek = out[0].copy()
    self.Psi = out[1].copy()
    self.Psi_GM = out[2].copy()

  def update(self, b=None, bs=None):
    if b is not None:
      self.b = make_func(b, self.z, 'b')
    if bs is not None:
      self.bs = make_func(bs, self.y, 'bs')
      self._bs_callable = callable(bs)
