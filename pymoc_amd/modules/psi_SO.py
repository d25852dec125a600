"""Drop-in `Psi_SO` with the reference's Python API, computing on the GPU.

Same constructor, attributes, methods and error text as `pymoc.modules.Psi_SO`
(src/pymoc/modules/psi_SO.py:8-375).  `b`, `bs`, `tau` given as callables are sampled on
their grid (`z` / `y`) and interpolated linearly in between; array and float inputs behave
exactly like the reference.  `ys` is the direct inverse of the piecewise-linear bs(y)
(the reference root-finds it with brentq to xtol=2e-12); the GM boundary-value problem
(`c` not None) is solved by 4th-order collocation on a refined fixed mesh instead of
SciPy's adaptive solve_bvp (agreement ~1e-6).
"""
import numpy as np

from .. import _lib
from ..device import DeviceArray
from ..psi_so import PsiSOBatch
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
    self._batch = None

  # ---- device plumbing
  def _run(self, ops, b=None):
    nz, ny = np.size(self.z), np.size(self.y)
    if self._batch is None:
      self._batch = PsiSOBatch(self.z, self.y, 1, tau=0.0, diagnostics=True)
      self._b = DeviceArray((1, nz))
      self._bs = DeviceArray((1, ny))
    t = self._batch
    t.opts = dict(f=self.f, rho=self.rho, L=self.L, c=self.c, bvp_with_Ek=self.bvp_with_Ek,
                  Hsill=self.Hsill, HEk=self.HEk, Htapertop=self.Htapertop,
                  Htaperbot=self.Htaperbot, smax=self.smax)
    t.set_KGM(float(self.KGM))
    if self._tau_is_float:
      t.set_tau(float(self._tau_float))
    else:
      t.set_tau(np.asarray(make_array(self.tau, self.y, 'tau'), dtype=np.float64)[None, :] +
                0 * self.y)
    barr = make_array(self.b, self.z, 'b') if b is None else b
    self._b.upload(np.asarray(barr, dtype=np.float64) + 0 * self.z)
    self._bs.upload(np.asarray(make_array(self.bs, self.y, 'bs'), dtype=np.float64) + 0 * self.y)
    if ops == _lib.PM_SO_OP_GM:
      t.Psi_Ek.upload(np.asarray(self.Psi_Ek, dtype=np.float64)[None, :])
    t.update(self._b, self._bs, ops=ops)
    return t

  # ---- API (psi_SO.py:106-375)
  def ys(self, b):
    t = self._run(_lib.PM_SO_OP_EKMAN, b=float(b) + 0 * self.z)
    return t.ys.download()[0, 0]

  def calc_N2(self):
    dz = self.z[1:] - self.z[:-1]
    N2 = np.zeros(np.size(self.z))
    b = self.b(self.z)
    N2[1:-1] = (b[2:] - b[:-2]) / (dz[1:] + dz[:-1])
    N2[0] = (b[1] - b[0]) / dz[0]
    N2[-1] = (b[-1] - b[-2]) / dz[-1]
    return make_func(N2, self.z, 'N2')

  def calc_Ekman(self):
    return self._run(_lib.PM_SO_OP_EKMAN).Ek_raw.download()[0]

  def calc_GM(self):
    return self._run(_lib.PM_SO_OP_GM).GM_raw.download()[0]

  def solve(self):
    t = self._run(_lib.PM_SO_OP_SOLVE)
    self.Psi_Ek = t.Psi_Ek.download()[0]
    self.Psi_GM = t.Psi_GM.download()[0]
    self.Psi = t.Psi.download()[0]

  def update(self, b=None, bs=None):
    if b is not None:
      self.b = make_func(b, self.z, 'b')
    if bs is not None:
      self.bs = make_func(bs, self.y, 'bs')
