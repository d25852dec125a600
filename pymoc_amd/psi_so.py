"""PsiSOBatch: Southern-Ocean residual overturning for an ensemble, on the GPU.

Arithmetic contract: Psi_SO.ys / calc_Ekman / calc_GM / solve of the reference
(src/pymoc/modules/psi_SO.py:106-354).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib, pm_psi_so
from .device import DeviceArray, _sh


def _ptr(x):
  if x is None:
    return None
  return x.ptr if isinstance(x, DeviceArray) else int(x)


class PsiSOBatch(object):
  """n members on shared grids z, y.  `b` [n,nz] and `bs` [n,ny] are device arrays (or raw
  device pointers into other batches' state) read at every update()."""

  def __init__(self, z, y, n, tau, KGM=1e3, f=1.2e-4, rho=1030, L=1e7, c=None,
               bvp_with_Ek=False, Hsill=None, HEk=None, Htapertop=None, Htaperbot=None,
               smax=0.01, bvp_refine=0, stream=None, z_dev=None, diagnostics=False):
    _lib.require_device()
    self.z_host = np.ascontiguousarray(z, dtype=np.float64)
    self.y_host = np.ascontiguousarray(y, dtype=np.float64)
    self.nz, self.ny, self.n = self.z_host.size, self.y_host.size, int(n)
    self.stream = stream
    self.z = z_dev if z_dev is not None else DeviceArray.from_host(self.z_host, stream=stream)
    self.y = DeviceArray.from_host(self.y_host, stream=stream)
    self.KGM = DeviceArray((self.n,))
    self.tau = None
    self.flags = 0
    self.set_tau(tau)
    self.set_KGM(KGM)
    self.opts = dict(f=f, rho=rho, L=L, c=c, bvp_with_Ek=bvp_with_Ek, Hsill=Hsill, HEk=HEk,
                     Htapertop=Htapertop, Htaperbot=Htaperbot, smax=smax)
    self.bvp_refine = int(bvp_refine)
    self.Psi = DeviceArray.zeros((self.n, self.nz), stream=stream)
    self.Psi_Ek = DeviceArray.zeros((self.n, self.nz), stream=stream)
    self.Psi_GM = DeviceArray.zeros((self.n, self.nz), stream=stream)
    self.status = DeviceArray.zeros((self.n,), np.int32, stream=stream)
    self.Ek_raw = self.GM_raw = self.ys = None
    if diagnostics:
      self.Ek_raw = DeviceArray.zeros((self.n, self.nz), stream=stream)
      self.GM_raw = DeviceArray.zeros((self.n, self.nz), stream=stream)
      self.ys = DeviceArray.zeros((self.n, self.nz), stream=stream)

  def set_tau(self, tau):
    """scalar or (n,): one wind stress per member; (n, ny): a profile on y per member."""
    t = np.asarray(tau, dtype=np.float64)
    if t.ndim == 0:
      t = np.full(self.n, t)
    if t.ndim == 1 and t.shape == (self.n,):
      self._tau_array = False
    elif t.ndim == 2 and t.shape == (self.n, self.ny):
      self._tau_array = True
    else:
      raise ValueError("tau must be scalar, (n,) or (n, ny)")
    if self.tau is None or self.tau.shape != t.shape:
      self.tau = DeviceArray(t.shape)
    self.tau.upload(t, self.stream)

  def set_KGM(self, KGM):
    k = np.asarray(KGM, dtype=np.float64)
    self.KGM.upload(np.full(self.n, k) if k.ndim == 0 else k, self.stream)

  def update(self, b, bs, ops=_lib.PM_SO_OP_SOLVE):
    d = self.descriptor(b, bs)
    check(lib.pm_psi_so_update(C.byref(d), int(ops), _sh(self.stream)))

  def descriptor(self, b, bs):
    """The pm_psi_so of an update of this batch (also a member of pm_jn2018_loop)."""
    o = self.opts
    d = pm_psi_so()
    d.n, d.nz, d.ny = self.n, self.nz, self.ny
    fl = _lib.PM_SO_TAU_ARRAY if self._tau_array else 0
    for name, bit in (("c", _lib.PM_SO_HAS_C), ("Hsill", _lib.PM_SO_HAS_HSILL),
                      ("HEk", _lib.PM_SO_HAS_HEK), ("Htapertop", _lib.PM_SO_HAS_HTAPERTOP),
                      ("Htaperbot", _lib.PM_SO_HAS_HTAPERBOT)):
      if o[name] is not None:
        fl |= bit
      setattr(d, name, float(o[name]) if o[name] is not None else 0.0)
    if o["bvp_with_Ek"]:
      fl |= _lib.PM_SO_BVP_WITH_EK
    d.flags, d.bvp_refine, d.reserved = fl, self.bvp_refine, 0
    d.z, d.y, d.b, d.bs = self.z.ptr, self.y.ptr, _ptr(b), _ptr(bs)
    d.tau, d.KGM = self.tau.ptr, self.KGM.ptr
    d.f, d.rho, d.L, d.smax = float(o["f"]), float(o["rho"]), float(o["L"]), float(o["smax"])
    d.Psi, d.Psi_Ek, d.Psi_GM = self.Psi.ptr, self.Psi_Ek.ptr, self.Psi_GM.ptr
    d.Ek_raw, d.GM_raw, d.ys = _ptr(self.Ek_raw), _ptr(self.GM_raw), _ptr(self.ys)
    d.status = self.status.ptr
    return d
