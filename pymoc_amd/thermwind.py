"""ThermwindBatch: thermal-wind overturning + isopycnal remap for an ensemble, on the GPU.

Arithmetic contract: Psi_Thermwind.solve / Psib / Psibz of the reference
(src/pymoc/modules/psi_thermwind.py:125-208).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib, pm_thermwind
from .device import DeviceArray, _sh


class ThermwindBatch(object):
  """n members on one grid z.  b1 / b2 are [n, nz] DeviceArrays (or raw device pointers
  into a ColumnBatch's state) that the batch reads but does not own."""

  def __init__(self, z, n, f=1.2e-4, nb=500, stream=None, z_dev=None):
    _lib.require_device()
    self.z_host = np.ascontiguousarray(z, dtype=np.float64)
    self.nz = self.z_host.size
    self.n, self.nb = int(n), int(nb)
    self.stream = stream
    self.z = z_dev if z_dev is not None else DeviceArray.from_host(self.z_host, stream=stream)
    fv = np.asarray(f, dtype=np.float64)
    self.f = DeviceArray.from_host(np.full(self.n, fv) if fv.ndim == 0 else fv, stream=stream)
    self.Psi = DeviceArray.zeros((self.n, self.nz), stream=stream)
    self.bgrid = DeviceArray.zeros((self.n, self.nb), stream=stream)
    self.psib = DeviceArray.zeros((self.n, self.nb), stream=stream)
    # Psi_iso of the basin and of the north in ONE [2n, nz] array (rows as a two-column
    # ColumnBatch has them: ColumnBatch.steps(psi_forcing=...) reads it as the columns' forcing)
    self.psibz = DeviceArray.zeros((2 * self.n, self.nz), stream=stream)
    self.psibz1 = self.psibz.view(0, self.n)
    self.psibz2 = self.psibz.view(self.n, self.n)

  @staticmethod
  def _ptr(x):
    if x is None:
      return None
    return x.ptr if isinstance(x, DeviceArray) else int(x)

  def descriptor(self, b1, b2, Psi_SO=None, wA1=None, wA2=None, nb=None, store_psib=True):
    """The pm_thermwind of an update of this batch (also a member of pm_twocol_loop /
    pm_jn2018_loop, the persistent run kernels)."""
    d = pm_thermwind()
    d.n, d.nz, d.nb, d.reserved = self.n, self.nz, int(nb or self.nb), 0
    if d.nb > self.nb:
      raise ValueError("nb exceeds the batch's allocation")
    d.z, d.b1, d.b2, d.f = self.z.ptr, self._ptr(b1), self._ptr(b2), self.f.ptr
    d.Psi = self.Psi.ptr
    d.bgrid, d.psib = (self.bgrid.ptr, self.psib.ptr) if store_psib else (None, None)
    d.psibz1, d.psibz2 = self.psibz1.ptr, self.psibz2.ptr
    d.Psi_SO, d.wA1, d.wA2 = self._ptr(Psi_SO), self._ptr(wA1), self._ptr(wA2)
    return d

  def update(self, b1, b2, ops=_lib.PM_TW_SOLVE | _lib.PM_TW_PSIB | _lib.PM_TW_PSIBZ,
             Psi_SO=None, wA1=None, wA2=None, nb=None, store_psib=True):
    """store_psib=False keeps `psib` / `bgrid` in the kernel's LDS only (the remap to the
    columns' levels does not need them in HBM: 2 x 8 nb bytes per member and update saved)."""
    d = self.descriptor(b1, b2, Psi_SO, wA1, wA2, nb, store_psib)
    check(lib.pm_thermwind_update(C.byref(d), int(ops), _sh(self.stream)))
