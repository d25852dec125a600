"""SOMLBatch: Southern-Ocean mixed-layer step for an ensemble, on the GPU.

Arithmetic contract: SO_ML.timestep / advdiff of the reference
(src/pymoc/modules/SO_ML.py:77-303), Crank-Nicolson diffusion by a Thomas sweep.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib, pm_so_ml
from .device import DeviceArray, _sh


def _ptr(x):
  if x is None:
    return None
  return x.ptr if isinstance(x, DeviceArray) else int(x)


def _rows(v, n, ny):
  a = np.asarray(v, dtype=np.float64)
  if a.ndim == 0:
    return np.full((n, ny), a)
  if a.ndim == 1:
    return np.broadcast_to(a, (n, ny)).copy()
  return np.ascontiguousarray(a)


class SOMLBatch(object):
  def __init__(self, y, nz, bs, surflux=0., rest_mask=0., b_rest=0., Ks=0., h=50., L=4e6,
               v_pist=1.5 / 86400., stream=None):
    _lib.require_device()
    self.y_host = np.ascontiguousarray(y, dtype=np.float64)
    self.ny, self.nz = self.y_host.size, int(nz)
    bs = np.atleast_2d(np.asarray(bs, dtype=np.float64))
    self.n = bs.shape[0]
    self.stream = stream
    self.y = DeviceArray.from_host(self.y_host, stream=stream)
    self.bs = DeviceArray.from_host(bs, stream=stream)
    self.Psi_s = DeviceArray.zeros((self.n, self.ny), stream=stream)
    self.surflux = DeviceArray.from_host(_rows(surflux, self.n, self.ny), stream=stream)
    self.rest_mask = DeviceArray.from_host(_rows(rest_mask, self.n, self.ny), stream=stream)
    self.b_rest = DeviceArray.from_host(_rows(b_rest, self.n, self.ny), stream=stream)
    self.status = DeviceArray.zeros((self.n,), np.int32, stream=stream)
    self.Ks, self.h, self.L, self.v_pist = float(Ks), float(h), float(L), float(v_pist)

  def step(self, b_basin, Psi_b, dt):
    d = pm_so_ml()
    d.n, d.nz, d.ny, d.reserved = self.n, self.nz, self.ny, 0
    d.y, d.bs, d.Psi_s = self.y.ptr, self.bs.ptr, self.Psi_s.ptr
    d.b_basin, d.Psi_b = _ptr(b_basin), _ptr(Psi_b)
    d.surflux, d.rest_mask, d.b_rest = self.surflux.ptr, self.rest_mask.ptr, self.b_rest.ptr
    d.Ks, d.h, d.L, d.v_pist = self.Ks, self.h, self.L, self.v_pist
    d.status = self.status.ptr
    check(lib.pm_so_ml_step(C.byref(d), float(dt), _sh(self.stream)))
