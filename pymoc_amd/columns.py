"""ColumnBatch: an ensemble-major batch of independent advective-diffusive columns
resident in HBM, stepped by the fused HIP kernel `pm_column_steps`.

Arithmetic contract: Column.convect / vertadvdiff / horadv / timestep of the
reference (src/pymoc/modules/column.py:210-348), bit for bit.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib, pm_columns
from .device import DeviceArray, _sh


def _per_col(v, ncols, dtype=np.float64):
  a = np.asarray(v, dtype=dtype)
  if a.ndim == 0:
    a = np.full(ncols, a, dtype=dtype)
  if a.shape != (ncols,):
    raise ValueError("per-column parameter must be scalar or shape (%d,)" % ncols)
  return np.ascontiguousarray(a)


def _per_col_profile(v, ncols, nz):
  a = np.asarray(v, dtype=np.float64)
  if a.ndim == 0:
    a = np.full((ncols, nz), a)
  elif a.ndim == 1:
    if a.shape[0] != nz:
      raise ValueError("profile must have nz=%d levels" % nz)
    a = np.broadcast_to(a, (ncols, nz))
  if a.shape != (ncols, nz):
    raise ValueError("profile must be scalar, (nz,) or (ncols, nz)")
  return np.ascontiguousarray(a)


def dAkappa_dz(area, kappa, z):
  """np.gradient(Area(z)*kappa(z), z) per column -- Column.dAkappa_dz, column.py:96-122.

  Static data: evaluated once on the host with NumPy itself (the reference re-evaluates
  it every step) and kept in HBM."""
  return np.gradient(area * kappa, z, axis=-1)


class ColumnBatch(object):
  def __init__(self, z, kappa, area, b, bs=0.025, bbot=0.0, bzbot=None, N2min=1e-7,
               do_conv=False, kappa_alt=None, stream=None, report_nonfinite=True):
    _lib.require_device()
    z = np.ascontiguousarray(z, dtype=np.float64)
    if z.ndim != 1 or z.size < 2:
      raise ValueError("z must be a 1-D grid with at least 2 levels")
    self.nz = nz = z.size
    b = np.asarray(b, dtype=np.float64)
    if b.ndim == 1:
      b = b[None, :]
    self.ncols = ncols = b.shape[0]
    if b.shape != (ncols, nz):
      raise ValueError("b must be (ncols, nz)")
    self.stream = stream
    self.z_host = z
    self.z = DeviceArray.from_host(z, stream=stream)
    self.b = DeviceArray.from_host(b, stream=stream)
    self.nsel = 2 if kappa_alt is not None else 1
    self.kappa = DeviceArray((self.nsel, ncols, nz))
    self.dAk = DeviceArray((self.nsel, ncols, nz))
    self.area = DeviceArray((ncols, nz))
    self.set_static(kappa, area, kappa_alt)
    self.bs = DeviceArray((ncols,))
    self.bbot = DeviceArray((ncols,))
    self.bzbot = DeviceArray((ncols,))
    self.N2min = DeviceArray((ncols,))
    self.flags = DeviceArray((ncols,), np.int32)
    self.ksel = DeviceArray.zeros((ncols,), np.int32, stream=stream)
    self.nonfinite = DeviceArray.zeros((ncols,), np.int32, stream=stream) if report_nonfinite else None
    self._flags_host = np.zeros(ncols, dtype=np.int32)
    self.set_params(bs=bs, bbot=bbot, bzbot=bzbot, N2min=N2min, do_conv=do_conv)
    self._wA = None
    self._vdx = None
    self._bin = None

  # ------------------------------------------------------------------ uploads
  def set_static(self, kappa, area, kappa_alt=None):
    ncols, nz = self.ncols, self.nz
    area = _per_col_profile(area, ncols, nz)
    ks = [_per_col_profile(kappa, ncols, nz)]
    if self.nsel == 2:
      if kappa_alt is None:
        raise ValueError("kappa_alt required for a two-set batch")
      ks.append(_per_col_profile(kappa_alt, ncols, nz))
    self.area.upload(area, self.stream)
    # every column's Area is constant in z (all reference scripts): lets the fused JN2018
    # kernel keep it in scalar registers (pm_jn2018.hints)
    self.uniform_area = bool(np.all(area == area[:, :1]))
    self.kappa.upload(np.stack(ks), self.stream)
    self.dAk.upload(np.stack([dAkappa_dz(area, k, self.z_host) for k in ks]), self.stream)

  def set_params(self, bs=None, bbot=None, bzbot=False, N2min=None, do_conv=None):
    """Per-column scalars; `bzbot=None` clears the bottom-stratification BC, the
    default `False` leaves it unchanged."""
    n = self.ncols
    if bs is not None:
      self.bs.upload(_per_col(bs, n), self.stream)
    if bbot is not None:
      self.bbot.upload(_per_col(bbot, n), self.stream)
    if N2min is not None:
      self.N2min.upload(_per_col(N2min, n), self.stream)
    touched = False
    if bzbot is None:
      self._flags_host &= ~_lib.PM_COL_BZBOT
      touched = True
    elif bzbot is not False:
      self.bzbot.upload(_per_col(bzbot, n), self.stream)
      self._flags_host |= _lib.PM_COL_BZBOT
      touched = True
    if do_conv is not None:
      dc = _per_col(do_conv, n, dtype=bool)
      self._flags_host = np.where(dc, self._flags_host | _lib.PM_COL_DO_CONV,
                                  self._flags_host & ~_lib.PM_COL_DO_CONV).astype(np.int32)
      touched = True
    if touched:
      self.flags.upload(self._flags_host, self.stream)

  def set_ksel(self, ksel):
    self.ksel.upload(_per_col(ksel, self.ncols, np.int32), self.stream)

  def set_b(self, b):
    self.b.upload(np.asarray(b, dtype=np.float64).reshape(self.ncols, self.nz),
                  self.stream)

  def get_b(self, out=None):
    return self.b.download(out, self.stream)

  def get_nonfinite(self):
    return self.nonfinite.download(stream=self.stream) if self.nonfinite else None

  def _dev(self, x, cache_name):
    """Accept a DeviceArray or host data for a [ncols, nz] field."""
    if x is None or isinstance(x, DeviceArray):
      return x
    buf = getattr(self, cache_name)
    if buf is None:
      buf = DeviceArray((self.ncols, self.nz))
      setattr(self, cache_name, buf)
    buf.upload(_per_col_profile(x, self.ncols, self.nz), self.stream)
    return buf

  # ------------------------------------------------------------------ compute
  def descriptor(self):
    d = pm_columns()
    d.ncols, d.nz, d.nsel, d.reserved = self.ncols, self.nz, self.nsel, 0
    d.z, d.b = self.z.ptr, self.b.ptr
    d.kappa, d.area, d.dAkappa = self.kappa.ptr, self.area.ptr, self.dAk.ptr
    d.bs, d.bbot, d.bzbot, d.N2min = self.bs.ptr, self.bbot.ptr, self.bzbot.ptr, self.N2min.ptr
    d.flags, d.ksel = self.flags.ptr, self.ksel.ptr
    d.nonfinite = self.nonfinite.ptr if self.nonfinite else None
    return d

  def kernel_shape(self, lanes_per_col=0):
    """(lanes per column, levels per lane) the library uses for this batch."""
    g, p = C.c_int32(0), C.c_int32(0)
    check(lib.pm_column_kernel_shape(self.ncols, self.nz, int(lanes_per_col), C.byref(g),
                                     C.byref(p)))
    return g.value, p.value

  def kernel_name(self, nsteps, lanes_per_col=0, ops=_lib.PM_OP_TIMESTEP, horadv=False):
    """The kernel instantiation `steps(...)` of this shape launches (reporting only)."""
    buf = C.create_string_buffer(96)
    check(lib.pm_column_kernel_name(self.ncols, self.nz, int(lanes_per_col), int(nsteps),
                                    int(ops), int(bool(horadv)), buf, 96))
    return buf.value.decode()

  def steps(self, wA, dt, nsteps=1, ops=_lib.PM_OP_TIMESTEP, vdx_in=None, b_in=None,
            lanes_per_col=0):
    """nsteps x (convect -> vertadvdiff -> horadv) with wA held fixed, one launch."""
    if vdx_in is not None and b_in is None:
      raise TypeError('b_in is needed if vdx_in is provided')  # column.py:348
    wA_d = self._dev(wA, "_wA")
    vdx_d = self._dev(vdx_in, "_vdx")
    bin_d = self._dev(b_in, "_bin") if vdx_in is not None else None
    d = self.descriptor()
    check(lib.pm_column_steps(C.byref(d), wA_d.ptr if wA_d else None,
                              vdx_d.ptr if vdx_d else None,
                              bin_d.ptr if bin_d else None, float(dt), int(nsteps),
                              int(ops), int(lanes_per_col), _sh(self.stream)))
