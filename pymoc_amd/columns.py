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


def _in_fast_range(x):
  """Elementwise: zero, or finite with magnitude in [2^-200, 2^200] -- the operand window of the
  kernels' exact-division shortcut (csrc/common.hip.h: in_fast_div_range)."""
  a = np.abs(np.asarray(x, dtype=np.float64))
  with np.errstate(invalid="ignore"):
    return (a == 0) | ((a >= 2.0**-200) & (a <= 2.0**200))


_HINT_DIV3_OFF = 1 << 20  # (use_hints(div3=False): a bit outside the per-column flag bits)


def div3_proven(denominators):
  """True when the kernels' 3-instruction quotient is correctly rounded for EVERY numerator over
  each of these denominators (`pm_div3_proven`, include/pymoc_hip.h: a host-side proof by
  enumeration of the only numerators that could fail)."""
  d = np.ascontiguousarray(denominators, dtype=np.float64).ravel()
  ok = C.c_int32(0)
  check(lib.pm_div3_proven(d.ctypes.data, d.size, C.byref(ok), None))
  return bool(ok.value)


def device_reciprocals_exact(denominators):
  """True when the DEVICE's 1.0 / d is the correctly rounded reciprocal of each denominator
  (`pm_recip_check`: compared with the host's IEEE quotient; the proofs of the kernels' exact
  quotients take y = RN(1/d), and the device's fp64 division is not correctly rounded in every
  case)."""
  d = np.ascontiguousarray(denominators, dtype=np.float64).ravel()
  ok = C.c_int32(0)
  check(lib.pm_recip_check(d.ctypes.data, d.size, C.byref(ok)))
  return bool(ok.value)


class ColumnBatch(object):
  def __init__(self, z, kappa, area, b, bs=0.025, bbot=0.0, bzbot=None, N2min=1e-7,
               do_conv=False, kappa_alt=None, stream=None, report_nonfinite=True,
               kappa_affine=None):
    """`kappa_affine=(kappa_base [ncols], kappa_profile [nz])`: the sweep's structure, when
    `kappa == kappa_base[:, None] + kappa_profile[None, :]` holds BIT FOR BIT (checked here; a
    ValueError otherwise).  One-step launches on a large batch then form kappa on the device
    instead of streaming it (pm_columns.kappa_base / kappa_profile)."""
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
    self.kappa_base = self.kappa_profile = None
    self._kappa_affine = kappa_affine
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
    # rows [0, n/2) and [n/2, n) of a two-column ensemble each repeat one coefficient profile
    # (a sweep over forcing / boundary values only): PM_JN_SHARED_COEF of the fused JN2018 kernel
    h = ncols // 2
    self.shared_halves = bool(ncols % 2 == 0 and h >= 1 and all(
        np.array_equal(a[:h], np.broadcast_to(a[0], (h, nz))) and
        np.array_equal(a[h:], np.broadcast_to(a[h], (h, nz))) for a in [area] + ks))
    self.kappa.upload(np.stack(ks), self.stream)
    aff = getattr(self, "_kappa_affine", None)
    self.kappa_base = self.kappa_profile = None
    if aff is not None and self.nsel == 1:
      kb = np.ascontiguousarray(aff[0], dtype=np.float64).reshape(ncols)
      kp = np.ascontiguousarray(aff[1], dtype=np.float64).reshape(nz)
      if not np.array_equal(kb[:, None] + kp[None, :], ks[0]):
        raise ValueError("kappa_affine: kappa != kappa_base[:, None] + kappa_profile[None, :] bit for bit")
      self.kappa_base = DeviceArray.from_host(kb, stream=self.stream)
      self.kappa_profile = DeviceArray.from_host(kp, stream=self.stream)
    self._kappa_affine = None  # (a later set_static without the pair drops the hint)
    dAks = [dAkappa_dz(area, k, self.z_host) for k in ks]
    self.dAk.upload(np.stack(dAks), self.stream)
    # PM_COL_STATIC_IN_RANGE (include/pymoc_hip.h): the static operands of a column lie inside
    # the window of the kernels' exact-division shortcut, so the one-step streaming kernel only
    # has to test the state and the forcing
    dz = np.diff(self.z_host)
    dzc = 0.5 * (dz[1:] + dz[:-1])  # column.py:238's spacing, as the kernels form it
    ok = (_in_fast_range(self.z_host).all() and _in_fast_range(dz).all() and (dz != 0).all() and
          _in_fast_range(dzc).all() and (dzc != 0).all())
    self._static_ok = np.full(ncols, bool(ok))
    for a in [area] + ks + dAks:
      self._static_ok &= _in_fast_range(a).all(axis=1)
    self._static_ok &= (area != 0).all(axis=1)
    # PM_COLS_DIV3_PROVEN: every static denominator of the step -- the grid spacings, the centred
    # spacings and (Area one number per column) every column's Area -- admits the 3-instruction
    # exact quotient; pm_div3_proven enumerates and tests the only numerators that could fail
    self.div3_proven = False
    if self.uniform_area and ok:
      den = np.concatenate([np.unique(dz), np.unique(dzc), np.unique(area[:, 0])])
      self.div3_proven = div3_proven(den) and device_reciprocals_exact(den)
    if hasattr(self, "_flags_host"):
      self._upload_flags()

  def set_params(self, bs=None, bbot=None, bzbot=False, N2min=None, do_conv=None):
    """Per-column scalars; `bzbot=None` clears the bottom-stratification BC, the
    default `False` leaves it unchanged."""
    n = self.ncols
    par = self.__dict__.setdefault("_par_ok", {})
    if bs is not None:
      v = _per_col(bs, n)
      self.bs.upload(v, self.stream)
      par["bs"] = _in_fast_range(v)
    if bbot is not None:
      v = _per_col(bbot, n)
      self.bbot.upload(v, self.stream)
      par["bbot"] = _in_fast_range(v)
    if N2min is not None:
      v = _per_col(N2min, n)
      self.N2min.upload(v, self.stream)
      par["N2min"] = _in_fast_range(v)
    if bzbot is None:
      self._flags_host &= ~_lib.PM_COL_BZBOT
      par.pop("bzbot", None)
    elif bzbot is not False:
      v = _per_col(bzbot, n)
      self.bzbot.upload(v, self.stream)
      self._flags_host |= _lib.PM_COL_BZBOT
      par["bzbot"] = _in_fast_range(v)
    if do_conv is not None:
      dc = _per_col(do_conv, n, dtype=bool)
      self._flags_host = np.where(dc, self._flags_host | _lib.PM_COL_DO_CONV,
                                  self._flags_host & ~_lib.PM_COL_DO_CONV).astype(np.int32)
    self._upload_flags()

  def _upload_flags(self):
    ok = self._static_ok.copy()
    for v in self.__dict__.get("_par_ok", {}).values():
      ok &= v
    bit = np.int32(_lib.PM_COL_STATIC_IN_RANGE)
    self._flags_host = np.where(ok, self._flags_host | bit, self._flags_host & ~bit).astype(np.int32)
    bit = np.int32(_lib.PM_COL_UNIFORM_AREA)  # every column's Area is one number (set_static)
    self._flags_host = (self._flags_host | bit if self.uniform_area
                        else self._flags_host & ~bit).astype(np.int32)
    off = np.int32(self.__dict__.get("_hints_off", 0))
    self.flags.upload((self._flags_host & ~off).astype(np.int32), self.stream)

  def use_hints(self, uniform_area=True, static_in_range=True, div3=True):
    """Switch the per-column hints the host derives from the static operands on or off
    (PM_COL_UNIFORM_AREA: a column's Area is one number, read with its scalars;
    PM_COL_STATIC_IN_RANGE: the static operands lie inside the exact-division window, so a launch
    tests only the state and the forcing).  Without them the kernels take the C-ABI's default
    path -- every array read, every operand tested; results are bit-identical either way.  The
    hints are re-derived whenever the static operands or parameters change."""
    self._hints_off = ((0 if uniform_area else _lib.PM_COL_UNIFORM_AREA) |
                       (0 if static_in_range else _lib.PM_COL_STATIC_IN_RANGE) |
                       (0 if div3 else _HINT_DIV3_OFF))
    self._upload_flags()

  @property
  def has_bzbot(self):
    """Does any column use the bottom-gradient boundary condition (column.py:232-233)?"""
    return bool(np.any(self._flags_host & _lib.PM_COL_BZBOT))

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
    d.ncols, d.nz, d.nsel = self.ncols, self.nz, self.nsel
    off = self.__dict__.get("_hints_off", 0)
    ua = self.uniform_area and not (off & _lib.PM_COL_UNIFORM_AREA)  # (use_hints(uniform_area=False))
    d.reserved = _lib.PM_COLS_ALL_UNIFORM_AREA if ua else 0
    if ua and self.div3_proven and not (off & _HINT_DIV3_OFF):
      d.reserved |= _lib.PM_COLS_DIV3_PROVEN
    d.z, d.b = self.z.ptr, self.b.ptr
    d.kappa, d.area, d.dAkappa = self.kappa.ptr, self.area.ptr, self.dAk.ptr
    d.bs, d.bbot, d.bzbot, d.N2min = self.bs.ptr, self.bbot.ptr, self.bzbot.ptr, self.N2min.ptr
    d.flags, d.ksel = self.flags.ptr, self.ksel.ptr
    d.nonfinite = self.nonfinite.ptr if self.nonfinite else None
    if self.kappa_base is not None:
      d.kappa_base, d.kappa_profile = self.kappa_base.ptr, self.kappa_profile.ptr
    return d

  def kernel_shape(self, lanes_per_col=0):
    """(lanes per column, levels per lane) the library uses for this batch."""
    g, p = C.c_int32(0), C.c_int32(0)
    check(lib.pm_column_kernel_shape(self.ncols, self.nz, int(lanes_per_col), C.byref(g),
                                     C.byref(p)))
    return g.value, p.value

  def kernel_name(self, nsteps, lanes_per_col=0, ops=_lib.PM_OP_TIMESTEP, horadv=False,
                  arith="exact"):
    """The kernel instantiation `steps(...)` of this shape launches (reporting only)."""
    if arith == "contracted":
      ops = ops | _lib.PM_OP_CONTRACTED
    buf = C.create_string_buffer(96)
    check(lib.pm_column_kernel_name(self.ncols, self.nz, int(lanes_per_col), int(nsteps),
                                    int(ops), int(bool(horadv)), buf, 96))
    name = buf.value.decode()
    # the batch-wide PM_COLS_ALL_UNIFORM_AREA hint selects the scalar-Area instantiation
    off = self.__dict__.get("_hints_off", 0)
    if (self.uniform_area and not (off & _lib.PM_COL_UNIFORM_AREA) and
        name.startswith("k_column_steps<64,") and name.endswith(",2,true>")):
      # ... and PM_COLS_DIV3_PROVEN the 3-instruction quotients (division form 6)
      div3 = self.div3_proven and not (off & _HINT_DIV3_OFF)
      name = (name[:-len(",2,true>")] + ",6,true,true>") if div3 else (name[:-1] + ",true>")
    return name

  def combine_forcing(self, wA, out=None):
    """weff = wA - d(A kappa)/dz of each column's coefficient set in use, on the device
    (`pm_column_weff`), for `steps(weff, ..., precombined=True)`: wA is static between two
    overturning updates, so loops that step once per launch need not re-read d(A kappa)/dz."""
    wA_d = self._dev(wA, "_wA")
    out = DeviceArray((self.ncols, self.nz)) if out is None else out
    d = self.descriptor()
    check(lib.pm_column_weff(C.byref(d), wA_d.ptr, out.ptr, _sh(self.stream)))
    return out

  def steps(self, wA, dt, nsteps=1, ops=_lib.PM_OP_TIMESTEP, vdx_in=None, b_in=None,
            lanes_per_col=0, precombined=False, arith="exact", psi_forcing=None,
            twobasin_forcing=None):
    """nsteps x (convect -> vertadvdiff -> horadv) with wA held fixed, one launch.
    precombined: `wA` is the output of `combine_forcing` (PM_OP_WEFF).
    psi_forcing=(Psi_iso [ncols, nz], Psi_SO [ncols/2, nz] or None) instead of `wA` (two-column
    ensembles, launches of >= 3 steps): the kernel forms wA_basin = (Psi_iso - Psi_SO) * 1e6 and
    wA_north = -Psi_iso * 1e6 itself (PM_OP_WA_PSI, example_twocol_plusSO.py:105-106).
    arith: "exact" (default: the reference's operation order, bit-identical to NumPy) or
    "contracted" (opt-in tolerance mode, PM_OP_CONTRACTED: ~1e-13 relative, ~3x faster)."""
    if precombined:
      ops = ops | _lib.PM_OP_WEFF
    if arith == "contracted":
      ops = ops | _lib.PM_OP_CONTRACTED
    elif arith != "exact":
      raise ValueError("arith must be 'exact' or 'contracted'")
    if vdx_in is not None and b_in is None:
      raise TypeError('b_in is needed if vdx_in is provided')  # column.py:348
    if twobasin_forcing is not None:
      # (iso [2n, nz], zon [2n, nz], Psi_SO [2n, nz]) of a three-column two-basin ensemble instead
      # of `wA`: the kernel forms the driver's forcing itself (PM_OP_WA_TWOBASIN, >= 3 steps)
      if vdx_in is not None or precombined or wA is not None or psi_forcing is not None:
        raise ValueError("twobasin_forcing replaces wA and excludes horadv / precombined forcing")
      iso, zon, pso = twobasin_forcing
      d = self.descriptor()
      check(lib.pm_column_steps(C.byref(d), iso.ptr, zon.ptr, pso.ptr, float(dt), int(nsteps),
                                int(ops | _lib.PM_OP_WA_TWOBASIN), int(lanes_per_col),
                                _sh(self.stream)))
      return
    if psi_forcing is not None:
      if vdx_in is not None or precombined or wA is not None:
        raise ValueError("psi_forcing replaces wA and excludes horadv / precombined forcing")
      psi_iso, psi_so = psi_forcing
      d = self.descriptor()
      check(lib.pm_column_steps(C.byref(d), psi_iso.ptr, None, psi_so.ptr if psi_so is not None else None,
                                float(dt), int(nsteps), int(ops | _lib.PM_OP_WA_PSI),
                                int(lanes_per_col), _sh(self.stream)))
      return
    wA_d = self._dev(wA, "_wA")
    vdx_d = self._dev(vdx_in, "_vdx")
    bin_d = self._dev(b_in, "_bin") if vdx_in is not None else None
    d = self.descriptor()
    check(lib.pm_column_steps(C.byref(d), wA_d.ptr if wA_d else None,
                              vdx_d.ptr if vdx_d else None,
                              bin_d.ptr if bin_d else None, float(dt), int(nsteps),
                              int(ops), int(lanes_per_col), _sh(self.stream)))
