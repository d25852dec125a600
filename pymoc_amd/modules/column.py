"""Drop-in `Column` with the reference's Python API, computing on the GPU.

Same constructor, attributes, method names, argument meaning and error text as
`pymoc.modules.Column` (src/pymoc/modules/column.py:6-348); the arithmetic of
convect / vertadvdiff / horadv / timestep runs in the HIP kernel `pm_column_steps`
on a one-column batch.  `self.b` is the caller's NumPy array, updated IN PLACE (the
reference's aliasing behaviour, column.py:249), so user loops that read `basin.b` or poke
`basin.bbot` / `basin.kappa` between steps run unchanged.

Lazy stepping.  A user loop calls `timestep` many times with the same forcing between two
overturning updates (examples/example_twocol.py:85-96: 24 steps per update).  Such calls are
QUEUED and executed as one fused launch (one H2D, one `pm_column_steps(nsteps=k)`, one D2H)
the moment the state is needed:
  * `col.b` is read (the attribute is a property) or assigned;
  * the next call differs in anything -- wA values, dt, do_conv, horadv inputs, bs / bbot /
    bzbot / N2min, a re-assigned (or, for arrays, edited) kappa / Area;
  * any other pymoc_amd module object computes (`flush_all()`: they may read the array
    through an alias such as `AMOC.update(b1=basin.b)`);
  * `vertadvdiff` / `convect` / `horadv` / `solve_equi` are called, the queue is
    `MAX_QUEUE` steps long, or the object is deleted.
The queued steps are bit-identical to stepping one by one (the kernel's step-splitting
invariance, tests/test_column_gpu.py).  The one observable difference to write-through:
a variable that ALIASES the array and is READ directly (not through `col.b` or another
pymoc_amd object) between steps shows the last flushed state.  WRITES through such an alias
are detected at the next `timestep` call (the array is compared with its state when the queue
was opened) and applied after the steps queued before them, as the reference would -- except a
write of the very value the entry already had, which leaves no trace.  `LAZY = False` (or the
environment variable PYMOC_EAGER=1) restores a launch per call.
For throughput use `pymoc_amd.ColumnBatch` / the ensemble drivers: state stays in HBM there.
"""
import os
import weakref

import numpy as np

from .. import _lib
from ..utils import make_func, make_array

LAZY = os.environ.get("PYMOC_EAGER", "0") != "1"
MAX_QUEUE = 1 << 16
_pending = weakref.WeakSet()  # Columns with queued steps


def flush_all():
  """Execute the queued steps of every Column (called by the other module classes before
  they read user arrays)."""
  if _pending:
    for col in list(_pending):
      col._flush()


def _static_token(fn, z):
  """Change detector for a kappa / Area callable: its VALUES on the grid.  The reference
  evaluates kappa(z) and Area(z) in every step (column.py:122, :241-248), so anything that changes
  what the callable returns counts -- an in-place edit of the array a make_func closure aliases,
  a user callable reading a global or a closure variable the loop updates.  make_func closures
  over a float or an array are recognised and compared without a call; every other callable is
  called on z (O(nz) host work per timestep call)."""
  src = getattr(fn, "_pm_source", None)
  if isinstance(src, np.ndarray):
    return (id(fn), src.tobytes())
  if src is not None:
    return (id(fn), src)
  return (id(fn), np.asarray(fn(z), dtype=np.float64).tobytes())


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
    self._arena = None
    self._static_tok = None
    self._q = None  # queued identical timesteps: [count, wA, dt, do_conv, params, b at opening]
    self._b = make_array(b, self.z, 'b')
    self.bz = np.gradient(self._b, z)

  # ---- state: the caller's array, synchronised on access
  @property
  def b(self):
    if self._q is not None:
      self._flush()
    return self._b

  @b.setter
  def b(self, value):
    if self._q is not None:
      self._flush()
    self._b = value

  def __del__(self):
    try:
      if self._q is not None:
        self._flush()
    except Exception:
      pass

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
    self._flush()
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
    self._out = np.empty(nz)
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

  def _sync_statics(self, params):
    """Evaluate kappa(z), Area(z) and upload them with d(A kappa)/dz when they changed (first
    call, or the user re-assigned / edited kappa or Area).  Only called with no step queued."""
    if getattr(self, "_arena", None) is None or self._nz != self.z.size:
      self._alloc()
      self._static_tok = None
    if self._static_tok != params[4:]:
      z = self.z
      kap = np.asarray(self.kappa(z), dtype=np.float64) + 0 * z
      area = np.asarray(self.Area(z), dtype=np.float64) + 0 * z
      self._kap.upload(kap)
      self._area.upload(area)
      self._dAk.upload(np.gradient(area * kap, z))  # dAkappa_dz, column.py:96-122
      self._static_tok = params[4:]

  def _params(self):
    return (self.bs, self.bbot, self.bzbot, self.N2min, _static_token(self.kappa, self.z),
            _static_token(self.Area, self.z))

  def _run(self, ops, do_conv, wA=None, dt=1., vdx_in=None, b_in=None, nsteps=1, params=None):
    z, nz, h = self.z, self.z.size, None
    if params is None:  # immediate call: the coefficients as they are now
      params = self._params()
      self._sync_statics(params)
    # (queued calls uploaded theirs when the queue was opened: kappa / Area may have been
    # re-assigned since)
    bs, bbot, bzbot, N2min = params[:4]
    h = self._host
    h[0:nz] = self._b
    h[nz:2 * nz] = 0. if wA is None else wA
    if vdx_in is not None:
      h[2 * nz:3 * nz] = vdx_in
      h[3 * nz:4 * nz] = b_in
    h[4 * nz:4 * nz + 4] = (bs, bbot, 0. if bzbot is None else bzbot, N2min)
    flags = (_lib.PM_COL_DO_CONV if do_conv else 0) | (
        _lib.PM_COL_BZBOT if bzbot is not None else 0)
    h[4 * nz + 4:].view(np.int32)[0] = flags
    self._arena.upload(h)
    _lib.check(_lib.lib.pm_column_steps(
        self._C.byref(self._desc), self._wA_ptr,
        self._vdx_ptr if vdx_in is not None else None,
        self._bin_ptr if vdx_in is not None else None, float(dt), int(nsteps), int(ops), 0,
        None))
    direct = (self._b.flags.c_contiguous and self._b.flags.writeable and
              self._b.dtype == np.float64 and self._b.size == nz)
    _lib.check(_lib.lib.pm_memcpy_d2h(self._b.ctypes.data if direct else self._out.ctypes.data,
                                      self._arena.ptr, nz * 8, None))
    if not direct:
      self._b[...] = self._out

  def _flush(self):
    """Run the queued timesteps as one fused launch."""
    q, self._q = self._q, None
    _pending.discard(self)
    if q is not None:
      count, wA, dt, do_conv, params, b0 = q
      # A write into the array through an alias while steps were queued (`arr = col.b` once,
      # `arr[k] = ...` inside the loop): the reference would have stepped first and then taken
      # the write.  The queued steps run from the state the queue was opened with and the
      # entries written since (those that differ from it) are re-applied on top.
      edited = None
      if not np.array_equal(self._b, b0, equal_nan=True):
        edited = ~((self._b == b0) | (np.isnan(self._b) & np.isnan(b0)))
        vals = self._b[edited].copy()
        self._b[...] = b0
      # PM_OP_TIMESTEP without horadv inputs = convect + vertadvdiff on the fused fast path
      self._run(_lib.PM_OP_TIMESTEP, do_conv, wA=wA, dt=dt, nsteps=count, params=params)
      if edited is not None:
        self._b[edited] = vals

  # ---- the time-stepping API (column.py:210-348)
  def vertadvdiff(self, wA, dt, do_conv=False):
    wA = make_array(wA, self.z, 'wA')
    self._flush()
    self._run(_lib.PM_OP_VERTADVDIFF, do_conv, wA=wA, dt=dt)

  def convect(self):
    self._flush()
    self._run(_lib.PM_OP_CONVECT, True)

  def horadv(self, vdx_in, b_in, dt):
    vdx_in = make_array(vdx_in, self.z, 'vdx_in')
    b_in = make_array(b_in, self.z, 'b_in')
    self._flush()
    self._run(_lib.PM_OP_HORADV, False, dt=dt, vdx_in=vdx_in, b_in=b_in)

  def timestep(self, wA=0., dt=1., do_conv=False, vdx_in=None, b_in=None):
    wA = make_array(wA, self.z, 'wA')
    ops = _lib.PM_OP_CONVECT | _lib.PM_OP_VERTADVDIFF
    if vdx_in is not None and b_in is None:
      # the reference has already convected and stepped when it raises (column.py:336-348)
      self._flush()
      self._run(ops, do_conv, wA=wA, dt=dt)
      raise TypeError('b_in is needed if vdx_in is provided')
    if vdx_in is not None:
      vdx_in = make_array(vdx_in, self.z, 'vdx_in')
      b_in = make_array(b_in, self.z, 'b_in')
      self._flush()
      self._run(ops | _lib.PM_OP_HORADV, do_conv, wA=wA, dt=dt, vdx_in=vdx_in, b_in=b_in)
      return
    if not LAZY:
      self._run(ops, do_conv, wA=wA, dt=dt)
      return
    # queue: identical consecutive steps fuse into one launch
    params = self._params()
    q = self._q
    if q is not None:
      if (q[0] < MAX_QUEUE and q[2] == dt and q[3] == do_conv and q[4] == params and
          (q[1] is wA or np.array_equal(q[1], wA)) and
          np.array_equal(self._b, q[5], equal_nan=True)):  # (no write through an alias since)
        q[0] += 1
        return
      self._flush()
    # a new queue: coefficients are evaluated and uploaded now (the device is idle), wA is
    # snapshotted (the caller may reuse its buffer; the reference reads it during the call)
    self._sync_statics(params)
    self._q = [1, np.array(wA, dtype=np.float64, copy=True), dt, bool(do_conv), params,
               np.array(self._b, dtype=np.float64, copy=True)]
    _pending.add(self)
