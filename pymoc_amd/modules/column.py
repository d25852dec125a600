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
from ..columns import ColumnBatch
from ..utils import make_func, make_array


class Column(object):
  def __init__(
      self,
      z=None,    # grid (input)
      kappa=None,    # diffusivity profile (input)
      bs=0.025,    # surface buoyancy bound. cond (input)
      bbot=0.0,    # bottom buoyancy boundary condition (input)
      bzbot=None,    # bottom strat. as alternative boundary condition (input)
      b=0.0,    # buoyancy profile (input, output)
      Area=None,    # horizontal area (can be function of depth)
      N2min=1e-7    # minimum strat. for conv adjustment
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
    self._batch = None
    self._kap_cached = None
    self._area_cached = None

  # ---- host-side helpers of the reference API (column.py:74-122)
  def Akappa(self, z):
    return self.Area(z) * self.kappa(z)

  def dAkappa_dz(self, z):
    return np.gradient(self.Akappa(z), z)

  def solve_equi(self, wA):
    raise NotImplementedError(
        'Column.solve_equi (SciPy solve_bvp equilibrium solver, column.py:187-208) is '
        'outside the timestep() path this engine replaces')

  # ---- device plumbing
  def _sync_to_device(self, do_conv):
    z = self.z
    kap = np.asarray(self.kappa(z), dtype=np.float64) + 0 * z
    area = np.asarray(self.Area(z), dtype=np.float64) + 0 * z
    if self._batch is None:
      self._batch = ColumnBatch(z, kap, area, np.asarray(self.b, dtype=np.float64),
                                report_nonfinite=False)
      self._kap_cached, self._area_cached = kap.copy(), area.copy()
    else:
      if not (np.array_equal(kap, self._kap_cached) and
              np.array_equal(area, self._area_cached)):
        self._batch.set_static(kap, area)
        self._kap_cached, self._area_cached = kap.copy(), area.copy()
      self._batch.set_b(self.b)
    self._batch.set_params(bs=float(self.bs), bbot=float(self.bbot),
                           bzbot=None if self.bzbot is None else float(self.bzbot),
                           N2min=float(self.N2min), do_conv=bool(do_conv))

  def _run(self, ops, do_conv, wA=None, dt=1., vdx_in=None, b_in=None):
    self._sync_to_device(do_conv)
    self._batch.steps(wA, dt, 1, ops, vdx_in, b_in)
    self.b[...] = self._batch.get_b()[0]

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
