"""Drop-in `Psi_Thermwind` with the reference's Python API, computing on the GPU.

Same constructor, attributes (`Psi`, `bgrid`, `b1`, `b2`, `f`, `z`, `sol_init`), methods
and error text as `pymoc.modules.Psi_Thermwind` (src/pymoc/modules/psi_thermwind.py:7-232).
Profiles given as callables are sampled on `z` (between grid nodes the engine interpolates
linearly, the reference would call the function at SciPy's collocation midpoints:
SURVEY hazard H7); array and float profiles behave exactly like the reference.
"""
import numpy as np

from .. import _lib
from ..device import DeviceArray
from ..thermwind import ThermwindBatch
from ..utils import make_func, make_array


class Psi_Thermwind(object):
  def __init__(
      self,
      f=1.2e-4,    # Coriolis parameter (input)
      z=None,    # grid (input)
      sol_init=None,    # initial conditions for the reference's ODE solver (unused here)
      b1=None,    # buoyancy in the basin (input, output)
      b2=0.,    # buoyancy in the deep water formation region (input, output)
  ):
    self.f = f
    if isinstance(z, np.ndarray):
      self.z = z
      nz = np.size(z)
    else:
      raise TypeError('z needs to be numpy array providing grid levels')
    self.b1 = make_func(b1, self.z, 'b1')
    self.b2 = make_func(b2, self.z, 'b2')
    self.sol_init = np.zeros((2, nz)) if sol_init is None else sol_init
    self._batch = None
    self._nb = 0

  def _device(self, nb):
    nz = np.size(self.z)
    if self._batch is None or self._nb < nb:
      self._nb = max(int(nb), 1)
      self._batch = ThermwindBatch(self.z, 1, f=float(self.f), nb=self._nb)
      self._b1 = DeviceArray((1, nz))
      self._b2 = DeviceArray((1, nz))
    self._batch.f.upload(np.array([float(self.f)]))
    self._b1.upload(np.asarray(make_array(self.b1, self.z, 'b1'), dtype=np.float64) + 0 * self.z)
    self._b2.upload(np.asarray(make_array(self.b2, self.z, 'b2'), dtype=np.float64) + 0 * self.z)
    return self._batch

  def solve(self):
    t = self._device(self._nb or 1)
    t.update(self._b1, self._b2, ops=_lib.PM_TW_SOLVE, nb=1)
    self.Psi = t.Psi.download()[0]

  def Psib(self, nb=500):
    t = self._device(nb)
    t.Psi.upload(np.asarray(self.Psi, dtype=np.float64)[None, :])
    t.update(self._b1, self._b2, ops=_lib.PM_TW_PSIB, nb=nb)
    self.bgrid = t.bgrid.download()[0, :nb].copy()
    return t.psib.download()[0, :nb].copy()

  def Psibz(self, nb=500):
    t = self._device(nb)
    t.Psi.upload(np.asarray(self.Psi, dtype=np.float64)[None, :])
    t.update(self._b1, self._b2, ops=_lib.PM_TW_PSIB | _lib.PM_TW_PSIBZ, nb=nb)
    self.bgrid = t.bgrid.download()[0, :nb].copy()
    return [t.psibz1.download()[0], t.psibz2.download()[0]]

  def update(self, b1=None, b2=None):
    if b1 is not None:
      self.b1 = make_func(b1, self.z, 'b1')
    if b2 is not None:
      self.b2 = make_func(b2, self.z, 'b2')
