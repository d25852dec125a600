"""Drop-in `SO_ML` with the reference's Python API, computing on the GPU.

Same constructor, attributes, methods and error text as `pymoc.modules.SO_ML`
(src/pymoc/modules/SO_ML.py:5-303).  `self.bs` and `self.Psi_s` are REBOUND to new arrays
every step, as in the reference (SO_ML.py:232,259,269).  The Crank-Nicolson solve is a
Thomas sweep instead of the reference's dense np.linalg.inv (difference ~1e-16).
"""
import numpy as np

from ..device import DeviceArray
from ..so_ml import SOMLBatch
from ..utils import make_array


class SO_ML(object):
  def __init__(
      self,
      y=None,
      Ks=0.,
      h=50.,
      L=4e6,
      surflux=0.,
      rest_mask=0.,
      b_rest=0.,
      v_pist=1.5 / 86400.,
      bs=0.0,
      Psi_s=None
  ):
    if isinstance(y, np.ndarray):
      self.y = y
    else:
      raise TypeError('y needs to be numpy array providing (regular) grid')
    self.Ks = Ks
    self.h = h
    self.L = L
    self.surflux = make_array(surflux, self.y, 'surflux')
    self.rest_mask = make_array(rest_mask, self.y, 'rest_mask')
    self.b_rest = make_array(b_rest, self.y, 'b_rest')
    self.v_pist = v_pist
    self.Psi_s = Psi_s
    self.bs = make_array(bs, self.y, 'bs')
    self._batch = None

  def advdiff(self, b_basin, Psi_b, dt):
    nz = np.size(b_basin)
    if self._batch is None or self._batch.nz != nz:
      self._batch = SOMLBatch(self.y, nz, np.asarray(self.bs, dtype=np.float64))
      self._bb = DeviceArray((1, nz))
      self._pb = DeviceArray((1, nz))
    t = self._batch
    t.Ks, t.h, t.L, t.v_pist = float(self.Ks), float(self.h), float(self.L), float(self.v_pist)
    y0 = 0 * self.y
    t.bs.upload(np.asarray(self.bs, dtype=np.float64)[None, :] + y0)
    t.surflux.upload(np.asarray(self.surflux, dtype=np.float64)[None, :] + y0)
    t.rest_mask.upload(np.asarray(self.rest_mask, dtype=np.float64)[None, :] + y0)
    t.b_rest.upload(np.asarray(self.b_rest, dtype=np.float64)[None, :] + y0)
    self._bb.upload(np.asarray(b_basin, dtype=np.float64)[None, :])
    self._pb.upload(np.asarray(Psi_b, dtype=np.float64)[None, :])
    t.step(self._bb, self._pb, dt)
    if t.status.download()[0] == 1:
      # np.nonzero(Psi_mod)[0][0] / np.argwhere(Psi_b > 0)[0][0] on an empty result
      raise IndexError('index 0 is out of bounds for axis 0 with size 0')
    self.Psi_s = t.Psi_s.download()[0]
    self.bs = t.bs.download()[0]

  def timestep(self, b_basin=None, Psi_b=None, dt=1.):
    if not isinstance(b_basin, np.ndarray):
      raise TypeError('b_basin needs to be numpy array providing buoyancy levels in basin')
    if not isinstance(Psi_b, np.ndarray):
      raise TypeError(
          'Psi_b needs to be numpy array providing overturning at buoyancy levels given by b_basin'
      )
    self.advdiff(b_basin=b_basin, Psi_b=Psi_b, dt=dt)
