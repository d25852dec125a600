"""Batched drivers: the coupled time loops of the reference's example scripts, run for a
whole parameter-sweep ensemble with all state resident in HBM.

The reference has no driver class: the loops live in examples/*.py.  The
classes here reproduce those loops' order of operations and cadence exactly:
  ColumnThermwindEnsemble  examples/example_timestepping.py:73-80   (BASELINE config 1)
  TwoColEnsemble           examples/example_twocol.py:85-96         (config 3)
                           examples/example_twocol_plusSO.py:99-115 (config 4, with_so)
Each member is independent; `cfg` is a dict from `pymoc_amd.configs`.
"""
import numpy as np

from . import _lib
from .columns import ColumnBatch
from .device import DeviceArray
from .psi_so import PsiSOBatch
from .thermwind import ThermwindBatch

_TW_ALL = _lib.PM_TW_SOLVE | _lib.PM_TW_PSIB | _lib.PM_TW_PSIBZ


def _rows(v, n, nz):
  a = np.asarray(v, dtype=np.float64)
  if a.ndim == 0:
    return np.full((n, nz), a)
  if a.ndim == 1 and a.shape[0] == n and n != nz:
    return np.repeat(a[:, None], nz, axis=1)
  if a.ndim == 1:
    return np.broadcast_to(a, (n, nz)).copy()
  return a


def _vec(v, n):
  a = np.asarray(v, dtype=np.float64)
  return np.full(n, a) if a.ndim == 0 else a


class ColumnThermwindEnsemble(object):
  """One column per member, thermal wind against b2 = 0 re-solved after EVERY step and
  applied in z-space: wA = Psi * 1e6 (example_timestepping.py:73-80)."""

  def __init__(self, cfg, n=None, stream=None, lanes_per_col=0):
    z = cfg['z']
    nz = z.size
    b0 = np.atleast_2d(cfg['b0'])
    n = b0.shape[0] if n is None else n
    self.n, self.nz, self.dt = n, nz, float(cfg['dt'])
    self.lanes = lanes_per_col
    self.cols = ColumnBatch(z, _rows(cfg['kappa'], n, nz), _rows(cfg['Area'], n, nz),
                            _rows(b0, n, nz), bs=_vec(cfg['bs'], n), bbot=_vec(cfg['bbot'], n),
                            stream=stream)
    self.tw = ThermwindBatch(z, n, f=cfg['f'], nb=1, stream=stream, z_dev=self.cols.z)
    self.b2 = DeviceArray.zeros((n, nz))
    self.wA = DeviceArray.zeros((n, nz))
    self._solve()

  def _solve(self):
    self.tw.update(self.cols.b, self.b2, ops=_lib.PM_TW_SOLVE | _lib.PM_TW_WA_PSI,
                   wA1=self.wA, nb=1)

  def run(self, nsteps):
    for _ in range(int(nsteps)):
      self.cols.steps(self.wA, self.dt, 1, lanes_per_col=self.lanes)
      self._solve()

  def state(self):
    return dict(b=self.cols.get_b(), Psi=self.tw.Psi.download())


class TwoColEnsemble(object):
  """Basin + northern sinking column per member, coupled by the thermal-wind overturning
  mapped to isopycnal space every MOC_up_iters steps."""

  def __init__(self, cfg, stream=None, lanes_per_col=0):
    z = cfg['z']
    nz = z.size
    n = np.atleast_2d(cfg['b_basin0']).shape[0]
    self.n, self.nz = n, nz
    self.dt, self.M, self.nb = float(cfg['dt']), int(cfg['MOC_up_iters']), int(cfg['nb'])
    self.lanes = lanes_per_col
    self.stream = stream
    kap = _rows(cfg['kappa'], n, nz)
    # rows [0, n): basin columns, rows [n, 2n): northern columns
    self.cols = ColumnBatch(
        z, np.concatenate([kap, kap]),
        np.concatenate([_rows(cfg['A_basin'], n, nz), _rows(cfg['A_north'], n, nz)]),
        np.concatenate([_rows(cfg['b_basin0'], n, nz), _rows(cfg['b_north0'], n, nz)]),
        bs=np.concatenate([_vec(cfg['bs'], n), _vec(cfg['bs_north'], n)]),
        bbot=np.concatenate([_vec(cfg['bbot'], n), _vec(cfg['bbot'], n)]),
        do_conv=np.concatenate([np.zeros(n, bool), np.ones(n, bool)]), stream=stream)
    self.tw = ThermwindBatch(z, n, f=cfg['f'], nb=self.nb, stream=stream, z_dev=self.cols.z)
    self.wA = DeviceArray.zeros((2 * n, nz))
    self._off = n * nz * 8
    self.ii = 0
    self.so = None
    if 'y' in cfg and 'bs_SO' in cfg:  # example_twocol_plusSO.py:69-81
      ny = cfg['y'].size
      self.so = PsiSOBatch(z, cfg['y'], n, tau=cfg['tau'], KGM=cfg['KGM'], f=cfg['f'],
                           L=cfg['L'], c=cfg.get('c'), bvp_with_Ek=cfg.get('bvp_with_Ek', False),
                           bvp_refine=cfg.get('bvp_refine', 0), stream=stream,
                           z_dev=self.cols.z)
      self.bs_SO = DeviceArray.from_host(_rows(cfg['bs_SO'], n, ny) if np.ndim(cfg['bs_SO']) == 1
                                         else cfg['bs_SO'])
    self._update()  # AMOC.solve(); AMOC.Psibz() [; SO.solve()] on the initial profiles

  # device views
  @property
  def _b_basin(self):
    return self.cols.b.ptr

  @property
  def _b_north(self):
    return self.cols.b.ptr + self._off

  def _psi_so(self):
    return self.so.Psi if self.so is not None else None

  def _update(self):
    # SO.solve() and AMOC.solve() both read basin.b only, so the SO update may run first
    # and feed wAb = (Psi_iso_b - SO.Psi)*1e6 inside the thermal-wind launch
    if self.so is not None:
      self.so.update(self._b_basin, self.bs_SO)
    self.tw.update(self._b_basin, self._b_north, ops=_TW_ALL, Psi_SO=self._psi_so(),
                   wA1=self.wA.ptr, wA2=self.wA.ptr + self._off)

  def run(self, nsteps):
    """`for ii in range(nsteps): step both columns; if ii % MOC_up_iters == 0: update`,
    with the steps between two updates fused into one launch (wA is constant there)."""
    remaining = int(nsteps)
    while remaining > 0:
      nxt = self.ii if self.ii % self.M == 0 else (self.ii // self.M + 1) * self.M
      n = min(nxt - self.ii + 1, remaining)
      self.cols.steps(self.wA, self.dt, n, lanes_per_col=self.lanes)
      self.ii += n
      remaining -= n
      if (self.ii - 1) % self.M == 0:
        self._update()

  def state(self):
    b = self.cols.get_b()
    out = dict(b_basin=b[:self.n], b_north=b[self.n:], Psi=self.tw.Psi.download(),
               Psi_iso_b=self.tw.psibz1.download(), Psi_iso_n=self.tw.psibz2.download())
    if self.so is not None:
      out.update(Psi_SO=self.so.Psi.download(), Psi_Ek=self.so.Psi_Ek.download(),
                 Psi_GM=self.so.Psi_GM.download())
    return out

  def nonfinite_members(self):
    nf = self.cols.get_nonfinite()
    return np.nonzero(nf[:self.n] | nf[self.n:])[0]
