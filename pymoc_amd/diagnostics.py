"""Diagnostics and pickup files in the reference's wire format.

`run_JansenNadeau_2018.py` writes two positional `.npz` files:
  * diagnostics (:192-226, :268-272): `np.savez(diagfile, AMOC_save, AMOC_b_save,
    b_basin_save, b_north_save, bs_SO_save, z, bgrid_save, y, Psi_SO_save, tau, kapGM)` ->
    arr_0 .. arr_10, every time series shaped (levels, n_diag); `examples/Plot_overturning.py`
    reads exactly these;
  * pickup (:266-267): `np.savez(file, basin.b, north.b, channel.bs)` -> arr_0, arr_1, arr_2,
    re-read by `--pickup` (:61-64, :135-138).
For an ensemble the same files are written per member (byte-compatible with the reference's
readers) or once for the whole ensemble with a leading member axis.
"""
import numpy as np


class JN2018Diagnostics(object):
  """Attach to a JN2018Ensemble: `ens.recorder = JN2018Diagnostics(ens, Diag_iters, total)`.

  Samples are taken where the reference takes them: at iterations that are multiples of
  Diag_iters, right after the MOC update and before the step (:204-226)."""

  def __init__(self, ens, Diag_iters, total_iters, members=None):
    if Diag_iters % ens.M != 0:
      raise ValueError("Diag_iters must be a multiple of MOC_up_iters (:96)")
    self.ens, self.Diag_iters = ens, int(Diag_iters)
    self.members = np.arange(ens.n) if members is None else np.asarray(members)
    nd = int(total_iters / Diag_iters)
    n, nz, ny, nb = self.members.size, ens.nz, ens.ny, ens.nb
    self.AMOC = np.zeros((n, nz, nd))
    self.AMOC_b = np.zeros((n, nb, nd))
    self.bgrid = np.zeros((n, nb, nd))
    self.b_basin = np.zeros((n, nz, nd))
    self.b_north = np.zeros((n, nz, nd))
    self.bs_SO = np.zeros((n, ny, nd))
    self.Psi_SO = np.zeros((n, nz, nd))
    self.nd = nd

  def maybe_record(self, ii):
    if ii % self.Diag_iters != 0:
      return
    k = int(ii / self.Diag_iters)
    if k >= self.nd:
      return
    e, m = self.ens, self.members
    b = e.cols.get_b()
    self.AMOC[:, :, k] = e.tw.Psi.download(stream=e.stream)[m]
    self.AMOC_b[:, :, k] = e.tw.psib.download(stream=e.stream)[m]
    self.bgrid[:, :, k] = e.tw.bgrid.download(stream=e.stream)[m]
    self.b_basin[:, :, k] = b[:e.n][m]
    self.b_north[:, :, k] = b[e.n:][m]
    self.bs_SO[:, :, k] = e.ml.bs.download(stream=e.stream)[m]
    self.Psi_SO[:, :, k] = e.so.Psi.download(stream=e.stream)[m]

  def save_member(self, path, j, tau, kapGM):
    """The reference's diagfile for recorded member j (positional arr_0..arr_10)."""
    np.savez(path, self.AMOC[j], self.AMOC_b[j], self.b_basin[j], self.b_north[j],
             self.bs_SO[j], self.ens.tw.z_host, self.bgrid[j], self.ens.so.y_host,
             self.Psi_SO[j], tau, kapGM)

  def save_ensemble(self, path, tau, kapGM):
    """Same positional layout with a leading member axis on every time series."""
    np.savez(path, self.AMOC, self.AMOC_b, self.b_basin, self.b_north, self.bs_SO,
             self.ens.tw.z_host, self.bgrid, self.ens.so.y_host, self.Psi_SO, tau, kapGM,
             self.members)


def save_pickup(ens, path, member=None):
  """np.savez(path, basin.b, north.b, channel.bs): one member in the reference's layout, or
  the whole ensemble with a leading member axis."""
  b = ens.cols.get_b()
  bs = ens.ml.bs.download(stream=ens.stream)
  if member is None:
    np.savez(path, b[:ens.n], b[ens.n:], bs)
  else:
    np.savez(path, b[member], b[ens.n + member], bs[member])


def load_pickup(cfg, path):
  """A copy of `cfg` restarted from a pickup file, the way the script does (:135-138):
  b_basin, b_north, bs_SO replaced; everything else (incl. bbot = b[0], kappaeff) as at a
  cold start.  Accepts the reference's 1-D arrays or ensemble arrays."""
  p = np.load(path)
  out = dict(cfg)
  out['b_basin0'] = 1.0 * np.atleast_2d(p['arr_0'])
  out['b_north0'] = 1.0 * np.atleast_2d(p['arr_1'])
  bs = 1.0 * np.atleast_2d(p['arr_2'])
  bs[:, -1] = np.asarray(cfg['bs'], dtype=np.float64)  # bs_SO[-1] = bs (:152)
  out['bs_SO0'] = bs
  return out
