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


class DeviceTimeSeries(object):
  """Diagnostic time series resident on the device ("diagnostic ring buffer
  on device"): the reference preallocates `(levels, n_diag)` arrays and fills column k at every
  sample (run_JansenNadeau_2018.py:192-198, 218-226); here record k is a row of ONE device
  array `[n_rec][field][selected member][levels]`, appended by ONE row-gather launch on the
  compute stream (pm_rows_pack: the selected members only, gathered on the device), and brought
  to page-locked host memory by asynchronous copies on a side stream -- every `flush_every`
  records, or once at the end.  The stepping never waits for the host.

  fields: [(name, nlev)];  sources at `append`: {name: DeviceArray | device address}, rows =
  members of the ensemble, `nlev` doubles apart."""

  def __init__(self, fields, n_rec, n_members, members=None, stream=None, flush_every=0):
    from .device import DeviceArray, Event, PinnedArray, Stream
    self.fields = [(str(k), int(v)) for k, v in fields]
    self.n_rec, self.stream = int(n_rec), stream
    members = np.arange(n_members) if members is None else np.asarray(members, dtype=np.int64)
    if members.size and (members.min() < 0 or members.max() >= n_members):
      raise ValueError("recorded member outside the ensemble")
    self.members = members
    self.nsel = int(members.size)
    all_rows = self.nsel == n_members and np.array_equal(members, np.arange(n_members))
    self._sel = None if all_rows else DeviceArray.from_host(members.astype(np.int32),
                                                            stream=stream)
    self.offsets, off = {}, 0
    for name, nlev in self.fields:
      self.offsets[name] = off
      off += self.nsel * nlev
    self.rec = off  # doubles per record
    self.flush_every = int(flush_every)
    self.dev = DeviceArray((max(self.n_rec, 1), max(self.rec, 1)))
    self.host = PinnedArray((max(self.n_rec, 1), max(self.rec, 1)))
    self._side = Stream()
    self._ev = Event()
    self.count = 0      # records appended
    self._flushed = 0   # records whose copy has been issued

  def append(self, sources, k=None):
    """Record k (default: the next one): one launch, no synchronisation."""
    from .device import rows_pack, _addr
    k = self.count if k is None else int(k)
    if not (0 <= k < self.n_rec):
      return False
    base = self.dev.ptr + 8 * k * self.rec
    rows_pack([(_addr(sources[name]), base + 8 * self.offsets[name], nlev, nlev)
               for name, nlev in self.fields], self.nsel, sel=self._sel, stream=self.stream)
    self.count = max(self.count, k + 1)
    if self.flush_every > 0 and self.count - self._flushed >= self.flush_every:
      self.flush()
    return True

  def flush(self):
    """Issue the device-to-host copy of the records appended since the last flush (side
    stream, behind an event of the compute stream); returns at once."""
    from ._lib import check, lib
    from .device import download_async
    if self.count <= self._flushed:
      return
    self._ev.record(self.stream)
    check(lib.pm_stream_wait_event(self._side.handle, self._ev.handle))
    lo, hi = self._flushed, self.count
    download_async(self.dev.ptr + 8 * lo * self.rec, 8 * (hi - lo) * self.rec, self.host,
                   self._side, offset=8 * lo * self.rec)
    self._flushed = hi

  def wait(self):
    """Every record appended so far is in host memory when this returns."""
    self.flush()
    self._side.sync()

  def series(self):
    """{name: [nsel, nlev, n_rec]} -- the reference's (levels, n_diag) layout per member; records
    never written stay zero, as in the reference's preallocated arrays."""
    self.wait()
    h = self.host.array
    out = {}
    for name, nlev in self.fields:
      a = np.zeros((self.nsel, nlev, self.n_rec))
      o = self.offsets[name]
      if self.count:
        blk = h[:self.count, o:o + self.nsel * nlev].reshape(self.count, self.nsel, nlev)
        a[:, :, :self.count] = np.transpose(blk, (1, 2, 0))
      out[name] = a
    return out


class JN2018Diagnostics(object):
  """Attach to a JN2018Ensemble: `ens.recorder = JN2018Diagnostics(ens, Diag_iters, total)`.

  Samples are taken where the reference takes them: at iterations that are multiples of
  Diag_iters, right after the MOC update and before the step (:204-226).  The seven sampled
  fields go into a device-resident time series (`DeviceTimeSeries`: one row-gather launch per
  sample, asynchronous copies to page-locked memory); `AMOC`, `AMOC_b`, `bgrid`, `b_basin`,
  `b_north`, `bs_SO`, `Psi_SO` -- arrays [recorded member, levels, n_diag] -- are assembled on
  first access.  `members`: record these members only (default all)."""

  _FIELDS = ("AMOC", "AMOC_b", "bgrid", "b_basin", "b_north", "bs_SO", "Psi_SO")

  def __init__(self, ens, Diag_iters, total_iters, members=None, flush_every=0):
    if Diag_iters % ens.M != 0:
      raise ValueError("Diag_iters must be a multiple of MOC_up_iters (:96)")
    self.ens, self.Diag_iters = ens, int(Diag_iters)
    self.members = np.arange(ens.n) if members is None else np.asarray(members)
    nd = int(total_iters / Diag_iters)
    nz, ny, nb = ens.nz, ens.ny, ens.nb
    self.nd = nd
    self.ts = DeviceTimeSeries(
        [("AMOC", nz), ("AMOC_b", nb), ("bgrid", nb), ("b_basin", nz), ("b_north", nz),
         ("bs_SO", ny), ("Psi_SO", nz)], nd, ens.n, members=self.members, stream=ens.stream,
        flush_every=flush_every)
    self._series, self._series_count = None, -1

  def maybe_record(self, ii):
    if ii % self.Diag_iters != 0:
      return
    k = int(ii / self.Diag_iters)
    if k >= self.nd:
      return
    e = self.ens
    self.ts.append(dict(AMOC=e.tw.Psi, AMOC_b=e.tw.psib, bgrid=e.tw.bgrid,
                        b_basin=e.cols.b.ptr, b_north=e.cols.b.ptr + e._off,
                        bs_SO=e.ml.bs, Psi_SO=e.so.Psi), k=k)

  def __getattr__(self, name):
    if name in JN2018Diagnostics._FIELDS:
      if self._series is None or self._series_count != self.ts.count:
        self._series, self._series_count = self.ts.series(), self.ts.count
      return self._series[name]
    raise AttributeError(name)

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
