"""diagnostics and pickup files in the reference's wire format."""
import numpy as np
import pytest

from conftest import load_golden, relerr
from pymoc_amd import configs, diagnostics

pytestmark = pytest.mark.gpu


def _cfg():
  m = configs.jn2018_member(nz=81, dt_days=30.)
  cfg = dict(m)
  for k in ("b_basin0", "b_north0", "bs_SO0", "surflux", "b_rest", "rest_mask"):
    cfg[k] = m[k][None]
  return m, cfg


@pytest.mark.parametrize("fused", [True, False])
def test_diagfile_and_pickup_match_the_reference_script(gpu, tmp_path, fused):
  g = load_golden("jn2018_files")
  total, Diag = int(g["total_iters"]), int(g["Diag_iters"])
  m, cfg = _cfg()
  ens = gpu.JN2018Ensemble(cfg, fused=fused)
  ens.recorder = diagnostics.JN2018Diagnostics(ens, Diag, total)
  ens.run(total)
  dfile, pfile = str(tmp_path / "diags.npz"), str(tmp_path / "pickup.npz")
  ens.recorder.save_member(dfile, 0, m["tau"], m["KGM"])
  diagnostics.save_pickup(ens, pfile, member=0)
  d = np.load(dfile)
  assert sorted(d.files) == ["arr_%d" % i for i in range(11)] or len(d.files) == 11
  for i in range(11):
    ref = g["diag_%d" % i]
    got = d["arr_%d" % i]
    assert got.shape == ref.shape, i
    assert relerr(got, ref) <= 1e-10 or np.abs(ref).max() == 0, i
  p = np.load(pfile)
  for i in range(3):
    assert relerr(p["arr_%d" % i], g["pickup_%d" % i]) <= 1e-10, i
  # restart from the pickup exactly like `--pickup` and run 240 more steps
  ens2 = gpu.JN2018Ensemble(diagnostics.load_pickup(cfg, pfile), fused=fused)
  ens2.run(240)
  st = ens2.state()
  for i, k in enumerate(("b_basin", "b_north", "bs_SO")):
    assert relerr(st[k][0], g["restart_%d" % i]) <= 1e-10, k


def test_ensemble_pickup_roundtrip(gpu, tmp_path):
  c = configs.config5(N=64, nz=81, dt_days=30.)
  c["rest_mask"] = np.repeat(c["rest_mask"][None], 64, axis=0)
  a = gpu.JN2018Ensemble(c)
  a.run(36)
  f = str(tmp_path / "p.npz")
  diagnostics.save_pickup(a, f)
  p = np.load(f)
  st = a.state()
  assert np.array_equal(p["arr_0"], st["b_basin"]) and np.array_equal(p["arr_2"], st["bs_SO"])
  b = gpu.JN2018Ensemble(diagnostics.load_pickup(c, f))
  assert np.array_equal(b.state()["b_north"], st["b_north"])


class _BlockingRecorder(object):
  """Round 4's recorder, kept here as the check: seven synchronous whole-ensemble downloads per
  sample, members picked on the host."""

  def __init__(self, ens, Diag_iters, members):
    self.ens, self.Diag_iters, self.members, self.samples = ens, Diag_iters, members, []

  def maybe_record(self, ii):
    if ii % self.Diag_iters:
      return
    e, m = self.ens, self.members
    b = e.cols.get_b()
    self.samples.append(dict(
        AMOC=e.tw.Psi.download()[m], AMOC_b=e.tw.psib.download()[m],
        bgrid=e.tw.bgrid.download()[m], b_basin=b[:e.n][m], b_north=b[e.n:][m],
        bs_SO=e.ml.bs.download()[m], Psi_SO=e.so.Psi.download()[m]))


@pytest.mark.parametrize("members,flush_every", [(None, 0), ([5, 0, 17, 17], 2)])
def test_device_time_series_equals_blocking_downloads(gpu, members, flush_every):
  """The device-resident diagnostic time series (one row-gather launch per sample on the compute
  stream, asynchronous copies to pinned memory on a side stream) holds exactly what blocking
  downloads at the same instants return -- all members, and a selection gathered on the device."""
  c = configs.config5(N=24, nz=81, dt_days=30.)
  c["rest_mask"] = np.repeat(c["rest_mask"][None], 24, axis=0)
  Diag, total = 24, 24 * 7 + 5
  s = gpu.Stream()
  a, b = gpu.JN2018Ensemble(c, stream=s), gpu.JN2018Ensemble(c, stream=s)
  a.recorder = diagnostics.JN2018Diagnostics(a, Diag, total, members=members,
                                             flush_every=flush_every)
  sel = np.arange(24) if members is None else np.asarray(members)
  b.recorder = _BlockingRecorder(b, Diag, sel)
  a.run(total)
  b.run(total)
  assert a.recorder.nd == 7 and len(b.recorder.samples) == 8  # (the reference drops sample nd)
  for k in diagnostics.JN2018Diagnostics._FIELDS:
    got = getattr(a.recorder, k)
    assert got.shape[0] == sel.size and got.shape[2] == 7
    for j in range(7):
      assert np.array_equal(got[:, :, j], b.recorder.samples[j][k], equal_nan=True), (k, j)
  # the series can be read mid-run and the run continued (records never written stay zero)
  a2 = gpu.JN2018Ensemble(c, stream=s)
  a2.recorder = diagnostics.JN2018Diagnostics(a2, Diag, total, members=members)
  a2.run(2 * Diag + 1)
  mid = a2.recorder.b_basin.copy()
  assert np.array_equal(mid[:, :, :3], a.recorder.b_basin[:, :, :3]) and not mid[:, :, 3:].any()
  a2.run(total - 2 * Diag - 1)
  assert np.array_equal(a2.recorder.AMOC_b, a.recorder.AMOC_b, equal_nan=True)
