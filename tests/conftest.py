import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
  return np.load(os.path.join(GOLDEN, name + ".npz"))


def relerr(a, ref):
  """max-norm error relative to max|ref| (NaNs must coincide)."""
  a, ref = np.asarray(a, dtype=float), np.asarray(ref, dtype=float)
  assert a.shape == ref.shape, (a.shape, ref.shape)
  assert np.array_equal(np.isnan(a), np.isnan(ref)), "NaN pattern differs"
  scale = np.nanmax(np.abs(ref)) if np.isfinite(ref).any() else 0.0
  diff = np.nanmax(np.abs(a - ref)) if np.isfinite(ref).any() else 0.0
  return diff / scale if scale > 0 else diff


@pytest.fixture(scope="session")
def gpu():
  """The engine on cuda:0 -- fails loudly (no skip, no fallback) when unusable."""
  import pymoc_amd
  pymoc_amd._lib.require_device()
  return pymoc_amd


def digest_err(x, ref_digest):
  """Largest deviation of the {sum, sum of squares} digests of the rows of `x` [n, nlev] from the
  reference's (fixture G20, ensemble_digests.npz), in units of what a relative error `1` of every
  level would move them by: |d sum| / (nlev * rms), |d sumsq| / (2 * sumsq).  Returns [n]."""
  x = np.asarray(x, dtype=np.float64)
  nlev = x.shape[1]
  s, q = np.sum(x, axis=1), np.sum(x * x, axis=1)
  rs, rq = ref_digest[:, 0], ref_digest[:, 1]
  rms = np.sqrt(np.maximum(rq, 1e-300) / nlev)
  return np.maximum(np.abs(s - rs) / (nlev * rms), np.abs(q - rq) / (2 * np.maximum(rq, 1e-300)))
