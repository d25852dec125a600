"""Executes the ctypes stub printed in INTEGRATION.md (the binding a reference maintainer
would add) against the built library, so the document cannot drift from the C-ABI."""
import os
import re
import types

import numpy as np
import pytest

import oracle as O
from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_integration_md_stub_runs_and_matches_the_oracle(gpu):
  text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
  blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
  stub = next(b for b in blocks if "def column_timestep" in b)
  stub = stub.replace('C.CDLL("libpymoc_hip.so")',
                      'C.CDLL(%r)' % os.path.join(ROOT, "pymoc_amd", "libpymoc_hip.so"))
  ns = {}
  exec(compile(stub, "INTEGRATION.md", "exec"), ns)
  z = np.linspace(-4000., 0., 80)
  b0 = 0.03 * np.exp(z / 300.) - 0.002
  col = types.SimpleNamespace(z=z, b=b0.copy(), kappa=lambda x: 2e-5 + 1e-4 * np.exp(x / 500.),
                              Area=lambda x: 6e13 + 0 * x, bs=0.025, bbot=-0.002, bzbot=None,
                              N2min=1e-7)
  wA = 6e13 * 2e-8 * np.sin(z / 700.)
  ns["column_timestep"](col, wA, 86400. * 30, True)
  ref = O.column_timestep(z, col.kappa(z), col.Area(z), b0, wA, 86400. * 30, do_conv=True,
                          bs=0.025, bbot=-0.002, N2min=1e-7)
  assert np.array_equal(col.b, ref)
