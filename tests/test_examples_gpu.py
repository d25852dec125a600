"""The scripts under examples/ run end to end (short settings)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args):
  p = subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True,
                     timeout=300)
  assert p.returncode == 0, p.stdout + p.stderr
  return p.stdout


def test_twocol_user_loop_example(gpu):
  out = _run(["examples/twocol_user_loop.py", "--years", "20"])
  assert "max overturning" in out


def test_jn2018_sweep_example(gpu, tmp_path):
  out = _run(["examples/jn2018_sweep.py", "--members", "32", "--years", "10", "--nz", "81",
              "--dt-days", "30", "--out", str(tmp_path)])
  assert "32 members on 1 GPU(s)" in out and "0 non-finite members" in out
  d = np.load(os.path.join(str(tmp_path), "diags_member0.npz"))
  assert len(d.files) == 11 and d["arr_0"].shape[0] == 81  # the reference's positional layout
  p = np.load(os.path.join(str(tmp_path), "pickup_rank0.npz"))
  assert p["arr_0"].shape == (32, 81) and p["arr_2"].shape == (32, 51)
