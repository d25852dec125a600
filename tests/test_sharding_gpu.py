"""N>1 with the REAL engine: two processes (gloo for the host-side gather, both on the one
GPU of the test box -- RCCL refuses two ranks on one device) each step their shard of a
coupled two-column + SO ensemble with the HIP kernels; the gathered result must equal the
single-process run bit for bit (members never interact, so sharding cannot change anything)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from test_sharding_cpu import _free_port

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["PM_ROOT"]); sys.path.insert(0, os.path.join(os.environ["PM_ROOT"], "tests"))
from gloo_comm import GlooCommunicator
import pymoc_amd
from pymoc_amd import configs, sharding
N, steps = int(os.environ["PM_N"]), 60
comm = GlooCommunicator()
lo, hi = sharding.member_range(N, comm.world, comm.rank)
cfg = dict(configs.config4(N=N, members=(lo, hi)), bvp_refine=8)
ens = pymoc_amd.TwoColEnsemble(cfg)
ens.run(steps)
st = ens.state()
comm.barrier()
full = {k: sharding.gather_members(comm, st[k], N) for k in ("b_basin", "b_north", "Psi", "Psi_SO")}
if comm.rank == 0:
  np.savez(os.environ["PM_OUT"], **full)
comm.close()
'''


@pytest.mark.parametrize("world,N", [(2, 96), (3, 50)])
def test_gloo_sharded_gpu_ensemble_matches_single_process(gpu, tmp_path, world, N):
  from pymoc_amd import configs
  out = str(tmp_path / "full.npz")
  port = _free_port()
  procs = []
  for r in range(world):
    env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PM_ROOT=ROOT, PM_OUT=out,
               PM_N=str(N), OMP_NUM_THREADS="1")
    procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env))
  for p in procs:
    assert p.wait(timeout=300) == 0
  ens = gpu.TwoColEnsemble(dict(configs.config4(N=N), bvp_refine=8))
  ens.run(60)
  st = ens.state()
  got = np.load(out)
  for k in ("b_basin", "b_north", "Psi", "Psi_SO"):
    assert np.array_equal(got[k], st[k]), k
