"""N>1 path on CPU: world_size-2 (and 3, ragged) gloo processes shard a config-2
ensemble with pymoc_amd.sharding, step their members with the ORACLE (the GPU engine
cannot run here), all-gather, and must reproduce the single-process result exactly."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from pymoc_amd import sharding

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["PM_ROOT"]); sys.path.insert(0, os.path.join(os.environ["PM_ROOT"], "tests"))
import oracle as O
from gloo_comm import GlooCommunicator
from pymoc_amd import configs, sharding
N, steps = int(os.environ["PM_N"]), 20
comm = GlooCommunicator()
lo, hi = sharding.member_range(N, comm.world, comm.rank)
c = configs.config2(N=N, members=(lo, hi))
b = O.column_ensemble_steps(c["z"], c["kappa"], c["Area"], c["b0"], c["wA"], c["dt"],
                            c["do_conv"], c["bs"], c["bbot"], c["N2min"], steps)
comm.barrier()
full = sharding.gather_members(comm, b, N)
tmax = comm.max_host(float(comm.rank + 1))
if comm.rank == 0:
  np.save(os.environ["PM_OUT"], full)
  assert tmax == comm.world
comm.close()
'''


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def test_member_range_partitions_exactly():
  for n in (0, 1, 7, 1024, 65536, 1000003):
    for w in (1, 2, 3, 8):
      blocks = [sharding.member_range(n, w, r) for r in range(w)]
      assert blocks[0][0] == 0 and blocks[-1][1] == n
      assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
      sizes = [hi - lo for lo, hi in blocks]
      assert max(sizes) - min(sizes) <= 1
  with pytest.raises(ValueError):
    sharding.member_range(10, 2, 2)


def test_single_communicator_gather():
  comm = sharding.SingleCommunicator()
  a = np.arange(12.).reshape(4, 3)
  assert np.array_equal(sharding.gather_members(comm, a, 4), a)


@pytest.mark.parametrize("world,N", [(2, 64), (3, 50)])
def test_gloo_sharded_ensemble_matches_single_process(tmp_path, world, N):
  import oracle as O
  from pymoc_amd import configs
  out = str(tmp_path / "full.npy")
  port = _free_port()
  procs = []
  for r in range(world):
    env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PM_ROOT=ROOT, PM_OUT=out,
               PM_N=str(N), OMP_NUM_THREADS="1")
    procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env))
  for p in procs:
    assert p.wait(timeout=300) == 0
  c = configs.config2(N=N)
  ref = O.column_ensemble_steps(c["z"], c["kappa"], c["Area"], c["b0"], c["wA"], c["dt"],
                                c["do_conv"], c["bs"], c["bbot"], c["N2min"], 20)
  assert np.array_equal(np.load(out), ref)
