"""bench.py's rank logic with WORLD_SIZE = 2 on CPU: both ranks must walk through the same
sequence of collectives (barrier / max_host / all-gather pairing) and only rank 0 may print.
The engine cannot run here, so the device objects are stubs (host arrays, no arithmetic) and
the communicator is torch.distributed/gloo; what is under test is bench.py's control flow --
`bench_config2` and `headline_coupled` -- i.e. that a first multi-GPU run cannot deadlock on a
collective only some ranks reach, and that the rank != 0 return paths are taken."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

WORKER = r'''
import json, os, sys, types
import numpy as np
sys.path.insert(0, os.environ["PM_ROOT"]); sys.path.insert(0, os.path.join(os.environ["PM_ROOT"], "tests"))
from gloo_comm import GlooCommunicator
from pymoc_amd import configs, sharding
import bench

LOG = []

class Comm(GlooCommunicator):
  """gloo + a log of every collective this rank entered"""
  def barrier(self, stream=None):
    LOG.append("barrier"); super().barrier(stream)
  def max_host(self, value):
    LOG.append("max"); return super().max_host(value)
  def allgather_host(self, arr):
    LOG.append("allgather"); return super().allgather_host(arr)
  def allgather_device(self, send, recv, stream=None):
    recv.a[...] = self.allgather_host(send.a).reshape(recv.a.shape)

class DeviceArray(object):
  def __init__(self, shape, dtype=np.float64):
    self.a = np.zeros(shape, dtype=dtype); self.shape = self.a.shape; self.nbytes = self.a.nbytes; self.ptr = 0
  @classmethod
  def from_host(cls, x, stream=None):
    d = cls(np.shape(x)); d.a[...] = x; return d
  def download(self, out=None, stream=None):
    return self.a.copy()

class Event(object):
  def record(self, stream): pass
  def elapsed_ms(self, other): return 1.0

class Stream(object):
  def sync(self): pass

class ColumnBatch(object):
  def __init__(self, z, kappa, area, b, **kw):
    self.b = DeviceArray.from_host(b); self.n = self.b.shape[0]
  def steps(self, wA, dt, n, lanes_per_col=0, **kw): pass
  def get_nonfinite(self): return np.zeros(self.n, dtype=np.int32)
  def get_b(self): return self.b.a.copy()
  def kernel_shape(self, lanes): return 64, 2
  def kernel_name(self, F, lanes, **kw): return "stub"

class Diag(object):
  def __init__(self, comm, n): self.comm, self.ngathers, self.bytes_per_rank, self.n = comm, 0, 8 * n, n
  def gather(self, sources=None, step=None):
    self.comm.allgather_host(np.zeros(self.n)); self.ngathers += 1
  def wait(self): pass

class Ensemble(object):
  M, nz, ny, nb, diag_iters = 36, 200, 51, 500, 360
  def __init__(self, cfg, comm=None, n_total=None, stream=None, **kw):
    self.n = len(cfg["members"]); self.ii = 0
    self.diag = Diag(comm, self.n) if comm is not None else None
  def run(self, nsteps):
    for _ in range(nsteps):
      if self.diag is not None and self.ii % self.diag_iters == 0:
        self.diag.gather()       # the cadence gather of the real drivers (DiagnosticGather.due)
      self.ii += 1
  def gather_diagnostics(self, step=None): self.diag.gather()
  def nonfinite_members(self): return np.zeros(0, dtype=int)

fake = types.SimpleNamespace(ColumnBatch=ColumnBatch, TwoColEnsemble=Ensemble, JN2018Ensemble=Ensemble,
                             synchronize=lambda: None)
comm = Comm()
env = dict(pymoc_amd=fake, configs=configs, DeviceArray=DeviceArray, Event=Event, stream=Stream(),
           comm=comm, rank=comm.rank, world=comm.world,
           make_gather=lambda cm, n, ntot, fields, st, mode, overlap: Diag(cm, n))
bench.kernel_breakdown = lambda config, env, members, nsteps, warm_blocks: ({}, {"bound": "stub"})
args = types.SimpleNamespace(members=8, nz=100, steps_per_launch=10, steps=3, warmup=1, lanes=0,
                             force_rccl=False, no_single_step=True, gather="root", gather_inline=False)
outs = {}
outs["config2"] = bench.bench_config2(args, env)
for c in (3, 4, 5):
  a = types.SimpleNamespace(**dict(vars(args), members=6))
  outs["config%d" % c] = bench.headline_coupled(c, a, env)
comm.barrier()
comm.close()
json.dump({"rank": comm.rank, "log": LOG,
           "printed": {k: (v is not None) for k, v in outs.items()},
           "value": {k: (v["value"] if v else None) for k, v in outs.items()},
           "n_gpus": {k: (v["n_gpus"] if v else None) for k, v in outs.items()}},
          open(os.environ["PM_OUT"] + ".%d" % comm.rank, "w"))
'''


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def test_bench_rank_logic_world2_collectives_pair_up(tmp_path):
  pytest.importorskip("torch")
  out = str(tmp_path / "ranks.json")
  port = _free_port()
  procs = []
  for r in range(2):
    env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PM_ROOT=ROOT, PM_OUT=out,
               OMP_NUM_THREADS="1")
    procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env))
  for p in procs:
    assert p.wait(timeout=300) == 0  # a collective reached by one rank only would hang here
  res = [json.load(open(out + ".%d" % r)) for r in range(2)]
  assert res[0]["log"] == res[1]["log"] and len(res[0]["log"]) > 20
  # rank 0 alone reports; whole-job value counts both ranks' members
  assert all(res[0]["printed"].values()) and not any(res[1]["printed"].values())
  assert all(v == 2 for v in res[0]["n_gpus"].values())
  assert res[0]["log"].count("allgather") >= 2 + 3 * 2  # config 2's warm-up + final gather, cadence / final gathers
