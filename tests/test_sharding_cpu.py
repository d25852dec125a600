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


# ------------------------------------------------- rendezvous, launcher, diagnostic gather
def test_rendezvous_key_uses_launcher_values_only(monkeypatch):
  """Ranks started by different parents (any spawner, one wrapper shell per rank) must meet
  at the same file: the key may depend on MASTER_ADDR/PORT and the run id only."""
  env = dict(MASTER_ADDR="127.0.0.1", MASTER_PORT="29511", TORCHELASTIC_RUN_ID="abc")
  p0 = sharding.rendezvous_path(env)
  monkeypatch.setattr(os, "getppid", lambda: 424242)
  assert sharding.rendezvous_path(env) == p0
  assert sharding.rendezvous_path(dict(env, MASTER_PORT="29512")) != p0
  assert sharding.rendezvous_path(dict(env, TORCHELASTIC_RUN_ID="abd")) != p0
  assert sharding.rendezvous_path(dict(env, PYMOC_RUN_ID="x")) != p0
  assert sharding.rendezvous_path(dict(env, TORCHELASTIC_RESTART_COUNT="1")) != p0
  assert sharding.rendezvous_path(dict(env, PYMOC_RENDEZVOUS="/tmp/given")) == "/tmp/given"


def test_stale_id_file_is_ignored_and_replaced(tmp_path):
  import threading
  import time
  path = str(tmp_path / "id")
  with open(path, "wb") as f:
    f.write(b"S" * 128 + sharding._TAG.pack(os.getppid() + 1, 0.))
  old = time.time() - 3600
  os.utime(path, (old, old))  # leftover of a crashed run, an hour old
  with pytest.raises(TimeoutError):
    sharding.wait_for_id(path, 128, time.time(), timeout_s=0.3, trust_path=False)
  t = threading.Timer(0.2, sharding.publish_id, (path, b"N" * 128))
  t.start()
  assert sharding.wait_for_id(path, 128, time.time(), timeout_s=10) == b"N" * 128
  t.join()


def test_id_of_another_launcher_is_stale_even_when_fresh(tmp_path):
  """torch.distributed.run's static rendezvous gives every launch the run id 'none', and the
  driver's N = 2, 4, 8 series reuses one MASTER_PORT: a launch that died seconds ago leaves a
  FRESH file at the same path.  It carries the pid of ITS launcher, so a rank of the next launch
  (another parent) does not take it once it is older than the start-up slack; the id written by
  a sibling (same parent) is taken whatever its age, also by a rank that starts minutes late."""
  import time
  path = str(tmp_path / "id")
  # (1) written 40 s ago by a rank whose launcher was another process
  with open(path, "wb") as f:
    f.write(b"S" * 128 + sharding._TAG.pack(os.getppid() + 1, time.time() - 40))
  old = time.time() - 40
  os.utime(path, (old, old))
  with pytest.raises(TimeoutError):
    sharding.wait_for_id(path, 128, time.time(), timeout_s=0.3, trust_path=False)
  # (2) same age, written by a sibling: valid for a late starter
  sharding.publish_id(path, b"N" * 128)
  os.utime(path, (old, old))
  assert sharding.wait_for_id(path, 128, time.time(), timeout_s=2, trust_path=False) == b"N" * 128
  # (3) a path unique to the launch needs no test at all
  with open(path, "wb") as f:
    f.write(b"U" * 128 + sharding._TAG.pack(1, 0.))
  os.utime(path, (old - 3600, old - 3600))
  assert sharding.wait_for_id(path, 128, time.time(), timeout_s=2, trust_path=True) == b"U" * 128
  assert sharding.unique_key({"PYMOC_RUN_ID": "ab12"}) and sharding.unique_key({"TORCHELASTIC_RUN_ID": "7"})
  assert not sharding.unique_key({"TORCHELASTIC_RUN_ID": "none"}) and not sharding.unique_key({})


def test_launcher_environment_and_exit_code(tmp_path):
  from pymoc_amd import launch
  script = tmp_path / "child.py"
  script.write_text(
      "import os, sys\n"
      "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
      "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
      "open(os.path.join(sys.argv[1], 'r%d_%s_%s' % (r, os.environ['MASTER_PORT'],\n"
      "     os.environ['PYMOC_RUN_ID'])), 'w').close()\n"
      "sys.exit(int(sys.argv[2]) if r == w - 1 else 0)\n")
  out = tmp_path / "out"
  out.mkdir()
  assert launch.spawn([sys.executable, str(script), str(out), "0"], 3) == 0
  names = sorted(os.listdir(out))
  assert [n.split("_")[0] for n in names] == ["r0", "r1", "r2"]
  assert len({n.split("_", 1)[1] for n in names}) == 1  # same port and run id for all ranks
  assert launch.spawn([sys.executable, str(script), str(out), "7"], 2) == 7


def test_diagnostic_gather_single_rank_host_path():
  g = sharding.DiagnosticGather(None, 5, 5, [("a", 3), ("b", 2)], keep_history=True)
  a, b = np.arange(15.).reshape(5, 3), -np.arange(10.).reshape(5, 2)
  assert g.due(0, 240) and g.due(480, 240) and not g.due(24, 240) and not g.due(5, None)
  g.gather(dict(a=a, b=b), step=0)
  last = g.last()
  assert np.array_equal(last["a"], a) and np.array_equal(last["b"], b)
  assert g.history[0][0] == 0 and g.ngathers == 1 and g.bytes_per_rank == 8 * 25
  g.wait()  # nothing in flight in host mode: a no-op
  with pytest.raises(ValueError):
    sharding.DiagnosticGather(None, 4, 5, [("a", 3)])
  with pytest.raises(ValueError):
    sharding.DiagnosticGather(None, 5, 5, [("a", 3)], mode="ring")
  with pytest.raises(ValueError):
    sharding.DiagnosticGather(None, 5, 5, [("a", 3)], mode="root", root=1)


DIAG_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["PM_ROOT"]); sys.path.insert(0, os.path.join(os.environ["PM_ROOT"], "tests"))
from oracle import drivers
from gloo_comm import GlooCommunicator
from pymoc_amd import configs, sharding
N, steps, D = int(os.environ["PM_N"]), 73, 24
comm = GlooCommunicator()
lo, hi = sharding.member_range(N, comm.world, comm.rank)
cfg = configs.config3(N=N, members=(lo, hi))
nz = cfg["z"].size
mode = os.environ.get("PM_MODE", "all")
diag = sharding.DiagnosticGather(comm, hi - lo, N, [(k, nz) for k in ("b_basin", "b_north", "Psi", "Psi_SO")],
                                 keep_history=True, mode=mode)
# the coupled loop of TwoColEnsemble with the ORACLE as the stepper (no GPU here): gather after
# the update of every step ii with ii % D == 0, and once more at the end
snaps = [ii + 1 for ii in range(steps) if ii % D == 0] + [steps]
runs = [drivers.run_twocol(configs.member(cfg, j, 3), steps, snaps) for j in range(hi - lo)]
for s in snaps:
  st = {k: np.stack([r[s][k] for r in runs]) if runs else np.zeros((0, nz))
        for k in ("b_basin", "b_north", "Psi", "Psi_SO")}
  diag.gather(st, step=s)
if mode == "root" and comm.rank != 0:  # only the writer rank holds gathered data
  assert diag.last() is None and all(h[1] is None for h in diag.history) and not diag.receives
if comm.rank == 0:
  np.savez(os.environ["PM_OUT"], steps=np.array([h[0] for h in diag.history]),
           **{"%s_%d" % (k, i): h[1][k] for i, h in enumerate(diag.history) for k in h[1]})
comm.close()
'''


@pytest.mark.parametrize("world,N,mode", [(2, 6, "all"), (3, 5, "all"), (3, 5, "root")])
def test_gloo_diagnostic_gather_cadence_matches_single_process(tmp_path, world, N, mode):
  """The coupled drivers' Diag_iters gather across ranks (ragged shards included) must
  deliver, at every cadence point, exactly the fields of the unsharded ensemble."""
  from oracle import drivers
  from pymoc_amd import configs
  out = str(tmp_path / "diag.npz")
  port = _free_port()
  procs = []
  for r in range(world):
    env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PM_ROOT=ROOT, PM_OUT=out,
               PM_N=str(N), PM_MODE=mode, OMP_NUM_THREADS="1")
    procs.append(subprocess.Popen([sys.executable, "-c", DIAG_WORKER], env=env))
  for p in procs:
    assert p.wait(timeout=300) == 0
  got = np.load(out)
  steps = list(got["steps"])
  assert steps == [1, 25, 49, 73, 73]
  cfg = configs.config3(N=N)
  runs = [drivers.run_twocol(configs.member(cfg, j, 3), 73, steps) for j in range(N)]
  for i, s in enumerate(steps):
    for k in ("b_basin", "b_north", "Psi", "Psi_SO"):
      ref = np.stack([r[s][k] for r in runs])
      assert np.array_equal(got["%s_%d" % (k, i)], ref), (s, k)


TWOBASIN_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["PM_ROOT"]); sys.path.insert(0, os.path.join(os.environ["PM_ROOT"], "tests"))
from oracle import drivers
from gloo_comm import GlooCommunicator
from pymoc_amd import configs, sharding
from pymoc_amd.ensembles import TwoBasinEnsemble
N, steps, D = int(os.environ["PM_N"]), 50, 24
comm = GlooCommunicator()
lo, hi = sharding.member_range(N, comm.world, comm.rank)
cfg = configs.config_twobasin(N=N, members=(lo, hi))
nz = cfg["z"].size
F = TwoBasinEnsemble.FIELDS
diag = sharding.DiagnosticGather(comm, hi - lo, N, [(k, nz) for k in F], keep_history=True,
                                 mode=os.environ["PM_MODE"])
# TwoBasinEnsemble.run's cadence with the ORACLE as the stepper: gather after the update that
# follows step ii when ii % D == 0, and once more at the end
snaps = [ii + 1 for ii in range(steps) if ii % D == 0] + [steps]
keys = ("tau", "K", "A_Pac", "A_Atl", "A_north")
def member(j):
  m = dict(cfg)
  for k in keys:
    m[k] = cfg[k][j]
  return m
runs = [drivers.run_twobasin(member(j), steps, set(snaps)) for j in range(hi - lo)]
for s in snaps:
  diag.gather({k: np.stack([r[s][k] for r in runs]) for k in F}, step=s)
if comm.rank == 0:
  np.savez(os.environ["PM_OUT"], steps=np.array([h[0] for h in diag.history]),
           **{"%s_%d" % (k, i): h[1][k] for i, h in enumerate(diag.history) for k in h[1]})
comm.close()
'''


@pytest.mark.parametrize("mode", ["all", "root"])
def test_gloo_twobasin_gather_world2(tmp_path, mode):
  """SURVEY 8f row N1 sharded: the seven fields twobasin_NadeauJansen.py samples (:124-133),
  gathered at the driver's cadence from two ranks, equal the unsharded members' fields."""
  from oracle import drivers
  from pymoc_amd import configs
  from pymoc_amd.ensembles import TwoBasinEnsemble
  N, world = 5, 2
  out = str(tmp_path / "tb.npz")
  port = _free_port()
  procs = []
  for r in range(world):
    env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PM_ROOT=ROOT, PM_OUT=out,
               PM_N=str(N), PM_MODE=mode, OMP_NUM_THREADS="1")
    procs.append(subprocess.Popen([sys.executable, "-c", TWOBASIN_WORKER], env=env))
  for p in procs:
    assert p.wait(timeout=300) == 0
  got = np.load(out)
  steps = list(got["steps"])
  assert steps == [1, 25, 49, 50]
  cfg = configs.config_twobasin(N=N)
  for j in range(N):
    m = dict(cfg)
    for k in ("tau", "K", "A_Pac", "A_Atl", "A_north"):
      m[k] = cfg[k][j]
    ref = drivers.run_twobasin(m, 50, set(steps))
    for i, s in enumerate(steps):
      for k in TwoBasinEnsemble.FIELDS:
        assert np.array_equal(got["%s_%d" % (k, i)][j], ref[s][k]), (j, s, k)
