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


DIAG_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["PM_ROOT"]); sys.path.insert(0, os.path.join(os.environ["PM_ROOT"], "tests"))
from gloo_comm import GlooCommunicator
import pymoc_amd
from pymoc_amd import configs, sharding
N, cfgno = int(os.environ["PM_N"]), int(os.environ["PM_CFG"])
comm = GlooCommunicator()
lo, hi = sharding.member_range(N, comm.world, comm.rank)
if cfgno == 4:
  cfg = dict(configs.config4(N=N, members=(lo, hi)), bvp_refine=8)
  ens = pymoc_amd.TwoColEnsemble(cfg, comm=comm, n_total=N, diag_iters=48, keep_history=True)
  ens.run(100)
elif cfgno == 6:
  cfg = configs.config_twobasin(N=N, members=(lo, hi))
  ens = pymoc_amd.TwoBasinEnsemble(cfg, comm=comm, n_total=N, diag_iters=48, keep_history=True)
  ens.run(100)
else:
  cfg = configs.config5(N=N, members=(lo, hi))
  ens = pymoc_amd.JN2018Ensemble(cfg, comm=comm, n_total=N, diag_iters=72, keep_history=True)
  ens.run(150)
ens.gather_diagnostics()
if comm.rank == 0:
  h = ens.diag.history
  np.savez(os.environ["PM_OUT"], steps=np.array([s for s, _ in h]),
           **{"%s_%d" % (k, i): d[k] for i, (s, d) in enumerate(h) for k in d})
comm.close()
'''


@pytest.mark.parametrize("cfgno,world,N", [(4, 2, 40), (5, 3, 10), (6, 2, 21)])
def test_sharded_coupled_driver_gathers_at_diag_cadence(gpu, tmp_path, cfgno, world, N):
  """Coupled drivers with a communicator: the Diag_iters gathers of a sharded run (2-3
  processes on the one GPU, gloo carrying the collective) equal, gather for gather and bit
  for bit, those of the unsharded ensemble (which packs on device and needs no collective)."""
  from pymoc_amd import configs
  out = str(tmp_path / "diag.npz")
  port = _free_port()
  procs = []
  for r in range(world):
    env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PM_ROOT=ROOT, PM_OUT=out,
               PM_N=str(N), PM_CFG=str(cfgno), OMP_NUM_THREADS="1")
    procs.append(subprocess.Popen([sys.executable, "-c", DIAG_WORKER], env=env))
  for p in procs:
    assert p.wait(timeout=300) == 0
  if cfgno == 4:
    ens = gpu.TwoColEnsemble(dict(configs.config4(N=N), bvp_refine=8), diag_iters=48,
                             keep_history=True)
    ens.run(100)
    want_steps = [0, 48, 96, 100]
  elif cfgno == 6:
    ens = gpu.TwoBasinEnsemble(configs.config_twobasin(N=N), diag_iters=48, keep_history=True)
    ens.run(100)
    want_steps = [0, 48, 96, 100]
  else:
    ens = gpu.JN2018Ensemble(configs.config5(N=N), diag_iters=72, keep_history=True)
    ens.run(150)
    want_steps = [0, 72, 144, 150]
  ens.gather_diagnostics()
  got = np.load(out)
  assert list(got["steps"]) == want_steps
  assert [s for s, _ in ens.diag.history] == want_steps
  fields = ens.FIELDS if cfgno == 6 else ("b_basin", "b_north", "Psi", "Psi_SO")
  for i, (s, d) in enumerate(ens.diag.history):
    for k in fields:
      # config 5 (N = 10) contains member 2, which the reference itself loses at step 37
      assert np.array_equal(got["%s_%d" % (k, i)], d[k], equal_nan=True), (s, k)
  # the last gather is the state itself
  st = ens.state()
  for k in fields:
    assert np.array_equal(ens.diag.history[-1][1][k], st[k], equal_nan=True), k


@pytest.mark.parametrize("cfgno", [4, 5])
def test_exchange_on_its_own_stream_equals_inline_exchange(gpu, cfgno):
  """Round 5: the diagnostic exchange runs on a communication stream behind an event of the
  compute stream, with two alternating send buffers, while the stepping goes on.  Every gather
  of such a run must equal, bit for bit, the gather of a run that packs and exchanges in line
  on the compute stream (round 4's behaviour) -- a gather every MOC interval here, so that a
  pack is always queued right behind the steps that follow the previous one."""
  from pymoc_amd import configs
  s = gpu.Stream()
  runs = []
  for overlap in (True, False):
    if cfgno == 4:
      ens = gpu.TwoColEnsemble(dict(configs.config4(N=64), bvp_refine=8), stream=s,
                               diag_iters=24, keep_history=True, gather_overlap=overlap)
      ens.run(200)
    else:
      ens = gpu.JN2018Ensemble(configs.config5(N=24), stream=s, diag_iters=36,
                               keep_history=True, gather_overlap=overlap)
      ens.run(300)
    ens.gather_diagnostics()
    runs.append(ens)
  ha, hb = runs[0].diag.history, runs[1].diag.history
  assert [x for x, _ in ha] == [x for x, _ in hb] and len(ha) >= 9
  for (sa, da), (_, db) in zip(ha, hb):
    for k in da:
      assert np.array_equal(da[k], db[k], equal_nan=True), (sa, k)
  st = runs[0].state()
  for k in ("b_basin", "b_north", "Psi", "Psi_SO"):
    assert np.array_equal(ha[-1][1][k], st[k], equal_nan=True), k
    assert np.array_equal(runs[0].diag.last()[k], st[k], equal_nan=True), k


def test_explicit_stream_equals_default_stream(gpu):
  """Every fill, upload, launch and download of a driver must be ordered on the stream the
  caller passes: a run on an explicit (non-blocking) Stream is bit-identical to the
  default-stream run (zero-fills on a private stream could be overtaken)."""
  from pymoc_amd import configs
  s = gpu.Stream()
  for make, steps in (
      (lambda st: gpu.TwoColEnsemble(dict(configs.config4(N=32), bvp_refine=8), stream=st), 60),
      (lambda st: gpu.JN2018Ensemble(configs.config5(N=16), stream=st), 80),
      (lambda st: gpu.ColumnThermwindEnsemble(configs.config1(), stream=st), 20),
      (lambda st: gpu.TwoBasinEnsemble(configs.config_twobasin(N=8), stream=st), 50)):
    a, b = make(None), make(s)
    a.run(steps)
    b.run(steps)
    sa, sb = a.state(), b.state()
    for k in sa:
      assert np.array_equal(sa[k], sb[k], equal_nan=True), k


@pytest.mark.parametrize("args,nfields", [
    (["--config", "4", "--members", "512", "--steps", "20", "--warmup", "2"], 4),
    (["--config", "4", "--members", "512", "--steps", "20", "--warmup", "2", "--gather", "all"], 4),
    (["--config", "5", "--members", "256", "--steps", "20", "--warmup", "2"], 4),
    (["--config", "5", "--members", "256", "--steps", "20", "--warmup", "2", "--gather-inline"], 4),
    (["--config", "2", "--steps", "3", "--warmup", "1", "--no-single-step"], 0),
])
def test_bench_under_torch_distributed_run_with_rccl_one_rank(gpu, args, nfields):
  """Rehearsal of the driver's N > 1 launch on the one GPU a test box has:
  `python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 --force-rccl ...` as a
  FRESH child process (the launcher starts the rank before anything in it touches the GPU).
  RCCL is initialised through the launcher's environment, its barrier / all-reduce bracket the
  timed region, and the Diag_iters all-gathers of {b_basin, b_north, Psi_AMOC, Psi_SO} run INSIDE
  the timed region as real RCCL collectives on device buffers, next to the two-stream update of
  config 4.  What this cannot show is more than one rank: no 1 -> 8 curve has been measured."""
  import importlib.util
  import json
  # (torch must NOT be imported into this process: its bundled ROCm runtime next to the system's
  # makes a later in-process RCCL initialisation fail with "no ROCm-capable device")
  if importlib.util.find_spec("torch") is None:
    pytest.skip("torch.distributed.run not available")
  port = _free_port()
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
         "--master-addr", "127.0.0.1", "--master-port", str(port),
         os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-rccl", "--no-cpu-baseline"] + args
  env = dict(os.environ, OMP_NUM_THREADS="1")
  for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
    env.pop(k, None)
  p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
  assert p.returncode == 0, p.stderr[-2000:]
  line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
  out = json.loads(line)
  assert out["n_gpus"] == 1 and out["value"] > 0 and out["roofline"]["frac"] > 0
  if nfields:
    members, nz = out["members_per_gpu"], out["nz"]
    assert out["gathers_in_timed_region"] >= 2   # the cadence gathers + the final one
    assert out["rccl_collectives_in_timed_region"] == out["gathers_in_timed_region"]
    # members_per_gpu x n_fields x nz x 8 B per rank and gather
    assert out["gather_bytes_per_rank"] == members * nfields * nz * 8
    assert set(out["nonfinite"]) <= {2, 1268}  # (config 5: the two members the REFERENCE loses)
