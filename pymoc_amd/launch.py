"""One process per GPU without torch: `python -m pymoc_amd.launch --nproc N script.py [args]`.

Starts N children of `script.py`, rank r with RANK=r, LOCAL_RANK=r, WORLD_SIZE=N,
MASTER_ADDR / MASTER_PORT and a fresh PYMOC_RUN_ID (the key of the RCCL id rendezvous,
`pymoc_amd.sharding.rendezvous_path`), i.e. the same environment contract as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N`, which remains usable.
The launcher itself never touches the GPU: every child is started before any HIP call is
made in this process, so no GPU-initialised process is ever replaced by an exec.
Exit code: the first non-zero child code (the remaining children are terminated), else 0.
"""
import argparse
import os
import signal
import socket
import subprocess
import sys
import time
import uuid


def free_port(addr="127.0.0.1"):
  s = socket.socket()
  s.bind((addr, 0))
  port = s.getsockname()[1]
  s.close()
  return port


def child_env(rank, world, addr, port, run_id, base=None):
  env = dict(os.environ if base is None else base)
  env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
             LOCAL_WORLD_SIZE=str(world), MASTER_ADDR=addr, MASTER_PORT=str(port),
             PYMOC_RUN_ID=run_id)
  env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs here
  env.setdefault("OMP_NUM_THREADS", "1")
  return env


def spawn(argv, nproc, addr="127.0.0.1", port=None, run_id=None, poll_s=0.05):
  """Run `argv` as nproc ranks; returns the job's exit code."""
  port = free_port(addr) if not port else int(port)
  run_id = run_id or uuid.uuid4().hex[:12]
  procs = [subprocess.Popen(argv, env=child_env(r, nproc, addr, port, run_id))
           for r in range(nproc)]
  code = 0
  try:
    live = list(procs)
    while live:
      for p in list(live):
        rc = p.poll()
        if rc is None:
          continue
        live.remove(p)
        if rc != 0 and code == 0:
          code = rc
          for q in live:  # one rank failed: the others would wait in a collective forever
            q.send_signal(signal.SIGTERM)
      time.sleep(poll_s)
  except KeyboardInterrupt:
    code = 130
    for p in procs:
      if p.poll() is None:
        p.send_signal(signal.SIGTERM)
  finally:
    deadline = time.time() + 10
    for p in procs:
      while p.poll() is None and time.time() < deadline:
        time.sleep(poll_s)
      if p.poll() is None:
        p.kill()
  return code


def main(argv=None):
  ap = argparse.ArgumentParser(prog="python -m pymoc_amd.launch")
  ap.add_argument("--nproc", type=int, required=True, help="ranks = GPUs of this node")
  ap.add_argument("--master-addr", default="127.0.0.1")
  ap.add_argument("--master-port", type=int, default=0, help="0 = pick a free port")
  ap.add_argument("script")
  ap.add_argument("args", nargs=argparse.REMAINDER)
  a = ap.parse_args(argv)
  return spawn([sys.executable, a.script] + a.args, a.nproc, a.master_addr, a.master_port)


if __name__ == "__main__":
  sys.exit(main())
