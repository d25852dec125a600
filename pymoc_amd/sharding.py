"""Ensemble sharding across the GPUs of one node: one process per GPU.

Members never interact, so rank r owns the contiguous block
`member_range(N, world, r)` and steps it with no communication.  The only collective
is the all-gather of per-member output at diagnostic time, done by RCCL over xGMI on
device buffers (`RcclCommunicator`, through the C-ABI `pm_comm_*`).  The host logic
here only needs a communicator with `rank`, `world`, `barrier`, `allgather_host` and
`max_host`; tests/gloo_comm.py supplies one over torch.distributed/gloo so the N>1 path
is covered on CPU (a test vehicle, not a compute fallback).
"""
import ctypes as C
import os
import time

import numpy as np


def world_info(env=None):
  env = os.environ if env is None else env
  return (int(env.get("RANK", "0")), int(env.get("WORLD_SIZE", "1")),
          int(env.get("LOCAL_RANK", env.get("RANK", "0"))))


def member_range(n_members, world, rank):
  """Contiguous block [lo, hi) of rank `rank`; blocks differ in size by at most one."""
  if not (0 <= rank < world):
    raise ValueError("rank %d outside world of %d" % (rank, world))
  base, extra = divmod(int(n_members), int(world))
  lo = rank * base + min(rank, extra)
  return lo, lo + base + (1 if rank < extra else 0)


def assemble(gathered, counts):
  """Drop the padding of an all-gather of equal-sized (padded) shards.

  gathered: [world, max_count, ...] ; counts: members actually owned by each rank."""
  return np.concatenate([gathered[r, :counts[r]] for r in range(len(counts))], axis=0)


class SingleCommunicator(object):
  rank, world = 0, 1

  def barrier(self, stream=None):
    pass

  def allgather_host(self, arr):
    return np.asarray(arr)[None, ...]

  def max_host(self, value):
    return float(value)

  def close(self):
    pass


class RcclCommunicator(object):
  """RCCL through the C-ABI.  The 128-byte unique id travels from rank 0 to the other
  ranks of the node through a file (single node: shared /tmp)."""

  def __init__(self, rank=None, world=None, stream=None, rendezvous_dir="/tmp",
               timeout_s=300.0):
    from . import _lib
    from .device import DeviceArray
    self._lib = _lib
    r, w, _ = world_info()
    self.rank = r if rank is None else rank
    self.world = w if world is None else world
    self.stream = stream
    _lib.require_device()
    key = "pymoc_rccl_%s_%s_%s" % (os.environ.get("MASTER_PORT", "0"),
                                   os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                                   os.getppid())
    path = os.path.join(rendezvous_dir, key)
    buf = C.create_string_buffer(128)
    if self.rank == 0:
      _lib.check(_lib.lib.pm_comm_unique_id(buf))
      tmp = path + ".tmp%d" % os.getpid()
      with open(tmp, "wb") as f:
        f.write(buf.raw)
      os.replace(tmp, path)
    else:
      t0 = time.time()
      while not os.path.exists(path):
        if time.time() - t0 > timeout_s:
          raise TimeoutError("no RCCL unique id at %s after %.0f s" % (path, timeout_s))
        time.sleep(0.05)
      with open(path, "rb") as f:
        buf.raw = f.read(128)
    h = C.c_void_p()
    _lib.check(_lib.lib.pm_comm_init(C.byref(h), self.world, self.rank, buf))
    self.handle = h
    self._scalar = DeviceArray((2,))
    self.barrier()
    if self.rank == 0:
      try:
        os.remove(path)
      except OSError:
        pass

  def _sh(self, stream):
    s = stream if stream is not None else self.stream
    return s.handle if s is not None else None

  def barrier(self, stream=None):
    self._lib.check(self._lib.lib.pm_comm_barrier(self.handle, self._sh(stream)))

  def allgather_device(self, send, recv, stream=None):
    """recv[world][count] <- send[count] (DeviceArrays, fp64)."""
    count = int(np.prod(send.shape))
    if recv.nbytes != send.nbytes * self.world:
      raise ValueError("recv must hold world x send")
    self._lib.check(self._lib.lib.pm_comm_allgather(self.handle, send.ptr, recv.ptr, count,
                                                    self._sh(stream)))

  def allgather_host(self, arr):
    from .device import DeviceArray
    arr = np.ascontiguousarray(arr, dtype=np.float64)
    send = DeviceArray.from_host(arr)
    recv = DeviceArray((self.world,) + arr.shape)
    self.allgather_device(send, recv)
    return recv.download(stream=self.stream)

  def max_host(self, value):
    self._scalar.upload(np.array([float(value), 0.0]), self.stream)
    self._lib.check(self._lib.lib.pm_comm_allreduce_max(
        self.handle, self._scalar.ptr, self._scalar.ptr + 8, 1, self._sh(None)))
    return float(self._scalar.download(stream=self.stream)[1])

  def close(self):
    if getattr(self, "handle", None):
      self._lib.lib.pm_comm_destroy(self.handle)
      self.handle = None


def gather_members(comm, local, n_members):
  """All-gather a per-member array [n_local, ...] into the full [n_members, ...]."""
  local = np.ascontiguousarray(local, dtype=np.float64)
  counts = [member_range(n_members, comm.world, r) for r in range(comm.world)]
  counts = [hi - lo for lo, hi in counts]
  pad = max(counts)
  if local.shape[0] != counts[comm.rank]:
    raise ValueError("rank %d owns %d members, got %d" %
                     (comm.rank, counts[comm.rank], local.shape[0]))
  padded = np.zeros((pad,) + local.shape[1:])
  padded[:local.shape[0]] = local
  return assemble(comm.allgather_host(padded), counts)


def make_communicator(**kw):
  """RCCL when launched with WORLD_SIZE > 1 (one process per GPU), else a no-op."""
  _, world, _ = world_info()
  return RcclCommunicator(**kw) if world > 1 else SingleCommunicator()
