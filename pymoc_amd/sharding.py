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
import struct
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


def rendezvous_path(env=None, rendezvous_dir="/tmp"):
  """File through which rank 0 hands the RCCL unique id to the other ranks of the node.

  The key is built from launcher-provided values only -- MASTER_ADDR, MASTER_PORT and a run
  id (torch.distributed.run's TORCHELASTIC_RUN_ID or pymoc_amd.launch's PYMOC_RUN_ID, plus
  the elastic restart count) -- so that any spawner works, including one wrapper shell per
  rank.  `PYMOC_RENDEZVOUS` overrides the whole path."""
  env = os.environ if env is None else env
  if env.get("PYMOC_RENDEZVOUS"):
    return env["PYMOC_RENDEZVOUS"]
  run_id = env.get("PYMOC_RUN_ID") or env.get("TORCHELASTIC_RUN_ID") or "none"
  key = "pymoc_rccl_%s_%s_%s_%s" % (env.get("MASTER_ADDR", "127.0.0.1"),
                                    env.get("MASTER_PORT", "0"), run_id,
                                    env.get("TORCHELASTIC_RESTART_COUNT", "0"))
  return os.path.join(rendezvous_dir, "".join(c if c.isalnum() or c in "._-" else "_"
                                               for c in key))


_TAG = struct.Struct("<qd")  # launcher pid (the parent all ranks of one launch share), time


def publish_id(path, payload):
  """Rank 0: replace whatever a crashed earlier run left at `path` with the new id, followed
  by the pid of this rank's parent (the launcher: torch.distributed.run's agent or
  pymoc_amd.launch start every rank of a launch from ONE process) and the time."""
  try:
    os.remove(path)
  except OSError:
    pass
  tmp = path + ".tmp%d" % os.getpid()
  with open(tmp, "wb") as f:
    f.write(payload + _TAG.pack(os.getppid(), time.time()))
  os.replace(tmp, path)


def unique_key(env=None):
  """True when the launcher gave this launch an id of its own (PYMOC_RUN_ID, or a
  TORCHELASTIC_RUN_ID other than torch.distributed.run's static-rendezvous default 'none'):
  the rendezvous path then cannot belong to an earlier launch."""
  env = os.environ if env is None else env
  return bool(env.get("PYMOC_RENDEZVOUS") or env.get("PYMOC_RUN_ID") or
              env.get("TORCHELASTIC_RUN_ID", "none") != "none")


def wait_for_id(path, nbytes, started, timeout_s=300.0, stale_slack_s=30.0, trust_path=None):
  """Other ranks: wait for a complete id file that is not the leftover of a crashed earlier
  launch.  A file is taken when (a) the path is unique to this launch (`unique_key`), or (b) it
  was written by a child of this rank's own parent -- the launcher's pid travels with the id --
  or, for spawners with one wrapper process per rank, (c) it is not older than this process by
  more than `stale_slack_s` (ranks of a launch start within seconds; rank 0 replaces leftovers
  when it starts).  Rule (b) is what `python -m torch.distributed.run` with its default run id
  relies on: consecutive launches on one MASTER_PORT (the driver's N = 1, 2, 4, 8 series) have
  different agents, so the id of a launch that died cannot be mistaken for this one's."""
  trust = unique_key() if trust_path is None else trust_path
  t0 = time.time()
  while True:
    try:
      st = os.stat(path)
      if st.st_size >= nbytes + _TAG.size:
        with open(path, "rb") as f:
          data = f.read(nbytes + _TAG.size)
        if len(data) == nbytes + _TAG.size:
          ppid, _ = _TAG.unpack(data[nbytes:])
          if trust or ppid == os.getppid() or st.st_mtime >= started - stale_slack_s:
            return data[:nbytes]
    except OSError:
      pass
    if time.time() - t0 > timeout_s:
      raise TimeoutError("no RCCL unique id at %s after %.0f s" % (path, timeout_s))
    time.sleep(0.02)


_PROCESS_START = time.time()


class RcclCommunicator(object):
  """RCCL through the C-ABI.  The 128-byte unique id travels from rank 0 to the other
  ranks of the node through a file (single node: shared /tmp), see `rendezvous_path`."""

  def __init__(self, rank=None, world=None, stream=None, rendezvous_dir="/tmp",
               timeout_s=300.0, path=None):
    from . import _lib
    from .device import DeviceArray
    self._lib = _lib
    r, w, _ = world_info()
    self.rank = r if rank is None else rank
    self.world = w if world is None else world
    self.stream = stream
    _lib.require_device()
    path = path or rendezvous_path(rendezvous_dir=rendezvous_dir)
    buf = C.create_string_buffer(128)
    if self.rank == 0:
      _lib.check(_lib.lib.pm_comm_unique_id(buf))
      publish_id(path, buf.raw)
    else:
      buf.raw = wait_for_id(path, 128, _PROCESS_START, timeout_s)
    h = C.c_void_p()
    _lib.check(_lib.lib.pm_comm_init(C.byref(h), self.world, self.rank, buf))
    self.handle = h
    # a communicator someone asked for explicitly runs its collectives even with ONE rank
    # (bench.py --force-rccl, the single-GPU rehearsal of the N > 1 path): DiagnosticGather
    self.always_collective = True
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

  def gather_root_device(self, send, recv, root=0, stream=None):
    """recv[world][count] on `root` <- send[count] of every rank (ncclSend / ncclRecv); `recv`
    may be None on the other ranks."""
    count = int(np.prod(send.shape))
    if self.rank == root and (recv is None or recv.nbytes != send.nbytes * self.world):
      raise ValueError("recv on the root must hold world x send")
    self._lib.check(self._lib.lib.pm_comm_gather_root(
        self.handle, send.ptr, recv.ptr if recv is not None else None, count, int(root),
        1 if self.always_collective else 0, self._sh(stream)))

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


class DiagnosticGather(object):
  """The one exchange step of the path: the gather of per-member diagnostic fields at the
  drivers' output cadence (`if ii % Diag_iters == 0`, run_JansenNadeau_2018.py:218-226) and
  at the end of a run.

  Every rank packs its fields into a send buffer laid out [field][pad][nlev] (pad = the
  largest shard, so ragged shards exchange equal counts) with ONE row-gather launch
  (pm_rows_pack) and issues ONE collective into recv[world][field][pad][nlev] -- fewer, larger
  messages suit the per-link-bound xGMI mesh; no host staging.

  OFF THE CRITICAL PATH (round 5).  The pack runs on the compute stream (it must see the state
  of this step); the collective runs on a COMMUNICATION STREAM of its own that waits for the
  pack's event, so stepping continues while the bytes move.  Two send buffers alternate: the
  pack of gather k+1 only waits for the collective of gather k-1 (which read the same buffer).
  `last()`, `history`, `wait()` synchronise with the communication stream.  `overlap=False`
  puts the collective back on the compute stream (the round-4 behaviour, kept for A/B runs).

  mode "all":  ncclAllGather -- every rank ends up with every member (the default; what a
               user who continues on all ranks with the gathered fields needs);
  mode "root": point-to-point gather to rank `root` (pm_comm_gather_root) -- what the
               reference's cadence needs (one process writes the output): each rank SENDS its
               block once instead of receiving world-1 blocks; `last()` / `history` hold data on
               the root only (None elsewhere).

  `keep_history`: every gather is also copied to page-locked host memory by an asynchronous
  device-to-host copy on the communication stream (no host synchronisation inside the loop).

  Communicators without `allgather_device` (the gloo test vehicle, which has no device) go
  through `allgather_host` on NumPy sources with the same layout, synchronously, which is what
  the CPU multi-process tests exercise.

  fields: sequence of (name, nlev)."""

  def __init__(self, comm, n_local, n_total, fields, stream=None, keep_history=False,
               mode="all", root=0, overlap=True):
    if mode not in ("all", "root"):
      raise ValueError("mode must be 'all' or 'root'")
    self.comm = comm if comm is not None else SingleCommunicator()
    self.fields = [(str(k), int(v)) for k, v in fields]
    self.n_local, self.n_total = int(n_local), int(n_total)
    self.stream = stream
    self.mode, self.root, self.overlap = mode, int(root), bool(overlap)
    world, rank = self.comm.world, self.comm.rank
    if not (0 <= self.root < world):
      raise ValueError("root %d outside world of %d" % (self.root, world))
    self.counts = [hi - lo for lo, hi in (member_range(self.n_total, world, r)
                                          for r in range(world))]
    if self.counts[rank] != self.n_local:
      raise ValueError("rank %d owns %d of %d members, not %d" %
                       (rank, self.counts[rank], self.n_total, self.n_local))
    self.pad = max(self.counts) if self.counts else 0
    self.offsets, off = {}, 0
    for name, nlev in self.fields:
      self.offsets[name] = off
      off += self.pad * nlev
    self.count = off  # doubles per rank
    self.keep_history = keep_history
    self._history = []   # finished records: (step, {field: [n_total, nlev]} | None)
    self._pending = []   # records still in flight: (step, PinnedArray)
    self.ngathers = 0
    self.ncollectives = 0  # device collectives issued (RCCL all-gathers / gathers to root)
    self._send = self._recv = None
    self._slot = 0
    self._last_dev = None  # device array holding the most recent gather
    self._host = None      # last gathered [world][count] in host mode
    self._host_valid = False
    self._comm_stream = None

  # ------------------------------------------------------------------- helpers
  @property
  def receives(self):
    """Does this rank end up with the gathered data?"""
    return self.mode == "all" or self.comm.rank == self.root

  def _collective_always(self):
    return bool(getattr(self.comm, "always_collective", False))

  def _collective(self):
    return self.comm.world > 1 or self._collective_always()

  def due(self, step, diag_iters):
    return diag_iters is not None and diag_iters > 0 and step % int(diag_iters) == 0

  # ---------------------------------------------------------------- device path
  def _device_setup(self):
    from .device import DeviceArray, Event, Stream
    if self._send is not None:
      return
    self._send = [DeviceArray.zeros((self.count,), stream=self.stream) for _ in range(2)]
    if self._collective() and self.receives:
      self._recv = DeviceArray((self.comm.world, self.count))
    self._comm_stream = Stream() if self.overlap else self.stream
    self._ev_packed = Event()
    self._ev_sent = [Event(), Event()]

  @staticmethod
  def _wait(stream, event):
    from ._lib import check, lib
    check(lib.pm_stream_wait_event(stream.handle if stream is not None else None, event.handle))

  def _gather_device(self, sources, step):
    from .device import PinnedArray, download_async, rows_pack, _addr
    self._device_setup()
    slot = self._slot
    self._slot ^= 1
    send, cs = self._send[slot], self._comm_stream
    if cs is not self.stream:
      # the collective (or host copy) that last read this send buffer, two gathers ago
      self._wait(self.stream, self._ev_sent[slot])
    rows_pack([(_addr(sources[name]), send.ptr + 8 * self.offsets[name], nlev, nlev)
               for name, nlev in self.fields], self.n_local, stream=self.stream)
    if cs is not self.stream:
      self._ev_packed.record(self.stream)
      self._wait(cs, self._ev_packed)
    out = send
    if self._collective():
      if self.mode == "all":
        self.comm.allgather_device(send, self._recv, cs)
      else:
        self.comm.gather_root_device(send, self._recv, self.root, cs)
      self.ncollectives += 1
      out = self._recv
    self._last_dev = out if self.receives else None
    if self.keep_history:
      if self.receives:
        pin = PinnedArray((out.nbytes // 8,))
        download_async(out.ptr, out.nbytes, pin, cs)
        self._pending.append((step, pin))
      else:
        self._pending.append((step, None))
    if cs is not self.stream:
      self._ev_sent[slot].record(cs)

  def wait(self):
    """Block until every gather issued so far has landed (end of a run; the bench's timed
    region ends behind it)."""
    if self._comm_stream is not None:
      self._comm_stream.sync()
    elif self._send is not None:
      from ._lib import check, lib
      check(lib.pm_stream_sync(None))

  # ------------------------------------------------------------------ host path
  def _to_host(self, src, nlev):
    """[n_local, nlev] host copy of a DeviceArray or of a raw device address."""
    from ._lib import check, lib
    from .device import _sh
    out = np.empty((self.n_local, nlev))
    ptr = src if isinstance(src, int) else src.ptr
    check(lib.pm_memcpy_d2h(out.ctypes.data, ptr, out.nbytes, _sh(self.stream)))
    return out

  def _gather_host(self, sources, step):
    buf = np.zeros(self.count)
    for name, nlev in self.fields:
      a = np.ascontiguousarray(sources[name], dtype=np.float64).reshape(self.n_local, nlev)
      o = self.offsets[name]
      buf[o:o + self.n_local * nlev] = a.ravel()
    g = np.asarray(self.comm.allgather_host(buf)).reshape(self.comm.world, self.count)
    self._host = g if self.receives else None
    self._host_valid = True
    if self.keep_history:
      self._history.append((step, self._assemble(g) if self.receives else None))

  def gather(self, sources, step=None):
    """sources: {field: DeviceArray | device address (int) | ndarray [n_local, nlev]}."""
    host = any(isinstance(sources[k], np.ndarray) for k, _ in self.fields)
    if host or (self.comm.world > 1 and not hasattr(self.comm, "allgather_device")):
      if not host:  # device state but a host-only communicator: stage through the host
        sources = {k: self._to_host(sources[k], nlev) for k, nlev in self.fields}
      self._gather_host(sources, step)
    else:
      self._host, self._host_valid = None, False
      self._gather_device(sources, step)
    self.ngathers += 1

  # --------------------------------------------------------------------- results
  def _assemble(self, g):
    g = np.asarray(g).reshape(-1, self.count)
    out = {}
    for name, nlev in self.fields:
      o = self.offsets[name]
      blk = g[:, o:o + self.pad * nlev].reshape(g.shape[0], self.pad, nlev)
      out[name] = assemble(blk, self.counts)
    return out

  def _drain(self):
    if self._pending:
      self.wait()
      for step, pin in self._pending:
        if pin is None:
          self._history.append((step, None))
        else:
          self._history.append((step, self._assemble(pin.array.copy())))
          pin.free()
      self._pending = []

  @property
  def history(self):
    """[(step, {field: [n_total, nlev]})] of every gather so far (keep_history); the dict is
    None on ranks that do not receive (mode 'root')."""
    self._drain()
    return self._history

  def last(self):
    """{field: [n_total, nlev]} of the most recent gather, shard padding removed (None on a
    rank that does not receive)."""
    if self._host_valid:
      return None if self._host is None else self._assemble(self._host)
    if self._send is None:
      raise RuntimeError("nothing gathered yet")
    if self._last_dev is None:
      return None
    self.wait()
    return self._assemble(self._last_dev.download(stream=self._comm_stream))

  @property
  def bytes_per_rank(self):
    return 8 * self.count


def make_communicator(**kw):
  """RCCL when launched with WORLD_SIZE > 1 (one process per GPU), else a no-op."""
  _, world, _ = world_info()
  return RcclCommunicator(**kw) if world > 1 else SingleCommunicator()
