"""torch.distributed (gloo) communicator for the CPU multi-process tests: same interface
as pymoc_amd.sharding.RcclCommunicator, host arrays only."""
import numpy as np


class GlooCommunicator(object):
  """torch.distributed (gloo) on host arrays -- used by the CPU multi-process tests."""

  def __init__(self):
    import torch.distributed as dist
    self._dist = dist
    if not dist.is_initialized():
      dist.init_process_group("gloo")
    self.rank, self.world = dist.get_rank(), dist.get_world_size()

  def barrier(self, stream=None):
    self._dist.barrier()

  def allgather_host(self, arr):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(arr))
    out = [torch.empty_like(t) for _ in range(self.world)]
    self._dist.all_gather(out, t)
    return np.stack([o.numpy() for o in out])

  def max_host(self, value):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
    return float(t[0])

  def close(self):
    if self._dist.is_initialized():
      self._dist.destroy_process_group()


