"""Task sharding of the multi-task objective over the GPUs of one node (one process per GPU).

The multi-task NLL (hyperbo/gp_utils/objectives.py:181-195) is a mean over independent
sub-datasets, so tasks are partitioned statically (longest-processing-time first on n^3) and the
only exchange is ONE sum-all-reduce of [nll_sum, n_tasks, grad_sum] (<= ~10 KB) per evaluation.
Communicators:
  RcclComm      -- libhbo's RCCL binding (hbo_comm_*), device buffers over xGMI; the unique id is
                   distributed by any bootstrap callable (e.g. torch.distributed broadcast_object).
  TorchDistComm -- torch.distributed all_reduce (gloo on CPU for tests, nccl==RCCL on GPU).
  LocalComm     -- single process.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, Hashable, List, Sequence

import numpy as np


def lpt_partition(sizes: Dict[Hashable, int], num_shards: int, power: float = 3.0) -> List[List[Hashable]]:
  """Greedy longest-processing-time assignment of tasks (cost n^power) to shards; deterministic."""
  if num_shards <= 0:
    raise ValueError('num_shards must be positive')
  order = sorted(sizes.items(), key=lambda kv: (-float(kv[1])**power, str(kv[0])))
  loads = [0.0] * num_shards
  shards: List[List[Hashable]] = [[] for _ in range(num_shards)]
  for key, n in order:
    s = min(range(num_shards), key=lambda i: (loads[i], i))
    shards[s].append(key)
    loads[s] += float(n)**power
  return shards


def shard_dataset(dataset, rank: int, world_size: int, exclude_aligned: bool = True):
  """Returns this rank's sub-dict of `dataset` (same selection rule as objectives.py:181-185)."""
  from hyperbo_amd.gp_utils import objectives
  items = objectives.included_sub_datasets(dataset, exclude_aligned)
  sizes = {k: s.x.shape[0] for k, s in items}
  mine = set(lpt_partition(sizes, world_size)[rank])
  return {k: s for k, s in items if k in mine}


class LocalComm:
  rank, world_size = 0, 1

  def allreduce_sum(self, buf: np.ndarray) -> np.ndarray:
    return np.asarray(buf, dtype=np.float64)


class TorchDistComm:
  """torch.distributed is plumbing only (rendezvous + all_reduce); no tensors elsewhere."""

  def __init__(self, device=None, group=None):
    """`group`: a torch.distributed process group (e.g. one created with backend='nccl', which is RCCL on ROCm --
    then pass the rank's `device` so that the buffer travels over xGMI); default: the default group, host tensors."""
    import torch.distributed as dist
    if not dist.is_initialized():
      raise RuntimeError('torch.distributed is not initialised')
    self._dist = dist
    self.rank, self.world_size = dist.get_rank(), dist.get_world_size()
    self.device = device
    self.group = group

  def allreduce_sum(self, buf: np.ndarray) -> np.ndarray:
    import torch
    t = torch.from_numpy(np.ascontiguousarray(buf, dtype=np.float64).copy())
    if self.device is not None:
      t = t.to(self.device)
    self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
    return t.cpu().numpy()


class RcclComm:
  """RCCL all-reduce through libhbo (ncclAllReduce on the context's stream, xGMI)."""

  def __init__(self, ctx, rank: int, world_size: int, bcast_bytes: Callable[[bytes], bytes]):
    from hyperbo_amd import _native as nat
    self._nat, self.ctx = nat, ctx
    self.rank, self.world_size = rank, world_size
    uid = C.create_string_buffer(nat.UNIQUE_ID_BYTES)
    if rank == 0:
      rc = nat.lib().hbo_comm_unique_id(uid)
      if rc != nat.HBO_OK:
        raise nat.HboError(rc, (nat.lib().hbo_last_error(None) or b'').decode())
    uid_bytes = bcast_bytes(bytes(uid.raw))
    buf = C.create_string_buffer(uid_bytes, nat.UNIQUE_ID_BYTES)
    ctx.check(nat.lib().hbo_comm_init(ctx.handle, rank, world_size, buf), allow_not_pd=False)

  def allreduce_sum(self, buf: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(buf, dtype=np.float64).copy()
    self.ctx.check(self._nat.lib().hbo_comm_allreduce_sum(
        self.ctx.handle, a.ctypes.data_as(C.POINTER(C.c_double)), a.size), allow_not_pd=False)
    return a

  def close(self):
    self._nat.lib().hbo_comm_destroy(self.ctx.handle)


def sharded_mean_nll(local_nll_sum: float, local_count: int, local_grad_sum: np.ndarray, comm):
  """All-reduce [nll_sum, count, grad_sum] and return the mean over ALL tasks (objectives.py:192-195)."""
  buf = np.concatenate([[float(local_nll_sum), float(local_count)], np.asarray(local_grad_sum, dtype=np.float64)])
  buf = comm.allreduce_sum(buf)
  total, count, grad = buf[0], buf[1], buf[2:]
  if count <= 0:
    return 0.0, grad * 0.0, 0
  return total / count, grad / count, int(round(count))
