"""Id-sharded embedding table over the GPUs of one node — the MI355X form of the reference's sync /
all-to-all path (native_training/distributed_ps_sync.py:95-287 lookup, :289-490 apply_gradients;
packing rules of runtime/ops/fused_reorder_by_indices.cc:38-123).

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).  GPU g owns
{fid : fid mod N == g} (distributed_ps.py:289, fused_reorder_by_indices.cc:121-123) as a complete
local MultiHashTable.  Per step and per rank, for a batch of B ids of one table:

  forward   dedup (first-occurrence order)            -> U unique ids, inverse[B]
            stable partition of the unique ids by shard (FusedReorderByIndices' shard-major order)
            all-to-all #1  per-shard counts  int64[N]   (distributed_ps_sync.py:132-159)
            all-to-all #2  ids               int64[M]
            local lookup of the M received ids (no insert)
            all-to-all #3  rows              fp32[M, D]  (:217-261)
            un-permute + scatter to the B occurrences
  backward  duplicate-gradient sum per unique id, permute to send order
            all-to-all #4  gradients         fp32[M, D]  (:357-479)
            owner applies the optimizer once per distinct id: contributions of different senders are
            added first (enable_grad_accumulation, tf_bridge.cc:270-310) — chosen over the
            reference default of N separate applications because it is order-independent.

xGMI is point-to-point (every GPU pair has its own link), so one all-to-all-v per tensor is the
natural pattern; the exchange moves (N-1)/N * (8 U + 2 * 4 D U) bytes per rank per step.

The exchange logic is device-agnostic torch code (plumbing); the table and dedup stages behind
``LocalBackend`` are the HIP kernels.  tests/test_sharded_gloo.py runs the same exchange code with
world_size 2 on CPU/gloo against a stand-in backend.
"""
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist


class LocalBackend:
  """What the exchange needs from the local engine.  The GPU implementation is HipBackend; the
  gloo test supplies a CPU stand-in with the same four methods."""

  dim: int

  def unique(self, ids: torch.Tensor):
    """-> (unique_ids[U] first-occurrence order, inverse[B])"""
    raise NotImplementedError

  def lookup(self, ids: torch.Tensor) -> torch.Tensor:
    raise NotImplementedError

  def segment_sum(self, grads: torch.Tensor, inverse: torch.Tensor, n_unique: int) -> torch.Tensor:
    raise NotImplementedError

  def optimize_accumulated(self, ids: torch.Tensor, grads: torch.Tensor, update_time: int,
                           global_step: int):
    """ids may repeat (one row per sender); gradients of equal ids are added, then ONE step."""
    raise NotImplementedError


class HipBackend(LocalBackend):
  """Local shard on this process's MI355X: MultiHashTable + DedupWorkspace (libmhte.so)."""

  def __init__(self, table, table_name: str):
    from monolith_amd import _lib
    from monolith_amd.distribution_ops import DedupWorkspace
    self._lib = _lib
    self.table = table
    self.name = table_name
    self.idx = table._index(table_name)  # pylint: disable=protected-access
    self.dim = table.get_table_dim_sizes()[self.idx]
    self.ws = DedupWorkspace(table._device)  # pylint: disable=protected-access
    lr0 = sum(table._slice_sizes[:self.idx])  # pylint: disable=protected-access
    self.lrs = np.ascontiguousarray(
        table.learning_rate[lr0:lr0 + table._slice_sizes[self.idx]])  # pylint: disable=protected-access
    self._u = None

  def unique(self, ids):
    self._u = self.ws.unique(ids, want_host_count=True)
    U = self._u.n_unique
    return self._u.unique_ids[:U], self._u.inverse

  def lookup(self, ids):
    out = torch.empty((ids.numel(), self.dim), dtype=torch.float32, device=ids.device)
    if ids.numel():
      self.table.table_lookup_n(self.idx, ids, None, out)
    return out

  def segment_sum(self, grads, inverse, n_unique):
    out = self.ws.segment_sum(grads, self._u, self.dim)
    return out[:n_unique]

  def optimize_accumulated(self, ids, grads, update_time, global_step):
    if ids.numel():
      self.table.table_optimize_n(self.idx, ids, None, grads, self.lrs, update_time, global_step,
                                  flags=self._lib.MHTE_SUM_DUPLICATES)


def shard_of(ids: torch.Tensor, num_shards: int) -> torch.Tensor:
  """floormod(id, N) — distributed_ps.py:289; equals the fused op's `val % N` for FIDs (bit 63 = 0)."""
  return torch.remainder(ids, num_shards)


class ShardedEmbedding:
  """All-to-all sharded lookup / apply_gradients for one table."""

  def __init__(self, backend: LocalBackend, group: Optional[dist.ProcessGroup] = None):
    self.backend = backend
    self.group = group
    self.world = dist.get_world_size(group)
    self.rank = dist.get_rank(group)
    self.dim = backend.dim
    self._ctx = None

  def _a2a(self, out, inp, out_splits, in_splits):
    dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits,
                           group=self.group)

  def lookup(self, ids: torch.Tensor) -> torch.Tensor:
    """ids int64 [B] on this rank -> rows fp32 [B, D]."""
    N, D = self.world, self.dim
    uids, inverse = self.backend.unique(ids)
    U = uids.numel()
    shard = shard_of(uids, N)
    # stable partition by shard == FusedReorderByIndices' shard-major, first-occurrence order
    order = torch.sort(shard, stable=True).indices
    send_ids = uids[order]
    send_counts = torch.bincount(shard, minlength=N)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=self.group)  # exchange #1: sizes
    sc = send_counts.cpu().tolist()
    rc = recv_counts.cpu().tolist()
    M = int(sum(rc))
    recv_ids = torch.empty(M, dtype=torch.int64, device=ids.device)
    self._a2a(recv_ids, send_ids, rc, sc)                               # exchange #2: ids
    rows = self.backend.lookup(recv_ids)                                # owner-side lookup
    back = torch.empty((U, D), dtype=torch.float32, device=ids.device)
    self._a2a(back, rows, sc, rc)                                       # exchange #3: rows
    # back is in send order; undo the partition, then scatter to occurrences
    pos_in_send = torch.empty_like(order)
    pos_in_send[order] = torch.arange(U, device=ids.device)
    out = back[pos_in_send[inverse.long()]]
    self._ctx = (inverse, order, sc, rc, recv_ids, U)
    return out

  def apply_gradients(self, grads: torch.Tensor, update_time: int, global_step: int = 0):
    """grads fp32 [B, D] for the ids of the preceding lookup()."""
    inverse, order, sc, rc, recv_ids, U = self._ctx
    D = self.dim
    gu = self.backend.segment_sum(grads, inverse, U)                    # [U, D], unique order
    send = gu[order].contiguous()
    recv = torch.empty((recv_ids.numel(), D), dtype=torch.float32, device=grads.device)
    self._a2a(recv, send, rc, sc)                                       # exchange #4: gradients
    self.backend.optimize_accumulated(recv_ids, recv, update_time, global_step)
    self._ctx = None
