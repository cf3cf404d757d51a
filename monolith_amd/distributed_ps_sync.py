"""Id-sharded embedding table over the GPUs of one node — the MI355X form of the reference's sync /
all-to-all path (native_training/distributed_ps_sync.py:95-287 lookup, :289-490 apply_gradients;
packing rules of runtime/ops/fused_reorder_by_indices.cc:38-123).

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).  GPU g owns
{fid : fid mod N == g} (distributed_ps.py:289, fused_reorder_by_indices.cc:121-123) as a complete
local MultiHashTable.  Per step and per rank, for a batch of B ids of one table:

  forward   dedup of the B ids (run dedup, csrc/mhte_step_kernels.h)  -> U unique ids
            shard-major packing of the unique ids (FusedReorderByIndices' layout)
            all-to-all #1  per-shard counts  int64[N]   (distributed_ps_sync.py:132-159)
            all-to-all #2  ids               int64[M]
            owner: lookup of the M received ids (no insert)
            all-to-all #3  rows              fp32[M, D]  (:217-261)
            scatter of the U returned rows to the B occurrences (MonolithFillWithOffsetMap)
  backward  duplicate-gradient sum per unique id, written in send order
            (MonolithFillWithOffsetMapGradient)
            all-to-all #4  gradients         fp32[M, D]  (:357-479)
            owner: contributions of different senders to one id are added (in rank order), then
            ONE optimizer step per distinct id (enable_grad_accumulation, tf_bridge.cc:270-310) —
            chosen over the reference default of N separate applications because it is
            order-independent.

Only unique ids and their rows / summed gradients cross xGMI: (N-1)/N * (8 U + 2 * 4 D U) bytes per
rank per step (≈6 MB at D = 64, U ≈ 12.8 k), which matters most at small N, where one pair of GPUs
shares a single link.  xGMI is point-to-point, so one all-to-all-v per tensor is the natural
pattern.  The split sizes are the one thing the host must know (a D2H copy of 2N+1 counts per
step); everything else stays on the device.

The exchange logic is device-agnostic; the local engine behind ``LocalBackend`` is the HIP library
(``HipBackend``).  tests/test_sharded_gloo.py runs the same exchange code with world_size 2 on
CPU/gloo against a stand-in backend, and — on a GPU box — two ranks sharing the GPU against
``HipBackend`` with the collectives staged through host memory.
"""
import ctypes as C
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist


class LocalBackend:
  """What the exchange needs from the local engine.  The GPU implementation is HipBackend; the
  gloo test supplies a CPU stand-in with the same methods.  ``cap`` = batch capacity B."""

  dim: int

  def dedup(self, ids: torch.Tensor):
    """Deduplicate this rank's batch.  -> (unique_ids[B] (first n_unique valid, any order),
    n_unique int32[1] on the ids' device).  The occurrence structure stays inside the backend."""
    raise NotImplementedError

  def partition(self, unique_ids: torch.Tensor, n_unique: torch.Tensor, num_shards: int):
    """-> (send_ids int64[B] shard-major, send_pos int32[B]: position of unique index u in it,
    counts int32[num_shards])"""
    raise NotImplementedError

  def scatter(self, rows: torch.Tensor, send_pos: torch.Tensor, n_out: int) -> torch.Tensor:
    """rows [*, D] in send order -> [n_out, D]: the row of every occurrence."""
    raise NotImplementedError

  def sum(self, grads: torch.Tensor, send_pos: torch.Tensor) -> torch.Tensor:
    """grads [B, D] per occurrence -> [B, D] buffer whose first U rows are the per-id sums in send
    order."""
    raise NotImplementedError

  def owner_lookup(self, ids: torch.Tensor) -> torch.Tensor:
    raise NotImplementedError

  def owner_apply(self, ids: torch.Tensor, grads: torch.Tensor, update_time: int, global_step: int):
    """ids may repeat (one row per sender); gradients of equal ids are added, then ONE step."""
    raise NotImplementedError


class HipBackend(LocalBackend):
  """Local shard on this process's MI355X (libmhte.so): run dedup + shard packing + scatter / sum
  on the sender side, MultiHashTable lookup and the fused sum + apply on the owner side."""

  MAX_STEP_BATCH = 65536

  def __init__(self, table, table_name: str):
    from monolith_amd import _lib
    from monolith_amd.distribution_ops import DedupWorkspace
    from monolith_amd.multi_hash_table_ops import _stream
    self._lib = _lib
    self._L = _lib.lib()
    self._stream = _stream
    self.table = table
    self.name = table_name
    self.idx = table._index(table_name)  # pylint: disable=protected-access
    self.dim = table.get_table_dim_sizes()[self.idx]
    self.dev = torch.device("cuda:%d" % table._device)  # pylint: disable=protected-access
    self.ws_s = DedupWorkspace(table._device)  # sender side  # pylint: disable=protected-access
    self.ws_o = DedupWorkspace(table._device)  # owner side  # pylint: disable=protected-access
    lr0 = sum(table._slice_sizes[:self.idx])  # pylint: disable=protected-access
    self.lrs = np.ascontiguousarray(
        table.learning_rate[lr0:lr0 + table._slice_sizes[self.idx]])  # pylint: disable=protected-access
    self._cap = 0
    self._ocap = 0

  def _sender_buffers(self, n):
    if n > self._cap:
      d = self.dev
      self.uids = torch.empty(n, dtype=torch.int64, device=d)
      self.nu = torch.zeros(1, dtype=torch.int32, device=d)
      self.send_ids = torch.empty(n, dtype=torch.int64, device=d)
      self.send_pos = torch.empty(n, dtype=torch.int32, device=d)
      self.gsum = torch.empty((n, self.dim), dtype=torch.float32, device=d)
      self._cap = n

  def _owner_buffers(self, m):
    if m > self._ocap:
      d = self.dev
      self.o_uids = torch.empty(m, dtype=torch.int64, device=d)
      self.o_nu = torch.zeros(1, dtype=torch.int32, device=d)
      self.o_grad_u = torch.empty((m, self.dim), dtype=torch.float32, device=d)
      self._ocap = m

  def dedup(self, ids):
    n = ids.numel()
    if n > self.MAX_STEP_BATCH:
      raise self._lib.InvalidArgumentError(self._lib.MHTE_INVALID_ARGUMENT,
                                           "sharded step: at most %d ids per rank and step" %
                                           self.MAX_STEP_BATCH)
    self._sender_buffers(n)
    self._n = n
    self.ws_s.step_dedup(ids, self.uids, self.nu)
    return self.uids, self.nu

  def partition(self, unique_ids, n_unique, num_shards):
    counts = torch.empty(num_shards, dtype=torch.int32, device=self.dev)
    vp, check = self._lib.vp, self._lib.check
    check(self._L.mhte_shard_partition(self.ws_s._h, vp(unique_ids), C.c_int64(self._n),  # pylint: disable=protected-access
                                       vp(n_unique), C.c_int32(num_shards), vp(self.send_ids),
                                       vp(self.send_pos), vp(counts), self._stream()))
    return self.send_ids, self.send_pos, counts

  def scatter(self, rows, send_pos, n_out):
    out = torch.empty((n_out, self.dim), dtype=torch.float32, device=self.dev)
    if rows.numel() == 0:  # nothing came back: every occurrence misses
      return out.zero_()
    vp, check = self._lib.vp, self._lib.check
    check(self._L.mhte_step_scatter(self.ws_s._h, vp(rows), vp(send_pos), C.c_int32(self.dim),  # pylint: disable=protected-access
                                    vp(out), self._stream()))
    return out

  def sum(self, grads, send_pos):
    vp, check = self._lib.vp, self._lib.check
    check(self._L.mhte_step_sum(self.ws_s._h, vp(grads), vp(send_pos), C.c_int32(self.dim),  # pylint: disable=protected-access
                                vp(self.gsum), self._stream()))
    return self.gsum

  def owner_lookup(self, ids):
    out = torch.empty((ids.numel(), self.dim), dtype=torch.float32, device=self.dev)
    if ids.numel():
      self.table.table_lookup_n(self.idx, ids, None, out)
    return out

  def owner_apply(self, ids, grads, update_time, global_step):
    m = ids.numel()
    if m == 0:
      return
    if m <= self.MAX_STEP_BATCH and self.table._lib.mhte_table_fused_backward_ok(  # pylint: disable=protected-access
        self.table.handle, self.idx):
      # the received ids are a batch with duplicates (one occurrence per sender): run dedup + the
      # fused sum / upsert / optimizer launch of the single-GPU step
      self._owner_buffers(m)
      self.ws_o.step_dedup(ids, self.o_uids[:m], self.o_nu)
      self.table.table_step_backward(self.idx, self.ws_o, None, self.o_uids[:m], self.o_nu, grads,
                                     self.o_grad_u, self.lrs, update_time, global_step)
    else:
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
    # a gloo group cannot move device tensors: stage them through host memory (test configuration:
    # several ranks sharing one GPU)
    self._gloo = dist.get_backend(group) == "gloo"
    self._ctx = None

  def _a2a(self, out, inp, out_splits=None, in_splits=None):
    if self._gloo and inp.is_cuda:
      o, i = out.cpu(), inp.cpu()
      dist.all_to_all_single(o, i, output_split_sizes=out_splits, input_split_sizes=in_splits,
                             group=self.group)
      out.copy_(o)
      return
    dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits,
                           group=self.group)

  def lookup(self, ids: torch.Tensor) -> torch.Tensor:
    """ids int64 [B] on this rank -> rows fp32 [B, D]."""
    N, D, be = self.world, self.dim, self.backend
    dev = ids.device
    uids, nu = be.dedup(ids)
    send_ids, send_pos, counts = be.partition(uids, nu, N)
    send_counts = counts.to(torch.int64)
    recv_counts = torch.empty_like(send_counts)
    self._a2a(recv_counts, send_counts)                                 # exchange #1: sizes
    both = torch.cat([send_counts, recv_counts]).cpu().tolist()         # the step's one D2H copy
    sc, rc = both[:N], both[N:]
    U, M = int(sum(sc)), int(sum(rc))
    recv_ids = torch.empty(M, dtype=torch.int64, device=dev)
    self._a2a(recv_ids, send_ids[:U], rc, sc)                           # exchange #2: ids
    rows = be.owner_lookup(recv_ids)                                    # owner-side lookup
    back = torch.empty((U, D), dtype=torch.float32, device=dev)
    self._a2a(back, rows, sc, rc)                                       # exchange #3: rows
    out = be.scatter(back, send_pos, ids.numel())                       # rows -> occurrences
    self._ctx = (send_pos, sc, rc, recv_ids, U, M)
    return out

  def apply_gradients(self, grads: torch.Tensor, update_time: int, global_step: int = 0):
    """grads fp32 [B, D] for the ids of the preceding lookup()."""
    send_pos, sc, rc, recv_ids, U, M = self._ctx
    D, be = self.dim, self.backend
    gsum = be.sum(grads, send_pos)                                      # [*, D], send order
    recv = torch.empty((M, D), dtype=torch.float32, device=grads.device)
    self._a2a(recv, gsum[:U], rc, sc)                                   # exchange #4: gradients
    be.owner_apply(recv_ids, recv, update_time, global_step)
    self._ctx = None
