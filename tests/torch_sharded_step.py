"""TEST HARNESS (not product): round 1's torch.distributed form of the id-sharded step for ONE table —
four ``all_to_all_single`` calls per step (sizes, ids, rows, gradients; reference
native_training/distributed_ps_sync.py:132-159,217-261,357-479), the split sizes through the host.

The product's sharded step is ``monolith_amd.distributed_ps_sync.ShardedMultiStep`` (C++,
csrc/mhte_shard_host.h).  This module is kept for
  * tests/test_sharded_gloo.py — world_size 2 over gloo: the exchange logic below against a CPU stand-in of
    the local engine (``OracleBackend`` there), i.e. the sharding rule fid mod N, the shard-major stable
    packing and the un-permute / scatter of THIS module, not the C++ step; and two ranks sharing the one
    GPU with ``HipBackend``;
  * tests/test_parity_gpu.py::test_sender_side_partition_scatter_sum — ``HipBackend`` is a thin adapter over
    the product's sender-side C entry points (mhte_shard_partition / mhte_step_scatter / mhte_step_sum);
  * bench.py's last-resort fallback (``--transport torch``, or when the C++ step cannot be created on some
    rank of an N-GPU run): ≈ 4x slower than the C++ step, and the line names it.

Owner semantics here: contributions of different senders to one id are added (in rank order), then ONE
optimizer step per distinct id (enable_grad_accumulation, tf_bridge.cc:270-310).
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
  on the sender side, MultiHashTable lookup and the fused sum + apply on the owner side.  Two
  slots of sender / owner state alternate (``use_slot``), so that the id dispatch of the next batch
  can be prepared while the current one is still being trained."""

  MAX_STEP_BATCH = 65536

  class _Slot:
    def __init__(self, device):
      from monolith_amd.distribution_ops import DedupWorkspace
      self.ws_s = DedupWorkspace(device)  # sender side
      self.ws_o = DedupWorkspace(device)  # owner side
      self.cap = 0
      self.ocap = 0
      self.n = 0
      self.owner_n = -1   # number of received ids whose run dedup ws_o holds (-1: none)

  def __init__(self, table, table_name: str):
    from monolith_amd import _lib
    from monolith_amd.multi_hash_table_ops import _stream
    self._lib = _lib
    self._L = _lib.lib()
    self._stream = _stream
    self.table = table
    self.name = table_name
    self.idx = table._index(table_name)  # pylint: disable=protected-access
    self.dim = table.get_table_dim_sizes()[self.idx]
    self.dev = torch.device("cuda:%d" % table._device)  # pylint: disable=protected-access
    self._slots = [HipBackend._Slot(table._device), HipBackend._Slot(table._device)]  # pylint: disable=protected-access
    self._s = self._slots[0]
    self._fused_ok = bool(table._lib.mhte_table_fused_backward_ok(table.handle, self.idx))  # pylint: disable=protected-access
    lr0 = sum(table._slice_sizes[:self.idx])  # pylint: disable=protected-access
    self.lrs = np.ascontiguousarray(
        table.learning_rate[lr0:lr0 + table._slice_sizes[self.idx]])  # pylint: disable=protected-access

  def use_slot(self, i: int):
    self._s = self._slots[i & 1]

  def _sender_buffers(self, n):
    sl = self._s
    if n > sl.cap:
      d = self.dev
      sl.uids = torch.empty(n, dtype=torch.int64, device=d)
      sl.nu = torch.zeros(1, dtype=torch.int32, device=d)
      sl.send_ids = torch.empty(n, dtype=torch.int64, device=d)
      sl.send_pos = torch.empty(n, dtype=torch.int32, device=d)
      sl.gsum = torch.empty((n, self.dim), dtype=torch.float32, device=d)
      sl.cap = n

  def _owner_buffers(self, m):
    sl = self._s
    if m > sl.ocap:
      d = self.dev
      sl.o_uids = torch.empty(m, dtype=torch.int64, device=d)
      sl.o_nu = torch.zeros(1, dtype=torch.int32, device=d)
      sl.o_grad_u = torch.empty((m, self.dim), dtype=torch.float32, device=d)
      sl.ocap = m

  def dedup(self, ids):
    n = ids.numel()
    if n > self.MAX_STEP_BATCH:
      raise self._lib.InvalidArgumentError(self._lib.MHTE_INVALID_ARGUMENT,
                                           "sharded step: at most %d ids per rank and step" %
                                           self.MAX_STEP_BATCH)
    self._sender_buffers(n)
    sl = self._s
    sl.n = n
    sl.owner_n = -1
    sl.ws_s.step_dedup(ids, sl.uids, sl.nu)
    return sl.uids, sl.nu

  def partition(self, unique_ids, n_unique, num_shards):
    sl = self._s
    counts = torch.empty(num_shards, dtype=torch.int32, device=self.dev)
    vp, check = self._lib.vp, self._lib.check
    check(self._L.mhte_shard_partition(sl.ws_s._h, vp(unique_ids), C.c_int64(sl.n),  # pylint: disable=protected-access
                                       vp(n_unique), C.c_int32(num_shards), vp(sl.send_ids),
                                       vp(sl.send_pos), vp(counts), self._stream()))
    return sl.send_ids, sl.send_pos, counts

  def scatter(self, rows, send_pos, n_out):
    out = torch.empty((n_out, self.dim), dtype=torch.float32, device=self.dev)
    if rows.numel() == 0:  # nothing came back: every occurrence misses
      return out.zero_()
    vp, check = self._lib.vp, self._lib.check
    check(self._L.mhte_step_scatter(self._s.ws_s._h, vp(rows), vp(send_pos), C.c_int32(self.dim),  # pylint: disable=protected-access
                                    vp(out), self._stream()))
    return out

  def sum(self, grads, send_pos):
    sl = self._s
    vp, check = self._lib.vp, self._lib.check
    check(self._L.mhte_step_sum(sl.ws_s._h, vp(grads), vp(send_pos), C.c_int32(self.dim),  # pylint: disable=protected-access
                                vp(sl.gsum), self._stream()))
    return sl.gsum

  def owner_lookup(self, ids):
    out = torch.empty((ids.numel(), self.dim), dtype=torch.float32, device=self.dev)
    if ids.numel():
      self.table.table_lookup_n(self.idx, ids, None, out)
    return out

  def _owner_fast(self, m):
    return 0 < m <= self.MAX_STEP_BATCH and self._fused_ok

  def owner_prepare(self, ids):
    """Run dedup of the received ids ahead of their gradients (it depends on the ids only)."""
    m = ids.numel()
    if self._owner_fast(m):
      self._owner_buffers(m)
      sl = self._s
      sl.ws_o.step_dedup(ids, sl.o_uids[:m], sl.o_nu)
      sl.owner_n = m

  def owner_apply(self, ids, grads, update_time, global_step):
    m = ids.numel()
    if m == 0:
      return
    sl = self._s
    if self._owner_fast(m):
      # the received ids are a batch with duplicates (one occurrence per sender): run dedup + the
      # fused sum / upsert / optimizer launch of the single-GPU step
      if sl.owner_n != m:
        self.owner_prepare(ids)
      self.table.table_step_backward(self.idx, sl.ws_o, None, sl.o_uids[:m], sl.o_nu, grads,
                                     sl.o_grad_u, self.lrs, update_time, global_step)
      # the displacement pass of this update reads the slot's o_uids / o_grad_u: run it now, in
      # stream order, so that the side stream's next owner_prepare into this slot (it waits for the
      # event recorded after apply_gradients) cannot overtake it
      self.table.table_finish_pending(self.idx)
      sl.owner_n = -1
    else:
      self.table.table_optimize_n(self.idx, ids, None, grads, self.lrs, update_time, global_step,
                                  flags=self._lib.MHTE_SUM_DUPLICATES)


def shard_of(ids: torch.Tensor, num_shards: int) -> torch.Tensor:
  """floormod(id, N) — distributed_ps.py:289; equals the fused op's `val % N` for FIDs (bit 63 = 0)."""
  return torch.remainder(ids, num_shards)


class ShardedEmbedding:
  """All-to-all sharded lookup / apply_gradients for one table.

  ``lookup(ids, next_ids=...)`` also starts the id dispatch of the FOLLOWING batch (dedup, shard
  packing, size + id exchanges, owner-side dedup — everything that depends on ids only) on a side
  stream, as the reference's prefetch queue does (distributed_ps_sync.py:199-203): it runs beside
  this step's row exchange, scatter, dense model and backward, and the next ``lookup`` (which must
  receive that same tensor) only has the owner lookup, the row exchange and the scatter on its
  critical path.  ``next_ids`` must already be materialised (it comes from the input pipeline): the
  side stream does not wait for work queued on the caller's stream, only for the previous user of
  the state slot it writes.  ``prefetch_on_side_stream=False`` runs that dispatch on the caller's
  stream instead (no overlap on the GPU, fewer cross-stream dependencies for the host: with one rank
  and no link latency to hide it is ≈10 % faster, 210-230 µs against 250 µs per step)."""

  def __init__(self, backend: LocalBackend, group: Optional[dist.ProcessGroup] = None,
               prefetch_on_side_stream: bool = True):
    self.backend = backend
    self.prefetch_on_side_stream = prefetch_on_side_stream
    self.group = group
    self.world = dist.get_world_size(group)
    self.rank = dist.get_rank(group)
    self.dim = backend.dim
    # a gloo group cannot move device tensors: stage them through host memory (test configuration:
    # several ranks sharing one GPU)
    self._gloo = dist.get_backend(group) == "gloo"
    self._ctx = None
    self._pre = None      # dispatch prepared ahead: (key, dispatch tuple, event)
    self._slot = 0
    self._side = None
    self._slot_free = [None, None]   # event: the step that last used the slot has been enqueued

  def _a2a(self, out, inp, out_splits=None, in_splits=None):
    if self._gloo and inp.is_cuda:
      o, i = out.cpu(), inp.cpu()
      dist.all_to_all_single(o, i, output_split_sizes=out_splits, input_split_sizes=in_splits,
                             group=self.group)
      out.copy_(o)
      return
    dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits,
                           group=self.group)

  def _dispatch(self, ids: torch.Tensor, slot: int):
    """Everything of a step that depends on the ids only."""
    N, be = self.world, self.backend
    if hasattr(be, "use_slot"):
      be.use_slot(slot)
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
    if hasattr(be, "owner_prepare"):
      be.owner_prepare(recv_ids)
    return (slot, send_pos, sc, rc, recv_ids, U, M)

  def lookup(self, ids: torch.Tensor, next_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ids int64 [B] on this rank -> rows fp32 [B, D]."""
    D, be = self.dim, self.backend
    dev = ids.device
    key = self._batch_key(ids)
    if self._pre is not None and self._same_batch(self._pre[0], key):
      _, disp, ev = self._pre
      if ev is not None:
        torch.cuda.current_stream().wait_event(ev)
    else:
      disp = self._dispatch(ids, self._slot)
    self._pre = None
    slot, send_pos, sc, rc, recv_ids, U, M = disp
    if hasattr(be, "use_slot"):
      be.use_slot(slot)
    rows = be.owner_lookup(recv_ids)                                    # owner-side lookup
    back = torch.empty((U, D), dtype=torch.float32, device=dev)
    self._a2a(back, rows, sc, rc)                                       # exchange #3: rows
    out = be.scatter(back, send_pos, ids.numel())                       # rows -> occurrences
    self._ctx = disp
    if next_ids is not None:
      self._prefetch(next_ids, 1 - slot)
    return out

  def apply_gradients(self, grads: torch.Tensor, update_time: int, global_step: int = 0,
                      next_ids: Optional[torch.Tensor] = None):
    """grads fp32 [B, D] for the ids of the preceding lookup()."""
    slot, send_pos, sc, rc, recv_ids, U, M = self._ctx
    D, be = self.dim, self.backend
    if hasattr(be, "use_slot"):
      be.use_slot(slot)
    gsum = be.sum(grads, send_pos)                                      # [*, D], send order
    recv = torch.empty((M, D), dtype=torch.float32, device=grads.device)
    self._a2a(recv, gsum[:U], rc, sc)                                   # exchange #4: gradients
    be.owner_apply(recv_ids, recv, update_time, global_step)
    self._ctx = None
    self._slot = 1 - slot
    if grads.is_cuda:
      ev = torch.cuda.Event()
      ev.record(torch.cuda.current_stream())
      self._slot_free[slot] = ev
    if next_ids is not None and self._pre is None:
      self._prefetch(next_ids, self._slot)

  @staticmethod
  def _batch_key(ids):
    # the tensor object and its version: a buffer refilled in place is a different batch
    return (ids, getattr(ids, "_version", 0), ids.data_ptr(), ids.numel())

  @staticmethod
  def _same_batch(a, b):
    # same memory, unmodified since (the key keeps the tensor alive); not the Python object: a
    # slice of a resident id array is a new view object every time
    return a[1] == b[1] and a[2] == b[2] and a[3] == b[3]

  def _prefetch(self, ids: torch.Tensor, slot: int):
    key = self._batch_key(ids)
    if not ids.is_cuda or not self.prefetch_on_side_stream:
      # (same stream: the dispatch simply runs behind this step's forward)
      self._pre = (key, self._dispatch(ids, slot), None)
      return
    if self._side is None:
      self._side = torch.cuda.Stream(device=ids.device)
    main = torch.cuda.current_stream()
    if self._slot_free[slot] is not None:   # the slot's previous step must have finished with it
      self._side.wait_event(self._slot_free[slot])
    with torch.cuda.stream(self._side):
      disp = self._dispatch(ids, slot)
      ev = torch.cuda.Event()
      ev.record(self._side)
    for t in (disp[1], disp[4]):     # consumed on the main stream later
      if isinstance(t, torch.Tensor) and t.is_cuda:
        t.record_stream(main)
    self._pre = (key, disp, ev)
