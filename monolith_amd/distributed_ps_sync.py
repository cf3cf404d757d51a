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


# =================================================================================================
# The multi-table sharded step, driven from C++ (csrc/mhte_shard_host.h)
# =================================================================================================
def _rccl_env():
  """Point the library at the RCCL this process already has (PyTorch ships its own copy)."""
  import os
  if "MHTE_RCCL_LIBRARY" not in os.environ:
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if os.path.exists(cand):
      os.environ["MHTE_RCCL_LIBRARY"] = cand


def shard_unique_id() -> bytes:
  """128-byte RCCL unique id (mhte_shard_unique_id); one rank makes it, the launcher hands it to
  all."""
  from monolith_amd import _lib
  _rccl_env()
  buf = C.create_string_buffer(128)
  _lib.check(_lib.lib().mhte_shard_unique_id(buf))
  return buf.raw


def shard_block_geometry(dims, batch_per_table: int, world: int, ids_per_peer_table: int = 0):
  """The wire format of the sharded multi-table step (csrc/mhte_shard_kernels.h), as ShardStep::init
  lays it out: every rank keeps one fixed-capacity block per peer and direction,

    id block    int64[hdr + T * cap]      word t < T: number of ids of table t, then cap id slots
                                          per table at id_off[t]
    row block   float32[sum_t cap * dim]  cap rows per table at row_off[t]; id slot s <-> row s

  -> dict(cap, hdr, id_off[T], row_off[T], ids_block (int64 words), rows_block (floats)).
  ``ShardedMultiStep.info()`` reports the same block sizes from the library."""
  T = len(dims)
  mb = int(batch_per_table)
  c = int(ids_per_peer_table) if ids_per_peer_table > 0 else mb   # default: the whole batch fits
  cap = (min(c, mb) + 3) & ~3
  hdr = (T + 7) & ~7
  id_off = [hdr + t * cap for t in range(T)]
  row_off, rw = [], 0
  for d in dims:
    row_off.append(rw)
    rw += cap * int(d)
  idw = hdr + T * cap
  return {"cap": cap, "hdr": hdr, "id_off": id_off, "row_off": row_off,
          "ids_block": (idw + 1) & ~1, "rows_block": rw}


class ShardedMultiStep:
  """All tables of the model, sharded by id over the ranks, one exchange per direction for all of
  them (native_training/distributed_ps_sync.py:95-287, :289-490) — mhte_shard_step_*: the whole
  step is enqueued from C++.

  ``table`` is this rank's MultiHashTable (the ids it owns).  ``group``: a torch.distributed group
  used ONCE, at creation, as the launcher's side channel (any backend: gloo when several ranks
  share one device).  ``transport``:

    "ipc"   peer stores: every rank maps the others' receive windows (hipIpc*), an exchange is one
            copy kernel storing the occupied part of every (peer, table) segment into the peers'
            windows, sized on the device — works across xGMI and between processes on one GPU;
    "rccl"  ncclSend / ncclRecv groups on a communicator of the library's own (rank 0's unique id
            is handed out through ``group``); needs one device per rank;
    "auto"  world 1: identity; else "ipc" if every rank's self test passes, else "rccl".

  Same call protocol as ``MultiSparseStep``: ``forward(ragged, next_ragged)`` returns the flat
  per-occurrence embedding and dispatches the next batch's ids ahead; ``backward(flat_grad,
  update_time)``."""

  def __init__(self, table, batch_per_table: int, group: Optional["dist.ProcessGroup"] = None,
               ids_per_peer_table: int = 0, use_rccl: Optional[bool] = None, transport: str = "auto",
               grad_fp16: Optional[bool] = None, overlap: Optional[bool] = None):
    from monolith_amd import _lib
    self._libmod = _lib
    self._lib = table._lib  # pylint: disable=protected-access
    self.table = table
    self.batch = int(batch_per_table)
    self._dims = table.get_table_dim_sizes()
    if group is not None or (dist.is_available() and dist.is_initialized()):
      self.world = dist.get_world_size(group)
      self.rank = dist.get_rank(group)
    else:
      self.world, self.rank = 1, 0
    if transport not in ("auto", "ipc", "rccl", "identity"):
      raise ValueError("transport must be auto | ipc | rccl | identity")
    if use_rccl:
      transport = "rccl"
    self._group = group
    self._h = None
    # the wire format / pipeline mode of a world > 1 are part of what the ranks agree on when the
    # windows are connected: chosen here, applied before the handles are taken
    self._grad_fp16, self._overlap = grad_fp16, overlap
    self.transport_note = None    # why the first choice of transport was not taken (auto)
    if transport == "ipc" or (transport == "auto" and self.world > 1):
      err = self._create_ipc(ids_per_peer_table)
      if err is None:
        return
      if transport == "ipc":
        raise err
      self.transport_note = "peer stores unavailable (%s): RCCL send / recv" % (str(err)[:200],)
      transport = "rccl"
    use_rccl = transport == "rccl"
    uid = None
    if use_rccl:
      _rccl_env()
      if self.world > 1:
        box = [shard_unique_id() if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0,
                                   group=group)
        uid = box[0]
      else:
        uid = shard_unique_id()
    h = C.c_void_p()
    _lib.check(self._lib.mhte_shard_step_create(
        table.handle, C.c_int64(self.batch), C.c_int32(self.rank), C.c_int32(self.world),
        C.c_int64(int(ids_per_peer_table)), uid, C.byref(h)))
    self._h = h
    self._ahead = None
    self._keep = None
    if self._grad_fp16 is not None and not (self.world == 1 and not use_rccl):
      self.set_grad_fp16(self._grad_fp16)
    if self._overlap is not None:
      self.set_overlap(self._overlap)

  def _all_true(self, ok: bool) -> bool:
    if self.world == 1:
      return ok
    box = [None] * self.world
    dist.all_gather_object(box, bool(ok), group=self._group)
    return all(box)

  def _create_ipc(self, ids_per_peer_table):
    """The peer-store transport: create -> gather every rank's window handle -> connect -> self test.
    Collective.  Returns None, or the error after EVERY rank has given the transport up."""
    _lib = self._libmod
    h = C.c_void_p()
    err = None
    blob = C.create_string_buffer(128)
    try:
      _lib.check(self._lib.mhte_shard_step_create_ipc(
          self.table.handle, C.c_int64(self.batch), C.c_int32(self.rank), C.c_int32(self.world),
          C.c_int64(int(ids_per_peer_table)), C.byref(h)))
      if self._grad_fp16 is not None:
        _lib.check(self._lib.mhte_shard_step_set_grad_bits(h, C.c_int32(16 if self._grad_fp16 else 32)))
      if self._overlap is not None:
        _lib.check(self._lib.mhte_shard_step_set_overlap(h, C.c_int32(1 if self._overlap else 0)))
      _lib.check(self._lib.mhte_shard_step_ipc_handle(h, blob))
    except _lib.MhteError as e:
      err = e
    blobs = [None] * self.world
    if self.world > 1:
      dist.all_gather_object(blobs, blob.raw if err is None else None, group=self._group)
    else:
      blobs[0] = blob.raw if err is None else None
    if err is None and all(b is not None for b in blobs):
      try:
        _lib.check(self._lib.mhte_shard_step_ipc_connect(h, b"".join(blobs), C.c_int32(self.world)))
      except _lib.MhteError as e:
        err = e
    elif err is None:
      err = _lib.MhteError(_lib.MHTE_UNAVAILABLE, "a peer could not create its window")
    if self._all_true(err is None):
      try:   # (every rank connected: the round trip is collective)
        _lib.check(self._lib.mhte_shard_step_ipc_selftest(h, self._stream()))
      except _lib.MhteError as e:
        err = e
    elif err is None:
      err = _lib.MhteError(_lib.MHTE_UNAVAILABLE, "a peer could not map the windows")
    if not self._all_true(err is None):
      if h:
        torch.cuda.synchronize()
        self._lib.mhte_shard_step_destroy(h)
      return err or _lib.MhteError(_lib.MHTE_UNAVAILABLE, "a peer failed the peer-store self test")
    self._h = h
    self._ahead = None
    self._keep = None
    return None

  def close(self, collective: bool = True):
    """Destroys the step.  With peers (world > 1) an explicit close is collective: every rank drains
    its stream and meets the others before any window is unmapped — a peer may still be storing
    into it."""
    if getattr(self, "_h", None):
      torch.cuda.synchronize()
      if collective and self.world > 1 and dist.is_available() and dist.is_initialized():
        dist.barrier(group=self._group)
      self._lib.mhte_shard_step_destroy(self._h)
      self._h = None

  def __del__(self):
    try:
      self.close(collective=False)
    except Exception:  # pylint: disable=broad-except
      pass

  def set_overlap(self, on: bool = True):
    """The next batch's dedup / numbering / id dispatch on a stream of the step's own, beside the
    dense model the caller runs between ``forward`` and ``backward`` (mhte_shard_step_set_overlap)."""
    self._libmod.check(self._lib.mhte_shard_step_set_overlap(self._h, C.c_int32(1 if on else 0)))
    return self

  def set_grad_fp16(self, on: bool = True):
    """The gradient exchange in fp16 (mhte_shard_step_set_grad_bits; the reference's optional cast of
    the gradient all-to-all): a numerics change, every rank must choose the same."""
    self._libmod.check(self._lib.mhte_shard_step_set_grad_bits(self._h, C.c_int32(16 if on else 32)))
    return self

  def info(self):
    out = (C.c_int64 * 4)()
    self._libmod.check(self._lib.mhte_shard_step_info(self._h, out))
    d = {"ids_per_peer_table": out[0], "id_block_bytes": out[1], "row_block_bytes": out[2],
         "transport": ("identity", "rccl", "group", "ipc", "ipc (coarse window)")[out[3]]}
    if self.transport_note:
      d["transport_note"] = self.transport_note
    ln = (C.c_int32 * 2)()
    self._libmod.check(self._lib.mhte_shard_step_launches(self._h, ln))
    d["launches_per_step"] = int(ln[0]) + int(ln[1])     # (of the last forward + backward)
    d["launches_forward"], d["launches_backward"] = int(ln[0]), int(ln[1])
    if out[3] == 1:
      d["rccl_ranks"], d["rccl_rank"] = self.comm_ranks()
    return d

  def comm_ranks(self):
    """(ncclCommCount, ncclCommUserRank) of the step's own RCCL communicator; (0, 0) without one."""
    out = (C.c_int32 * 2)()
    self._libmod.check(self._lib.mhte_shard_step_comm_ranks(self._h, out))
    return int(out[0]), int(out[1])

  @staticmethod
  def _key(r):
    return (r.values, r.values._version, r.row_splits.tobytes())  # pylint: disable=protected-access

  @staticmethod
  def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)

  def forward(self, ragged, next_ragged=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _lib = self._libmod
    lens = ragged.row_lengths()
    total = int(sum(int(l) * d for l, d in zip(lens, self._dims)))
    if out is None:
      out = torch.empty(total, dtype=torch.float32, device=ragged.values.device)
    a = self._ahead
    v = ragged.values
    pre = (a is not None and a[0].data_ptr() == v.data_ptr() and a[0].numel() == v.numel() and
           a[1] == v._version and a[2] == ragged.row_splits.tobytes())  # pylint: disable=protected-access
    sp = np.ascontiguousarray(ragged.row_splits, dtype=np.int64)
    if next_ragged is not None:
      nsp = np.ascontiguousarray(next_ragged.row_splits, dtype=np.int64)
      nv, nsp_p, nsp_n = _lib.vp(next_ragged.values), nsp.ctypes.data_as(C.POINTER(C.c_int64)), nsp.size
    else:
      nv, nsp_p, nsp_n = C.c_void_p(0), None, 0
    _lib.check(self._lib.mhte_shard_step_forward(
        self._h, _lib.vp(ragged.values), sp.ctypes.data_as(C.POINTER(C.c_int64)), C.c_int64(sp.size),
        _lib.vp(out), C.c_int64(out.numel()), nv, nsp_p, C.c_int64(nsp_n),
        C.c_int32(1 if pre else 0), self._stream()))
    self._ahead = self._key(next_ragged) if next_ragged is not None else None
    self._keep = (ragged, next_ragged, out)
    return out

  def backward(self, flat_grad: torch.Tensor, update_time: int, global_step: int = 0):
    _lib = self._libmod
    lrs = np.ascontiguousarray(self.table.learning_rate, dtype=np.float32)
    _lib.check(self._lib.mhte_shard_step_backward(
        self._h, _lib.vp(flat_grad), C.c_int64(flat_grad.numel()),
        lrs.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(lrs.size), C.c_int64(int(update_time)),
        C.c_int64(int(global_step)), self._stream()))

  def unique_counts(self) -> np.ndarray:
    """Distinct ids per table of the batch last given to ``forward`` (synchronises)."""
    out = np.zeros(len(self._dims), dtype=np.int64)
    self._libmod.check(self._lib.mhte_shard_step_unique_counts(
        self._h, out.ctypes.data_as(C.POINTER(C.c_int64)), self._stream()))
    return out

  def check(self):
    """Wait for the stream; raises ResourceExhausted if a peer block overflowed."""
    self._libmod.check(self._lib.mhte_shard_step_check(self._h, self._stream()))


class ShardedStepGroup:
  """All ranks of a world inside ONE process on one GPU (mhte_shard_group_*): device copies stand
  in for the links.  For tests of the N-rank protocol on a single MI355X."""

  def __init__(self, tables, batch_per_table: int, ids_per_peer_table: int = 0):
    from monolith_amd import _lib
    self._libmod = _lib
    self.tables = list(tables)
    self.world = len(self.tables)
    self._lib = self.tables[0]._lib  # pylint: disable=protected-access
    self._dims = self.tables[0].get_table_dim_sizes()
    self._hs = []
    for r, t in enumerate(self.tables):
      h = C.c_void_p()
      _lib.check(self._lib.mhte_shard_step_create(
          t.handle, C.c_int64(int(batch_per_table)), C.c_int32(r), C.c_int32(self.world),
          C.c_int64(int(ids_per_peer_table)), None, C.byref(h)))
      self._hs.append(h)
    self._arr = (C.c_void_p * self.world)(*[h.value for h in self._hs])
    self._keep = None

  def close(self):
    if getattr(self, "_hs", None):
      torch.cuda.synchronize()
      for h in self._hs:
        self._lib.mhte_shard_step_destroy(h)
      self._hs = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass

  def launches(self):
    """Per rank: (kernel launches + exchanges) the last forward and the last backward enqueued
    (mhte_shard_step_launches)."""
    out = []
    for h in self._hs:
      ln = (C.c_int32 * 2)()
      self._libmod.check(self._lib.mhte_shard_step_launches(h, ln))
      out.append((int(ln[0]), int(ln[1])))
    return out

  def forward(self, raggeds, next_raggeds=None, prefetched=False):
    """raggeds: one Ragged per rank -> list of flat embeddings."""
    _lib, N = self._libmod, self.world
    I64P = C.POINTER(C.c_int64)
    outs, sps = [], []
    for r in raggeds:
      lens = r.row_lengths()
      outs.append(torch.empty(int(sum(int(l) * d for l, d in zip(lens, self._dims))),
                              dtype=torch.float32, device=r.values.device))
      sps.append(np.ascontiguousarray(r.row_splits, dtype=np.int64))
    ids = (C.c_void_p * N)(*[r.values.data_ptr() for r in raggeds])
    spl = (I64P * N)(*[s.ctypes.data_as(I64P) for s in sps])
    emb = (C.c_void_p * N)(*[o.data_ptr() for o in outs])
    elen = (C.c_int64 * N)(*[o.numel() for o in outs])
    if next_raggeds is not None:
      nsps = [np.ascontiguousarray(r.row_splits, dtype=np.int64) for r in next_raggeds]
      nids = (C.c_void_p * N)(*[r.values.data_ptr() for r in next_raggeds])
      nspl = (I64P * N)(*[s.ctypes.data_as(I64P) for s in nsps])
      n_next = nsps[0].size
    else:
      nsps, nids, nspl, n_next = None, None, None, 0
    _lib.check(self._lib.mhte_shard_group_forward(
        self._arr, C.c_int32(N), ids, spl, C.c_int64(sps[0].size), emb, elen, nids, nspl,
        C.c_int64(n_next), C.c_int32(1 if prefetched else 0),
        C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    self._keep = (raggeds, next_raggeds, outs, sps, nsps)
    return outs

  def backward(self, flat_grads, update_time: int, global_step: int = 0):
    _lib, N = self._libmod, self.world
    lrs = np.ascontiguousarray(self.tables[0].learning_rate, dtype=np.float32)
    val = (C.c_void_p * N)(*[g.data_ptr() for g in flat_grads])
    vlen = (C.c_int64 * N)(*[g.numel() for g in flat_grads])
    _lib.check(self._lib.mhte_shard_group_backward(
        self._arr, C.c_int32(N), val, vlen, lrs.ctypes.data_as(C.POINTER(C.c_float)),
        C.c_int64(lrs.size), C.c_int64(int(update_time)), C.c_int64(int(global_step)),
        C.c_void_p(torch.cuda.current_stream().cuda_stream)))

  def check(self):
    for h in self._hs:
      self._libmod.check(self._lib.mhte_shard_step_check(
          h, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
