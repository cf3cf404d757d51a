"""Id-sharded embedding tables over the GPUs of one node — the MI355X form of the reference's sync /
all-to-all path (native_training/distributed_ps_sync.py:95-287 lookup, :289-490 apply_gradients;
packing rules of runtime/ops/fused_reorder_by_indices.cc:38-123).

One process per GPU.  GPU g owns {fid : fid mod N == g} of every table (distributed_ps.py:289,
fused_reorder_by_indices.cc:121-123) as a complete local MultiHashTable.  The step is enqueued from C++
(csrc/mhte_shard_host.h, C ABI mhte_shard_step_*): ``ShardedMultiStep`` below mirrors it — one launch per
stage for all tables, the exchanges as direct peer stores into hipIpc windows or RCCL send / recv groups,
no count ever reaching the host.  ``ShardedStepGroup`` drives N ranks inside one process
(tests: device copies stand in for the links).

Round 1's torch.distributed form of the step for ONE table (``ShardedEmbedding``: four all_to_all_single
calls per step, split sizes through the host) is no longer part of the product: it lives in
tests/torch_sharded_step.py as the harness of the world-2 gloo test and as bench.py's last-resort
fallback when the C++ step cannot be created on an N-GPU node.
"""
import ctypes as C
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist


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

  def set_exact_order(self, on: bool = True):
    """Every duplicate list of this rank summed strictly in occurrence order (mhte_shard_step_set_exact_order):
    the owners' rows are then the reference's bit for bit.  Takes effect at the next backward."""
    self._libmod.check(self._lib.mhte_shard_step_set_exact_order(self._h, C.c_int32(1 if on else 0)))

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

  def wire_stats(self):
    """(send / recv pairs, exchanges, most pairs in one exchange, host waits for counts) of the last forward +
    backward (mhte_shard_step_wire_stats): the exact-size RCCL form packs a peer's segments into one pair."""
    out = (C.c_int64 * 4)()
    self._libmod.check(self._lib.mhte_shard_step_wire_stats(self._h, out))
    return {"pairs": int(out[0]), "exchanges": int(out[1]), "pairs_max": int(out[2]), "host_waits": int(out[3])}

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

  def set_exact_order(self, on: bool = True):
    """Every rank's duplicate lists summed strictly in occurrence order (mhte_shard_step_set_exact_order)."""
    for h in self._hs:
      self._libmod.check(self._lib.mhte_shard_step_set_exact_order(h, C.c_int32(1 if on else 0)))

  def wire_stats(self):
    """Per rank: mhte_shard_step_wire_stats of the last forward + backward (device copies stand for the pairs)."""
    out = []
    for h in self._hs:
      w = (C.c_int64 * 4)()
      self._libmod.check(self._lib.mhte_shard_step_wire_stats(h, w))
      out.append({"pairs": int(w[0]), "exchanges": int(w[1]), "pairs_max": int(w[2]), "host_waits": int(w[3])})
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
