"""One sparse training step of a MonolithModel on a single table, as the reference's worker + PS
execute it (native_training/distributed_ps.py:282-329 lookup, :489-514 apply_gradients), with every
stage on the GPU and no host round trip inside the step:

   dedup (first-occurrence order)                unique_key_with_value_and_offset
   lookup of the unique ids                       MonolithMultiHashTableLookup
   scatter of unique rows to every occurrence     MonolithFillWithOffsetMap
   ... dense forward / backward (the caller's) ...
   duplicate-gradient sum (occurrence order)      MonolithFillWithOffsetMapGradient   } one fused
   optimizer apply on the unique ids              MonolithMultiHashTableOptimize      } launch

The unique-id count stays in device memory; kernels are launched with fixed grids and mask / loop
themselves.

Pipelining.  The dedup depends on the ids only, not on the table, so — like the reference's
prefetch queue in front of the lookup (distributed_ps_sync.py:199-203) — the dedup of batch s+1 is
done while batch s is looked up and updated: ``forward(ids, next_ids=...)``.  A step is then exactly
TWO launches on one queue (mhte_table_step_forward / _backward, csrc/mhte_step_kernels.h):

   forward   lookup(s)  | run dedup(s+1) | displacement pass of update s-1 (usually idle)
   backward  gradient sum + upsert + optimizer(s) | numbering, heavy work list and table probe(s+1)
             (the probe leaves every distinct id's row handle / slot, or a reserved row for an id the
              table lacks: update s+1 reads no bucket and allocates nothing)

different workgroups of one kernel doing the jobs (a dependency between two HIP queues costs ~10 us
on this part, a kernel boundary on one queue ~2 us).  Batches of up to 65 536 ids.

Two batches of look-ahead: ``forward(ids, next_ids=b1, ahead_ids=b2)``.  The run dedup's chain (ids,
LDS set, scratch CAS, count bump, position lists) is what ends the forward launch; given the batch
after the next as well, it moves into the BACKWARD launch, whose own chains are longer:

   forward   lookup(s) | displacement pass of update s-1
   backward  gradient sum + upsert + optimizer(s) | numbering + table probe(s+1) | run dedup(s+2)

Three dedup workspaces / result buffers rotate so the batches in flight never share scratch.

Without ``next_ids`` (and for rows too wide for the fused kernels) the step runs unpipelined: the
list-building dedup (mhte_unique_unordered / mhte_unique) on a side stream beside the lookup, then
mhte_table_sum_optimize_n."""
from typing import Optional

import numpy as np
import torch

from monolith_amd import _lib
from monolith_amd.distribution_ops import DedupWorkspace, UniqueResult
from monolith_amd.multi_hash_table_ops import MultiHashTable

MAX_PIPELINED_BATCH = 65536  # 64 dedup workgroups x 1024 positions (mhte_step_kernels.h)


class SparseStep:

  def __init__(self, table: MultiHashTable, table_name: str, batch: int,
               exact_order: bool = False, direct: bool = True, fused_backward: bool = True,
               ordered_unique: bool = False):
    self.direct = direct
    self.fused_backward = fused_backward
    # the reference's first-occurrence numbering of the unique ids is only needed by the unfused
    # three-op forward (its gather indexes rows by that numbering) and by wide rows
    self.ordered_unique = ordered_unique or not (direct and fused_backward)
    self.table = table
    self.name = table_name
    self.idx = table._index(table_name)  # pylint: disable=protected-access
    self.dim = table.get_table_dim_sizes()[self.idx]
    ok = int(table._lib.mhte_table_fused_backward_ok(table.handle, self.idx))  # pylint: disable=protected-access
    self.fusable = ok != 0          # 1: SGD / Adagrad / FTRL, 2: any per-element optimizer
    self._full_opts = ok == 2
    if not self.fusable:
      self.ordered_unique = True  # wide rows: segment sum + optimize on CSR lists
    # an occurrence filter: the UNPIPELINED update walks CSR occurrence lists (the admission decision
    # needs each id's count), so its dedup is the ordered one; the pipelined step consults the filter
    # itself and keeps its run dedup
    # (likewise optimizers beyond SGD / Adagrad / FTRL: the unpipelined one-launch backward is
    # compiled for those three; the others take segment sum + update on CSR lists there)
    self._plain_ordered = (self.ordered_unique or getattr(table, "_hash_filter", None) is not None or
                           self._full_opts)
    self.batch = batch
    self.exact_order = exact_order
    dev = torch.device("cuda:%d" % table._device)  # pylint: disable=protected-access
    self.side = torch.cuda.Stream(device=dev)
    n = batch

    # unpipelined path: list-building dedup
    self._ws0 = DedupWorkspace(dev.index)
    self._u0 = UniqueResult(torch.empty(n, dtype=torch.int64, device=dev),
                            torch.empty(n, dtype=torch.int32, device=dev),
                            torch.empty(n + 1, dtype=torch.int32, device=dev),
                            torch.empty(n, dtype=torch.int32, device=dev),
                            torch.zeros(1, dtype=torch.int32, device=dev), None,
                            None if self._plain_ordered else
                            torch.empty(n + 1, dtype=torch.int32, device=dev))
    # pipelined path: three slots — the batch being trained, the batch deduplicated ahead of it, and
    # (two batches of look-ahead) the one after that
    self._ws = [DedupWorkspace(dev.index) for _ in range(3)]
    self._uids = [torch.empty(n, dtype=torch.int64, device=dev) for _ in range(3)]
    self._nu = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(3)]
    self._key = [None, None, None]    # (data_ptr, numel) of the ids whose run dedup the slot holds
    self._cur = 0
    self._nxt = None
    self._ahead = None          # the batch two steps on, handed to forward(): its dedup rides in backward
    self._mode = None           # "pipe" / "plain": how the current batch was deduplicated
    self._done = None           # event: the side-stream dedup of the current batch finished
    self._joined = True
    self.emb_u = torch.empty((n, self.dim), dtype=torch.float32, device=dev)
    self.emb = torch.empty((n, self.dim), dtype=torch.float32, device=dev)
    self.grad_u = torch.empty((n, self.dim), dtype=torch.float32, device=dev)
    lr0 = sum(table._slice_sizes[:self.idx])  # pylint: disable=protected-access
    self.lrs = np.ascontiguousarray(
        table.learning_rate[lr0:lr0 + table._slice_sizes[self.idx]])  # pylint: disable=protected-access

  # unpipelined dedup state (used by the bench's per-kernel timing pass too)
  @property
  def ws(self) -> DedupWorkspace:
    return self._ws0

  @property
  def u(self) -> UniqueResult:
    return self._u0

  def _unique(self, ids):
    if self._plain_ordered:
      self._ws0.unique(ids, want_host_count=False, out=self._u0)
    else:
      self._ws0.unique_unordered(ids, want_host_count=False, out=self._u0)

  def _slot_of(self, key):
    for i in range(3):
      if self._same_batch(self._key[i], key):
        return i
    return None

  def _free_slot(self, *busy):
    for i in range(3):
      if i not in busy:
        return i
    raise AssertionError("no free dedup slot")

  def forward(self, ids: torch.Tensor, next_ids: Optional[torch.Tensor] = None,
              ahead_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Rows for every occurrence.  ``direct`` (default): ONE probe+gather kernel over the B
    occurrences (duplicates of a Zipf head key are served from L2) — the same values as the
    reference's dedup -> lookup(unique) -> FillWithOffsetMap, without waiting for the dedup.
    ``direct=False`` keeps the reference's three-op shape.
    ``next_ids``: the following batch; its dedup is folded into this step's launches and picked
    up by the next ``forward`` (which must receive that same tensor).
    ``ahead_ids``: the batch after ``next_ids``; its run dedup rides in this step's BACKWARD launch,
    so that the next step's forward launch carries lookups only."""
    assert ids.numel() == self.batch
    main = torch.cuda.current_stream()
    if not self.direct:
      self._unique(ids)
      self.table.table_lookup_n(self.idx, self.u.unique_ids, self.u.n_unique_dev, self.emb_u,
                                n_max=self.batch)
      self.ws.gather_rows(self.emb_u, self.u.inverse, self.batch, self.dim, out=self.emb)
      self._joined = True
      self._mode = "plain"
      return self.emb
    key = self._batch_key(ids)
    pipelined = (self.fusable and self.fused_backward and not self.ordered_unique and
                 self.batch <= MAX_PIPELINED_BATCH)
    at = self._slot_of(key) if pipelined else None      # deduplicated ahead by an earlier step?
    if at is not None or (pipelined and next_ids is not None):
      if at is not None:
        self._cur = at
      else:  # first step of a pipeline: dedup in stream order
        # The previous backward may have left ids for the displacement pass, which reads the
        # unique ids and summed gradients of THIS slot's last batch: it must run before the
        # slot's buffers are rewritten (a restarted pipeline lost those updates).
        self.table.table_finish_pending(self.idx)
        self._key = [None, None, None]     # (whatever was prepared ahead belongs to another stream of batches)
        self._ws[self._cur].step_dedup(ids, self._uids[self._cur], self._nu[self._cur])
        self._key[self._cur] = key
      self._nxt = None
      if next_ids is not None:
        assert next_ids.numel() == self.batch
        nkey = self._batch_key(next_ids)
        nxt = self._slot_of(nkey)
        if nxt is None:
          # not deduplicated yet (no look-ahead of two, or the pipeline's first steps): its run dedup
          # rides in this forward launch
          nxt = self._free_slot(self._cur)
          self.table.table_step_forward(self.idx, ids, self.emb, self._ws[nxt], next_ids,
                                        self._uids[nxt], self._nu[nxt])
          self._key[nxt] = nkey
        else:
          self.table.table_step_forward(self.idx, ids, self.emb)     # lookups alone
        self._nxt = nxt
      else:
        self.table.table_step_forward(self.idx, ids, self.emb)
      # slots that hold neither this batch nor the next are stale
      for i in range(3):
        if i != self._cur and i != self._nxt:
          self._key[i] = None
      self._ahead = ahead_ids if (next_ids is not None and ahead_ids is not None) else None
      if self._ahead is not None:
        assert self._ahead.numel() == self.batch
      self._mode = "pipe"
      self._joined = True
      return self.emb
    # unpipelined: the dedup (needed by backward only) runs on the side stream beside the lookup
    here = torch.cuda.Event()
    here.record(main)
    self.side.wait_event(here)
    with torch.cuda.stream(self.side):
      self._unique(ids)
      ev = torch.cuda.Event()
      ev.record(self.side)
    self._done = ev
    self._joined = False
    self._mode = "plain"
    self.table.table_lookup_n(self.idx, ids, None, self.emb, n_max=self.batch)
    return self.emb

  @staticmethod
  def _batch_key(ids):
    """Identity of a batch handed over as ``next_ids``: its memory (address, length) and version
    counter.  The key keeps the tensor referenced, so the address cannot be handed to another
    allocation meanwhile, and a buffer refilled in place bumps the version (views share their
    base's counter) — either way the batch deduplicated ahead is not taken for a different one
    (ADVICE r1).  Not the Python object: ``ids_all[s]`` makes a new view object on every call."""
    return (ids, ids._version, ids.data_ptr(), ids.numel())  # pylint: disable=protected-access

  @staticmethod
  def _same_batch(a, b):
    return (a is not None and a[1] == b[1] and a[2] == b[2] and a[3] == b[3] and
            a[0].device == b[0].device)

  def _join(self):
    if not self._joined:
      torch.cuda.current_stream().wait_event(self._done)
      self._joined = True

  def backward(self, grads: torch.Tensor, update_time: int, global_step: int = 0):
    self._join()
    if self._mode == "pipe":
      cur, nxt = self._cur, self._nxt
      ws_next = self._ws[nxt] if nxt is not None else None
      kw = {}
      if self._ahead is not None and nxt is not None:
        ah = self._free_slot(cur, nxt)
        kw = dict(ws_ahead=self._ws[ah], ahead_ids=self._ahead, uids_ahead=self._uids[ah],
                  n_unique_ahead=self._nu[ah])
        self._key[ah] = self._batch_key(self._ahead)
      self.table.table_step_backward(self.idx, self._ws[cur], ws_next, self._uids[cur],
                                     self._nu[cur], grads, self.grad_u, self.lrs, update_time,
                                     global_step, exact_order=self.exact_order, **kw)
      self._ahead = None
      self._key[cur] = None  # the slot's runs are consumed
    elif self.fused_backward:
      # one launch: per-id gradient sum + upsert + optimizer (mhte_table_sum_optimize_n)
      self.table.table_sum_optimize_n(self.idx, self.ws, self.u, grads, self.grad_u, self.lrs,
                                      update_time, global_step, exact_order=self.exact_order,
                                      n_max=self.batch)
    else:
      self.ws.segment_sum(grads, self.u, self.dim, out=self.grad_u, exact_order=self.exact_order)
      self.table.table_optimize_n(self.idx, self.u.unique_ids, self.u.n_unique_dev, self.grad_u,
                                  self.lrs, update_time, global_step, flags=_lib.MHTE_IDS_UNIQUE,
                                  n_max=self.batch)

  def c_loop(self, ids_all: torch.Tensor, lo: int, hi: int, grad_pool, update_time0: int):
    """Steps [lo, hi) of the contiguous batch array ``ids_all`` [n_batches, batch] enqueued by a plain C
    loop over the C ABI (csrc/eager_loop.c, ``mhte_eager_step_loop``): forward(batch s, next = s + 1) and
    backward(grad_pool[s % len], update_time0 + s) per step, no interpreter in between.  Batch ``lo`` must
    have been deduplicated ahead by the previous step (``forward(.., next_ids=ids_all[lo])``); on return
    batch ``hi`` is, so ``forward(ids_all[hi], ..)`` continues the pipeline."""
    C = _lib.C
    assert self._mode == "pipe" and ids_all.is_contiguous() and ids_all.shape[1] == self.batch
    at = self._slot_of(self._batch_key(ids_all[lo]))
    assert at is not None, "batch lo was not deduplicated ahead"
    E = _lib.eager_lib()
    ws = (C.c_void_p * 3)(*[w._h for w in self._ws])  # pylint: disable=protected-access
    uid = (C.c_void_p * 3)(*[u.data_ptr() for u in self._uids])
    nu = (C.c_void_p * 3)(*[u.data_ptr() for u in self._nu])
    gp = (C.c_void_p * len(grad_pool))(*[g.data_ptr() for g in grad_pool])
    lrs = np.ascontiguousarray(self.lrs, dtype=np.float32)
    r = E.mhte_eager_step_loop(
        self.table.handle, C.c_int32(self.idx), ws, uid, nu, C.c_int32(at), _lib.vp(ids_all),
        C.c_int64(self.batch), C.c_int64(lo), C.c_int64(hi), _lib.vp(self.emb), gp,
        C.c_int32(len(grad_pool)), _lib.vp(self.grad_u), lrs.ctypes.data_as(C.POINTER(C.c_float)),
        C.c_int64(lrs.size), C.c_int64(int(update_time0)),
        C.c_int32(_lib.MHTE_EXACT_ORDER if self.exact_order else 0),
        C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if r < 0:
      _lib.check(-r - 1)
    self._key = [None, None, None]
    self._key[r] = self._batch_key(ids_all[hi])
    self._cur, self._nxt, self._ahead = r, None, None

  def quiesce(self):
    """Host-synchronise and forget the side-stream event.  Call before capturing steps into a
    hipGraph: a capture must not wait on events recorded outside it."""
    torch.cuda.synchronize()
    self._done = None
    self._joined = True

  def n_unique(self) -> int:
    self._join()
    if self._mode == "pipe":
      return int(self._nu[self._cur].item())
    return int(self.u.n_unique_dev.item())


class MultiSparseStep:
  """The sparse step of a model over EVERY table of a ``MultiHashTable`` — what the reference runs
  per step as ``MultiHashTable.lookup`` / ``apply_gradients`` on the ragged (id, id_split) batch of
  its feature tables (native_training/multi_hash_table_ops.py:349-413;
  multi_type_hash_table.py:253-303 builds that layout for a model) — as ONE forward and ONE
  backward launch for all tables together (mhte_multi_step_forward / _backward,
  csrc/mhte_mstep_kernels.h), the dedup of the next batch riding in them:

     forward   per table: lookup(s) | run dedup(s+1)
     backward  per table: gradient sum + upsert + optimizer(s) | numbering of batch s+1
               (+ a displacement launch, usually idle)

  ``forward(ragged, next_ragged)`` returns the flat embedding of ``MultiHashTable.raw_lookup``
  (tables in sorted-name order, rows per OCCURRENCE); ``backward(flat_grad, ...)`` takes the
  gradient in the same layout.  A batch handed over as ``next_ragged`` is picked up by the next
  ``forward`` when it receives that same ``Ragged`` object, unmodified (tensor identity and version
  are checked); any other batch is deduplicated on the spot (two more launches)."""

  def __init__(self, table: MultiHashTable, batch_per_table: int, exact_order: bool = False):
    self.table = table
    self.batch = int(batch_per_table)
    self.exact_order = exact_order
    self._lib = table._lib  # pylint: disable=protected-access
    self._dims = table.get_table_dim_sizes()
    h = _lib.C.c_void_p()
    _lib.check(self._lib.mhte_multi_step_create(table.handle, _lib.C.c_int64(self.batch),
                                                _lib.C.byref(h)))
    self._h = h
    self._ahead = None   # (values tensor, its _version, row_splits bytes) deduplicated ahead
    self._keep = None    # arguments of the launches in flight

  def close(self):
    if getattr(self, "_h", None):
      torch.cuda.synchronize()
      self._lib.mhte_multi_step_destroy(self._h)
      self._h = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass

  @staticmethod
  def _key(r):
    return (r.values, r.values._version, r.row_splits.tobytes())  # pylint: disable=protected-access

  def _stream(self):
    return _lib.C.c_void_p(torch.cuda.current_stream().cuda_stream)

  def forward(self, ragged, next_ragged=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    C = _lib.C
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
      nsp = None
      nv, nsp_p, nsp_n = C.c_void_p(0), None, 0
    _lib.check(self._lib.mhte_multi_step_forward(
        self._h, _lib.vp(ragged.values), sp.ctypes.data_as(C.POINTER(C.c_int64)), C.c_int64(sp.size),
        _lib.vp(out), C.c_int64(out.numel()), nv, nsp_p, C.c_int64(nsp_n),
        C.c_int32(1 if pre else 0), self._stream()))
    self._ahead = self._key(next_ragged) if next_ragged is not None else None
    self._keep = (ragged, next_ragged, out)
    return out

  def backward(self, flat_grad: torch.Tensor, update_time: int, global_step: int = 0):
    C = _lib.C
    lrs = np.ascontiguousarray(self.table.learning_rate, dtype=np.float32)
    _lib.check(self._lib.mhte_multi_step_backward(
        self._h, _lib.vp(flat_grad), C.c_int64(flat_grad.numel()),
        lrs.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(lrs.size), C.c_int64(int(update_time)),
        C.c_int64(int(global_step)), C.c_int32(_lib.MHTE_EXACT_ORDER if self.exact_order else 0),
        self._stream()))

  def unique_counts(self) -> np.ndarray:
    """Distinct ids per table of the batch last given to ``forward`` (synchronises)."""
    out = np.zeros(len(self._dims), dtype=np.int64)
    _lib.check(self._lib.mhte_multi_step_unique_counts(
        self._h, out.ctypes.data_as(_lib.C.POINTER(_lib.C.c_int64)), self._stream()))
    return out
