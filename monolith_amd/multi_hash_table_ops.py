"""MultiHashTable on MI355X — host-side mirror of monolith/native_training/multi_hash_table_ops.py.

Same class, method names, argument meaning and error behaviour as the reference's Python wrapper
(reference multi_hash_table_ops.py:131-548) so that the parity tests read like
multi_hash_table_ops_test.py; the TF custom ops it calls are replaced by the C ABI of
include/monolith_amd_hash_table.h (libmhte.so).  Differences that follow from leaving TF:
  * tensors are torch CUDA tensors; a RaggedTensor is ``Ragged(values, row_splits)`` with
    row_splits on the host (the TF kernels take them as HostMemory);
  * ops are enqueued on the current torch CUDA stream and ordered by it; methods still return the
    table so that call sites written as ``table = table.assign_add(...)`` work unchanged
    (reference ``_copy_with_new_table`` :425-428).
"""
import ctypes as C
import time
from typing import Dict, List, NamedTuple, Optional, Sequence, Tuple

import numpy as np
import torch

from monolith_amd import _lib
from monolith_amd import entry
from monolith_amd._lib import check, vp


class Ragged(NamedTuple):
  """values: int64 CUDA tensor [n]; row_splits: host int64 array [T+1]"""
  values: torch.Tensor
  row_splits: np.ndarray

  def row_lengths(self):
    return np.diff(self.row_splits)


def _stream():
  return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _i64p(a: np.ndarray):
  return a.ctypes.data_as(C.POINTER(C.c_int64))


def _i32p(a: np.ndarray):
  return a.ctypes.data_as(C.POINTER(C.c_int32))


def _f32p(a: np.ndarray):
  return a.ctypes.data_as(C.POINTER(C.c_float))


def infer_dim_size(table_config: entry.EmbeddingHashTableConfig) -> int:
  return table_config.dim_size


def _lower_config(name: str, cfg: entry.HashTableConfigInstance, keep: list):
  tc = cfg.table_config
  segs = (_lib.SegmentConfig * len(tc.segments))()
  for i, s in enumerate(tc.segments):
    segs[i].dim_size = s.dim_size
    segs[i].opt_type = s.optimizer.opt_type
    for k, v in enumerate(s.optimizer.params()):
      segs[i].opt_params[k] = v
    segs[i].init_type = s.initializer.init_type
    segs[i].init_value = s.initializer.value
    segs[i].init_value2 = getattr(s.initializer, "value2", 0.0)
  c = _lib.TableConfig()
  bname = name.encode()
  c.name = bname
  c.n_segments = len(tc.segments)
  c.segments = segs
  c.initial_capacity = int(tc.initial_capacity)
  c.reserve_rows = int(tc.reserve_rows)
  c.max_load_factor = float(tc.max_load_factor)
  se = tc.slot_expire_time_config
  c.default_expire_days = int(se.default_expire_time) if int(se.default_expire_time) != 0 else -1
  slots = np.ascontiguousarray(list(se.slot_expire_times.keys()), dtype=np.int64)
  days = np.ascontiguousarray(list(se.slot_expire_times.values()), dtype=np.int32)
  c.n_slot_expire = int(slots.size)
  c.expire_slots = _i64p(slots)
  c.expire_days = _i32p(days)
  so = tc.slot_occurrence_threshold_config
  c.default_occurrence_threshold = int(so.default_occurrence_threshold)
  oslots = np.ascontiguousarray(list(so.slot_occurrence_thresholds.keys()), dtype=np.int64)
  othr = np.ascontiguousarray(list(so.slot_occurrence_thresholds.values()), dtype=np.int32)
  c.n_slot_occurrence = int(oslots.size)
  c.occurrence_slots = _i64p(oslots)
  c.occurrence_thresholds = _i32p(othr)
  c.enable_feature_eviction = 1 if tc.enable_feature_eviction else 0
  c.feature_evict_every_n_hours = int(tc.feature_evict_every_n_hours)
  keep.extend([segs, bname, slots, days, oslots, othr])
  return c


class HashFilter:
  """The hash filter resource of the reference (hash_filter_ops.create_hash_filter; the
  ``filter_handle`` input of CreateMonolithMultiHashTable): a counting admission filter in HBM,
  shared by the tables of the MultiHashTable it is attached to."""

  def __init__(self, capacity: int = 300000000, split_num: int = 7, device: Optional[int] = None,
               config: Optional[bytes] = None):
    """``config``: serialized SlotOccurrenceThresholdConfig, the hash filter op's attr."""
    if not torch.cuda.is_available():
      raise _lib.MhteError(_lib.MHTE_UNAVAILABLE, "HashFilter needs a HIP device")
    self._lib = _lib.lib()
    self._device = torch.cuda.current_device() if device is None else int(device)
    h = C.c_void_p()
    if config is None:
      check(self._lib.mhte_hash_filter_create(C.c_uint64(int(capacity)), C.c_int32(int(split_num)),
                                              C.c_int32(self._device), C.byref(h)))
    else:
      check(self._lib.mhte_hash_filter_create_from_proto(
          C.c_uint64(int(capacity)), C.c_int32(int(split_num)), config, C.c_int64(len(config)),
          C.c_int32(self._device), C.byref(h)))
    self._h = h

  def get(self, ids: torch.Tensor) -> torch.Tensor:
    """Seen counts (0..15) of ``ids`` (Filter::get)."""
    ids = ids.to(device="cuda:%d" % self._device, dtype=torch.int64).contiguous()
    out = torch.empty(ids.numel(), dtype=torch.int32, device=ids.device)
    check(self._lib.mhte_hash_filter_get(self._h, vp(ids), C.c_int64(ids.numel()), vp(out),
                                         _stream()))
    return out

  def _stats(self):
    out = (C.c_int64 * (4 + 64))()
    check(self._lib.mhte_hash_filter_stats(self._h, out, C.c_int32(4 + 64), _stream()))
    return out

  def num_elements(self) -> List[int]:
    """elements per split (HashFilter::estimated_total_element of each)."""
    s = self._stats()
    return [int(s[4 + i]) for i in range(int(s[3]))]

  def estimated_total_element(self) -> int:
    return sum(self.num_elements())

  def failure_count(self) -> int:
    return int(self._stats()[2])

  def save(self, basename: str) -> "HashFilter":
    """hash_filter_ops.save_hash_filter (MonolithHashFilterSave): one file per split."""
    import os
    d = os.path.dirname(basename)
    if d:
      os.makedirs(d, exist_ok=True)
    check(self._lib.mhte_hash_filter_save(self._h, basename.encode("utf-8"), _stream()))
    return self

  def restore(self, basename: str) -> "HashFilter":
    """hash_filter_ops.restore_hash_filter (MonolithHashFilterRestore)."""
    check(self._lib.mhte_hash_filter_restore(self._h, basename.encode("utf-8"), _stream()))
    return self

  def close(self):
    if getattr(self, "_h", None):
      torch.cuda.synchronize(self._device)
      self._lib.mhte_hash_filter_destroy(self._h)
      self._h = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass


class ProbabilisticFilter(HashFilter):
  """hash_filter_ops.create_probabilistic_filter (MonolithProbabilisticFilter): stateless admission
  with probability count / threshold (``equal_probability``: 1 - (1 - p)^count)."""

  def __init__(self, equal_probability: bool = False, device: Optional[int] = None,
               config: Optional[bytes] = None, seed: int = 0):  # pylint: disable=super-init-not-called
    if not torch.cuda.is_available():
      raise _lib.MhteError(_lib.MHTE_UNAVAILABLE, "ProbabilisticFilter needs a HIP device")
    self._lib = _lib.lib()
    self._device = torch.cuda.current_device() if device is None else int(device)
    h = C.c_void_p()
    check(self._lib.mhte_hash_filter_create_probabilistic(
        C.c_int32(1 if equal_probability else 0), C.c_uint64(int(seed)), config,
        C.c_int64(len(config) if config else 0), C.c_int32(self._device), C.byref(h)))
    self._h = h


class MultiHashTable:
  """The GPU-resident equivalent of the reference's ``MultiHashTable`` resource
  (runtime/ops/multi_hash_table.h:29-72): T embedding tables ordered by sorted name."""
  NAME_PREFIX = "MonolithMultiHashTable"
  _names_in_use = set()

  def __init__(self, configs: Dict[str, entry.HashTableConfigInstance], name_suffix: str = "",
               device: Optional[int] = None, hash_filter: Optional["HashFilter"] = None):
    if not torch.cuda.is_available():
      raise _lib.MhteError(_lib.MHTE_UNAVAILABLE,
                           "MultiHashTable needs a HIP device; there is no CPU fallback")
    self._lib = _lib.lib()
    self._device = torch.cuda.current_device() if device is None else int(device)
    self._table_names = tuple(sorted(configs.keys()))
    self._configs = configs
    self._dims = tuple(infer_dim_size(configs[n].table_config) for n in self._table_names)
    lrs: List[float] = []
    for n in self._table_names:
      cfg = configs[n]
      if len(cfg.learning_rate_fns) != len(cfg.table_config.segments):
        raise ValueError("Size of learning_rate_fns and size of segments must be equal.")
      lrs.extend(cfg.call_learning_rate_fns())
    self._learning_rate = np.ascontiguousarray(lrs, dtype=np.float32)
    self._shared_name = "_".join([MultiHashTable.NAME_PREFIX, name_suffix])
    if self._shared_name in MultiHashTable._names_in_use:
      raise ValueError("shared_name {} has already been used.".format(self._shared_name))
    keep = []
    arr = (_lib.TableConfig * len(self._table_names))(
        *[_lower_config(n, configs[n], keep) for n in self._table_names])
    h = C.c_void_p()
    check(self._lib.mhte_multi_table_create(arr, len(self._table_names), self._device,
                                            self._shared_name.encode(), C.byref(h)))
    self._h = h
    self._hash_filter = hash_filter   # (kept alive with the table)
    if hash_filter is not None:
      check(self._lib.mhte_multi_table_set_filter(self._h, hash_filter._h))  # pylint: disable=protected-access
    MultiHashTable._names_in_use.add(self._shared_name)
    self._slice_sizes = tuple(
        self._lib.mhte_table_slice_size(self._h, i) for i in range(len(self._table_names)))
    # Feature eviction (tf_bridge.cc:73-104) lives behind the boundary: the library checks the
    # cadence on its update entry points and enqueues the scan on their stream.

  @classmethod
  def from_configs(cls, configs: Dict[str, entry.HashTableConfigInstance], *args, **kwargs):
    return cls(configs, *args, **kwargs)

  def close(self):
    if getattr(self, "_h", None):
      torch.cuda.synchronize(self._device)
      self._lib.mhte_multi_table_destroy(self._h)
      self._h = None
      MultiHashTable._names_in_use.discard(self._shared_name)

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass

  # ------------------------------------------------------------------ properties
  @property
  def table_names(self):
    return self._table_names

  @property
  def handle(self):
    return self._h

  @property
  def shared_name(self):
    return self._shared_name

  @property
  def learning_rate(self):
    """concat over tables (sorted by name) of per-segment learning rates (reference :271-281)"""
    return self._learning_rate

  def set_learning_rate(self, lrs: Sequence[float]):
    lrs = np.ascontiguousarray(lrs, dtype=np.float32)
    assert lrs.size == self._learning_rate.size
    self._learning_rate = lrs

  def get_table_dim_sizes(self):
    return self._dims

  def _dev(self, t: torch.Tensor, dtype) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
      t = torch.as_tensor(np.asarray(t))
    return t.to(device="cuda:%d" % self._device, dtype=dtype).contiguous()

  # ------------------------------------------------------------------ BaseMultiHashTable
  def assign(self, slot_to_id_and_value: Dict[str, Tuple[torch.Tensor, torch.Tensor]],
             req_time: int = 0) -> "MultiHashTable":
    ragged_id = self.get_ragged_id({k: v[0] for k, v in slot_to_id_and_value.items()})
    flat_value = self.get_flat_value({k: v[1] for k, v in slot_to_id_and_value.items()})
    return self.raw_assign(ragged_id, flat_value, req_time)

  def assign_add(self, slot_to_id_and_value: Dict[str, Tuple[torch.Tensor, torch.Tensor]],
                 req_time: int = 0) -> "MultiHashTable":
    ragged_id = self.get_ragged_id({k: v[0] for k, v in slot_to_id_and_value.items()})
    flat_value = self.get_flat_value({k: v[1] for k, v in slot_to_id_and_value.items()})
    check(self._lib.mhte_assign_add(self._h, vp(ragged_id.values), _i64p(ragged_id.row_splits),
                                    C.c_int64(ragged_id.row_splits.size), vp(flat_value),
                                    C.c_int64(flat_value.numel()), C.c_int64(int(req_time)),
                                    C.c_int32(0), _stream()))
    return self

  def reinitialize(self, slot: str, ids: torch.Tensor,
                   now: int = 0) -> Tuple["MultiHashTable", torch.Tensor]:
    ids = self._dev(ids, torch.int64)
    status = torch.empty(ids.numel(), dtype=torch.int32, device=ids.device)
    check(self._lib.mhte_reinitialize(self._h, slot.encode(), vp(ids), C.c_int64(ids.numel()),
                                      vp(status), C.c_int64(int(now)), _stream()))
    return self, status

  def lookup(self, slot_to_id: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    ragged_id = self.get_ragged_id(slot_to_id)
    flat_embedding = self.raw_lookup(ragged_id)
    slot_to_embeddings = self.get_embeddings(ragged_id, flat_embedding)
    return {k: v for k, v in slot_to_embeddings.items() if k in slot_to_id}

  def apply_gradients(self, slot_to_id_and_grad: Dict[str, Tuple[torch.Tensor, torch.Tensor]],
                      global_step: int = 0, req_time: int = 0,
                      ids_unique: bool = False) -> "MultiHashTable":
    ragged_id = self.get_ragged_id({k: v[0] for k, v in slot_to_id_and_grad.items()})
    flat_grad = self.get_flat_value({k: v[1] for k, v in slot_to_id_and_grad.items()})
    return self.raw_apply_gradients(ragged_id, flat_grad, global_step, req_time, ids_unique)

  # ------------------------------------------------------------------ fused ops (sync training)
  def fused_lookup(self, ids: torch.Tensor, fused_slot_size, num_of_shards: int, req_time: int = 0):
    """reference :442-454 -> (embeddings, embedding_splits, id_offsets, embedding_offsets, ids)"""
    ids = self._dev(ids, torch.int64)
    fss = np.ascontiguousarray(
        fused_slot_size.cpu().numpy() if isinstance(fused_slot_size, torch.Tensor)
        else fused_slot_size, dtype=np.int32)
    T = len(self._table_names)
    if fss.size != T * num_of_shards:
      raise _lib.InvalidArgumentError(_lib.MHTE_INVALID_ARGUMENT,
                                      "fused_slot_size must have num_of_shards*num_tables entries")
    splits = np.empty(num_of_shards, np.int32)
    id_off = np.empty(T * num_of_shards + 1, np.int32)
    emb_off = np.empty(T * num_of_shards + 1, np.int32)
    tk, te = C.c_int64(0), C.c_int64(0)
    check(self._lib.mhte_compute_fused_offsets(self._h, _i32p(fss), C.c_int32(num_of_shards),
                                               _i32p(id_off), _i32p(emb_off), _i32p(splits),
                                               C.byref(tk), C.byref(te)))
    if tk.value > ids.numel():
      raise _lib.InvalidArgumentError(_lib.MHTE_INVALID_ARGUMENT, "ids shorter than fused_slot_size")
    emb = torch.empty(te.value, dtype=torch.float32, device=ids.device)
    check(self._lib.mhte_fused_lookup(self._h, vp(ids), _i32p(fss), C.c_int32(num_of_shards),
                                      C.c_int64(int(req_time)), vp(emb), C.c_int64(emb.numel()),
                                      _i32p(splits), _i32p(id_off), _i32p(emb_off), _stream()))
    return emb, splits, id_off, emb_off, ids

  def fused_apply_gradient(self, ids: torch.Tensor, indices, fused_slot_size, id_grads: torch.Tensor,
                           id_offsets, grad_offsets, global_step: int, req_time: int,
                           num_of_shards: int, enable_grad_accumulation: bool = False,
                           ids_unique_per_segment: bool = False) -> "MultiHashTable":
    """reference :458-483.  ``indices`` is accepted for signature parity (the CPU kernel ignores
    it too, multi_hash_table_update_op.cc:271)."""
    ids = self._dev(ids, torch.int64)
    id_grads = self._dev(id_grads, torch.float32)
    fss = np.ascontiguousarray(fused_slot_size, dtype=np.int32)
    ko = np.ascontiguousarray(id_offsets, dtype=np.int32)
    go = np.ascontiguousarray(grad_offsets, dtype=np.int32)
    flags = (_lib.MHTE_SUM_DUPLICATES if enable_grad_accumulation else 0) | (
        _lib.MHTE_IDS_UNIQUE if ids_unique_per_segment else 0)
    check(self._lib.mhte_fused_optimize(self._h, vp(ids), _i32p(fss), vp(id_grads),
                                        C.c_int64(id_grads.numel()), _i32p(ko), _i32p(go),
                                        _f32p(self._learning_rate),
                                        C.c_int64(self._learning_rate.size),
                                        C.c_int64(int(req_time)), C.c_int64(int(global_step)),
                                        C.c_int32(num_of_shards), C.c_int32(flags), _stream()))
    return self

  # ------------------------------------------------------------------ RawMultiTypeHashTable
  def raw_lookup(self, ragged_id: Ragged) -> torch.Tensor:
    lens = ragged_id.row_lengths()
    total = int(sum(int(l) * d for l, d in zip(lens, self._dims))) if lens.size == len(
        self._dims) else 0
    out = torch.empty(total, dtype=torch.float32, device=ragged_id.values.device)
    check(self._lib.mhte_lookup(self._h, vp(ragged_id.values), _i64p(ragged_id.row_splits),
                                C.c_int64(ragged_id.row_splits.size), vp(out),
                                C.c_int64(out.numel()), _stream()))
    return out

  def maybe_evict(self, force_check: bool = False) -> List[str]:
    """Deprecated (round-1 API).  The eviction cadence is checked inside the library now (every
    update entry point, at most every 10 s; mhte_table_config.enable_feature_eviction), so a plain
    call has nothing to do.  ``force_check=True`` still means "scan now": every table is scanned
    against its TTLs (mhte_table_evict) and the names are returned."""
    import warnings
    warnings.warn("MultiHashTable.maybe_evict is deprecated: the library runs the eviction cadence "
                  "itself; use evict(name) for an immediate scan", DeprecationWarning, stacklevel=2)
    if not force_check:
      return []
    for name in self._table_names:
      self.evict(name)
    return list(self._table_names)

  def raw_apply_gradients(self, ragged_id: Ragged, flat_grad: torch.Tensor, global_step: int = 0,
                          req_time: int = 0, ids_unique: bool = False) -> "MultiHashTable":
    flat_grad = self._dev(flat_grad, torch.float32)
    check(self._lib.mhte_optimize(self._h, vp(ragged_id.values), _i64p(ragged_id.row_splits),
                                  C.c_int64(ragged_id.row_splits.size), vp(flat_grad),
                                  C.c_int64(flat_grad.numel()), _f32p(self._learning_rate),
                                  C.c_int64(self._learning_rate.size), C.c_int64(int(req_time)),
                                  C.c_int64(int(global_step)),
                                  C.c_int32(_lib.MHTE_IDS_UNIQUE if ids_unique else 0), _stream()))
    return self

  def raw_assign(self, ragged_id: Ragged, flat_value: torch.Tensor,
                 req_time: int = 0) -> "MultiHashTable":
    flat_value = self._dev(flat_value, torch.float32)
    check(self._lib.mhte_assign(self._h, vp(ragged_id.values), _i64p(ragged_id.row_splits),
                                C.c_int64(ragged_id.row_splits.size), vp(flat_value),
                                C.c_int64(flat_value.numel()), C.c_int64(int(req_time)),
                                C.c_int32(0), _stream()))
    return self

  def get_embeddings(self, ragged_id: Ragged, value: torch.Tensor) -> Dict[str, torch.Tensor]:
    d, off = {}, 0
    for name, n, dim in zip(self._table_names, ragged_id.row_lengths(), self._dims):
      d[name] = value[off:off + int(n) * dim].view(int(n), dim)
      off += int(n) * dim
    return d

  def get_ragged_id(self, slot_to_id: Dict[str, torch.Tensor]) -> Ragged:
    unknown = set(slot_to_id) - set(self._table_names)
    if unknown:
      raise KeyError("unknown tables: %s" % sorted(unknown))
    dev = "cuda:%d" % self._device
    tensors, splits = [], [0]
    for name in self._table_names:
      t = slot_to_id.get(name)
      t = torch.empty(0, dtype=torch.int64, device=dev) if t is None else self._dev(
          t, torch.int64).reshape(-1)
      tensors.append(t)
      splits.append(splits[-1] + t.numel())
    return Ragged(torch.cat(tensors) if tensors else torch.empty(0, dtype=torch.int64, device=dev),
                  np.ascontiguousarray(splits, dtype=np.int64))

  def get_flat_value(self, slot_to_value: Dict[str, torch.Tensor]) -> torch.Tensor:
    dev = "cuda:%d" % self._device
    tensors = []
    for name in self._table_names:
      v = slot_to_value.get(name)
      if v is not None:
        tensors.append(self._dev(v, torch.float32).reshape(-1))
    return torch.cat(tensors) if tensors else torch.empty(0, dtype=torch.float32, device=dev)

  # ------------------------------------------------------------------ boundary completion
  @classmethod
  def from_serialized_config(cls, config: bytes, name_suffix: str = "", device: Optional[int] = None,
                             hash_filter: Optional["HashFilter"] = None, reserve_rows: int = 0,
                             max_load_factor: float = 0.0) -> "MultiHashTable":
    """CreateMonolithMultiHashTable with its ``config`` input as the reference passes it: a
    serialized MultiEmbeddingHashTableConfig (mhte_multi_table_create_from_proto)."""
    self = cls.__new__(cls)
    self._lib = _lib.lib()
    self._device = torch.cuda.current_device() if device is None else int(device)
    self._shared_name = "_".join([MultiHashTable.NAME_PREFIX, name_suffix])
    if self._shared_name in MultiHashTable._names_in_use:
      raise ValueError("shared_name {} has already been used.".format(self._shared_name))
    lrs = np.zeros(1 << 16, dtype=np.float32)   # (one float per segment of the model; checked below)
    h = C.c_void_p()
    check(self._lib.mhte_multi_table_create_from_proto(
        config, C.c_int64(len(config)), hash_filter._h if hash_filter is not None else C.c_void_p(0),  # pylint: disable=protected-access
        C.c_uint64(int(reserve_rows)), C.c_float(float(max_load_factor)), C.c_int32(self._device),
        self._shared_name.encode(), _f32p(lrs), C.c_int32(lrs.size), C.byref(h)))
    self._h = h
    self._hash_filter = hash_filter
    MultiHashTable._names_in_use.add(self._shared_name)
    T = self._lib.mhte_num_tables(self._h)
    self._table_names = tuple(self._lib.mhte_table_name(self._h, i).decode() for i in range(T))
    self._dims = tuple(self._lib.mhte_table_dim(self._h, i) for i in range(T))
    self._slice_sizes = tuple(self._lib.mhte_table_slice_size(self._h, i) for i in range(T))
    if sum(self._slice_sizes) > lrs.size:
      self._lib.mhte_multi_table_destroy(self._h)
      self._h = None
      MultiHashTable._names_in_use.discard(self._shared_name)
      raise _lib.InvalidArgumentError(
          _lib.MHTE_INVALID_ARGUMENT, "config has %d segments: more than the %d learning rates this "
          "binding reads back" % (sum(self._slice_sizes), lrs.size))
    self._learning_rate = np.ascontiguousarray(lrs[:sum(self._slice_sizes)])
    self._configs = None
    return self

  @staticmethod
  def is_initialized(shared_name: str) -> bool:
    """IsHashTableInitialized (multi_hash_table_op.cc:145-166)."""
    return bool(_lib.lib().mhte_multi_table_is_initialized(shared_name.encode()))

  def lookup_entry(self, slot_to_id: Dict[str, torch.Tensor]) -> Dict[str, List[bytes]]:
    """MonolithMultiHashTableLookupEntry: the serialized EntryDump of every id (b"" when absent)."""
    ragged_id = self.get_ragged_id(slot_to_id)
    n = int(ragged_id.row_splits[-1])
    offs = np.zeros(n + 1, dtype=np.int64)
    need = C.c_int64(0)
    cap = max(1, n) * 64
    while True:
      buf = C.create_string_buffer(cap)
      rc = self._lib.mhte_lookup_entry(self._h, vp(ragged_id.values), _i64p(ragged_id.row_splits),
                                       C.c_int64(ragged_id.row_splits.size), buf, C.c_int64(cap),
                                       _i64p(offs), C.byref(need), _stream())
      if rc == _lib.MHTE_INVALID_ARGUMENT and need.value > cap:
        cap = int(need.value)
        continue
      check(rc)
      break
    raw = buf.raw
    out, k = {}, 0
    for name, m in zip(self._table_names, ragged_id.row_lengths()):
      out[name] = [raw[offs[k + i]:offs[k + i + 1]] for i in range(int(m))]
      k += int(m)
    return {k_: v for k_, v in out.items() if k_ in slot_to_id}

  def save_as_tensor(self, name_or_idx, shard_idx: int, num_shards: int, limit: int,
                     offset: int) -> Tuple[int, List[bytes]]:
    """MonolithHashTableSaveAsTensor (hash_table_ops.py:335-361 of the reference): up to ``limit``
    serialized EntryDump strings of one table, shard ``shard_idx`` of ``num_shards`` of its bucket
    array, resumed ``offset`` slots into the shard -> (new_offset, entries).  Iterate as
    hash_table_utils.iterate_table_and_apply does: until a call returns fewer than ``limit``."""
    i = name_or_idx if isinstance(name_or_idx, int) else self._index(name_or_idx)
    n_off = max(int(limit), 1) + 1
    offs = np.zeros(n_off, dtype=np.int64)
    new_offset, n_ent, need = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    cap = max(1, int(limit)) * 64
    while True:
      buf = C.create_string_buffer(cap)
      rc = self._lib.mhte_table_save_as_tensor(
          self._h, C.c_int32(i), C.c_int32(int(shard_idx)), C.c_int32(int(num_shards)), C.c_int64(int(limit)),
          C.c_int64(int(offset)), C.byref(new_offset), buf, C.c_int64(cap), _i64p(offs), C.c_int64(n_off),
          C.byref(n_ent), C.byref(need), _stream())
      if rc == _lib.MHTE_INVALID_ARGUMENT and need.value > cap:
        cap = int(need.value)
        continue
      check(rc)
      break
    raw = buf.raw
    return int(new_offset.value), [raw[offs[k]:offs[k + 1]] for k in range(int(n_ent.value))]

  @staticmethod
  def feature_stat(basename: str) -> Dict[str, int]:
    """MonolithMultiHashTableFeatureStat: entries per table name in a checkpoint's .meta files."""
    L = _lib.lib()
    names = C.create_string_buffer(1 << 16)
    counts = (C.c_uint64 * 1024)()
    n = C.c_int32(0)
    check(L.mhte_feature_stat(basename.encode("utf-8"), names, C.c_int64(len(names)), counts,
                              C.c_int32(1024), C.byref(n)))
    parts = names.raw.split(b"\0")[:n.value]
    return {p.decode(): int(counts[i]) for i, p in enumerate(parts)}

  # ------------------------------------------------------------------ introspection / maintenance
  def _index(self, name: str) -> int:
    i = self._lib.mhte_table_index(self._h, name.encode())
    if i < 0:
      raise KeyError(name)
    return i

  def size(self, name: str) -> int:
    n = C.c_int64(0)
    check(self._lib.mhte_table_size(self._h, C.c_int32(self._index(name)), C.byref(n), _stream()))
    return int(n.value)

  def stats(self, name: str) -> _lib.TableStats:
    s = _lib.TableStats()
    check(self._lib.mhte_table_get_stats(self._h, C.c_int32(self._index(name)), C.byref(s),
                                         _stream()))
    return s

  def set_count_hits(self, name: str, enable: bool = True) -> "MultiHashTable":
    check(self._lib.mhte_table_set_count_hits(self._h, C.c_int32(self._index(name)),
                                              C.c_int32(1 if enable else 0)))
    return self

  def contains(self, name: str, ids: torch.Tensor) -> torch.Tensor:
    ids = self._dev(ids, torch.int64)
    out = torch.empty(ids.numel(), dtype=torch.int32, device=ids.device)
    check(self._lib.mhte_table_contains(self._h, C.c_int32(self._index(name)), vp(ids),
                                        C.c_int64(ids.numel()), vp(out), _stream()))
    return out.bool()

  def save(self, basename: str, nshards: int = -1) -> "MultiHashTable":
    """MultiHashTable.save (NT/multi_hash_table_ops.py:478-497): the reference's on-disk format
    (``<basename>-%05d-of-%05d`` TFRecord+SNAPPY EntryDump shards and ``.meta`` sidecars); rows
    expired relative to the table's max update time are dropped."""
    import os
    d = os.path.dirname(basename)
    if d:
      os.makedirs(d, exist_ok=True)
    check(self._lib.mhte_multi_table_save(self._h, basename.encode("utf-8"), C.c_int32(int(nshards)),
                                          _stream()))
    return self

  def restore(self, basename: str) -> "MultiHashTable":
    """MultiHashTable.restore (NT/multi_hash_table_ops.py:499-513): tables of the checkpoint this
    MultiHashTable does not have are skipped, tables it has but the checkpoint lacks stay as they
    are."""
    check(self._lib.mhte_multi_table_restore(self._h, basename.encode("utf-8"), _stream()))
    return self

  def save_table(self, name: str, basename: str, nshards: int = -1) -> "MultiHashTable":
    """HashTable.save of ONE table (NT/hash_table_ops.py:313-320, MonolithHashTableSave): the
    single-table layout — ``<basename>-%05d-of-%05d`` uncompressed TFRecord files of EntryDump, no
    ``.meta`` (what the reference's exported saved_models carry in ``assets/``)."""
    import os
    d = os.path.dirname(basename)
    if d:
      os.makedirs(d, exist_ok=True)
    check(self._lib.mhte_table_save(self._h, C.c_int32(self._index(name)), basename.encode("utf-8"),
                                    C.c_int32(int(nshards)), _stream()))
    return self

  def restore_table(self, name: str, basename: str) -> "MultiHashTable":
    """HashTable.restore of ONE table (NT/hash_table_ops.py:322-325, MonolithHashTableRestore):
    validates the shard set, CLEARS the table, then upserts every record of every shard."""
    check(self._lib.mhte_table_restore(self._h, C.c_int32(self._index(name)), basename.encode("utf-8"),
                                       _stream()))
    return self

  def clear_table(self, name: str) -> "MultiHashTable":
    """EmbeddingHashTableInterface::Clear: no entries, capacity kept."""
    check(self._lib.mhte_table_clear(self._h, C.c_int32(self._index(name)), _stream()))
    return self

  def evict(self, name: str, max_update_time: int = -1) -> "MultiHashTable":
    check(self._lib.mhte_table_evict(self._h, C.c_int32(self._index(name)),
                                     C.c_int64(int(max_update_time)), _stream()))
    return self

  def dump(self, name: str, with_rows: bool = True):
    """(ids, positions=bucket*4+slot, ts, rows[n, dim+state]) in bucket-major order."""
    i = self._index(name)
    n = self.size(name)
    dev = "cuda:%d" % self._device
    rf = self._lib.mhte_table_row_floats(self._h, i)
    ids = torch.empty(n + 1, dtype=torch.int64, device=dev)
    pos = torch.empty(n + 1, dtype=torch.int64, device=dev)
    ts = torch.empty(n + 1, dtype=torch.int32, device=dev)
    rows = torch.empty((n + 1, rf), dtype=torch.float32, device=dev) if with_rows else None
    m = C.c_int64(0)
    check(self._lib.mhte_table_dump(self._h, C.c_int32(i), C.c_int64(n + 1), vp(ids), vp(pos),
                                    vp(ts), vp(rows), C.byref(m), _stream()))
    m = int(m.value)
    return ids[:m], pos[:m], ts[:m], (rows[:m] if with_rows else None)

  # single-table, device-side-count forms used by the fused step and the bench
  def table_lookup_n(self, name_or_idx, ids: torch.Tensor, n_dev: Optional[torch.Tensor],
                     out: torch.Tensor, n_max: Optional[int] = None):
    i = name_or_idx if isinstance(name_or_idx, int) else self._index(name_or_idx)
    check(self._lib.mhte_table_lookup_n(self._h, C.c_int32(i), vp(ids),
                                        C.c_int64(ids.numel() if n_max is None else n_max),
                                        vp(n_dev), vp(out), _stream()))
    return out

  def table_optimize_n(self, name_or_idx, ids: torch.Tensor, n_dev: Optional[torch.Tensor],
                       grads: torch.Tensor, lrs: np.ndarray, update_time: int, global_step: int = 0,
                       flags: int = _lib.MHTE_IDS_UNIQUE, n_max: Optional[int] = None):
    i = name_or_idx if isinstance(name_or_idx, int) else self._index(name_or_idx)
    lrs = np.ascontiguousarray(lrs, dtype=np.float32)
    check(self._lib.mhte_table_optimize_n(self._h, C.c_int32(i), vp(ids),
                                          C.c_int64(ids.numel() if n_max is None else n_max),
                                          vp(n_dev), vp(grads), _f32p(lrs), C.c_int64(lrs.size),
                                          C.c_int64(int(update_time)), C.c_int64(int(global_step)),
                                          C.c_int32(flags), _stream()))
    return self

  def table_sum_optimize_n(self, name_or_idx, ws, u, grads: torch.Tensor, grad_unique: torch.Tensor,
                           lrs: np.ndarray, update_time: int, global_step: int = 0,
                           exact_order: bool = False, n_max: Optional[int] = None,
                           defer_slowpath: bool = False):
    """Fused backward (mhte_table_sum_optimize_n): duplicate-gradient sum over the occurrence
    lists of ``u`` (the UniqueResult of ``ws``'s most recent ``unique``) + optimizer apply on the
    unique ids, one launch."""
    i = name_or_idx if isinstance(name_or_idx, int) else self._index(name_or_idx)
    lrs = np.ascontiguousarray(lrs, dtype=np.float32)
    n = u.inverse.numel() if n_max is None else n_max
    list_end = u.list_end if u.list_end is not None else u.seg_off[1:]
    check(self._lib.mhte_table_sum_optimize_n(
        self._h, C.c_int32(i), ws._h, vp(u.unique_ids), C.c_int64(n), vp(u.n_unique_dev),  # pylint: disable=protected-access
        vp(grads), vp(u.inverse), vp(u.seg_off), vp(list_end), vp(u.seg_pos), C.c_int64(n),
        vp(grad_unique),
        _f32p(lrs), C.c_int64(lrs.size), C.c_int64(int(update_time)), C.c_int64(int(global_step)),
        C.c_int32((_lib.MHTE_EXACT_ORDER if exact_order else 0) |
                  (_lib.MHTE_DEFER_SLOWPATH if defer_slowpath else 0)), _stream()))
    return self

  def table_finish_pending(self, name_or_idx):
    i = name_or_idx if isinstance(name_or_idx, int) else self._index(name_or_idx)
    check(self._lib.mhte_table_finish_pending(self._h, C.c_int32(i), _stream()))
    return self

  # pipelined step: the dedup of the next batch rides in the two launches of the current one
  def table_step_forward(self, name_or_idx, ids: torch.Tensor, out: torch.Tensor, ws_next=None,
                         next_ids: Optional[torch.Tensor] = None,
                         uids_next: Optional[torch.Tensor] = None,
                         n_unique_next: Optional[torch.Tensor] = None, ws_cur=None):
    """mhte_table_step_forward: lookup of ``ids`` + run dedup of ``next_ids`` into ``ws_next``
    (+ the displacement pass the previous ``table_step_backward`` left).  ``ws_cur``: the workspace
    holding the run dedup of ``ids`` itself — the launch then also reserves the row handles its
    update will need."""
    i = name_or_idx if isinstance(name_or_idx, int) else self._index(name_or_idx)
    check(self._lib.mhte_table_step_forward(
        self._h, C.c_int32(i), vp(ids), C.c_int64(ids.numel()), vp(out),
        ws_next._h if ws_next is not None else C.c_void_p(0),  # pylint: disable=protected-access
        vp(next_ids), C.c_int64(next_ids.numel() if next_ids is not None else 0), vp(uids_next),
        vp(n_unique_next),
        ws_cur._h if ws_cur is not None else C.c_void_p(0),  # pylint: disable=protected-access
        _stream()))
    return out

  def table_step_backward(self, name_or_idx, ws, ws_next, uids: torch.Tensor,
                          n_unique_dev: torch.Tensor, grads: torch.Tensor,
                          grad_unique: torch.Tensor, lrs: np.ndarray, update_time: int,
                          global_step: int = 0, exact_order: bool = False, ws_ahead=None,
                          ahead_ids: Optional[torch.Tensor] = None,
                          uids_ahead: Optional[torch.Tensor] = None,
                          n_unique_ahead: Optional[torch.Tensor] = None):
    """mhte_table_step_backward(_ahead): gradient sum + upsert + optimizer of the batch ``ws`` holds
    (+ the numbering and table probe of the batch ``ws_next`` holds, + the run dedup of
    ``ahead_ids`` — the batch two steps on — into ``ws_ahead``)."""
    i = name_or_idx if isinstance(name_or_idx, int) else self._index(name_or_idx)
    lrs = np.ascontiguousarray(lrs, dtype=np.float32)
    n = grads.shape[0]
    check(self._lib.mhte_table_step_backward_ahead(
        self._h, C.c_int32(i), ws._h,  # pylint: disable=protected-access
        ws_next._h if ws_next is not None else C.c_void_p(0),  # pylint: disable=protected-access
        vp(uids), C.c_int64(uids.numel()), vp(n_unique_dev), vp(grads), C.c_int64(n),
        vp(grad_unique), _f32p(lrs), C.c_int64(lrs.size), C.c_int64(int(update_time)),
        C.c_int64(int(global_step)), C.c_int32(_lib.MHTE_EXACT_ORDER if exact_order else 0),
        ws_ahead._h if ws_ahead is not None else C.c_void_p(0),  # pylint: disable=protected-access
        vp(ahead_ids), C.c_int64(ahead_ids.numel() if ahead_ids is not None else 0), vp(uids_ahead),
        vp(n_unique_ahead), _stream()))
    return self
