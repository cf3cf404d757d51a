"""Device-side dedup / packing ops around the table — host-side mirror of
monolith/native_training/distribution_ops.py (reference :80-190) for the ops that sit on the hot
path: ``unique_key_with_value_and_offset``, ``fill_with_offset_map`` and its gradient.

Two API levels:
  * the reference signatures (ragged key, value_offset ragged-of-ragged, value_buffer), for drop-in
    call sites and parity tests against the reference's docstring examples;
  * ``DedupWorkspace`` — the same computation in the form the fused MI355X step uses
    (unique ids, inverse index, CSR occurrence lists, unique count left on the device so that
    dedup -> lookup -> scatter -> segment-sum -> optimize runs without a host round trip).
"""
import ctypes as C
from typing import List, NamedTuple, Optional, Tuple

import numpy as np
import torch

from monolith_amd import _lib
from monolith_amd._lib import check, vp
from monolith_amd.multi_hash_table_ops import Ragged, _stream


class UniqueResult(NamedTuple):
  unique_ids: torch.Tensor    # int64 [n]   (first n_unique entries valid), first-occurrence order
  inverse: torch.Tensor       # int32 [n]   unique index of every position
  seg_off: torch.Tensor       # int32 [n+1] CSR offsets of each unique id's occurrence list
  seg_pos: torch.Tensor       # int32 [n]   positions, grouped by unique id, occurrence order
  n_unique_dev: torch.Tensor  # int32 [1]
  n_unique: Optional[int]     # host copy when requested
  list_end: Optional[torch.Tensor] = None  # int32 [n]: set by unique_unordered, where seg_off holds
                                           # the list STARTS and the lists are not in CSR order


class DedupWorkspace:
  """Scratch + entry points for the device dedup (libmhte.so: mhte_unique & co)."""

  def __init__(self, device: Optional[int] = None):
    if not torch.cuda.is_available():
      raise _lib.MhteError(_lib.MHTE_UNAVAILABLE, "DedupWorkspace needs a HIP device")
    self._lib = _lib.lib()
    self._device = torch.cuda.current_device() if device is None else int(device)
    h = C.c_void_p()
    check(self._lib.mhte_dedup_ws_create(C.c_int32(self._device), C.byref(h)))
    self._h = h

  def close(self):
    if getattr(self, "_h", None):
      torch.cuda.synchronize(self._device)
      self._lib.mhte_dedup_ws_destroy(self._h)
      self._h = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass

  def unique(self, ids: torch.Tensor, want_host_count: bool = True,
             out: Optional[UniqueResult] = None) -> UniqueResult:
    """First-occurrence-order unique with occurrence lists (unique_mapping_ops.cc:82-114)."""
    assert ids.is_cuda and ids.dtype == torch.int64 and ids.is_contiguous()
    n = ids.numel()
    dev = ids.device
    if out is None:
      uids = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
      inverse = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
      seg_off = torch.empty(n + 1, dtype=torch.int32, device=dev)
      seg_pos = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
      nu = torch.zeros(1, dtype=torch.int32, device=dev)
    else:
      uids, inverse, seg_off, seg_pos, nu = out[:5]
    host = C.c_int64(0)
    check(self._lib.mhte_unique(self._h, vp(ids), C.c_int64(n), vp(uids), vp(inverse), vp(seg_off),
                                vp(seg_pos), vp(nu), C.byref(host) if want_host_count else None,
                                _stream()))
    return UniqueResult(uids, inverse, seg_off, seg_pos, nu,
                        int(host.value) if want_host_count else None)

  def step_dedup(self, ids: torch.Tensor, uids: torch.Tensor, n_unique_dev: torch.Tensor):
    """Run dedup of the FIRST batch of a pipelined step (mhte_step_dedup): unique ids in
    unspecified order + count on the device; the occurrence runs stay in this workspace for
    ``MultiHashTable.table_step_backward``."""
    assert ids.is_cuda and ids.dtype == torch.int64 and ids.is_contiguous()
    check(self._lib.mhte_step_dedup(self._h, vp(ids), C.c_int64(ids.numel()), vp(uids),
                                    vp(n_unique_dev), _stream()))

  def unique_unordered(self, ids: torch.Tensor, want_host_count: bool = False,
                       out: Optional[UniqueResult] = None) -> UniqueResult:
    """Same key set and occurrence lists as ``unique`` with an unspecified numbering of the unique
    ids (mhte_unique_unordered, 3 launches instead of 5) — what the unpipelined fused backward needs.  In the
    result ``seg_off[u]`` / ``list_end[u]`` bound the positions of unique id u inside seg_pos."""
    assert ids.is_cuda and ids.dtype == torch.int64 and ids.is_contiguous()
    n = ids.numel()
    dev = ids.device
    if out is None or out.list_end is None:
      uids = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
      inverse = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
      lst_start = torch.empty(n + 1, dtype=torch.int32, device=dev)
      lst_end = torch.empty(n + 1, dtype=torch.int32, device=dev)
      seg_pos = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
      nu = torch.zeros(1, dtype=torch.int32, device=dev)
    else:
      uids, inverse, lst_start, seg_pos, nu = out[:5]
      lst_end = out.list_end
    host = C.c_int64(0)
    check(self._lib.mhte_unique_unordered(self._h, vp(ids), C.c_int64(n), vp(uids), vp(inverse),
                                          vp(lst_start), vp(lst_end), vp(seg_pos), vp(nu),
                                          C.byref(host) if want_host_count else None, _stream()))
    return UniqueResult(uids, inverse, lst_start, seg_pos, nu,
                        int(host.value) if want_host_count else None, lst_end)

  def gather_rows(self, src: torch.Tensor, index: torch.Tensor, n: int, dim: int,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[p] = src[index[p]] — FillWithOffsetMap in gather form (unique_mapping_ops.cc:225-242)."""
    if out is None:
      out = torch.empty((n, dim), dtype=torch.float32, device=src.device)
    check(self._lib.mhte_gather_rows(vp(src), vp(index), C.c_int64(n), C.c_int32(dim), vp(out),
                                     _stream()))
    return out

  def segment_sum(self, grads: torch.Tensor, u: UniqueResult, dim: int,
                  out: Optional[torch.Tensor] = None, exact_order: bool = False) -> torch.Tensor:
    """out[k] = sum of grads over the occurrence list of unique id k
    (FillWithOffsetMapGradient, unique_mapping_ops.cc:307-324).  Rows >= n_unique are untouched."""
    n = grads.numel() // dim
    if out is None:
      out = torch.zeros((max(n, 1), dim), dtype=torch.float32, device=grads.device)
    check(self._lib.mhte_segment_sum(self._h, vp(grads), vp(u.inverse), vp(u.seg_off),
                                     vp(u.seg_pos), vp(u.n_unique_dev), C.c_int64(n),
                                     C.c_int32(dim), vp(out), C.c_int32(1 if exact_order else 0),
                                     _stream()))
    return out


_default_ws = {}


def _ws(device) -> DedupWorkspace:
  d = device.index if isinstance(device, torch.device) else int(device)
  if d not in _default_ws:
    _default_ws[d] = DedupWorkspace(d)
  return _default_ws[d]


class _UniqueKeyWithValueAndOffsetResult(NamedTuple):
  unique_key: Ragged                 # values int64 [U], row_splits host [T+1]
  value_offset: torch.Tensor         # int64 [n]: float offsets into value_buffer
  value_offset_split: torch.Tensor   # int64 [U+1]: list boundaries per unique key
  value_buffer: Optional[torch.Tensor]


def unique_key_with_value_and_offset(key: Ragged, dims: List[int], generate_buffer=True):
  """reference distribution_ops.py:86-118 / ops/unique_mapping_ops.cc:51-155.

  key = [[0, 1, 0], [0]], dims = [2, 3] =>
    unique_key = [[0, 1], [0]], value_offset = [[[0, 4], [2]], [[6]]], buffer length 9.
  The ragged-of-ragged value_offset is returned flat (values + per-unique-key splits); the outer
  split is unique_key.row_splits, exactly the three tensors the reference op emits."""
  T = len(dims)
  if key.row_splits.size != T + 1:
    raise _lib.InvalidArgumentError(
        _lib.MHTE_INVALID_ARGUMENT,
        "RaggedKey should have %d but got %d" % (T, key.row_splits.size - 1))
  dev = key.values.device
  ws = _ws(dev)
  L = _lib.lib()
  n = key.values.numel()
  uniq_parts, splits = [], [0]
  value_offset = torch.empty(n, dtype=torch.int64, device=dev)
  vo_split_parts = []
  value_base = 0
  for t in range(T):
    lo, hi = int(key.row_splits[t]), int(key.row_splits[t + 1])
    ids = key.values[lo:hi].contiguous()
    r = ws.unique(ids, want_host_count=True)
    U = r.n_unique
    uniq_parts.append(r.unique_ids[:U])
    if hi > lo:
      vos = torch.empty(hi - lo + 1, dtype=torch.int64, device=dev)
      check(L.mhte_value_offsets(vp(r.seg_off), vp(r.seg_pos), vp(r.n_unique_dev),
                                 C.c_int64(hi - lo), C.c_int64(value_base), C.c_int64(dims[t]),
                                 C.c_int64(lo), vp(value_offset[lo:hi]), vp(vos), _stream()))
      # the reference emits cumulative list ends after a leading 0
      vo_split_parts.append(vos[1:U + 1])
    splits.append(splits[-1] + U)
    value_base += (hi - lo) * dims[t]
  zero = torch.zeros(1, dtype=torch.int64, device=dev)
  value_offset_split = torch.cat([zero] + vo_split_parts) if vo_split_parts else zero
  unique_key = Ragged(torch.cat(uniq_parts) if uniq_parts else key.values[:0],
                      np.ascontiguousarray(splits, dtype=np.int64))
  buf = torch.zeros(value_base, dtype=torch.float32, device=dev) if generate_buffer else None
  return _UniqueKeyWithValueAndOffsetResult(unique_key, value_offset, value_offset_split, buf)


def _all_vec4(dims):
  return all(d % 4 == 0 for d in dims)


def fill_with_offset_map(pos: Ragged, value: torch.Tensor, value_offset_map: torch.Tensor,
                         value_offset_map_split: torch.Tensor, value_buffer: torch.Tensor,
                         dims: List[int]) -> torch.Tensor:
  """reference distribution_ops.py:121-148 / ops/unique_mapping_ops.cc:204-268."""
  T = len(dims)
  if pos.row_splits.size != T + 1:
    raise _lib.InvalidArgumentError(
        _lib.MHTE_INVALID_ARGUMENT, "Pos's first dim doesn't match dim size. %d v.s. %d" %
        (pos.row_splits.size - 1, T))
  expected = int(sum(int(pos.row_splits[t + 1] - pos.row_splits[t]) * dims[t] for t in range(T)))
  if value.numel() < expected:
    raise _lib.InvalidArgumentError(
        _lib.MHTE_INVALID_ARGUMENT,
        "Value size doesn't match expected size. expected: %d, actual: %d. " %
        (expected, value.numel()))
  L = _lib.lib()
  vec = 1 if _all_vec4(dims) else 0
  voff = 0
  for t in range(T):
    lo, hi = int(pos.row_splits[t]), int(pos.row_splits[t + 1])
    if hi > lo:
      check(L.mhte_fill_with_offset_map(vp(pos.values[lo:hi]), C.c_int64(hi - lo),
                                        vp(value[voff:]), vp(value_offset_map),
                                        vp(value_offset_map_split), C.c_int32(dims[t]),
                                        C.c_int32(vec), vp(value_buffer), _stream()))
    voff += (hi - lo) * dims[t]
  return value_buffer


def fill_with_offset_map_gradient(pos: Ragged, grad: torch.Tensor, grad_offset_map: torch.Tensor,
                                  grad_offset_map_split: torch.Tensor,
                                  dims: List[int]) -> torch.Tensor:
  """ops/unique_mapping_ops.cc:284-329: backprop_grad[i] = sum over offsets of pos[i]."""
  T = len(dims)
  total = int(sum(int(pos.row_splits[t + 1] - pos.row_splits[t]) * dims[t] for t in range(T)))
  out = torch.empty(total, dtype=torch.float32, device=grad.device)
  L = _lib.lib()
  vec = 1 if _all_vec4(dims) else 0
  ooff = 0
  for t in range(T):
    lo, hi = int(pos.row_splits[t]), int(pos.row_splits[t + 1])
    if hi > lo:
      check(L.mhte_fill_with_offset_map_gradient(vp(pos.values[lo:hi]), C.c_int64(hi - lo),
                                                 vp(grad), vp(grad_offset_map),
                                                 vp(grad_offset_map_split), C.c_int32(dims[t]),
                                                 C.c_int32(vec), vp(out[ooff:]), _stream()))
    ooff += (hi - lo) * dims[t]
  return out


# ---------------------------------------------------------------------------------------------
# the step right after the exchange (reference distribution_ops.py:671-760, embedding_combiners.py)
# ---------------------------------------------------------------------------------------------
def _ptr_array(tensors):
  arr = (C.c_void_p * len(tensors))(*[C.c_void_p(t.data_ptr()) for t in tensors])
  return arr


def fused_gather_embeddings_by_input(fused_embeddings: torch.Tensor,
                                     fused_embedding_offsets: List[torch.Tensor],
                                     embedding_dims: List[int]) -> List[torch.Tensor]:
  """distribution_ops.fused_gather_embeddings_by_input (reference :671-675): for every merged slot i,
  rows of ``embedding_dims[i]`` floats gathered from the flat fused buffer at the given offsets."""
  assert fused_embeddings.is_cuda and fused_embeddings.dtype == torch.float32
  offs = [o.to(device=fused_embeddings.device, dtype=torch.int32).contiguous()
          for o in fused_embedding_offsets]
  outs = [torch.empty((o.numel(), d), dtype=torch.float32, device=fused_embeddings.device)
          for o, d in zip(offs, embedding_dims)]
  n = (C.c_int64 * len(offs))(*[o.numel() for o in offs])
  dims = (C.c_int32 * len(offs))(*[int(d) for d in embedding_dims])
  check(_lib.lib().mhte_fused_gather_embeddings_by_input(vp(fused_embeddings.contiguous()),
                                                         C.c_int32(len(offs)), _ptr_array(offs), n,
                                                         dims, _ptr_array(outs), _stream()))
  return outs


def fused_gather_embeddings_by_input_gradient(fused_embeddings_size: int, grads: List[torch.Tensor],
                                              embedding_offsets: List[torch.Tensor],
                                              embedding_dims: List[int], scale: float = 1.0):
  """reference :678-686: the flat gradient of the fused buffer (float atomics, like the reference)."""
  dev = grads[0].device
  offs = [o.to(device=dev, dtype=torch.int32).contiguous() for o in embedding_offsets]
  gs = [g.to(device=dev, dtype=torch.float32).contiguous() for g in grads]
  out = torch.empty(int(fused_embeddings_size), dtype=torch.float32, device=dev)
  n = (C.c_int64 * len(offs))(*[o.numel() for o in offs])
  dims = (C.c_int32 * len(offs))(*[int(d) for d in embedding_dims])
  check(_lib.lib().mhte_fused_gather_embeddings_by_input_gradient(
      vp(out), C.c_int64(out.numel()), C.c_int32(len(offs)), _ptr_array(gs), _ptr_array(offs), n,
      dims, C.c_float(float(scale)), _stream()))
  return out


def _reduce(id_indices, id_values, id_length, mode, indices_sorted):
  idx = id_indices.reshape(-1).to(dtype=torch.int64).contiguous()
  vals = id_values.to(dtype=torch.float32).contiguous()
  assert vals.is_cuda and idx.is_cuda and vals.dim() == 2 and idx.numel() == vals.shape[0]
  batch = int(id_length[0]) if not isinstance(id_length, int) else id_length
  out = torch.empty((batch, vals.shape[1]), dtype=torch.float32, device=vals.device)
  check(_lib.lib().mhte_reduce_rows(vp(idx), vp(vals), C.c_int64(idx.numel()),
                                    C.c_int32(vals.shape[1]), C.c_int64(batch), C.c_int32(mode),
                                    C.c_int32(1 if indices_sorted else 0), vp(out), _stream()))
  return out


def reduce_sum(id_indices, id_values, id_length, indices_sorted: bool = True):
  """distribution_ops.reduce_sum (reduce_op.cc:29-51).  ``indices_sorted``: row indices ascend, as
  a sparse / ragged input's do — sequential, bit-identical accumulation."""
  return _reduce(id_indices, id_values, id_length, 0, indices_sorted)


def reduce_mean(id_indices, id_values, id_length, indices_sorted: bool = True):
  """distribution_ops.reduce_mean (reduce_op.cc:55-87)."""
  return _reduce(id_indices, id_values, id_length, 1, indices_sorted)


def reduce_sqrtn(id_indices, id_values, id_length, indices_sorted: bool = True):
  """distribution_ops.reduce_sqrtn (ReduceSquareNorm, reduce_op.cc:91-125): sqrt of the sum of
  squares."""
  return _reduce(id_indices, id_values, id_length, 2, indices_sorted)


# ---------------------------------------------------------------------------------------------
# fused_embedding_to_layout (reference distribution_ops.fused_embedding_to_layout and its gradient;
# configuration messages of idl/matrix/proto/example.proto:176-221 as plain Python objects)
# ---------------------------------------------------------------------------------------------
class PoolingType:
  SUM, MEAN, FIRSTN = 0, 1, 3


class OutType:
  CONCAT, STACK, ADDN, NONE = 0, 1, 2, 3


class SliceConfig:
  def __init__(self, feature_name: str, start: int, end: int):
    self.feature_name, self.start, self.end = feature_name, int(start), int(end)


class OutConfig:
  """slice_configs in order; shape: list of per-tensor dims (first dim -1 = batch): one tensor for
  CONCAT / STACK / ADDN, one per slice for NONE."""

  def __init__(self, slice_configs: List[SliceConfig], out_type: int, shape: List[List[int]]):
    self.slice_configs, self.out_type, self.shape = list(slice_configs), out_type, [list(s) for s in shape]


class FeatureConfig:
  def __init__(self, table: str, pooling_type: int = PoolingType.SUM, slice_dims=(),
               max_sequence_length: int = 0):
    self.table, self.pooling_type = table, pooling_type
    self.slice_dims, self.max_sequence_length = list(slice_dims), int(max_sequence_length)


class FeatureConfigs:
  def __init__(self, feature_configs, out_configs):
    self.feature_configs, self.out_configs = dict(feature_configs), dict(out_configs)


def _layout_plan(cfgs: FeatureConfigs, batch_size: int):
  """-> (slices ctypes array, output shapes in op order).  Mirrors the op's constructor
  (runtime/ops/fused_embedding_to_layout.cc:240-284): features are indexed by sorted name, layouts
  are emitted in sorted-name order, a slice takes its feature's pooling type."""
  feature_names = sorted(cfgs.feature_configs)
  slices, shapes = [], []
  for ln in sorted(cfgs.out_configs):
    oc = cfgs.out_configs[ln]
    base = len(shapes)
    for sh in oc.shape:
      shapes.append([batch_size if (i == 0 and d == -1) else d for i, d in enumerate(sh)])
    offset = 0
    for i, sc in enumerate(oc.slice_configs):
      fc = cfgs.feature_configs[sc.feature_name]
      dim = sc.end - sc.start
      s = _lib.LayoutSlice()
      s.feature_idx = feature_names.index(sc.feature_name)
      s.start, s.dim = sc.start, dim
      s.pooling, s.max_sequence_length = fc.pooling_type, fc.max_sequence_length
      s.out_type = oc.out_type
      if len(oc.shape) == 1:
        s.out_index = base
        row = int(np.prod(shapes[base][1:]))
        if oc.out_type == OutType.CONCAT:
          s.out_offset = offset
          offset += dim
        elif oc.out_type == OutType.STACK:
          s.out_offset = i * dim
        else:
          s.out_offset = 0
        s.out_row_floats = row
      else:                          # NONE: one tensor per slice
        s.out_index = base + i
        s.out_offset = 0
        s.out_row_floats = int(np.prod(shapes[base + i][1:]))
      slices.append(s)
  return (_lib.LayoutSlice * len(slices))(*slices), shapes


def _layout_common(embeddings_list, fid_offset, feature_offset, nfl_offset):
  dev = embeddings_list[0].device
  embs = [e.to(device=dev, dtype=torch.float32).contiguous() for e in embeddings_list]
  stride = (C.c_int32 * len(embs))(*[int(e.shape[1]) if e.dim() == 2 else 1 for e in embs])
  count = (C.c_int64 * len(embs))(*[e.numel() for e in embs])
  fo = fid_offset.to(device=dev).contiguous()
  assert fo.dtype in (torch.int64, torch.uint64)
  fe = feature_offset.to(device=dev, dtype=torch.int32).contiguous()
  nf = nfl_offset.to(device=dev).contiguous()
  assert nf.dtype in (torch.int32, torch.uint32)
  return dev, embs, stride, count, fo, fe, nf


def fused_embedding_to_layout(embeddings_list: List[torch.Tensor], fid_offset: torch.Tensor,
                              feature_offset: torch.Tensor, nfl_offset: torch.Tensor, batch_size: int,
                              feature_cfgs: FeatureConfigs,
                              one_fid_unique_rows: bool = False) -> List[torch.Tensor]:
  """MonolithEmbeddingToLayout: the layouts' tensors, layouts in sorted-name order.
  ``one_fid_unique_rows``: MHTE_LAYOUT_ONE_FID_UNIQUE_ROWS (float4 copies)."""
  dev, embs, stride, count, fo, fe, nf = _layout_common(embeddings_list, fid_offset, feature_offset,
                                                        nfl_offset)
  slices, shapes = _layout_plan(feature_cfgs, int(batch_size))
  outs = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in shapes]
  lens = (C.c_int64 * len(outs))(*[o.numel() for o in outs])
  check(_lib.lib().mhte_embedding_to_layout(
      _ptr_array(embs), stride, count, C.c_int32(len(embs)), vp(fo), C.c_int64(fo.numel()), vp(fe),
      C.c_int64(fe.numel()), vp(nf), C.c_int32(nf.numel()), C.c_int32(int(batch_size)), slices,
      C.c_int32(len(slices)), _ptr_array(outs), lens, C.c_int32(len(outs)),
      C.c_int32(_lib.MHTE_LAYOUT_ONE_FID_UNIQUE_ROWS if one_fid_unique_rows else 0), _stream()))
  return outs


def fused_embedding_to_layout_grad(embeddings_list: List[torch.Tensor], fid_offset: torch.Tensor,
                                   feature_offset: torch.Tensor, nfl_offset: torch.Tensor,
                                   batch_size: int, tensors_grad: List[torch.Tensor],
                                   feature_cfgs: FeatureConfigs, one_fid_unique_rows: bool = False,
                                   out: List[torch.Tensor] = None) -> List[torch.Tensor]:
  """MonolithEmbeddingToLayoutGrad: gradients of ``embeddings_list`` (same shapes; ``out``: buffers
  to write them into, e.g. views of one flat gradient)."""
  dev, embs, stride, count, fo, fe, nf = _layout_common(embeddings_list, fid_offset, feature_offset,
                                                        nfl_offset)
  slices, shapes = _layout_plan(feature_cfgs, int(batch_size))
  assert len(tensors_grad) == len(shapes)
  tg = [g.to(device=dev, dtype=torch.float32).contiguous() for g in tensors_grad]
  grads = out if out is not None else [torch.empty_like(e) for e in embs]
  lens = (C.c_int64 * len(tg))(*[g.numel() for g in tg])
  check(_lib.lib().mhte_embedding_to_layout_grad(
      _ptr_array(grads), stride, count, C.c_int32(len(embs)), vp(fo), C.c_int64(fo.numel()), vp(fe),
      C.c_int64(fe.numel()), vp(nf), C.c_int32(nf.numel()), C.c_int32(int(batch_size)), slices,
      C.c_int32(len(slices)), _ptr_array(tg), lens, C.c_int32(len(tg)),
      C.c_int32(_lib.MHTE_LAYOUT_ONE_FID_UNIQUE_ROWS if one_fid_unique_rows else 0), _stream()))
  return grads


def lookup_gradient(id_indices: torch.Tensor, id_values: torch.Tensor,
                    input_grads: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
  """MonolithHashTableLookupGradient (runtime/ops/hash_table_lookup_op.cc:110-147): the gradient of a
  lookup gathered back per (batch row, id) pair of a sparse id tensor — ``ids[i] = id_values[i]``,
  ``output_grads[i] = input_grads[id_indices[i, 0]]``.  ``id_indices`` [n, k] int64 (column 0 = the
  batch row), ``id_values`` [n] int64, ``input_grads`` [rows, dim] float32."""
  assert id_indices.dim() == 2 and id_values.dim() == 1 and input_grads.dim() == 2
  if id_indices.shape[0] != id_values.shape[0]:
    raise _lib.InvalidArgumentError(
        _lib.MHTE_INVALID_ARGUMENT, "id_indices's first dim and id_values dim should be same. Got %dv.s. %d" %
        (id_indices.shape[0], id_values.shape[0]))
  dev = input_grads.device
  idx = id_indices.to(dev, torch.int64).contiguous()
  val = id_values.to(dev, torch.int64).contiguous()
  g = input_grads.to(torch.float32).contiguous()
  n, dim = int(val.numel()), int(g.shape[1])
  out_ids = torch.empty(n, dtype=torch.int64, device=dev)
  out = torch.empty((n, dim), dtype=torch.float32, device=dev)
  _lib.check(_lib.lib().mhte_lookup_gradient(
      _lib.vp(idx), _lib.C.c_int64(n), _lib.C.c_int64(int(idx.shape[1])), _lib.vp(val), _lib.vp(g),
      _lib.C.c_int64(int(g.shape[0])), _lib.C.c_int32(dim), _lib.vp(out_ids), _lib.vp(out),
      _lib.C.c_void_p(torch.cuda.current_stream().cuda_stream)))
  return out_ids, out
