"""TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.

ctypes bindings for
  * ``liboracle.so``               — the plain-C restatement of the reference hot path
                                     (oracle/mhte_oracle.c), and
  * ``_ref/libmonolith_ref*.so``   — the reference's own cuckoohash_map.hpp / avx_utils.h, and
                                     (``_filter``) its hash_filter sources, compiled in place from
                                     /root/reference by oracle/Makefile.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; ``monolith_amd`` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))

OPT_SGD, OPT_ADAGRAD, OPT_FTRL = 0, 1, 2
OPT_MOMENTUM, OPT_ADADELTA, OPT_RMSPROP, OPT_RMSPROPV2, OPT_ADAM, OPT_AMSGRAD = 3, 4, 5, 6, 7, 8
OPT_MOVING_AVERAGE, OPT_BATCH_SOFTMAX, OPT_GROUP_ADAGRAD = 9, 10, 11
INIT_ZEROS, INIT_ONES, INIT_CONSTANT = 0, 1, 2


def build(force=False):
  """Builds liboracle.so (always possible) and _ref (only where /root/reference exists)."""
  # (make compares time stamps: an edited restatement is rebuilt, an up-to-date one is left alone)
  subprocess.check_call(["make", "-s", "-C", _DIR] + (["-B"] if force else []) + ["liboracle.so"])
  if os.path.isdir("/root/reference") and (
      force or not all(os.path.exists(os.path.join(_DIR, "_ref", n)) for n in
                       ("libmonolith_ref.so", "libmonolith_ref_avx.so", "libmonolith_ref_filter.so",
                        "libmonolith_ref_opt.so", "libmonolith_ref_opt_avx.so"))):
    subprocess.check_call(["make", "-s", "-C", _DIR, "ref"])


class Segment(C.Structure):
  _fields_ = [("dim", C.c_int32), ("opt", C.c_int32), ("p", C.c_float * 8),
              ("init", C.c_int32), ("init_value", C.c_float)]


def segment(dim, opt=OPT_SGD, p=(), init=INIT_ZEROS, init_value=0.0):
  s = Segment()
  s.dim, s.opt, s.init, s.init_value = dim, opt, init, init_value
  for i, v in enumerate(p):
    s.p[i] = v
  return s


def _p(a, t):
  return a.ctypes.data_as(C.POINTER(t))


def _i64(a):
  return np.ascontiguousarray(a, dtype=np.int64)


def _f32(a):
  return np.ascontiguousarray(a, dtype=np.float32)


_lib = None


def lib():
  global _lib
  if _lib is None:
    build()
    L = C.CDLL(os.path.join(_DIR, "liboracle.so"))
    L.mo_table_new.restype = C.c_void_p
    L.mo_table_new.argtypes = [C.c_int32, C.POINTER(Segment), C.c_uint64]
    L.mo_hash.restype = C.c_uint64
    L.mo_hash.argtypes = [C.c_int64]
    L.mo_partial.restype = C.c_uint8
    L.mo_partial.argtypes = [C.c_uint64]
    L.mo_alt_index.restype = C.c_uint64
    L.mo_alt_index.argtypes = [C.c_int, C.c_uint8, C.c_uint64]
    for name in ("mo_size", "mo_lookup", "mo_dump", "mo_locate",
                 "mo_unique_key_with_value_and_offset", "mo_fused_reorder_by_indices"):
      getattr(L, name).restype = C.c_int64
    _lib = L
  return _lib


class Table:
  """Restated single EmbeddingHashTable (cuckoo_embedding_hash_table.cc:117-362)."""

  def __init__(self, segments, initial_capacity=1):
    if isinstance(segments, Segment):
      segments = [segments]
    self.L = lib()
    arr = (Segment * len(segments))(*segments)
    self.h = C.c_void_p(self.L.mo_table_new(len(segments), arr, C.c_uint64(initial_capacity)))
    self.dim = self.L.mo_dim(self.h)
    self.row_floats = self.L.mo_row_floats(self.h)
    self.nseg = len(segments)

  def __del__(self):
    try:
      self.L.mo_table_free(self.h)
    except Exception:  # pylint: disable=broad-except
      pass

  def size(self):
    return int(self.L.mo_size(self.h))

  def hashpower(self):
    return int(self.L.mo_hashpower(self.h))

  def contains(self, i):
    return bool(self.L.mo_contains(self.h, C.c_int64(int(i))))

  def locate(self, i):
    return int(self.L.mo_locate(self.h, C.c_int64(int(i))))

  def lookup(self, ids):
    ids = _i64(ids)
    out = np.empty((ids.size, self.dim), np.float32)
    hits = self.L.mo_lookup(self.h, _p(ids, C.c_int64), C.c_int64(ids.size), _p(out, C.c_float))
    return out, int(hits)

  def assign(self, ids, values, update_time=0):
    ids, values = _i64(ids), _f32(values)
    self.L.mo_assign(self.h, _p(ids, C.c_int64), C.c_int64(ids.size), _p(values, C.c_float),
                     C.c_int64(update_time))

  def assign_add(self, ids, values, update_time=0):
    ids, values = _i64(ids), _f32(values)
    self.L.mo_assign_add(self.h, _p(ids, C.c_int64), C.c_int64(ids.size), _p(values, C.c_float),
                         C.c_int64(update_time))

  def reinitialize(self, ids, now=0):
    ids = _i64(ids)
    st = np.empty(ids.size, np.int32)
    self.L.mo_reinitialize(self.h, _p(ids, C.c_int64), C.c_int64(ids.size), _p(st, C.c_int32),
                           C.c_int64(now))
    return st

  def optimize(self, ids, grads, lrs, update_time=0, global_step=0):
    ids, grads = _i64(ids), _f32(grads)
    lrs = _f32(np.atleast_1d(lrs))
    assert lrs.size == self.nseg
    self.L.mo_optimize(self.h, _p(ids, C.c_int64), C.c_int64(ids.size), _p(grads, C.c_float),
                       _p(lrs, C.c_float), C.c_int64(update_time), C.c_int64(global_step))

  def set_ttl(self, default_days, slot_to_days=None):
    slot_to_days = slot_to_days or {}
    s = _i64(list(slot_to_days.keys()))
    d = np.ascontiguousarray(list(slot_to_days.values()), dtype=np.int32)
    self.L.mo_set_ttl(self.h, C.c_int64(default_days), C.c_int32(s.size), _p(s, C.c_int64),
                      _p(d, C.c_int32))

  def evict(self, max_update_time):
    self.L.mo_evict(self.h, C.c_int64(max_update_time))

  def dump(self, with_rows=True):
    n = self.size()
    ids = np.empty(n, np.int64)
    pos = np.empty(n, np.int64)
    ts = np.empty(n, np.uint32)
    rows = np.empty((n, self.row_floats), np.float32) if with_rows else None
    m = self.L.mo_dump(self.h, C.c_int64(n), _p(ids, C.c_int64), _p(pos, C.c_int64),
                       _p(ts, C.c_uint32), _p(rows, C.c_float) if with_rows else None)
    assert m == n
    return ids, pos, ts, rows


def sgd(num, grad, lr):
  num, grad = _f32(num).copy(), _f32(grad)
  lib().mo_sgd(_p(num, C.c_float), _p(grad, C.c_float), C.c_int64(num.size), C.c_float(lr))
  return num


def adagrad(num, norm, grad, lr, wd):
  num, norm, grad = _f32(num).copy(), _f32(norm).copy(), _f32(grad)
  lib().mo_adagrad(_p(num, C.c_float), _p(norm, C.c_float), _p(grad, C.c_float),
                   C.c_int64(num.size), C.c_float(lr), C.c_float(wd))
  return num, norm


def adagrad_avx(num, norm, grad, lr, wd):
  """avx_utils.h:96-119 restated (blocks of 8 with fused multiply-adds, raw gradient in the weight
  step; the tail as the baseline)."""
  num, norm, grad = _f32(num).copy(), _f32(norm).copy(), _f32(grad)
  lib().mo_adagrad_avx(_p(num, C.c_float), _p(norm, C.c_float), _p(grad, C.c_float),
                       C.c_int64(num.size), C.c_float(lr), C.c_float(wd))
  return num, norm


def unique_key_with_value_and_offset(key, key_split, dims):
  key, key_split = _i64(key), _i64(key_split)
  dims = np.ascontiguousarray(dims, dtype=np.int32)
  T, n = dims.size, key.size
  uk = np.empty(n, np.int64)
  uks = np.empty(T + 1, np.int64)
  vo = np.empty(n, np.int64)
  vos = np.empty(n + 1, np.int64)
  blen = C.c_int64(0)
  nu = lib().mo_unique_key_with_value_and_offset(
      _p(key, C.c_int64), _p(key_split, C.c_int64), C.c_int32(T), _p(dims, C.c_int32),
      _p(uk, C.c_int64), _p(uks, C.c_int64), _p(vo, C.c_int64), _p(vos, C.c_int64),
      C.byref(blen))
  return uk[:nu].copy(), uks, vo, vos[:nu + 1].copy(), int(blen.value)


def fill_with_offset_map(pos, pos_split, value, value_offset_map, value_offset_map_split, dims,
                         buffer_len):
  pos, pos_split = _i64(pos), _i64(pos_split)
  value = _f32(value)
  vom, voms = _i64(value_offset_map), _i64(value_offset_map_split)
  dims = np.ascontiguousarray(dims, dtype=np.int32)
  buf = np.zeros(buffer_len, np.float32)
  rc = lib().mo_fill_with_offset_map(_p(pos, C.c_int64), _p(pos_split, C.c_int64),
                                     C.c_int32(dims.size), _p(dims, C.c_int32),
                                     _p(value, C.c_float), C.c_int64(value.size),
                                     _p(vom, C.c_int64), C.c_int64(vom.size),
                                     _p(voms, C.c_int64), _p(buf, C.c_float))
  if rc:
    raise ValueError("InvalidArgument(%d)" % rc)
  return buf


def fill_with_offset_map_gradient(pos, pos_split, grad, grad_offset_map, grad_offset_map_split,
                                  dims):
  pos, pos_split = _i64(pos), _i64(pos_split)
  grad = _f32(grad)
  gom, goms = _i64(grad_offset_map), _i64(grad_offset_map_split)
  dims = np.ascontiguousarray(dims, dtype=np.int32)
  bsize = int(sum(int(dims[j]) * int(pos_split[j + 1] - pos_split[j]) for j in range(dims.size)))
  out = np.empty(bsize, np.float32)
  rc = lib().mo_fill_with_offset_map_gradient(_p(pos, C.c_int64), _p(pos_split, C.c_int64),
                                              C.c_int32(dims.size), _p(dims, C.c_int32),
                                              _p(grad, C.c_float), _p(gom, C.c_int64),
                                              C.c_int64(gom.size), _p(goms, C.c_int64),
                                              _p(out, C.c_float))
  if rc:
    raise ValueError("InvalidArgument(%d)" % rc)
  return out


def compute_fused_offsets(slot_size_vec, table_dims, num_tables, num_shards):
  ss = np.ascontiguousarray(slot_size_vec, dtype=np.int32)
  td = np.ascontiguousarray(table_dims, dtype=np.int32)
  ko = np.empty(num_tables * num_shards + 1, np.int32)
  eo = np.empty(num_tables * num_shards + 1, np.int32)
  kpt = np.empty(num_tables, np.int32)
  es = np.empty(num_shards, np.int32)
  tk, te = C.c_int32(0), C.c_int32(0)
  lib().mo_compute_fused_offsets(_p(ss, C.c_int32), _p(td, C.c_int32), C.c_int32(num_tables),
                                 C.c_int32(num_shards), _p(ko, C.c_int32), _p(eo, C.c_int32),
                                 _p(kpt, C.c_int32), _p(es, C.c_int32), C.byref(tk), C.byref(te))
  return ko, eo, kpt, es, tk.value, te.value


def fused_reorder_by_indices(inputs, num_shards, dims, rank0_empty=False):
  """inputs: list of M int64 id vectors."""
  M = len(inputs)
  split = np.zeros(M + 1, np.int64)
  for i, a in enumerate(inputs):
    split[i + 1] = split[i] + len(a)
  flat = _i64(np.concatenate([_i64(a) for a in inputs]) if M else np.zeros(0, np.int64))
  dims = np.ascontiguousarray(dims, dtype=np.int32)
  total = int(split[M])
  out = np.empty(total, np.int64)
  shard_sizes = np.empty(num_shards, np.int32)
  sss = np.empty(num_shards * M, np.int32)
  eos = np.empty(M, np.int32)
  feo = np.empty(total, np.int32)
  nu = lib().mo_fused_reorder_by_indices(_p(flat, C.c_int64), _p(split, C.c_int64), C.c_int32(M),
                                         C.c_int32(num_shards), _p(dims, C.c_int32),
                                         C.c_int32(int(rank0_empty)), _p(out, C.c_int64),
                                         _p(shard_sizes, C.c_int32), _p(sss, C.c_int32),
                                         _p(eos, C.c_int32), _p(feo, C.c_int32))
  return out[:nu].copy(), shard_sizes, sss, eos, feo


# ---------------------------------------------------------------- reference-built library
_ref = {}


def ref_available(avx=False):
  return os.path.exists(
      os.path.join(_DIR, "_ref", "libmonolith_ref_avx.so" if avx else "libmonolith_ref.so"))


def ref_lib(avx=False):
  if avx not in _ref:
    build()
    L = C.CDLL(os.path.join(_DIR, "_ref",
                            "libmonolith_ref_avx.so" if avx else "libmonolith_ref.so"))
    L.ref_table_new.restype = C.c_void_p
    L.ref_table_new.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_uint64]
    L.ref_ps_new.restype = C.c_void_p
    L.ref_ps_new_shared.restype = C.c_void_p
    L.ref_ps_new_shared.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                    C.c_uint64]
    L.ref_ps_breakdown.restype = None
    L.ref_ps_breakdown.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.ref_ps_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                             C.c_uint64]
    for name in ("ref_size", "ref_hashpower", "ref_lookup", "ref_dump", "ref_ps_size",
                 "ref_ps_step", "ref_ps_lookup"):
      getattr(L, name).restype = C.c_int64
    L.ref_adagrad_optimize.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float),
                                       C.POINTER(C.c_float), C.c_int64, C.c_float, C.c_float]
    _ref[avx] = L
  return _ref[avx]


class RefTable:
  """The reference's cuckoohash_map (compiled from /root/reference) + restated accessor."""

  def __init__(self, dim, opt=OPT_SGD, init_acc=0.1, wd=0.0, init_value=0.0, initial_capacity=1,
               avx=False):
    self.L = ref_lib(avx)
    self.dim = dim
    self.row_floats = dim + (dim if opt == OPT_ADAGRAD else 0)
    self.h = C.c_void_p(self.L.ref_table_new(dim, opt, init_acc, wd, init_value,
                                             initial_capacity))

  def __del__(self):
    try:
      self.L.ref_table_free(self.h)
    except Exception:  # pylint: disable=broad-except
      pass

  def size(self):
    return int(self.L.ref_size(self.h))

  def hashpower(self):
    return int(self.L.ref_hashpower(self.h))

  def contains(self, i):
    return bool(self.L.ref_contains(self.h, C.c_int64(int(i))))

  def lookup(self, ids):
    ids = _i64(ids)
    out = np.empty((ids.size, self.dim), np.float32)
    hits = self.L.ref_lookup(self.h, _p(ids, C.c_int64), C.c_int64(ids.size), _p(out, C.c_float))
    return out, int(hits)

  def assign(self, ids, values, update_time=0):
    ids, values = _i64(ids), _f32(values)
    self.L.ref_assign(self.h, _p(ids, C.c_int64), C.c_int64(ids.size), _p(values, C.c_float),
                      C.c_int64(update_time))

  def assign_add(self, ids, values, update_time=0):
    ids, values = _i64(ids), _f32(values)
    self.L.ref_assign_add(self.h, _p(ids, C.c_int64), C.c_int64(ids.size), _p(values, C.c_float),
                          C.c_int64(update_time))

  def reinitialize(self, ids, now=0):
    ids = _i64(ids)
    st = np.empty(ids.size, np.int32)
    self.L.ref_reinitialize(self.h, _p(ids, C.c_int64), C.c_int64(ids.size), _p(st, C.c_int32),
                            C.c_int64(now))
    return st

  def optimize(self, ids, grads, lr, update_time=0):
    ids, grads = _i64(ids), _f32(grads)
    self.L.ref_optimize(self.h, _p(ids, C.c_int64), C.c_int64(ids.size), _p(grads, C.c_float),
                        C.c_float(lr), C.c_int64(update_time))

  def set_ttl(self, default_days, slot_to_days=None):
    slot_to_days = slot_to_days or {}
    s = _i64(list(slot_to_days.keys()))
    d = np.ascontiguousarray(list(slot_to_days.values()), dtype=np.int32)
    self.L.ref_set_ttl(self.h, C.c_int64(default_days), C.c_int(s.size), _p(s, C.c_int64),
                       _p(d, C.c_int32))

  def evict(self, max_update_time):
    self.L.ref_evict(self.h, C.c_int64(max_update_time))

  def dump(self, with_rows=True):
    n = self.size()
    ids = np.empty(n, np.int64)
    pos = np.empty(n, np.int64)
    ts = np.empty(n, np.uint32)
    rows = np.empty((n, self.row_floats), np.float32) if with_rows else None
    m = self.L.ref_dump(self.h, C.c_int64(n), _p(ids, C.c_int64), _p(pos, C.c_int64),
                        _p(ts, C.c_uint32), _p(rows, C.c_float) if with_rows else None)
    assert m == n, (m, n)
    return ids, pos, ts, rows


def ref_adagrad(num, norm, grad, lr, wd, avx=False):
  num, norm, grad = _f32(num).copy(), _f32(norm).copy(), _f32(grad)
  ref_lib(avx).ref_adagrad_optimize(_p(num, C.c_float), _p(norm, C.c_float), _p(grad, C.c_float),
                                    num.size, lr, wd)
  return num, norm


class RefPs:
  """CPU baseline: P single-threaded reference-map shards + worker-side dedup."""

  PHASES = ("dedup", "partition", "lookup", "scatter", "grad_sum", "optimize")

  def __init__(self, P, dim, opt, init_acc=0.1, wd=0.0, init_value=0.0, initial_capacity=1,
               avx=True, shared=False):
    """shared=False: variant (i), P single-threaded shards (fid mod P); shared=True: variant (ii),
    one table, P threads over contiguous chunks of the distinct ids."""
    self.L = ref_lib(avx)
    self.P, self.dim = P, dim
    new = self.L.ref_ps_new_shared if shared else self.L.ref_ps_new
    self.h = C.c_void_p(new(P, dim, opt, init_acc, wd, init_value, initial_capacity))

  def breakdown(self):
    """Seconds the last step spent in each phase (PHASES)."""
    out = (C.c_double * 6)()
    self.L.ref_ps_breakdown(self.h, out)
    return dict(zip(self.PHASES, [float(x) for x in out]))

  def __del__(self):
    try:
      self.L.ref_ps_free(self.h)
    except Exception:  # pylint: disable=broad-except
      pass

  def size(self):
    return int(self.L.ref_ps_size(self.h))

  def hashpower(self, shard=0):
    self.L.ref_ps_hashpower.restype = C.c_int64
    return int(self.L.ref_ps_hashpower(self.h, C.c_int(shard)))

  def step(self, ids, grads, lr, update_time, want_emb=True):
    ids, grads = _i64(ids), _f32(grads)
    emb = np.empty((ids.size, self.dim), np.float32) if want_emb else None
    u = self.L.ref_ps_step(self.h, _p(ids, C.c_int64), C.c_int64(ids.size), _p(grads, C.c_float),
                           C.c_float(lr), C.c_int64(update_time),
                           _p(emb, C.c_float) if want_emb else None)
    return emb, int(u)

  def lookup(self, ids):
    ids = _i64(ids)
    out = np.empty((ids.size, self.dim), np.float32)
    hits = self.L.ref_ps_lookup(self.h, _p(ids, C.c_int64), C.c_int64(ids.size),
                                _p(out, C.c_float))
    return out, int(hits)


# ---------------------------------------------------------------- occurrence filter
class SlidingFilter:
  """The restatement of SlidingHashFilter / HashFilter<uint16_t> (oracle/mhte_filter_oracle.c): the
  checker of the device filter.  ``defer_advance``: the window moves only in ``advance_if_full()``
  (the device checks between launches)."""

  def __init__(self, capacity, split_num, defer_advance=False):
    L = lib()
    L.mo_filter_new.restype = C.c_void_p
    L.mo_filter_new.argtypes = [C.c_uint64, C.c_int]
    L.mo_filter_estimated_total_element.restype = C.c_uint64
    L.mo_filter_split_words.restype = C.c_uint64
    self._L = L
    self._h = C.c_void_p(L.mo_filter_new(int(capacity), int(split_num)))
    if defer_advance:
      L.mo_filter_set_defer_advance(self._h, 1)

  def __del__(self):
    try:
      if self._h:
        self._L.mo_filter_free(self._h)
        self._h = None
    except Exception:  # pylint: disable=broad-except
      pass

  def add(self, fid, count=1):
    return int(self._L.mo_filter_add(self._h, C.c_uint64(int(fid) & 0xFFFFFFFFFFFFFFFF), C.c_uint32(int(count))))

  def get(self, fid):
    return int(self._L.mo_filter_get(self._h, C.c_uint64(int(fid) & 0xFFFFFFFFFFFFFFFF)))

  def should_be_filtered(self, fid, count, threshold):
    return bool(self._L.mo_filter_should_be_filtered(self._h, C.c_int64(int(fid)), C.c_int64(int(count)),
                                                     C.c_int64(int(threshold))))

  def advance_if_full(self):
    self._L.mo_filter_advance_if_full(self._h)

  def estimated_total_element(self):
    return int(self._L.mo_filter_estimated_total_element(self._h))

  def state(self):
    """-> dict(head, head_increment, failure_count, num_elements[nsplit])"""
    out = (C.c_uint64 * (4 + 64))()
    self._L.mo_filter_state(self._h, out)
    n = int(out[3])
    return {"head": int(out[0]), "head_increment": int(out[1]), "failure_count": int(out[2]),
            "num_elements": [int(out[4 + i]) for i in range(n)]}

  def split_words(self, split):
    n = int(self._L.mo_filter_split_words(self._h, C.c_int(split), None, C.c_uint64(0)))
    out = np.zeros(n, dtype=np.uint32)
    self._L.mo_filter_split_words(self._h, C.c_int(split), out.ctypes.data_as(C.POINTER(C.c_uint32)),
                                  C.c_uint64(n))
    return out

_ref_opt = {}


def ref_opt_available(avx=False):
  return os.path.exists(os.path.join(_DIR, "_ref", "libmonolith_ref_opt%s.so" % ("_avx" if avx else "")))


def ref_opt_lib(avx=False):
  if avx not in _ref_opt:
    build()
    L = C.CDLL(os.path.join(_DIR, "_ref", "libmonolith_ref_opt%s.so" % ("_avx" if avx else "")))
    L.ref_opt_new.restype = C.c_void_p
    L.ref_opt_new.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.ref_opt_free.argtypes = [C.c_void_p]
    L.ref_opt_ctx_bytes.restype = C.c_int64
    L.ref_opt_ctx_bytes.argtypes = [C.c_void_p]
    L.ref_opt_slice_size.argtypes = [C.c_void_p]
    L.ref_opt_init.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.ref_opt_optimize.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                   C.POINTER(C.c_float), C.c_float, C.c_int64]
    L.ref_opt_save_restore.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ref_dc_sgd.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int,
                             C.c_float, C.c_float]
    _ref_opt[avx] = L
  return _ref_opt[avx]


class RefOptimizer:
  """ONE optimizer object of the reference, made by its own factory (New*Optimizer, the
  optimizer/*_optimizer.cc files compiled in place: oracle/ref_opt_driver.cc) — Init() and Optimize()
  on a weight vector and an optimizer context held here.  `p` is the oracle's parameter layout for
  the optimizer (mhte_oracle.h, mo_segment.p)."""

  def __init__(self, opt, dim, p=(), avx=False):
    self.L = ref_opt_lib(avx)
    pp = (C.c_float * 8)(*([float(x) for x in p] + [0.0] * (8 - len(p))))
    self.h = C.c_void_p(self.L.ref_opt_new(int(opt), int(dim), pp))
    assert self.h, "reference factory refused the optimizer"
    self.dim = dim
    nb = int(self.L.ref_opt_ctx_bytes(self.h))
    self.ctx = np.zeros(max(1, (nb + 3) // 4), np.float32)
    self.ctx_floats = nb // 4
    self.num = np.zeros(dim, np.float32)
    self.L.ref_opt_init(self.h, _p(self.ctx, C.c_float))

  def __del__(self):
    try:
      if self.h:
        self.L.ref_opt_free(self.h)
        self.h = None
    except Exception:  # pylint: disable=broad-except
      pass

  def optimize(self, grad, lr, global_step=0):
    g = _f32(grad)
    assert g.size == self.dim
    rc = self.L.ref_opt_optimize(self.h, _p(self.ctx, C.c_float), _p(self.num, C.c_float), _p(g, C.c_float),
                                 C.c_float(lr), C.c_int64(global_step))
    assert rc == 0
    return self.num.copy(), self.ctx[:self.ctx_floats].copy()

  def save_restore(self):
    out = np.zeros_like(self.ctx)
    assert self.L.ref_opt_save_restore(self.h, _p(self.ctx, C.c_float), _p(out, C.c_float)) == 0
    return out[:self.ctx_floats].copy()


def ref_dc_sgd(num, grad, latest, lr, lambda_):
  """DcOptimizer (dc_optimizer.cc:28-43) around the reference's SGD: one OptimizeWithLatestValue."""
  L = ref_opt_lib()
  num, grad, latest = _f32(num).copy(), _f32(grad), _f32(latest)
  assert L.ref_dc_sgd(_p(num, C.c_float), _p(grad, C.c_float), _p(latest, C.c_float), C.c_int(num.size),
                      C.c_float(lr), C.c_float(lambda_)) == 0
  return num


_ref_filter = None


def ref_filter_available():
  return os.path.exists(os.path.join(_DIR, "_ref", "libmonolith_ref_filter.so"))


def ref_filter_lib():
  global _ref_filter
  if _ref_filter is None:
    build()
    L = C.CDLL(os.path.join(_DIR, "_ref", "libmonolith_ref_filter.so"))
    L.rf_create.restype = C.c_void_p
    L.rf_create.argtypes = [C.c_uint64, C.c_int]
    L.rf_destroy.argtypes = [C.c_void_p]
    L.rf_add.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
    L.rf_add.restype = C.c_uint32
    L.rf_get.argtypes = [C.c_void_p, C.c_uint64]
    L.rf_get.restype = C.c_uint32
    L.rf_should_be_filtered.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
    for n in ("rf_estimated_total_element", "rf_failure_count", "rf_split_num"):
      getattr(L, n).restype = C.c_uint64
      getattr(L, n).argtypes = [C.c_void_p]
    L.rf_save_split.restype = C.c_int64
    L.rf_save_split.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_int64]
    L.rf_restore_split.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_int64]
    L.rf_add_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.rf_get_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    _ref_filter = L
  return _ref_filter


class RefSlidingFilter:
  """The reference's own SlidingHashFilter (sliding_hash_filter.cc + hash_filter.{h,cc} compiled in
  place, oracle/ref_filter_driver.cc), absl::Hash replaced by the engine's fixed slot hash."""
  META = ("failure_count", "total_size", "num_elements", "fill_rate_e6", "split_num", "max_forward_step",
          "max_backward_step", "max_step", "head", "head_increment", "sliding_failure_count")

  def __init__(self, capacity, split_num):
    self._L = ref_filter_lib()
    self._h = C.c_void_p(self._L.rf_create(int(capacity), int(split_num)))

  def __del__(self):
    try:
      if self._h:
        self._L.rf_destroy(self._h)
        self._h = None
    except Exception:  # pylint: disable=broad-except
      pass

  def add(self, fid, count=1):
    return int(self._L.rf_add(self._h, int(fid) & 0xFFFFFFFFFFFFFFFF, int(count)))

  def get(self, fid):
    return int(self._L.rf_get(self._h, int(fid) & 0xFFFFFFFFFFFFFFFF))

  def add_many(self, fids, counts):
    fids = np.ascontiguousarray(fids, dtype=np.uint64)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    out = np.zeros(fids.size, dtype=np.uint32)
    self._L.rf_add_many(self._h, fids.ctypes.data, counts.ctypes.data, fids.size, out.ctypes.data)
    return out

  def get_many(self, fids):
    fids = np.ascontiguousarray(fids, dtype=np.uint64)
    out = np.zeros(fids.size, dtype=np.uint32)
    self._L.rf_get_many(self._h, fids.ctypes.data, fids.size, out.ctypes.data)
    return out

  def should_be_filtered(self, fid, count, threshold):
    return bool(self._L.rf_should_be_filtered(self._h, int(fid), int(count), int(threshold)))

  def estimated_total_element(self):
    return int(self._L.rf_estimated_total_element(self._h))

  def failure_count(self):
    return int(self._L.rf_failure_count(self._h))

  def save_split(self, split):
    """-> (meta dict, words uint32[total_size + 64]) — Filter::Save of one split"""
    meta = (C.c_uint64 * 11)()
    n = int(self._L.rf_save_split(self._h, split, meta, None, 0))
    words = np.zeros(n, dtype=np.uint32)
    self._L.rf_save_split(self._h, split, meta, words.ctypes.data_as(C.POINTER(C.c_uint32)), n)
    return dict(zip(self.META, [int(x) for x in meta])), words

  def restore_split(self, split, meta, words):
    m = (C.c_uint64 * 11)(*[int(meta[k]) for k in self.META])
    words = np.ascontiguousarray(words, dtype=np.uint32)
    return int(self._L.rf_restore_split(self._h, split, m, words.ctypes.data_as(C.POINTER(C.c_uint32)),
                                        words.size)) == 0
