"""ctypes binding of libmhte.so (include/monolith_amd_hash_table.h).  Fails loudly when the HIP
extension is missing: the product has no CPU or PyTorch fallback."""
import ctypes as C
import os
import subprocess

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "libmhte.so")
_SRC = os.path.join(_DIR, "csrc", "mhte.hip")
# every header under csrc/ (a new one must not leave a stale library looking current)
_DEPS = [_SRC] + sorted(os.path.join(_DIR, "csrc", h) for h in os.listdir(os.path.join(_DIR, "csrc"))
                        if h.endswith(".h")) + [os.path.join(_DIR, "..", "include", "monolith_amd_hash_table.h")]
# A/B measurements: MHTE_LIBRARY=<other build of libmhte.so> (same ABI) is loaded instead
_OVERRIDE = os.environ.get("MHTE_LIBRARY")

MHTE_OK = 0
MHTE_INVALID_ARGUMENT = 3
MHTE_NOT_FOUND = 5
MHTE_RESOURCE_EXHAUSTED = 8
MHTE_FAILED_PRECONDITION = 9
MHTE_INTERNAL = 13
MHTE_UNAVAILABLE = 14

MHTE_IDS_UNIQUE = 1
MHTE_SUM_DUPLICATES = 2
MHTE_EXACT_ORDER = 1       # flags of mhte_table_sum_optimize_n
MHTE_DEFER_SLOWPATH = 2
MHTE_LAYOUT_ONE_FID_UNIQUE_ROWS = 1
ABI_VERSION = 17           # MHTE_ABI_VERSION of include/monolith_amd_hash_table.h

OPT_SGD, OPT_ADAGRAD, OPT_FTRL = 0, 1, 2
OPT_MOMENTUM, OPT_ADADELTA, OPT_RMSPROP, OPT_RMSPROPV2, OPT_ADAM, OPT_AMSGRAD = 3, 4, 5, 6, 7, 8
OPT_MOVING_AVERAGE, OPT_BATCH_SOFTMAX, OPT_GROUP_ADAGRAD = 9, 10, 11
OPT_FLAG_STOCHASTIC_ROUNDING_FP16 = 0x100   # OR-ed into opt_type
INIT_ZEROS, INIT_ONES, INIT_CONSTANT, INIT_RANDOM_UNIFORM = 0, 1, 2, 3


class MhteError(RuntimeError):
  """Raised for every non-OK status; ``code`` is the TensorFlow error code the reference op would
  have produced (InvalidArgument = 3, ResourceExhausted = 8, ...)."""

  def __init__(self, code, msg):
    super().__init__("[mhte status %d] %s" % (code, msg))
    self.code = code


class InvalidArgumentError(MhteError):
  pass


class ResourceExhaustedError(MhteError):
  pass


class SegmentConfig(C.Structure):
  _fields_ = [("dim_size", C.c_int32), ("opt_type", C.c_int32), ("opt_params", C.c_float * 8),
              ("init_type", C.c_int32), ("init_value", C.c_float), ("init_value2", C.c_float)]


class TableConfig(C.Structure):
  _fields_ = [("name", C.c_char_p), ("n_segments", C.c_int32),
              ("segments", C.POINTER(SegmentConfig)), ("initial_capacity", C.c_uint64),
              ("reserve_rows", C.c_uint64), ("max_load_factor", C.c_float),
              ("default_expire_days", C.c_int64), ("n_slot_expire", C.c_int32),
              ("expire_slots", C.POINTER(C.c_int64)), ("expire_days", C.POINTER(C.c_int32)),
              ("default_occurrence_threshold", C.c_int32), ("n_slot_occurrence", C.c_int32),
              ("occurrence_slots", C.POINTER(C.c_int64)),
              ("occurrence_thresholds", C.POINTER(C.c_int32)),
              ("enable_feature_eviction", C.c_int32), ("feature_evict_every_n_hours", C.c_int32)]


class LayoutSlice(C.Structure):
  _fields_ = [("feature_idx", C.c_int32), ("start", C.c_int32), ("dim", C.c_int32),
              ("pooling", C.c_int32), ("max_sequence_length", C.c_int32), ("out_type", C.c_int32),
              ("out_index", C.c_int32), ("out_offset", C.c_int32), ("out_row_floats", C.c_int32)]


class TableStats(C.Structure):
  _fields_ = [("size", C.c_int64), ("hashpower", C.c_int32), ("rows_allocated", C.c_int64),
              ("lookup_hits", C.c_int64), ("dropped", C.c_int64), ("evicted", C.c_int64),
              ("max_update_ts", C.c_int64), ("bytes_buckets", C.c_int64),
              ("bytes_rows", C.c_int64)]


def library_path():
  return _SO


def _source_hash():
  """sha256 over the library's sources (names and contents)."""
  import hashlib
  h = hashlib.sha256()
  for d in sorted(_DEPS):
    if os.path.exists(d):
      h.update(os.path.basename(d).encode())
      with open(d, "rb") as f:
        h.update(f.read())
  return h.hexdigest()


_HASH_FILE = _SO + ".srchash"


def _stale():
  """The library is stale when it was built from other sources: by CONTENT (a sidecar keeps the hash
  of the sources of the last build) — time stamps do not survive a checkout or the copy to a GPU box.
  A library without a sidecar (built by hand) falls back to time stamps."""
  if not os.path.exists(_SO):
    return True
  if os.path.exists(_HASH_FILE):
    with open(_HASH_FILE) as f:
      return f.read().strip() != _source_hash()
  t = os.path.getmtime(_SO)
  return any(os.path.exists(d) and os.path.getmtime(d) > t for d in _DEPS)


def build_library(force=False, verbose=False):
  """hipcc --offload-arch=gfx950 cross-compiles without a GPU; output stays in-tree."""
  if not force and not _stale():
    _try_build_eager_loop(verbose)
    return _SO
  cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-shared",
         "-fPIC", "-o", _SO, _SRC]
  if verbose:
    print(" ".join(cmd))
  if os.path.exists(_HASH_FILE):
    os.remove(_HASH_FILE)
  src_hash = _source_hash()   # (of what the compiler is about to read: an edit during the minutes of the
                              # build must leave the library stale, not blessed)
  subprocess.check_call(cmd)
  with open(_HASH_FILE, "w") as f:
    f.write(src_hash + "\n")
  _try_build_eager_loop(verbose)
  return _SO


# the plain-C step loop of bench.py's `eager_cpp` timing (csrc/eager_loop.c): gcc, the public header,
# linked against libmhte.so beside it — a measurement aid, not part of the op surface
_EAGER_SO = os.path.join(_DIR, "libmhte_eager.so")
_EAGER_SRC = os.path.join(_DIR, "csrc", "eager_loop.c")


_HEADER = os.path.join(_DIR, "..", "include", "monolith_amd_hash_table.h")


def _try_build_eager_loop(verbose=False):
  """The helper is a benchmark aid: its build must not make the engine unavailable — a
  failure is reported and eager_lib() tries again when the helper is asked for."""
  try:
    build_eager_loop()
  except (OSError, subprocess.CalledProcessError) as e:
    if verbose:
      print("libmhte_eager.so not built (%s); bench.py's eager_cpp timing will retry" % e)


def build_eager_loop(force=False):
  # stale when older than its source, the public header it is compiled against, or the engine it binds
  deps = [d for d in (_EAGER_SRC, _HEADER, _SO) if os.path.exists(d)]
  if (not force and os.path.exists(_EAGER_SO) and
      all(os.path.getmtime(_EAGER_SO) >= os.path.getmtime(d) for d in deps)):
    return _EAGER_SO
  subprocess.check_call(["gcc", "-O2", "-std=c99", "-Wall", "-shared", "-fPIC",
                         "-I", os.path.join(_DIR, "..", "include"), _EAGER_SRC, "-o", _EAGER_SO,
                         "-L", _DIR, "-lmhte", "-Wl,-rpath,$ORIGIN"])
  return _EAGER_SO


_eager = None


def eager_lib():
  """libmhte_eager.so (mhte_eager_step_loop), loaded after libmhte.so."""
  global _eager
  if _eager is None:
    lib()
    if _OVERRIDE:   # (libmhte_eager.so binds the in-tree libmhte.so: two copies of the engine in one process)
      raise MhteError(MHTE_UNAVAILABLE, "libmhte_eager.so is not used with MHTE_LIBRARY")
    build_eager_loop()
    E = C.CDLL(_EAGER_SO)
    if not hasattr(E, "mhte_eager_abi_version") or E.mhte_eager_abi_version() != ABI_VERSION:
      build_eager_loop(force=True)   # (built against another version of the header)
      raise MhteError(MHTE_UNAVAILABLE, "libmhte_eager.so was stale (ABI); rebuilt — start the process again")
    E.mhte_eager_step_loop.restype = C.c_int32
    _eager = E
  return _eager


# every symbol include/monolith_amd_hash_table.h declares
EXPORTS = [
    "mhte_last_error", "mhte_abi_version", "mhte_multi_table_create", "mhte_multi_table_destroy",
    "mhte_num_tables", "mhte_table_name", "mhte_table_dim", "mhte_table_slice_size",
    "mhte_table_index", "mhte_shared_name", "mhte_lookup", "mhte_optimize", "mhte_assign",
    "mhte_assign_add", "mhte_reinitialize", "mhte_compute_fused_offsets", "mhte_fused_lookup",
    "mhte_fused_optimize", "mhte_table_size", "mhte_table_contains", "mhte_table_evict",
    "mhte_table_get_stats", "mhte_table_dump", "mhte_table_row_floats", "mhte_dedup_ws_create",
    "mhte_dedup_ws_destroy", "mhte_unique", "mhte_gather_rows", "mhte_segment_sum",
    "mhte_table_lookup_n", "mhte_table_optimize_n", "mhte_value_offsets",
    "mhte_fill_with_offset_map", "mhte_fill_with_offset_map_gradient", "mhte_table_set_count_hits",
    "mhte_table_sum_optimize_n", "mhte_unique_unordered", "mhte_table_fused_backward_ok", "mhte_table_finish_pending",
    "mhte_table_step_forward", "mhte_table_step_backward", "mhte_table_step_backward_ahead", "mhte_profile_arm", "mhte_profile_read",
    "mhte_trace_begin", "mhte_trace_end", "mhte_step_dedup", "mhte_shard_partition",
    "mhte_step_scatter", "mhte_step_sum", "mhte_multi_table_save", "mhte_multi_table_restore", "mhte_table_save", "mhte_table_restore", "mhte_table_clear",
    "mhte_hash_filter_create", "mhte_hash_filter_destroy", "mhte_multi_table_set_filter",
    "mhte_hash_filter_get", "mhte_hash_filter_stats", "mhte_hash_filter_create_probabilistic", "mhte_fused_gather_embeddings_by_input",
    "mhte_fused_gather_embeddings_by_input_gradient", "mhte_reduce_rows",
    "mhte_multi_step_create", "mhte_multi_step_destroy", "mhte_multi_step_forward",
    "mhte_multi_step_backward", "mhte_multi_step_unique_counts",
    "mhte_shard_unique_id", "mhte_shard_step_create", "mhte_shard_step_destroy",
    "mhte_shard_step_create_ipc", "mhte_shard_step_ipc_handle", "mhte_shard_step_ipc_connect",
    "mhte_shard_step_ipc_selftest", "mhte_shard_step_set_overlap", "mhte_shard_step_set_grad_bits", "mhte_shard_step_set_exact_order", "mhte_shard_step_forward", "mhte_shard_step_backward", "mhte_shard_step_check",
    "mhte_shard_step_info", "mhte_shard_step_launches", "mhte_shard_step_wire_stats", "mhte_shard_step_comm_ranks", "mhte_shard_step_unique_counts", "mhte_shard_group_forward", "mhte_shard_group_backward",
    "mhte_multi_table_create_from_proto", "mhte_multi_table_find", "mhte_multi_table_is_initialized",
    "mhte_hash_filter_create_from_proto", "mhte_lookup_entry", "mhte_feature_stat",
    "mhte_table_save_as_tensor", "mhte_lookup_gradient",
    "mhte_advance_clock_for_testing", "mhte_hash_filter_save", "mhte_hash_filter_restore",
    "mhte_embedding_to_layout", "mhte_embedding_to_layout_grad",
    "mhte_dense_mlp_create", "mhte_dense_mlp_destroy", "mhte_dense_mlp_set_params",
    "mhte_dense_mlp_get_params", "mhte_dense_mlp_forward", "mhte_dense_mlp_backward",
]

_lib = None


def lib():
  """Loads libmhte.so (building it first if the sources are newer).  Raises if unavailable."""
  global _lib
  if _lib is None:
    # torch ships its own libamdhip64/libhsa-runtime64; loading libmhte.so first would pull
    # /opt/rocm's copies into the process as well and the second HSA runtime then finds no device.
    # Import torch first so libmhte.so binds to the runtime that owns the tensors and streams.
    import torch  # noqa: F401  pylint: disable=unused-import,import-outside-toplevel
    # MHTE_LIBRARY (another build, A/B runs) and MHTE_NO_REBUILD=1 (use the .so as it is) are the
    # explicit opt-outs; otherwise a library older than its sources is rebuilt, and a failed
    # rebuild is an error — never a silent fall back to the stale binary
    if not _OVERRIDE and os.environ.get("MHTE_NO_REBUILD") != "1" and _stale():
      try:
        build_library()
      except Exception as e:  # pylint: disable=broad-except
        raise MhteError(MHTE_UNAVAILABLE,
                        "libmhte.so is %s and could not be rebuilt (%s); the MI355X engine has no "
                        "fallback path (MHTE_NO_REBUILD=1 loads an existing library as it is)" %
                        ("older than its sources" if os.path.exists(_SO) else "missing", e))
    L = C.CDLL(_OVERRIDE or _SO)
    for name in EXPORTS:
      if not hasattr(L, name):
        raise MhteError(MHTE_INTERNAL, "libmhte.so does not export %s" % name)
    L.mhte_last_error.restype = C.c_char_p
    for name in ("mhte_table_name", "mhte_shared_name"):
      getattr(L, name).restype = C.c_char_p
    L.mhte_multi_table_destroy.restype = None
    L.mhte_dedup_ws_destroy.restype = None
    L.mhte_hash_filter_destroy.restype = None
    L.mhte_multi_table_find.restype = C.c_void_p
    L.mhte_multi_table_find.argtypes = [C.c_char_p]
    L.mhte_multi_table_is_initialized.argtypes = [C.c_char_p]
    L.mhte_advance_clock_for_testing.restype = None
    L.mhte_advance_clock_for_testing.argtypes = [C.c_double]
    L.mhte_multi_step_destroy.restype = None
    L.mhte_multi_step_destroy.argtypes = [C.c_void_p]
    L.mhte_shard_step_destroy.restype = None
    L.mhte_shard_step_destroy.argtypes = [C.c_void_p]
    L.mhte_dense_mlp_destroy.restype = None
    L.mhte_dense_mlp_destroy.argtypes = [C.c_void_p]
    L.mhte_hash_filter_destroy.argtypes = [C.c_void_p]
    L.mhte_multi_table_destroy.argtypes = [C.c_void_p]
    L.mhte_dedup_ws_destroy.argtypes = [C.c_void_p]
    for name in ("mhte_num_tables",):
      getattr(L, name).argtypes = [C.c_void_p]
    for name in ("mhte_table_name", "mhte_table_dim", "mhte_table_slice_size",
                 "mhte_table_row_floats", "mhte_table_fused_backward_ok"):
      getattr(L, name).argtypes = [C.c_void_p, C.c_int32]
    L.mhte_table_index.argtypes = [C.c_void_p, C.c_char_p]
    L.mhte_shared_name.argtypes = [C.c_void_p]
    if L.mhte_abi_version() != ABI_VERSION:
      raise MhteError(MHTE_INTERNAL, "libmhte.so ABI version mismatch")
    _lib = L
  return _lib


def check(status):
  if status == MHTE_OK:
    return
  msg = lib().mhte_last_error().decode("utf-8", "replace")
  if status == MHTE_INVALID_ARGUMENT:
    raise InvalidArgumentError(status, msg)
  if status == MHTE_RESOURCE_EXHAUSTED:
    raise ResourceExhaustedError(status, msg)
  raise MhteError(status, msg)


def vp(x):
  """torch tensor / int / None -> c_void_p"""
  if x is None:
    return C.c_void_p(0)
  if isinstance(x, int):
    return C.c_void_p(x)
  return C.c_void_p(x.data_ptr())


PROFILE_TAGS = {1: "lookup_kernel", 2: "sum_apply_kernel", 6: "slowpath_kernel", 7: "dd_kernels",
                8: "upsert_kernel", 9: "step_fwd_kernel", 10: "step_bwd_kernel",
                11: "mstep_fwd_kernel", 12: "mstep_bwd_kernel", 13: "shard_build_kernel",
                14: "shard_lookup_kernel", 15: "shard_scatter_kernel", 16: "shard_upsert_kernel",
                17: "shard_push_kernel", 18: "shard_wait_kernel", 19: "gemm_nt_bf16_kernel"}
TRACE_WORDS = 8
TRACE_ROLES = {3: "run_dedup", 4: "displacement", 5: "lookup", 6: "work_list", 7: "apply_items",
               8: "apply_ids", 9: "reserve_rows"}


def profile_arm(n):
  """Kernel-exact timing of the next ``n`` hot launches of this thread (mhte_profile_arm)."""
  check(lib().mhte_profile_arm(C.c_int32(int(n))))


def profile_read(cap=65536):
  """-> [(kernel name, microseconds)] of the launches recorded since profile_arm; disarms."""
  tags = (C.c_int32 * cap)()
  us = (C.c_float * cap)()
  n = C.c_int32(0)
  check(lib().mhte_profile_read(C.c_int32(cap), tags, us, C.byref(n)))
  m = min(cap, n.value)
  return [(PROFILE_TAGS.get(tags[i], str(tags[i])), float(us[i])) for i in range(m)]


def trace_begin(buf, cap_records):
  """Per-wavefront timeline of the step kernels into ``buf`` (uint64 cuda tensor, TRACE_WORDS
  words per record) — mhte_trace_begin."""
  check(lib().mhte_trace_begin(vp(buf), C.c_int64(int(cap_records))))


def trace_end(cap=4096):
  """-> [(kernel name, grid, block, record offset)] of the traced launches; stops tracing."""
  tag = (C.c_int32 * cap)()
  grid = (C.c_int32 * cap)()
  block = (C.c_int32 * cap)()
  off = (C.c_int64 * cap)()
  n = C.c_int32(0)
  check(lib().mhte_trace_end(C.c_int32(cap), tag, grid, block, off, C.byref(n)))
  m = min(cap, n.value)
  return [(PROFILE_TAGS.get(tag[i], str(tag[i])), grid[i], block[i], off[i]) for i in range(m)]
