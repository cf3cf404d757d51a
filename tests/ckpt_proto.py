"""Test infrastructure: the reference's checkpoint messages as protobuf runtime classes, built
from descriptors that restate the .proto definitions (no protoc in this image), plus an independent
pure-Python TFRecord / TF-snappy-stream reader and writer.  Used to check the C++ codec
(monolith_amd/csrc/mhte_ckpt.h) byte for byte.

  EntryDump, MultiHashTableMetadata   runtime/hash_table/embedding_hash_table.proto:45-50,139-142
  OptimizerDump & co                  runtime/hash_table/optimizer/optimizer.proto:28-31,56-58,
                                      69-72,231-252
"""
import struct

from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

F = descriptor_pb2.FieldDescriptorProto


def _build():
  fd = descriptor_pb2.FileDescriptorProto()
  fd.name = "mhte_ckpt_test.proto"
  fd.package = "monolith.hash_table"
  fd.syntax = "proto2"

  def msg(name, fields, oneof=None):
    m = fd.message_type.add()
    m.name = name
    if oneof:
      m.oneof_decl.add().name = oneof
    for (fname, num, ftype, label, tname, in_oneof) in fields:
      f = m.field.add()
      f.name, f.number, f.type, f.label = fname, num, ftype, label
      if tname:
        f.type_name = ".monolith.hash_table." + tname
      if in_oneof:
        f.oneof_index = 0
    return m

  OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
  msg("AdagradOptimizerDump", [("norm", 1, F.TYPE_FLOAT, REP, None, False)])
  msg("SgdOptimizerDump", [])
  msg("FtrlOptimizerDump", [("zero", 1, F.TYPE_FLOAT, REP, None, False),
                            ("norm", 2, F.TYPE_FLOAT, REP, None, False)])
  fl = lambda name, num: (name, num, F.TYPE_FLOAT, REP, None, False)
  sc = lambda name, num: (name, num, F.TYPE_FLOAT, OPT, None, False)
  msg("AdadeltaOptimizerDump", [fl("accum", 1), fl("accum_update", 2)])          # optimizer.proto:113-116
  msg("AdamOptimizerDump", [fl("m", 1), fl("v", 2), sc("beta1_power", 3), sc("beta2_power", 4)])  # :130-135
  msg("AmsgradOptimizerDump", [fl("m", 1), fl("v", 2), fl("vhat", 3), sc("beta1_power", 4),
                               sc("beta2_power", 5)])                             # :148-154
  msg("MomentumOptimizerDump", [fl("n", 1)])                                      # :165-167
  msg("RmspropOptimizerDump", [fl("n", 1)])                                       # :193-195
  msg("RmspropV2OptimizerDump", [fl("n", 1)])                                     # :204-206
  msg("BatchSoftmaxOptimizerDump", [("global_step", 1, F.TYPE_INT64, OPT, None, False)])  # :179-181
  msg("GroupAdaGradOptimizerDump", [sc("grad_square_sum", 1)])                    # :100-102
  msg("SingleOptimizerDump", [("adagrad", 1, F.TYPE_MESSAGE, OPT, "AdagradOptimizerDump", True),
                              ("sgd", 2, F.TYPE_MESSAGE, OPT, "SgdOptimizerDump", True),
                              ("ftrl", 3, F.TYPE_MESSAGE, OPT, "FtrlOptimizerDump", True),
                              ("adadelta", 6, F.TYPE_MESSAGE, OPT, "AdadeltaOptimizerDump", True),
                              ("adam", 7, F.TYPE_MESSAGE, OPT, "AdamOptimizerDump", True),
                              ("amsgrad", 8, F.TYPE_MESSAGE, OPT, "AmsgradOptimizerDump", True),
                              ("momentum", 9, F.TYPE_MESSAGE, OPT, "MomentumOptimizerDump", True),
                              ("rmsprop", 11, F.TYPE_MESSAGE, OPT, "RmspropOptimizerDump", True),
                              ("rmspropv2", 12, F.TYPE_MESSAGE, OPT, "RmspropV2OptimizerDump", True),
                              ("batch_softmax", 14, F.TYPE_MESSAGE, OPT, "BatchSoftmaxOptimizerDump", True),
                              ("group_adagrad", 15, F.TYPE_MESSAGE, OPT, "GroupAdaGradOptimizerDump", True)],
      oneof="type")
  msg("OptimizerDump", [("dump", 1, F.TYPE_MESSAGE, REP, "SingleOptimizerDump", False)])
  msg("EntryDump", [("id", 1, F.TYPE_SFIXED64, OPT, None, False),
                    ("num", 2, F.TYPE_FLOAT, REP, None, False),
                    ("opt", 3, F.TYPE_MESSAGE, OPT, "OptimizerDump", False),
                    ("last_update_ts_sec", 4, F.TYPE_INT64, OPT, None, False)])
  msg("MultiHashTableMetadata", [("table_name", 1, F.TYPE_STRING, OPT, None, False),
                                 ("num_entries", 2, F.TYPE_UINT64, OPT, None, False)])
  # hash filter dumps (embedding_hash_table.proto:112-137)
  u32f = lambda name, num: (name, num, F.TYPE_UINT32, OPT, None, False)
  u64f = lambda name, num: (name, num, F.TYPE_UINT64, OPT, None, False)
  msg("SlidingHashFilterMetaDump", [u32f("split_num", 1), u32f("max_forward_step", 2), u32f("max_backward_step", 3),
                                    u32f("max_step", 4), u32f("head", 5), u32f("head_increment", 6),
                                    u64f("failure_count", 7)])
  msg("HashFilterSplitMetaDump", [u64f("failure_count", 1), u64f("total_size", 2), u64f("num_elements", 3),
                                  ("fill_rate", 4, F.TYPE_DOUBLE, OPT, None, False),
                                  ("sliding_hash_filter_meta", 5, F.TYPE_MESSAGE, OPT, "SlidingHashFilterMetaDump",
                                   False)])
  msg("HashFilterSplitDataDump", [u32f("offset", 1), ("data", 2, F.TYPE_UINT32, REP, None, False)])
  # ---- configuration messages (embedding_hash_table.proto:23-43,54-96,100-110; optimizer.proto;
  # initializer/initializer_config.proto), for the C ABI's proto-config reader
  i32 = lambda name, num: (name, num, F.TYPE_INT32, OPT, None, False)
  i64 = lambda name, num: (name, num, F.TYPE_INT64, OPT, None, False)
  u32 = lambda name, num: (name, num, F.TYPE_UINT32, OPT, None, False)
  bl = lambda name, num: (name, num, F.TYPE_BOOL, OPT, None, False)
  msg("AdagradOptimizerConfig", [i32("dim_size", 1), sc("learning_rate", 2), sc("initial_accumulator_value", 3),
                                 i32("hessian_compression_times", 4), sc("weight_decay_factor", 5),
                                 i64("warmup_steps", 6)])
  msg("SgdOptimizerConfig", [i32("dim_size", 1), sc("learning_rate", 2), i64("warmup_steps", 6)])
  msg("FtrlOptimizerConfig", [i32("dim_size", 1), sc("learning_rate", 2), sc("beta", 3),
                              sc("initial_accumulator_value", 4), sc("l1_regularization_strength", 5),
                              sc("l2_regularization_strength", 6), i64("warmup_steps", 7)])
  msg("AdamOptimizerConfig", [i32("dim_size", 1), sc("learning_rate", 2), sc("beta1", 3), sc("beta2", 4),
                              bl("use_beta1_warmup", 5), sc("weight_decay_factor", 6), bl("use_nesterov", 7),
                              sc("epsilon", 8), i64("warmup_steps", 9)])
  msg("MomentumOptimizerConfig", [i32("dim_size", 1), sc("learning_rate", 2), sc("weight_decay_factor", 3),
                                  bl("use_nesterov", 4), sc("momentum", 5), i64("warmup_steps", 6)])
  msg("DcOptimizerConfig", [i32("dim_size", 1), sc("lambda_", 2)])
  msg("GroupAdaGradOptimizerConfig", [i32("dim_size", 1), sc("learning_rate", 2), sc("beta", 3),
                                      sc("initial_accumulator_value", 4), sc("l2_regularization_strength", 5),
                                      sc("weight_decay_factor", 6), i64("warmup_steps", 7)])
  msg("OptimizerConfig", [("adagrad", 1, F.TYPE_MESSAGE, OPT, "AdagradOptimizerConfig", True),
                          ("sgd", 2, F.TYPE_MESSAGE, OPT, "SgdOptimizerConfig", True),
                          ("ftrl", 3, F.TYPE_MESSAGE, OPT, "FtrlOptimizerConfig", True),
                          ("adam", 7, F.TYPE_MESSAGE, OPT, "AdamOptimizerConfig", True),
                          ("momentum", 9, F.TYPE_MESSAGE, OPT, "MomentumOptimizerConfig", True),
                          ("dc", 13, F.TYPE_MESSAGE, OPT, "DcOptimizerConfig", True),
                          ("group_adagrad", 16, F.TYPE_MESSAGE, OPT, "GroupAdaGradOptimizerConfig", True),
                          ("stochastic_rounding_float16", 4, F.TYPE_BOOL, OPT, None, False)], oneof="type")
  msg("ZerosInitializerConfig", [i32("dim_size", 1)])
  msg("OnesInitializerConfig", [i32("dim_size", 1)])
  msg("ConstantsInitializerConfig", [i32("dim_size", 1), sc("constant", 2)])
  msg("RandomUniformInitializerConfig", [i32("dim_size", 1), sc("minval", 2), sc("maxval", 3)])
  msg("InitializerConfig", [("zeros", 1, F.TYPE_MESSAGE, OPT, "ZerosInitializerConfig", True),
                            ("random_uniform", 2, F.TYPE_MESSAGE, OPT, "RandomUniformInitializerConfig", True),
                            ("ones", 3, F.TYPE_MESSAGE, OPT, "OnesInitializerConfig", True),
                            ("constants", 15, F.TYPE_MESSAGE, OPT, "ConstantsInitializerConfig", True)],
      oneof="type")
  msg("Segment", [("init_config", 1, F.TYPE_MESSAGE, OPT, "InitializerConfig", False),
                  ("opt_config", 2, F.TYPE_MESSAGE, OPT, "OptimizerConfig", False),
                  i32("dim_size", 7)])
  msg("EntryConfig", [("segments", 1, F.TYPE_MESSAGE, REP, "Segment", False), i32("entry_type", 2)])
  msg("SlotExpireTime", [u32("slot", 1), u32("expire_time", 2)])
  msg("SlotExpireTimeConfig", [("slot_expire_times", 1, F.TYPE_MESSAGE, REP, "SlotExpireTime", False),
                               u32("default_expire_time", 2)])
  msg("CuckooEmbeddingHashTableConfig", [])
  msg("EmbeddingHashTableConfig", [("entry_config", 1, F.TYPE_MESSAGE, OPT, "EntryConfig", False),
                                   ("initial_capacity", 2, F.TYPE_UINT64, OPT, None, False),
                                   ("slot_expire_time_config", 3, F.TYPE_MESSAGE, OPT, "SlotExpireTimeConfig", False),
                                   ("cuckoo", 5, F.TYPE_MESSAGE, OPT, "CuckooEmbeddingHashTableConfig", False),
                                   i32("entry_type", 6), bl("enable_feature_eviction", 7),
                                   i32("feature_evict_every_n_hours", 8), bl("skip_zero_embedding", 10)])
  msg("MultiEmbeddingHashTableConfig", [("names", 1, F.TYPE_STRING, REP, None, False),
                                        ("configs", 2, F.TYPE_MESSAGE, REP, "EmbeddingHashTableConfig", False)])
  msg("SlotOccurrenceThreshold", [u32("slot", 1), u32("occurrence_threshold", 2)])
  msg("SlotOccurrenceThresholdConfig",
      [("slot_occurrence_thresholds", 1, F.TYPE_MESSAGE, REP, "SlotOccurrenceThreshold", False),
       u32("default_occurrence_threshold", 2)])
  pool = descriptor_pool.DescriptorPool()
  pool.Add(fd)
  get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName("monolith.hash_table." + n))
  return get


_get = _build()
EntryDump, MultiHashTableMetadata = _get("EntryDump"), _get("MultiHashTableMetadata")
HashFilterSplitMetaDump, HashFilterSplitDataDump = _get("HashFilterSplitMetaDump"), _get("HashFilterSplitDataDump")


def read_filter_split(path):
  """One split file of MonolithHashFilterSave (hash_filter.cc:29-58): the first record is the
  HashFilterSplitMetaDump, the rest HashFilterSplitDataDump of <= 10 000 words each -> (meta, words)."""
  import numpy as np
  recs = unframe(open(path, "rb").read())
  meta = HashFilterSplitMetaDump.FromString(recs[0])
  words = np.zeros(int(meta.total_size) + 64, dtype=np.uint32)
  for r in recs[1:]:
    d = HashFilterSplitDataDump.FromString(r)
    assert len(d.data) <= 10000
    words[d.offset:d.offset + len(d.data)] = np.array(d.data, dtype=np.uint32)
  return meta, words
MultiEmbeddingHashTableConfig = _get("MultiEmbeddingHashTableConfig")
EmbeddingHashTableConfig = _get("EmbeddingHashTableConfig")
SlotOccurrenceThresholdConfig = _get("SlotOccurrenceThresholdConfig")


# ---------------------------------------------------------------------------------- crc32c / TFRecord
_TABLE = []
for _i in range(256):
  _c = _i
  for _ in range(8):
    _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
  _TABLE.append(_c)


def crc32c(b: bytes) -> int:
  c = 0xFFFFFFFF
  for x in b:
    c = _TABLE[(c ^ x) & 0xFF] ^ (c >> 8)
  return c ^ 0xFFFFFFFF


def masked(b: bytes) -> int:
  c = crc32c(b)
  return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def frame(rec: bytes) -> bytes:
  hdr = struct.pack("<Q", len(rec))
  return hdr + struct.pack("<I", masked(hdr)) + rec + struct.pack("<I", masked(rec))


def unframe(stream: bytes):
  out, p = [], 0
  while p < len(stream):
    (n,) = struct.unpack_from("<Q", stream, p)
    (c1,) = struct.unpack_from("<I", stream, p + 8)
    assert c1 == masked(stream[p:p + 8]), "length crc"
    rec = stream[p + 12:p + 12 + n]
    (c2,) = struct.unpack_from("<I", stream, p + 12 + n)
    assert c2 == masked(rec), "data crc"
    out.append(rec)
    p += 16 + n
  return out


# ---------------------------------------------------------------------------------- snappy (raw block)
def _varint(n):
  out = bytearray()
  while n >= 0x80:
    out.append((n & 0x7F) | 0x80)
    n >>= 7
  out.append(n)
  return bytes(out)


def snappy_uncompress(b: bytes) -> bytes:
  p, n, shift = 0, 0, 0
  while True:
    x = b[p]
    p += 1
    n |= (x & 0x7F) << shift
    shift += 7
    if not x & 0x80:
      break
  out = bytearray()
  while p < len(b):
    tag = b[p]
    p += 1
    t = tag & 3
    if t == 0:
      ln = tag >> 2
      if ln >= 60:
        nb = ln - 59
        ln = int.from_bytes(b[p:p + nb], "little")
        p += nb
      ln += 1
      out += b[p:p + ln]
      p += ln
    else:
      if t == 1:
        ln, off = 4 + ((tag >> 2) & 7), ((tag >> 5) << 8) | b[p]
        p += 1
      elif t == 2:
        ln, off = 1 + (tag >> 2), int.from_bytes(b[p:p + 2], "little")
        p += 2
      else:
        ln, off = 1 + (tag >> 2), int.from_bytes(b[p:p + 4], "little")
        p += 4
      for _ in range(ln):
        out.append(out[-off])
  assert len(out) == n
  return bytes(out)


def snappy_compress_with_copies(data: bytes) -> bytes:
  """A small greedy compressor that DOES emit copy elements (2-byte offsets), so that files made
  here exercise the reader's copy paths."""
  out = bytearray(_varint(len(data)))
  lit = bytearray()
  table = {}

  def flush():
    nonlocal lit
    i = 0
    while i < len(lit):
      chunk = lit[i:i + 60]
      out.append((len(chunk) - 1) << 2)
      out.extend(chunk)
      i += 60
    lit = bytearray()

  i = 0
  while i < len(data):
    key = data[i:i + 4]
    j = table.get(key) if len(key) == 4 else None
    table[key] = i
    if j is not None and 0 < i - j < 65536:
      ln = 4
      while ln < 64 and i + ln < len(data) and data[j + ln] == data[i + ln]:
        ln += 1
      flush()
      out.append(((ln - 1) << 2) | 2)
      out += struct.pack("<H", i - j)
      i += ln
    else:
      lit.append(data[i])
      i += 1
  flush()
  return bytes(out)


def read_tf_snappy(stream: bytes) -> bytes:
  out, p = bytearray(), 0
  while p < len(stream):
    (cl,) = struct.unpack_from(">I", stream, p)
    out += snappy_uncompress(stream[p + 4:p + 4 + cl])
    p += 4 + cl
  return bytes(out)


def write_tf_snappy(raw: bytes, block=262144) -> bytes:
  out = bytearray()
  for i in range(0, len(raw), block):
    c = snappy_compress_with_copies(raw[i:i + block])
    out += struct.pack(">I", len(c)) + c
  return bytes(out)
