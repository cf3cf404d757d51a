// Host-side codec of the reference's MultiHashTable checkpoint format
// (monolith/native_training/runtime/ops/multi_hash_table_save_restore_ops.cc:107-238,323-404):
//
//   <basename>-%05d-of-%05d        TFRecord stream, SNAPPY-compressed, of serialized EntryDump
//                                  (hash_table/embedding_hash_table.proto:45-50), the tables of
//                                  the MultiHashTable one after another (sorted by name)
//   <basename>.meta-%05d-of-%05d   TFRecord stream, uncompressed, one MultiHashTableMetadata
//                                  {table_name, num_entries} (:139-142) per table
//
// The container formats are TensorFlow 2.4.0's (WORKSPACE:61-67; not under /root/reference), restated
// from their published definitions:
//   * TFRecord (tensorflow/core/lib/io/record_writer.cc): uint64 length | uint32 masked crc32c of
//     the length bytes | data | uint32 masked crc32c of the data; little-endian;
//     mask(c) = ((c >> 15) | (c << 17)) + 0xa282ead8; crc32c = CRC-32C (Castagnoli).
//   * SNAPPY compression of a record file (tensorflow/core/lib/io/snappy/snappy_outputbuffer.cc):
//     the record stream is cut into blocks of <= 256 KiB, each written as a 4-byte BIG-endian
//     compressed length followed by one raw snappy block (varint uncompressed length + elements,
//     google/snappy format_description.txt).  This writer emits literal elements only (a valid
//     snappy stream; fp32 payload does not compress anyway); the reader handles every element type.
//   * protobuf wire format (proto2: repeated scalars unpacked on write, either form accepted on read).
#ifndef MHTE_CKPT_H_
#define MHTE_CKPT_H_

#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <errno.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>

#include <stdlib.h>

#include <algorithm>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <type_traits>
#include <cstring>
#include <string>
#include <vector>

namespace mhte {
namespace ckpt {

// ------------------------------------------------------------------------------------ crc32c
// Slicing-by-8 tables (8 bytes per step); the SSE4.2 crc32 instruction (the same polynomial) is
// used when the host has it — the record stream of a large table is gigabytes of this.
struct Crc32cTables {
  uint32_t t[8][256];
  Crc32cTables() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int k = 1; k < 8; ++k) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xffu];
  }
};
inline const Crc32cTables& crc32c_tables() {
  static const Crc32cTables tabs;  // (thread-safe initialisation)
  return tabs;
}
inline uint32_t crc32c_soft(uint32_t c, const uint8_t* p, size_t n) {
  const Crc32cTables& T = crc32c_tables();
  while (n >= 8) {
    uint64_t v;
    memcpy(&v, p, 8);
    v ^= c;
    c = T.t[7][v & 0xff] ^ T.t[6][(v >> 8) & 0xff] ^ T.t[5][(v >> 16) & 0xff] ^
        T.t[4][(v >> 24) & 0xff] ^ T.t[3][(v >> 32) & 0xff] ^ T.t[2][(v >> 40) & 0xff] ^
        T.t[1][(v >> 48) & 0xff] ^ T.t[0][v >> 56];
    p += 8;
    n -= 8;
  }
  while (n--) c = T.t[0][(c ^ *p++) & 0xffu] ^ (c >> 8);
  return c;
}
#if defined(__x86_64__)
__attribute__((target("sse4.2"))) inline uint32_t crc32c_hw(uint32_t c, const uint8_t* p, size_t n) {
  uint64_t c64 = c;
  while (n >= 8) {
    uint64_t v;
    memcpy(&v, p, 8);
    c64 = __builtin_ia32_crc32di(c64, v);
    p += 8;
    n -= 8;
  }
  c = uint32_t(c64);
  while (n--) c = __builtin_ia32_crc32qi(c, *p++);
  return c;
}
#endif
inline uint32_t crc32c(const void* data, size_t n) {
  const uint8_t* p = static_cast<const uint8_t*>(data);
#if defined(__x86_64__)
  static const bool hw = __builtin_cpu_supports("sse4.2");
  if (hw) return crc32c_hw(0xffffffffu, p, n) ^ 0xffffffffu;
#endif
  return crc32c_soft(0xffffffffu, p, n) ^ 0xffffffffu;
}
inline uint32_t masked_crc(const void* data, size_t n) {
  const uint32_t c = crc32c(data, n);
  return ((c >> 15) | (c << 17)) + 0xa282ead8u;
}

// ------------------------------------------------------------------------------------ varint / proto
inline void put_varint(std::string& out, uint64_t v) {
  while (v >= 0x80) {
    out.push_back(char((v & 0x7f) | 0x80));
    v >>= 7;
  }
  out.push_back(char(v));
}
inline bool get_varint(const uint8_t*& p, const uint8_t* end, uint64_t* v) {
  uint64_t r = 0;
  for (int shift = 0; shift < 64 && p < end; shift += 7) {
    const uint8_t b = *p++;
    r |= uint64_t(b & 0x7f) << shift;
    if (!(b & 0x80)) {
      *v = r;
      return true;
    }
  }
  return false;
}
inline void put_f32(std::string& out, uint8_t tag, float f) {
  out.push_back(char(tag));
  uint32_t u;
  memcpy(&u, &f, 4);
  char b[4] = {char(u), char(u >> 8), char(u >> 16), char(u >> 24)};
  out.append(b, 4);
}

// Optimizer kinds of a segment as the codec sees them (SingleOptimizerDump's oneof,
// hash_table/optimizer/optimizer.proto:231-247)
// (values = the engine's OptType)
enum SegKind {
  kSegSgd = 0, kSegAdagrad = 1, kSegFtrl = 2, kSegMomentum = 3, kSegAdadelta = 4, kSegRmsprop = 5,
  kSegRmspropV2 = 6, kSegAdam = 7, kSegAmsgrad = 8, kSegMovingAverage = 9, kSegBatchSoftmax = 10,
  kSegGroupAdagrad = 11
};
// wire description of an optimizer's dump: field number in SingleOptimizerDump's oneof
// (optimizer.proto:231-247), the dump message's repeated-float fields in the order of the engine's
// state vectors, and the field numbers of the two scalars (0: none)
// oneof_field 0: the optimizer's Save() returns an EMPTY OptimizerDump (moving average,
// moving_average_optimizer.cc:54-57) — the segment contributes no SingleOptimizerDump at all and
// OptimizerCombination::Restore hands it none (optimizer_combination.cc:86-97, dump_size 0).
// step_field: the dump is one int64 varint (batch softmax's global_step, field 1), kept in the
// first two words of the segment's 4-float slot.
struct DumpSpec {
  int oneof_field;
  int nvec;
  int vec_field[3];
  int scal_field[2];
  int step_field;
  bool has_slot() const { return scal_field[0] != 0 || step_field != 0; }
};
inline DumpSpec dump_spec(int kind) {
  switch (kind) {
    case kSegSgd: return {2, 0, {0, 0, 0}, {0, 0}, 0};
    case kSegAdagrad: return {1, 1, {1, 0, 0}, {0, 0}, 0};          // norm
    case kSegFtrl: return {3, 2, {2, 1, 0}, {0, 0}, 0};             // engine: norm | zero; wire zero=1 norm=2
    case kSegMomentum: return {9, 1, {1, 0, 0}, {0, 0}, 0};         // n
    case kSegAdadelta: return {6, 2, {1, 2, 0}, {0, 0}, 0};         // accum, accum_update
    case kSegRmsprop: return {11, 1, {1, 0, 0}, {0, 0}, 0};         // n
    case kSegRmspropV2: return {12, 1, {1, 0, 0}, {0, 0}, 0};       // n
    case kSegAdam: return {7, 2, {1, 2, 0}, {3, 4}, 0};             // m, v, beta1_power, beta2_power
    case kSegMovingAverage: return {0, 0, {0, 0, 0}, {0, 0}, 0};    // (no dump)
    case kSegBatchSoftmax: return {14, 0, {0, 0, 0}, {0, 0}, 1};    // global_step
    case kSegGroupAdagrad: return {15, 0, {0, 0, 0}, {1, 0}, 0};    // grad_square_sum
    default: return {8, 3, {1, 2, 3}, {4, 5}, 0};                   // amsgrad: m, v, vhat, powers
  }
}
struct SegLayout {
  int dim;
  int kind;    // SegKind
  int w_off;   // float offsets inside the engine's row
  int st_off;
};

// EntryDump of one row (embedding_hash_table.proto:45-50; EntryAccessor::Save, entry_accessor.cc
// :218-226; optimizer Save()s: sgd_optimizer.cc:50-54, adagrad_optimizer.cc:62-70,
// ftrl_optimizer.cc:78-88; one SingleOptimizerDump per segment, optimizer_combination.cc:73-84)
inline int varint_len(uint64_t v) {
  int n = 1;
  while (v >= 0x80) {
    v >>= 7;
    ++n;
  }
  return n;
}
inline char* put_varint_raw(char* p, uint64_t v) {
  while (v >= 0x80) {
    *p++ = char((v & 0x7f) | 0x80);
    v >>= 7;
  }
  *p++ = char(v);
  return p;
}
inline char* put_f32_raw(char* p, uint8_t tag, float f) {
  *p++ = char(tag);
  memcpy(p, &f, 4);  // (little-endian host)
  return p + 4;
}
// (sizes first, then one pass over a buffer of the final size: a checkpoint is one of these per row)
inline void encode_entry(std::string& out, int64_t id, const float* row,
                         const std::vector<SegLayout>& segs, int dim, uint32_t ts) {
  auto step_of = [&](const SegLayout& s) {
    uint64_t a;
    memcpy(&a, row + s.st_off, 8);
    return a;
  };
  auto dump_len = [&](const SegLayout& s, const DumpSpec& ds) -> size_t {
    size_t m = 5u * (size_t(ds.nvec) * s.dim + (ds.scal_field[0] ? 1 : 0) + (ds.scal_field[1] ? 1 : 0));
    // (proto2 optional: Save() always sets the field, so it is written even when 0)
    if (ds.step_field) m += 1 + varint_len(step_of(s));
    return m;
  };
  size_t opt_size = 0;
  for (const SegLayout& s : segs) {
    const DumpSpec ds = dump_spec(s.kind);
    if (!ds.oneof_field) continue;
    const size_t m = dump_len(s, ds);
    const size_t single = 1 + varint_len(m) + m;
    opt_size += 1 + varint_len(single) + single;
  }
  out.resize(9 + 5u * size_t(dim) + 1 + varint_len(opt_size) + opt_size + 1 + varint_len(ts));
  char* p = &out[0];
  *p++ = char(0x09);  // id: field 1, sfixed64
  const uint64_t u = uint64_t(id);
  memcpy(p, &u, 8);
  p += 8;
  for (int i = 0; i < dim; ++i) p = put_f32_raw(p, 0x15, row[i]);  // num: field 2, float, unpacked
  *p++ = char(0x1a);  // opt: field 3
  p = put_varint_raw(p, opt_size);
  for (const SegLayout& s : segs) {
    const DumpSpec ds = dump_spec(s.kind);
    if (!ds.oneof_field) continue;
    const size_t m = dump_len(s, ds);
    *p++ = char(0x0a);  // OptimizerDump.dump: field 1
    p = put_varint_raw(p, 1 + varint_len(m) + m);
    *p++ = char((ds.oneof_field << 3) | 2);
    p = put_varint_raw(p, m);
    // protobuf serialises in field-number order: vectors (and scalars) sorted by field number
    for (int f = 1; f <= 5; ++f) {
      for (int k = 0; k < ds.nvec; ++k)
        if (ds.vec_field[k] == f) {
          const float* v = row + s.st_off + k * s.dim;
          for (int i = 0; i < s.dim; ++i) p = put_f32_raw(p, uint8_t((f << 3) | 5), v[i]);
        }
      for (int k = 0; k < 2; ++k)
        if (ds.scal_field[k] == f)
          p = put_f32_raw(p, uint8_t((f << 3) | 5), row[s.st_off + ds.nvec * s.dim + k]);
      if (ds.step_field == f) {
        *p++ = char((f << 3) | 0);
        p = put_varint_raw(p, step_of(s));
      }
    }
  }
  *p++ = char(0x20);  // last_update_ts_sec: field 4, varint
  p = put_varint_raw(p, ts);
}

struct ProtoError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

inline bool skip_field(const uint8_t*& p, const uint8_t* end, uint32_t wt) {
  uint64_t v;
  switch (wt) {
    case 0: return get_varint(p, end, &v);
    case 1: if (end - p < 8) return false; p += 8; return true;
    case 2: if (!get_varint(p, end, &v) || uint64_t(end - p) < v) return false; p += v; return true;
    case 5: if (end - p < 4) return false; p += 4; return true;
    default: return false;
  }
}
inline float rd_f32(const uint8_t* p) {
  uint32_t u = uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24);
  float f;
  memcpy(&f, &u, 4);
  return f;
}
// repeated float field (packed or not) -> appends to dst (at most cap values kept)
inline bool read_floats(const uint8_t*& p, const uint8_t* end, uint32_t wt, float* dst, int cap,
                        int* n) {
  if (wt == 5) {
    if (end - p < 4) return false;
    if (*n < cap) dst[*n] = rd_f32(p);
    ++*n;
    p += 4;
    return true;
  }
  if (wt == 2) {
    uint64_t len;
    if (!get_varint(p, end, &len) || uint64_t(end - p) < len || (len & 3)) return false;
    for (uint64_t i = 0; i < len; i += 4) {
      if (*n < cap) dst[*n] = rd_f32(p + i);
      ++*n;
    }
    p += len;
    return true;
  }
  return false;
}

// Parses one EntryDump into the engine's row layout.  Missing optimizer state keeps the values
// already in `row` (the caller pre-fills Init()); missing timestamp -> 0
// (multi_hash_table_save_restore_ops.cc:384-386).
inline void decode_entry(const uint8_t* p, size_t n, const std::vector<SegLayout>& segs, int dim,
                         int64_t* id, float* row, uint32_t* ts) {
  const uint8_t* end = p + n;
  *id = 0;
  *ts = 0;
  int nnum = 0;
  size_t seg_i = 0;
  while (p < end) {
    uint64_t key;
    if (!get_varint(p, end, &key)) throw ProtoError("EntryDump: bad tag");
    const uint32_t field = uint32_t(key >> 3), wt = uint32_t(key & 7);
    if (field == 1 && wt == 1) {
      if (end - p < 8) throw ProtoError("EntryDump: truncated id");
      uint64_t u = 0;
      for (int i = 0; i < 8; ++i) u |= uint64_t(p[i]) << (8 * i);
      *id = int64_t(u);
      p += 8;
    } else if (field == 2 && (wt == 5 || wt == 2)) {
      if (!read_floats(p, end, wt, row, dim, &nnum)) throw ProtoError("EntryDump: bad num");
    } else if (field == 3 && wt == 2) {
      uint64_t len;
      if (!get_varint(p, end, &len) || uint64_t(end - p) < len) throw ProtoError("EntryDump: bad opt");
      const uint8_t* q = p;
      const uint8_t* qend = p + len;
      p = qend;
      while (q < qend) {  // OptimizerDump: repeated SingleOptimizerDump dump = 1
        uint64_t k2;
        if (!get_varint(q, qend, &k2)) throw ProtoError("OptimizerDump: bad tag");
        if ((k2 >> 3) != 1 || (k2 & 7) != 2) {
          if (!skip_field(q, qend, uint32_t(k2 & 7))) throw ProtoError("OptimizerDump: bad field");
          continue;
        }
        uint64_t l2;
        if (!get_varint(q, qend, &l2) || uint64_t(qend - q) < l2) throw ProtoError("OptimizerDump: bad dump");
        const uint8_t* r = q;
        const uint8_t* rend = q + l2;
        q = rend;
        while (seg_i < segs.size() && dump_spec(segs[seg_i].kind).oneof_field == 0) ++seg_i;
        if (seg_i >= segs.size()) continue;  // more dumps than segments: ignored
        const SegLayout& sg = segs[seg_i++];
        while (r < rend) {  // SingleOptimizerDump: oneof
          uint64_t k3;
          if (!get_varint(r, rend, &k3)) throw ProtoError("SingleOptimizerDump: bad tag");
          const uint32_t f3 = uint32_t(k3 >> 3), w3 = uint32_t(k3 & 7);
          if (w3 != 2) {
            if (!skip_field(r, rend, w3)) throw ProtoError("SingleOptimizerDump: bad field");
            continue;
          }
          uint64_t l3;
          if (!get_varint(r, rend, &l3) || uint64_t(rend - r) < l3) throw ProtoError("SingleOptimizerDump: bad len");
          const uint8_t* m = r;
          const uint8_t* mend = r + l3;
          r = mend;
          const DumpSpec ds = dump_spec(sg.kind);
          if (int(f3) == ds.oneof_field) {
            int cnt[3] = {0, 0, 0};
            while (m < mend) {
              uint64_t k4;
              if (!get_varint(m, mend, &k4)) throw ProtoError("optimizer dump: bad tag");
              const int f4 = int(k4 >> 3);
              bool done = false;
              for (int k = 0; k < ds.nvec && !done; ++k) {
                if (ds.vec_field[k] == f4) {
                  if (!read_floats(m, mend, uint32_t(k4 & 7), row + sg.st_off + k * sg.dim, sg.dim, &cnt[k]))
                    throw ProtoError("optimizer dump: bad state vector");
                  done = true;
                }
              }
              if (ds.step_field == f4 && (k4 & 7) == 0) {
                uint64_t a;
                if (!get_varint(m, mend, &a)) throw ProtoError("optimizer dump: bad step");
                memcpy(row + sg.st_off, &a, 8);
                done = true;
              }
              for (int k = 0; k < 2 && !done; ++k) {
                if (ds.scal_field[k] == f4 && (k4 & 7) == 5) {
                  if (mend - m < 4) throw ProtoError("optimizer dump: bad scalar");
                  row[sg.st_off + ds.nvec * sg.dim + k] = rd_f32(m);
                  m += 4;
                  done = true;
                }
              }
              if (!done && !skip_field(m, mend, uint32_t(k4 & 7))) throw ProtoError("optimizer dump: bad field");
            }
          }
          // sgd (field 2) carries nothing; a dump of another optimizer type is ignored
        }
      }
    } else if (field == 4 && wt == 0) {
      uint64_t v;
      if (!get_varint(p, end, &v)) throw ProtoError("EntryDump: bad timestamp");
      *ts = uint32_t(v);
    } else if (!skip_field(p, end, wt)) {
      throw ProtoError("EntryDump: bad field");
    }
  }
}

inline void encode_meta(std::string& out, const std::string& table_name, uint64_t num_entries) {
  out.clear();
  out.push_back(char(0x0a));
  put_varint(out, table_name.size());
  out += table_name;
  out.push_back(char(0x10));
  put_varint(out, num_entries);
}
inline void decode_meta(const uint8_t* p, size_t n, std::string* name, uint64_t* num_entries) {
  const uint8_t* end = p + n;
  name->clear();
  *num_entries = 0;
  while (p < end) {
    uint64_t key;
    if (!get_varint(p, end, &key)) throw ProtoError("MultiHashTableMetadata: bad tag");
    if ((key >> 3) == 1 && (key & 7) == 2) {
      uint64_t len;
      if (!get_varint(p, end, &len) || uint64_t(end - p) < len) throw ProtoError("metadata: bad name");
      name->assign(reinterpret_cast<const char*>(p), len);
      p += len;
    } else if ((key >> 3) == 2 && (key & 7) == 0) {
      if (!get_varint(p, end, num_entries)) throw ProtoError("metadata: bad count");
    } else if (!skip_field(p, end, uint32_t(key & 7))) {
      throw ProtoError("metadata: bad field");
    }
  }
}

// ------------------------------------------------------------------------------------ snappy
inline void snappy_compress_literals(const char* in, size_t n, std::string& out) {
  put_varint(out, n);
  size_t i = 0;
  while (i < n) {
    const size_t len = std::min<size_t>(n - i, 65536);
    const size_t l1 = len - 1;
    if (l1 < 60) {
      out.push_back(char(l1 << 2));
    } else if (l1 < 256) {
      out.push_back(char(60 << 2));
      out.push_back(char(l1));
    } else {
      out.push_back(char(61 << 2));
      out.push_back(char(l1 & 0xff));
      out.push_back(char(l1 >> 8));
    }
    out.append(in + i, len);
    i += len;
  }
}
inline bool snappy_uncompress(const uint8_t* p, size_t n, std::string& out) {
  const uint8_t* end = p + n;
  uint64_t ulen;
  if (!get_varint(p, end, &ulen)) return false;
  const size_t base = out.size();
  out.reserve(base + ulen);
  while (p < end) {
    const uint8_t tag = *p++;
    const uint32_t type = tag & 3u;
    if (type == 0) {
      uint64_t len = (tag >> 2);
      if (len >= 60) {
        const int nb = int(len) - 59;
        if (end - p < nb) return false;
        len = 0;
        for (int i = 0; i < nb; ++i) len |= uint64_t(p[i]) << (8 * i);
        p += nb;
      }
      len += 1;
      if (uint64_t(end - p) < len) return false;
      out.append(reinterpret_cast<const char*>(p), len);
      p += len;
    } else {
      uint64_t len, off;
      if (type == 1) {
        if (end - p < 1) return false;
        len = 4 + ((tag >> 2) & 7u);
        off = (uint64_t(tag >> 5) << 8) | p[0];
        p += 1;
      } else if (type == 2) {
        if (end - p < 2) return false;
        len = 1 + (tag >> 2);
        off = uint64_t(p[0]) | (uint64_t(p[1]) << 8);
        p += 2;
      } else {
        if (end - p < 4) return false;
        len = 1 + (tag >> 2);
        off = uint64_t(p[0]) | (uint64_t(p[1]) << 8) | (uint64_t(p[2]) << 16) | (uint64_t(p[3]) << 24);
        p += 4;
      }
      const size_t cur = out.size() - base;
      if (off == 0 || off > cur) return false;
      for (uint64_t i = 0; i < len; ++i) out.push_back(out[out.size() - off]);  // may overlap
    }
  }
  return out.size() - base == ulen;
}

// The same into memory the caller owns: exactly `ulen` bytes at dst (the block's own varint must say
// ulen).  Blocks of one file decode independently of each other — on any thread.
inline bool snappy_block_length(const uint8_t* p, size_t n, uint64_t* ulen) {
  const uint8_t* q = p;
  return get_varint(q, p + n, ulen);
}
inline bool snappy_uncompress_to(const uint8_t* p, size_t n, char* dst, size_t ulen) {
  const uint8_t* end = p + n;
  uint64_t said;
  if (!get_varint(p, end, &said) || said != ulen) return false;
  size_t cur = 0;
  while (p < end) {
    const uint8_t tag = *p++;
    const uint32_t type = tag & 3u;
    if (type == 0) {
      uint64_t len = (tag >> 2);
      if (len >= 60) {
        const int nb = int(len) - 59;
        if (end - p < nb) return false;
        len = 0;
        for (int i = 0; i < nb; ++i) len |= uint64_t(p[i]) << (8 * i);
        p += nb;
      }
      len += 1;
      if (uint64_t(end - p) < len || len > ulen - cur) return false;
      memcpy(dst + cur, p, len);
      p += len;
      cur += len;
    } else {
      uint64_t len, off;
      if (type == 1) {
        if (end - p < 1) return false;
        len = 4 + ((tag >> 2) & 7u);
        off = (uint64_t(tag >> 5) << 8) | p[0];
        p += 1;
      } else if (type == 2) {
        if (end - p < 2) return false;
        len = 1 + (tag >> 2);
        off = uint64_t(p[0]) | (uint64_t(p[1]) << 8);
        p += 2;
      } else {
        if (end - p < 4) return false;
        len = 1 + (tag >> 2);
        off = uint64_t(p[0]) | (uint64_t(p[1]) << 8) | (uint64_t(p[2]) << 16) | (uint64_t(p[3]) << 24);
        p += 4;
      }
      if (off == 0 || off > cur || len > ulen - cur) return false;
      if (off >= len) {
        memcpy(dst + cur, dst + cur - off, len);
      } else {
        for (uint64_t i = 0; i < len; ++i) dst[cur + i] = dst[cur + i - off];  // overlapping run
      }
      cur += len;
    }
  }
  return cur == ulen;
}

// Growable byte buffer whose tail is handed out UNINITIALISED (std::string::resize would zero the
// bytes a decoder is about to overwrite — one more pass over a gigabyte stream).
class ByteArena {
 public:
  ByteArena() = default;
  ByteArena(const ByteArena&) = delete;
  ByteArena& operator=(const ByteArena&) = delete;
  ~ByteArena() { free(p_); }
  const char* data() const { return p_; }
  char* data() { return p_; }
  size_t size() const { return n_; }
  void clear() { n_ = 0; }
  void truncate(size_t n) { n_ = std::min(n_, n); }
  char* grow(size_t add) {
    if (n_ + add > cap_) {
      // 2 MiB-aligned and advised for huge pages: a stretch of tens of megabytes is first touched
      // by the decoders, and 4 KiB faults of fresh anonymous memory cost more than the copy
      // (measured: 0.2 s against 0.04 s for 340 MB)
      constexpr size_t kHuge = size_t(2) << 20;
      size_t nc = std::max(n_ + add, cap_ + cap_ / 2);
      nc = (nc + kHuge - 1) & ~(kHuge - 1);
      void* q = nullptr;
      if (posix_memalign(&q, kHuge, nc) != 0 || !q) throw std::bad_alloc();
      (void)madvise(q, nc, MADV_HUGEPAGE);
      if (n_) memcpy(q, p_, n_);
      free(p_);
      p_ = static_cast<char*>(q);
      cap_ = nc;
    }
    char* r = p_ + n_;
    n_ += add;
    return r;
  }
  void append(const char* s, size_t n) {
    if (n) memcpy(grow(n), s, n);
  }

 private:
  char* p_ = nullptr;
  size_t n_ = 0, cap_ = 0;
};

// fn(lo, hi) over [0, n) on however many threads the caller has for it (empty: this thread)
using ParallelFor = std::function<void(size_t n, const std::function<void(size_t lo, size_t hi)>& fn)>;

// ------------------------------------------------------------------------------------ record files
constexpr size_t kSnappyBlock = 262144;  // RecordWriterOptions' snappy input buffer, TF 2.4

class RecordWriter {
 public:
  RecordWriter(const std::string& path, bool snappy) : snappy_(snappy), path_(path) {
    fd_ = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
    if (fd_ < 0) throw std::runtime_error("cannot create " + path);
  }
  ~RecordWriter() {   // (without close(): what was collected goes out, errors are not reported)
    if (fd_ < 0) return;
    try {
      flush_out();
    } catch (...) {
    }
    ::close(fd_);
  }
  void write(const std::string& rec) {
    char hdr[12];
    const uint64_t len = rec.size();
    for (int i = 0; i < 8; ++i) hdr[i] = char(len >> (8 * i));
    const uint32_t c1 = masked_crc(hdr, 8);
    for (int i = 0; i < 4; ++i) hdr[8 + i] = char(c1 >> (8 * i));
    const uint32_t c2 = masked_crc(rec.data(), rec.size());
    char ftr[4];
    for (int i = 0; i < 4; ++i) ftr[i] = char(c2 >> (8 * i));
    emit(hdr, 12);
    emit(rec.data(), rec.size());
    emit(ftr, 4);
  }
  // the framing of write(), appended to a buffer (any thread); write_framed() then passes whole
  // buffers of framed records through the file's byte stream
  static void frame(std::string& out, const std::string& rec) {
    char hdr[12];
    const uint64_t len = rec.size();
    for (int i = 0; i < 8; ++i) hdr[i] = char(len >> (8 * i));
    const uint32_t c1 = masked_crc(hdr, 8);
    for (int i = 0; i < 4; ++i) hdr[8 + i] = char(c1 >> (8 * i));
    const uint32_t c2 = masked_crc(rec.data(), rec.size());
    char ftr[4];
    for (int i = 0; i < 4; ++i) ftr[i] = char(c2 >> (8 * i));
    out.append(hdr, 12);
    out.append(rec);
    out.append(ftr, 4);
  }
  void write_framed(const std::string& bytes) { emit(bytes.data(), bytes.size()); }
  void close() {
    if (fd_ < 0) return;
    if (snappy_) flush_block();
    flush_out();
    const int fd = fd_;
    fd_ = -1;
    if (::close(fd) != 0) throw std::runtime_error("checkpoint write failed (close) " + path_);
  }

 private:
  // Output goes to the descriptor directly.  Small pieces (record files without compression, block
  // headers) collect in out_; the payload of whole snappy blocks is handed to writev() from where
  // it lies — the caller's buffer — so a gigabyte stream is copied once, into the page cache.
  void write_all(const struct iovec* iov_in, int cnt) {
    std::vector<struct iovec> iov(iov_in, iov_in + cnt);
    size_t at = 0;
    while (at < iov.size()) {
      const int batch = int(std::min<size_t>(iov.size() - at, 512));
      const ssize_t w = ::writev(fd_, iov.data() + at, batch);
      if (w < 0) {
        if (errno == EINTR) continue;
        throw std::runtime_error("checkpoint write failed: " + path_);
      }
      size_t left = size_t(w);
      while (left && at < iov.size()) {   // (a short write: resume inside the vector)
        if (left >= iov[at].iov_len) {
          left -= iov[at].iov_len;
          ++at;
        } else {
          iov[at].iov_base = static_cast<char*>(iov[at].iov_base) + left;
          iov[at].iov_len -= left;
          left = 0;
        }
      }
      while (at < iov.size() && iov[at].iov_len == 0) ++at;
    }
  }
  void flush_out() {
    if (out_.empty()) return;
    struct iovec v;
    v.iov_base = &out_[0];
    v.iov_len = out_.size();
    write_all(&v, 1);
    out_.clear();
  }
  void emit(const char* p, size_t n) {
    if (!snappy_) {
      if (n >= (size_t(1) << 16)) {   // (large pieces skip the collecting buffer)
        flush_out();
        struct iovec v;
        v.iov_base = const_cast<char*>(p);
        v.iov_len = n;
        write_all(&v, 1);
        return;
      }
      out_.append(p, n);
      if (out_.size() >= (size_t(1) << 20)) flush_out();
      return;
    }
    // top up a partly filled block first
    if (!buf_.empty()) {
      const size_t take = std::min(n, kSnappyBlock - buf_.size());
      buf_.append(p, take);
      p += take;
      n -= take;
      if (buf_.size() == kSnappyBlock) flush_block();
    }
    // whole blocks straight from the caller's memory
    if (n >= kSnappyBlock) {
      const size_t blocks = n / kSnappyBlock;
      write_blocks(p, blocks);
      p += blocks * kSnappyBlock;
      n -= blocks * kSnappyBlock;
    }
    if (n) buf_.append(p, n);
  }
  static size_t literal_tag(size_t len, char* t) {
    const size_t l1 = len - 1;
    if (l1 < 60) {
      t[0] = char(l1 << 2);
      return 1;
    }
    if (l1 < 256) {
      t[0] = char(60 << 2);
      t[1] = char(l1);
      return 2;
    }
    t[0] = char(61 << 2);
    t[1] = char(l1 & 0xff);
    t[2] = char(l1 >> 8);
    return 3;
  }
  // [4-byte big-endian packed length | varint n | literal elements of <= 64 KiB] for n bytes at p,
  // as iovecs: the small parts are appended to `meta` (reserved by the caller: no reallocation)
  void block_iov(const char* p, size_t n, std::string& meta, std::vector<struct iovec>& iov) {
    char t[3];
    size_t cl = size_t(varint_len(n));
    for (size_t i = 0; i < n; i += 65536) cl += literal_tag(std::min<size_t>(n - i, 65536), t) + std::min<size_t>(n - i, 65536);
    char head[4 + 10];
    head[0] = char(cl >> 24);
    head[1] = char(cl >> 16);
    head[2] = char(cl >> 8);
    head[3] = char(cl);
    char* e = put_varint_raw(head + 4, n);
    auto small = [&](const char* q, size_t m) {
      struct iovec v;
      v.iov_base = &meta[0] + meta.size();
      v.iov_len = m;
      meta.append(q, m);
      iov.push_back(v);
    };
    small(head, size_t(e - head));
    for (size_t i = 0; i < n; i += 65536) {
      const size_t len = std::min<size_t>(n - i, 65536);
      small(t, literal_tag(len, t));
      struct iovec v;
      v.iov_base = const_cast<char*>(p + i);
      v.iov_len = len;
      iov.push_back(v);
    }
  }
  void write_blocks(const char* p, size_t blocks) {
    flush_out();
    std::string meta;
    std::vector<struct iovec> iov;
    const size_t kPer = 32;   // blocks per writev round (8 MiB)
    for (size_t b0 = 0; b0 < blocks; b0 += kPer) {
      const size_t nb = std::min(kPer, blocks - b0);
      meta.clear();
      meta.reserve(nb * 32);
      iov.clear();
      for (size_t b = 0; b < nb; ++b) block_iov(p + (b0 + b) * kSnappyBlock, kSnappyBlock, meta, iov);
      write_all(iov.data(), int(iov.size()));
    }
  }
  // one raw snappy block made of literal elements (snappy_compress_literals' bytes) from buf_
  void flush_block() {
    if (buf_.empty()) return;
    flush_out();
    std::string meta;
    meta.reserve(64);
    std::vector<struct iovec> iov;
    block_iov(buf_.data(), buf_.size(), meta, iov);
    write_all(iov.data(), int(iov.size()));
    buf_.clear();
  }
  int fd_ = -1;
  bool snappy_;
  std::string path_, buf_, out_;
};

class RecordReader {
 public:
  RecordReader(const std::string& path, bool snappy) : snappy_(snappy), path_(path) {
    if (snappy) {
      // the block stream is decoded straight out of a read-only mapping: no copy into a staging
      // buffer in front of the decoder (falls back to stdio when the file cannot be mapped)
      const int fd = open(path.c_str(), O_RDONLY);
      if (fd < 0) throw std::runtime_error("cannot open " + path);
      struct stat sb;
      if (fstat(fd, &sb) == 0 && sb.st_size > 0) {
        void* m = mmap(nullptr, size_t(sb.st_size), PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) {
          map_ = static_cast<const uint8_t*>(m);
          map_size_ = size_t(sb.st_size);
          (void)madvise(m, map_size_, MADV_SEQUENTIAL);
        }
      }
      close(fd);
      if (map_) return;
    }
    fp_ = fopen(path.c_str(), "rb");
    if (!fp_) throw std::runtime_error("cannot open " + path);
  }
  ~RecordReader() {
    if (fp_) fclose(fp_);
    if (map_) munmap(const_cast<uint8_t*>(map_), map_size_);
  }
  // false at a clean end of file; throws on corruption (errors::DataLoss in the reference)
  bool read(std::string* rec) {
    char hdr[12];
    const size_t got = fetch(hdr, 12);
    if (got == 0) return false;
    if (got != 12) throw std::runtime_error("truncated record header in " + path_);
    uint64_t len = 0;
    for (int i = 0; i < 8; ++i) len |= uint64_t(uint8_t(hdr[i])) << (8 * i);
    uint32_t c1 = 0;
    for (int i = 0; i < 4; ++i) c1 |= uint32_t(uint8_t(hdr[8 + i])) << (8 * i);
    if (c1 != masked_crc(hdr, 8)) throw std::runtime_error("corrupted record length in " + path_);
    rec->resize(len);
    if (fetch(&(*rec)[0], len) != len) throw std::runtime_error("truncated record in " + path_);
    char ftr[4];
    if (fetch(ftr, 4) != 4) throw std::runtime_error("truncated record in " + path_);
    uint32_t c2 = 0;
    for (int i = 0; i < 4; ++i) c2 |= uint32_t(uint8_t(ftr[i])) << (8 * i);
    if (c2 != masked_crc(rec->data(), rec->size()))
      throw std::runtime_error("corrupted record data in " + path_);
    return true;
  }
  // read() without the data checksum's verification: *crc receives the stored (masked) value for
  // verify(), which any thread may run
  bool read_raw(std::string* rec, uint32_t* crc) {
    char hdr[12];
    const size_t got = fetch(hdr, 12);
    if (got == 0) return false;
    if (got != 12) throw std::runtime_error("truncated record header in " + path_);
    uint64_t len = 0;
    for (int i = 0; i < 8; ++i) len |= uint64_t(uint8_t(hdr[i])) << (8 * i);
    uint32_t c1 = 0;
    for (int i = 0; i < 4; ++i) c1 |= uint32_t(uint8_t(hdr[8 + i])) << (8 * i);
    if (c1 != masked_crc(hdr, 8)) throw std::runtime_error("corrupted record length in " + path_);
    rec->resize(len);
    if (fetch(&(*rec)[0], len) != len) throw std::runtime_error("truncated record in " + path_);
    char ftr[4];
    if (fetch(ftr, 4) != 4) throw std::runtime_error("truncated record in " + path_);
    uint32_t c2 = 0;
    for (int i = 0; i < 4; ++i) c2 |= uint32_t(uint8_t(ftr[i])) << (8 * i);
    *crc = c2;
    return true;
  }
  void verify(const std::string& rec, uint32_t crc) const {
    if (crc != masked_crc(rec.data(), rec.size()))
      throw std::runtime_error("corrupted record data in " + path_);
  }
  void verify(const char* data, size_t n, uint32_t crc) const {
    if (crc != masked_crc(data, n)) throw std::runtime_error("corrupted record data in " + path_);
  }

  // Batch form: the next stretch of the file's byte stream (about target_bytes of it) lands in
  // `arena` as ONE contiguous buffer and `refs` lists the whole records inside it (data offset,
  // length, stored data checksum — unverified: verify() on any thread).  A record cut by the end of
  // the stretch is carried into the next call.  No per-record allocation or copy: the decoders
  // read the arena in place.  false at a clean end of file.
  struct RecRef {
    size_t off;
    uint32_t len;
    uint32_t crc;
  };
  bool read_batch(ByteArena& arena, size_t target_bytes, std::vector<RecRef>& refs,
                  const ParallelFor& par = ParallelFor()) {
    refs.clear();
    arena.clear();
    arena.append(carry_.data(), carry_.size());
    carry_.clear();
    if (pos_ < buf_.size()) arena.append(buf_.data() + pos_, buf_.size() - pos_);  // (after read() calls)
    buf_.clear();
    pos_ = 0;
    bool eof = false;
    size_t pos = 0;
    for (;;) {
      if (!eof && arena.size() < target_bytes) eof = !append_stretch(arena, target_bytes - arena.size(), par);
      while (arena.size() - pos >= 12) {
        const char* h = arena.data() + pos;
        uint64_t len = 0;
        for (int i = 0; i < 8; ++i) len |= uint64_t(uint8_t(h[i])) << (8 * i);
        uint32_t c1 = 0;
        for (int i = 0; i < 4; ++i) c1 |= uint32_t(uint8_t(h[8 + i])) << (8 * i);
        if (c1 != masked_crc(h, 8)) throw std::runtime_error("corrupted record length in " + path_);
        if (len > 0xffffffffull) throw std::runtime_error("oversized record in " + path_);
        if (arena.size() - pos < 16 + len) break;
        uint32_t c2 = 0;
        for (int i = 0; i < 4; ++i) c2 |= uint32_t(uint8_t(h[12 + len + i])) << (8 * i);
        refs.push_back(RecRef{pos + 12, uint32_t(len), c2});
        pos += 16 + len;
      }
      if (!refs.empty() || eof) break;
      target_bytes = arena.size() + kSnappyBlock;  // (one record longer than the stretch)
    }
    if (pos < arena.size()) {
      if (eof) throw std::runtime_error("truncated record in " + path_);
      carry_.assign(arena.data() + pos, arena.size() - pos);
    }
    return !refs.empty();
  }

 private:
  size_t fetch(char* dst, size_t n) {
    if (!snappy_) return fread(dst, 1, n, fp_);
    size_t done = 0;
    while (done < n) {
      if (pos_ == buf_.size()) {
        if (!next_block()) break;
      }
      const size_t take = std::min(n - done, buf_.size() - pos_);
      memcpy(dst + done, buf_.data() + pos_, take);
      pos_ += take;
      done += take;
    }
    return done;
  }
  // At least `want` more bytes of the byte stream (less at the end of the file) appended to `out`;
  // false when nothing was left.  Snappy files: the blocks that cover the stretch are located first
  // (their headers say how long each is, packed and unpacked), then unpacked side by side — every
  // block straight to its place in `out`, on the caller's threads (`par`).
  bool append_stretch(ByteArena& out, size_t want, const ParallelFor& par) {
    if (!snappy_) {
      want = std::max(want, size_t(1) << 20);
      const size_t base = out.size();
      char* dst = out.grow(want);
      const size_t got = fread(dst, 1, want, fp_);
      out.truncate(base + got);
      return got != 0;
    }
    struct Blk {
      const uint8_t* p;
      uint32_t cl;
      size_t ulen, off;
    };
    std::vector<Blk> blks;
    size_t total = 0;
    bool got_block = false;   // (a stretch of empty blocks is not the end of the file)
    while (total < want) {
      Blk b;
      if (!next_compressed(&b.p, &b.cl)) break;
      got_block = true;
      uint64_t ulen;
      if (!snappy_block_length(b.p, b.cl, &ulen) || ulen > (uint64_t(1) << 32))
        throw std::runtime_error("corrupted snappy block in " + path_);
      b.ulen = size_t(ulen);
      b.off = total;
      total += b.ulen;
      if (!map_) {   // (stdio fallback: the block lives in comp_ until the next one is read)
        if (!snappy_uncompress_to(b.p, b.cl, out.grow(b.ulen), b.ulen))
          throw std::runtime_error("corrupted snappy block in " + path_);
        continue;
      }
      blks.push_back(b);
    }
    if (!got_block) return false;
    if (blks.empty()) return true;
    char* dst = out.grow(total);
    std::vector<uint8_t> bad(blks.size(), 0);
    auto body = [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; ++i)
        bad[i] = snappy_uncompress_to(blks[i].p, blks[i].cl, dst + blks[i].off, blks[i].ulen) ? 0 : 1;
    };
    if (par && blks.size() > 1) par(blks.size(), body);
    else body(0, blks.size());
    for (uint8_t b : bad)
      if (b) throw std::runtime_error("corrupted snappy block in " + path_);
    return true;
  }
  // the next [4-byte big-endian length | block] of the file: *blk points into the mapping (or comp_)
  bool next_compressed(const uint8_t** blk, uint32_t* len) {
    uint8_t be[4];
    if (map_) {
      if (map_pos_ == map_size_) return false;
      if (map_size_ - map_pos_ < 4) throw std::runtime_error("truncated snappy block header in " + path_);
      memcpy(be, map_ + map_pos_, 4);
    } else {
      const size_t got = fread(be, 1, 4, fp_);
      if (got == 0) return false;
      if (got != 4) throw std::runtime_error("truncated snappy block header in " + path_);
    }
    const uint32_t cl = (uint32_t(be[0]) << 24) | (uint32_t(be[1]) << 16) | (uint32_t(be[2]) << 8) | be[3];
    if (map_) {
      if (map_size_ - map_pos_ - 4 < cl) throw std::runtime_error("truncated snappy block in " + path_);
      *blk = map_ + map_pos_ + 4;
      map_pos_ += size_t(4) + cl;
    } else {
      comp_.resize(cl);
      if (fread(&comp_[0], 1, cl, fp_) != cl) throw std::runtime_error("truncated snappy block in " + path_);
      *blk = reinterpret_cast<const uint8_t*>(comp_.data());
    }
    *len = cl;
    return true;
  }
  bool next_block() {
    const uint8_t* blk;
    uint32_t cl;
    if (!next_compressed(&blk, &cl)) return false;
    buf_.clear();
    pos_ = 0;
    if (!snappy_uncompress(blk, cl, buf_)) throw std::runtime_error("corrupted snappy block in " + path_);
    return true;
  }
  FILE* fp_ = nullptr;
  const uint8_t* map_ = nullptr;   // snappy files: the whole file, read-only
  size_t map_size_ = 0, map_pos_ = 0;
  bool snappy_;
  std::string path_, buf_, comp_, carry_;
  size_t pos_ = 0;
};

// Three stages over chunks 0 .. n-1 — the save path's scan | encode | write, the restore path's
// decode | upsert.  Every stage takes the chunks in order on a thread of its own (stage A on the
// caller's); neighbouring stages hand over through TWO buffer sets (slot = chunk & 1), so
//     A(c) starts after B(c-2),   B(c) after A(c) and C(c-2),   C(c) after B(c),
// and while one chunk is written the next is encoded and the one after it scanned.  Stage A may
// return bool: false says "the stream ended, there is no chunk c" (n is then only an upper bound).
// The first exception of any stage stops the others at their next chunk and is rethrown here,
// after both helper threads have ended (nothing keeps running into the caller's buffers).
template <class FA, class FB, class FC>
inline void run_pipeline3(size_t n, FA&& stage_a, FB&& stage_b, FC&& stage_c) {
  if (n == 0) return;
  std::mutex mu;
  std::condition_variable cv;
  size_t done_a = 0, done_b = 0, done_c = 0;   // chunks each stage has finished
  size_t total = n;                            // (lowered by a stage A that reports the end)
  std::exception_ptr err;
  auto fail = [&] {
    std::lock_guard<std::mutex> g(mu);
    if (!err) err = std::current_exception();
    cv.notify_all();
  };
  // waits until chunk c may be taken; false: there is no chunk c, or another stage failed
  auto wait_for = [&](size_t c, const std::function<bool()>& ready) {
    std::unique_lock<std::mutex> g(mu);
    cv.wait(g, [&] { return err || c >= total || ready(); });
    return !err && c < total;
  };
  auto finished = [&](size_t& counter) {
    std::lock_guard<std::mutex> g(mu);
    ++counter;
    cv.notify_all();
  };
  auto call_a = [&](size_t c, int slot) -> bool {
    if constexpr (std::is_void<decltype(stage_a(c, slot))>::value) {
      stage_a(c, slot);
      return true;
    } else {
      return stage_a(c, slot);
    }
  };
  auto run_b = [&] {
    try {
      for (size_t c = 0;; ++c) {
        if (!wait_for(c, [&] { return done_a > c && done_c + 2 > c; })) return;
        stage_b(c, int(c & 1));
        finished(done_b);
      }
    } catch (...) {
      fail();
    }
  };
  auto run_c = [&] {
    try {
      for (size_t c = 0;; ++c) {
        if (!wait_for(c, [&] { return done_b > c; })) return;
        stage_c(c, int(c & 1));
        finished(done_c);
      }
    } catch (...) {
      fail();
    }
  };
  std::thread tb, tc;
  try {
    tb = std::thread(run_b);
    tc = std::thread(run_c);
    for (size_t c = 0;; ++c) {
      if (!wait_for(c, [&] { return done_b + 2 > c; })) break;
      if (!call_a(c, int(c & 1))) {
        std::lock_guard<std::mutex> g(mu);
        total = c;
        cv.notify_all();
        break;
      }
      finished(done_a);
    }
  } catch (...) {   // (a stage A failure, or a helper thread that could not be started)
    fail();
  }
  if (tb.joinable()) tb.join();
  if (tc.joinable()) tc.join();
  if (err) std::rethrow_exception(err);
}

inline std::string shard_name(const std::string& base, const char* infix, int shard, int total) {
  char buf[64];
  snprintf(buf, sizeof(buf), "%s-%05d-of-%05d", infix, shard, total);
  return base + buf;
}

}  // namespace ckpt
}  // namespace mhte
#endif  // MHTE_CKPT_H_
