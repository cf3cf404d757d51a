// Reader for the serialized configuration protos the reference hands to its create ops, so that a
// TF shim can pass the `config` string input of CreateMonolithMultiHashTable
// (RT/ops/multi_hash_table_op.cc:44-112) and of the hash filter op straight through:
//   MultiEmbeddingHashTableConfig   RT/hash_table/embedding_hash_table.proto:93-96
//   EmbeddingHashTableConfig        :70-91    EntryConfig / Segment :23-43
//   SlotExpireTimeConfig            :54-64    SlotOccurrenceThresholdConfig :100-110
//   OptimizerConfig and the per-optimizer configs   RT/hash_table/optimizer/optimizer.proto
//   InitializerConfig               RT/hash_table/initializer/initializer_config.proto
// Hand-written protobuf wire decoding (no protobuf runtime in the engine); field numbers and
// defaults restated from the .proto files.  Host-only, included by mhte.hip.
#ifndef MHTE_PROTO_CONFIG_H_
#define MHTE_PROTO_CONFIG_H_

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/monolith_amd_hash_table.h"
#include "mhte_ckpt.h"

namespace mhte {
namespace pcfg {

struct Field {
  uint32_t num, wt;
  uint64_t v;             // varint / fixed value
  const uint8_t* p;       // length-delimited payload
  size_t n;
};

// iterates the fields of one message
struct Msg {
  const uint8_t* p;
  const uint8_t* end;
  Msg(const uint8_t* b, size_t n) : p(b), end(b + n) {}
  bool next(Field* f) {
    if (p >= end) return false;
    uint64_t key;
    if (!ckpt::get_varint(p, end, &key)) throw ckpt::ProtoError("config proto: bad tag");
    f->num = uint32_t(key >> 3);
    f->wt = uint32_t(key & 7);
    f->v = 0;
    f->p = nullptr;
    f->n = 0;
    switch (f->wt) {
      case 0:
        if (!ckpt::get_varint(p, end, &f->v)) throw ckpt::ProtoError("config proto: bad varint");
        break;
      case 1:
        if (end - p < 8) throw ckpt::ProtoError("config proto: short fixed64");
        memcpy(&f->v, p, 8);
        p += 8;
        break;
      case 5: {
        if (end - p < 4) throw ckpt::ProtoError("config proto: short fixed32");
        uint32_t w;
        memcpy(&w, p, 4);
        f->v = w;
        p += 4;
        break;
      }
      case 2: {
        uint64_t len;
        if (!ckpt::get_varint(p, end, &len) || uint64_t(end - p) < len)
          throw ckpt::ProtoError("config proto: bad length");
        f->p = p;
        f->n = size_t(len);
        p += len;
        break;
      }
      default: throw ckpt::ProtoError("config proto: unsupported wire type");
    }
    return true;
  }
};

inline float f32_of(const Field& f) {
  if (f.wt != 5) throw ckpt::ProtoError("config proto: float field with wrong wire type");
  const uint32_t w = uint32_t(f.v);
  float x;
  memcpy(&x, &w, 4);
  return x;
}

struct Segment {
  mhte_segment_config c;
  float learning_rate;   // the config's own (the ops take the live value as an input)
};
struct TableCfg {
  std::string name;
  std::vector<Segment> segs;
  uint64_t initial_capacity = 1;
  int64_t default_expire = 36500;
  std::vector<int64_t> expire_slots;
  std::vector<int32_t> expire_days;
  bool enable_eviction = false;
  int32_t evict_every_n_hours = 240;
};

// one optimizer config message: {field number -> param slot}, defaults per optimizer.proto
struct OptSpec {
  int opt;
  float lr_default;
  int lr_field;
  struct { int field, slot; float dflt; } par[6];
  int npar;
};

inline const OptSpec* opt_spec(uint32_t oneof_field) {
  static const OptSpec specs[] = {
      // adagrad = 1: {initial_accumulator_value(3) 0.1, weight_decay_factor(5) 0}
      {MHTE_OPT_ADAGRAD, 0.001f, 2, {{3, 0, 0.1f}, {5, 1, 0.f}}, 2},
      // sgd = 2
      {MHTE_OPT_SGD, 0.01f, 2, {}, 0},
      // ftrl = 3: {initial_accumulator_value(4) 0.1, beta(3) 0, l1(5) 0, l2(6) 0}
      {MHTE_OPT_FTRL, 0.01f, 2, {{4, 0, 0.1f}, {3, 1, 0.f}, {5, 2, 0.f}, {6, 3, 0.f}}, 4},
      // adadelta = 6: {averaging_ratio(4) 0.9, epsilon(5) 0.01, weight_decay_factor(3) 0}
      {MHTE_OPT_ADADELTA, 0.01f, 2, {{4, 0, 0.9f}, {5, 1, 0.01f}, {3, 2, 0.f}}, 3},
      // adam = 7: {beta1(3) .9, beta2(4) .99, epsilon(8) .01, weight_decay_factor(6) 0, use_nesterov(7)}
      {MHTE_OPT_ADAM, 0.01f, 2, {{3, 0, 0.9f}, {4, 1, 0.99f}, {8, 2, 0.01f}, {6, 3, 0.f}, {7, 4, 0.f}}, 5},
      // amsgrad = 8: same numbering
      {MHTE_OPT_AMSGRAD, 0.01f, 2, {{3, 0, 0.9f}, {4, 1, 0.99f}, {8, 2, 0.01f}, {6, 3, 0.f}, {7, 4, 0.f}}, 5},
      // momentum = 9: {momentum(5) .9, weight_decay_factor(3) 0, use_nesterov(4)}
      {MHTE_OPT_MOMENTUM, 0.01f, 2, {{5, 0, 0.9f}, {3, 1, 0.f}, {4, 2, 0.f}}, 3},
      // moving_average = 10: {momentum(2) .9}; no learning rate
      {MHTE_OPT_MOVING_AVERAGE, 0.f, 0, {{2, 0, 0.9f}}, 1},
      // rmsprop = 11: {momentum(4) .9, weight_decay_factor(3) 0, learning_rate(2) -> slot 2}
      {MHTE_OPT_RMSPROP, 0.01f, 2, {{4, 0, 0.9f}, {3, 1, 0.f}, {2, 2, 0.01f}}, 3},
      // rmspropv2 = 12
      {MHTE_OPT_RMSPROPV2, 0.01f, 2, {{4, 0, 0.9f}, {3, 1, 0.f}, {2, 2, 0.01f}}, 3},
      // batch_softmax = 15
      {MHTE_OPT_BATCH_SOFTMAX, 0.1f, 2, {}, 0},
      // group_adagrad = 16: {initial_accumulator_value(4) .1, beta(3) 0, l2(5) 0, weight_decay_factor(6) 0}
      {MHTE_OPT_GROUP_ADAGRAD, 0.01f, 2, {{4, 0, 0.1f}, {3, 1, 0.f}, {5, 2, 0.f}, {6, 3, 0.f}}, 4},
  };
  switch (oneof_field) {
    case 1: return &specs[0];
    case 2: return &specs[1];
    case 3: return &specs[2];
    case 6: return &specs[3];
    case 7: return &specs[4];
    case 8: return &specs[5];
    case 9: return &specs[6];
    case 10: return &specs[7];
    case 11: return &specs[8];
    case 12: return &specs[9];
    case 15: return &specs[10];
    case 16: return &specs[11];
    default: return nullptr;
  }
}

inline void parse_optimizer(const uint8_t* b, size_t n, Segment* s) {
  Msg m(b, n);
  Field f;
  bool have = false, sr16 = false;
  while (m.next(&f)) {
    if (f.num == 4 && f.wt == 0) {  // stochastic_rounding_float16 (optimizer.proto:228)
      sr16 = f.v != 0;
      continue;
    }
    if (f.wt != 2) continue;
    const OptSpec* sp = opt_spec(f.num);
    if (!sp)
      throw ckpt::ProtoError("config: optimizer (OptimizerConfig field " + std::to_string(f.num) +
                             ") is not implemented (the reference's own factory throws for "
                             "dynamic_wd_adagrad / group_ftrl; dc is out of scope)");
    have = true;
    s->c.opt_type = sp->opt;
    for (int k = 0; k < 8; ++k) s->c.opt_params[k] = 0.f;
    for (int k = 0; k < sp->npar; ++k) s->c.opt_params[sp->par[k].slot] = sp->par[k].dflt;
    s->learning_rate = sp->lr_default;
    Msg o(f.p, f.n);
    Field g;
    while (o.next(&g)) {
      if (g.num == 1 && g.wt == 0 && s->c.dim_size == 0) s->c.dim_size = int32_t(g.v);  // (Segment.dim_size wins)
      if (sp->lr_field && int(g.num) == sp->lr_field && g.wt == 5) s->learning_rate = f32_of(g);
      for (int k = 0; k < sp->npar; ++k)
        if (int(g.num) == sp->par[k].field)
          s->c.opt_params[sp->par[k].slot] = g.wt == 5 ? f32_of(g) : float(g.v);  // (bool fields: varint)
    }
  }
  if (!have) throw ckpt::ProtoError("config: segment without an optimizer");
  // A table created from the serialized config IS the reference's op: its Adagrad is the arithmetic the
  // reference's own build runs (.bazelrc:63-68 -mavx -mfma, optimizer/BUILD -D_ENABLE_AVX: AdagradOptimize
  // dispatches to Avx256AdagradOptimize, avx_utils.h:96-119,238-245 — fused multiply-adds and, with a
  // weight decay, the raw gradient in the weight step inside a block of 8), opt_params[2] = 1
  // (SegDesc.p[2], adagrad_step_avx).  MHTE_ADAGRAD_SCALAR=1 keeps the scalar loop of avx_utils.h:29-38
  // (what the reference's unit tests pin, and the default of entry.AdagradOptimizer).
  if ((s->c.opt_type & 0xff) == MHTE_OPT_ADAGRAD) {
    const char* e = getenv("MHTE_ADAGRAD_SCALAR");
    s->c.opt_params[2] = (e && atoi(e) != 0) ? 0.f : 1.f;
  }
  if (sr16) s->c.opt_type |= MHTE_OPT_FLAG_STOCHASTIC_ROUNDING_FP16;
}

inline void parse_initializer(const uint8_t* b, size_t n, Segment* s) {
  Msg m(b, n);
  Field f;
  s->c.init_type = MHTE_INIT_ZEROS;
  while (m.next(&f)) {
    if (f.wt != 2) continue;
    Msg o(f.p, f.n);
    Field g;
    switch (f.num) {
      case 1: s->c.init_type = MHTE_INIT_ZEROS; break;
      case 3: s->c.init_type = MHTE_INIT_ONES; break;
      case 15:
        s->c.init_type = MHTE_INIT_CONSTANT;
        while (o.next(&g))
          if (g.num == 2) s->c.init_value = f32_of(g);
        break;
      case 2:
        s->c.init_type = MHTE_INIT_RANDOM_UNIFORM;
        s->c.init_value = -0.05f;
        s->c.init_value2 = 0.05f;
        while (o.next(&g)) {
          if (g.num == 2) s->c.init_value = f32_of(g);
          if (g.num == 3) s->c.init_value2 = f32_of(g);
        }
        break;
      default: throw ckpt::ProtoError("config: unknown initializer");
    }
  }
}

inline void parse_table(const uint8_t* b, size_t n, TableCfg* t) {
  Msg m(b, n);
  Field f;
  bool skip_zero = false;
  while (m.next(&f)) {
    switch (f.num) {
      case 1: {  // entry_config
        Msg e(f.p, f.n);
        Field g;
        while (e.next(&g)) {
          if (g.num == 2 && g.wt == 0 && g.v == 2)
            throw ckpt::ProtoError("config: SERVING entries (compressed rows) are out of scope");
          if (g.num != 1 || g.wt != 2) continue;
          Segment s{};
          s.learning_rate = 0.f;
          Msg sm(g.p, g.n);
          Field h;
          bool has_opt = false;
          while (sm.next(&h)) {
            if (h.num == 1 && h.wt == 2) parse_initializer(h.p, h.n, &s);
            else if (h.num == 2 && h.wt == 2) { parse_optimizer(h.p, h.n, &s); has_opt = true; }
            else if (h.num == 7 && h.wt == 0) s.c.dim_size = int32_t(h.v);
          }
          if (!has_opt) throw ckpt::ProtoError("config: segment without opt_config");
          t->segs.push_back(s);
        }
        break;
      }
      case 2: t->initial_capacity = f.v; break;
      case 3: {  // slot_expire_time_config
        Msg e(f.p, f.n);
        Field g;
        while (e.next(&g)) {
          if (g.num == 2 && g.wt == 0) t->default_expire = int64_t(g.v);
          if (g.num == 1 && g.wt == 2) {
            Msg se(g.p, g.n);
            Field h;
            int64_t slot = 0;
            int32_t days = 0;
            while (se.next(&h)) {
              if (h.num == 1) slot = int64_t(h.v);
              if (h.num == 2) days = int32_t(h.v);
            }
            t->expire_slots.push_back(slot);
            t->expire_days.push_back(days);
          }
        }
        break;
      }
      case 7: t->enable_eviction = f.v != 0; break;
      case 8: t->evict_every_n_hours = int32_t(f.v); break;
      case 10: skip_zero = f.wt == 0 && f.v != 0; break;   // skip_zero_embedding (embedding_hash_table.proto:90)
      default: break;  // cuckoo (5), entry_type (6: PACKED / RAW are the same here)
    }
  }
  // Assign of an all-zero row erases the key, restore skips zero rows (cuckoo_embedding_hash_table.cc:192-197,
  // 304-310) — serving only: the reference's factory refuses the flag on any other entry type with this
  // message (embedding_hash_table_factory.cc:30-34), and SERVING entries were refused above.
  if (skip_zero)
    throw ckpt::ProtoError("config: skip_zero_embedding: Only EntryConfig_EntryType_SERVING supports skip_zero_embedding!");
  if (t->segs.empty()) throw ckpt::ProtoError("config: table without segments");
}

inline std::vector<TableCfg> parse_multi(const void* data, size_t n) {
  std::vector<std::string> names;
  std::vector<TableCfg> tables;
  Msg m(static_cast<const uint8_t*>(data), n);
  Field f;
  while (m.next(&f)) {
    if (f.num == 1 && f.wt == 2) names.emplace_back(reinterpret_cast<const char*>(f.p), f.n);
    if (f.num == 2 && f.wt == 2) {
      TableCfg t;
      parse_table(f.p, f.n, &t);
      tables.push_back(std::move(t));
    }
  }
  if (names.size() != tables.size())  // multi_hash_table_op.cc:50-53
    throw ckpt::ProtoError("config: names and configs differ in size (" + std::to_string(names.size()) +
                           " vs " + std::to_string(tables.size()) + ")");
  for (size_t i = 0; i < names.size(); ++i) tables[i].name = names[i];
  return tables;
}

// SlotOccurrenceThresholdConfig (the hash filter op's `config` attr)
inline void parse_occurrence(const void* data, size_t n, int32_t* dflt, std::vector<int64_t>* slots,
                             std::vector<int32_t>* thr) {
  *dflt = 0;
  Msg m(static_cast<const uint8_t*>(data), n);
  Field f;
  while (m.next(&f)) {
    if (f.num == 2 && f.wt == 0) *dflt = int32_t(f.v);
    if (f.num == 1 && f.wt == 2) {
      Msg e(f.p, f.n);
      Field g;
      int64_t slot = 0;
      int32_t t = 0;
      while (e.next(&g)) {
        if (g.num == 1) slot = int64_t(g.v);
        if (g.num == 2) t = int32_t(g.v);
      }
      slots->push_back(slot);
      thr->push_back(t);
    }
  }
}

}  // namespace pcfg
}  // namespace mhte
#endif  // MHTE_PROTO_CONFIG_H_
