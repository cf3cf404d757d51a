// Shared host/device core of the MI355X embedding-table engine: table geometry, the fixed hash,
// the serial cuckoo displacement used by the slow path, and the per-row optimizer arithmetic.
//
// Reference behaviour being matched (paths relative to
// /root/reference/monolith/native_training/runtime/):
//   hash_table/cuckoohash/cuckoohash_map.hpp:860-888   partial_key / index_hash / alt_index
//   hash_table/cuckoohash/cuckoohash_map.hpp:1398-1418 slot choice (LAST empty slot of b1, else b2)
//   hash_table/cuckoohash/cuckoohash_map.hpp:1445-1762 BFS displacement, MAX_BFS_PATH_LEN = 5
//   hash_table/cuckoohash/cuckoohash_map.hpp:1850-1894 bucket split on doubling
//   hash_table/entry_accessor.cc:113-195               row = float num[dim] | optimizer ctx
//   hash_table/optimizer/{sgd,adagrad,ftrl}_optimizer.cc, optimizer/avx_utils.h:29-38
//
// This header is compiled by hipcc (device + host) and, for the CPU-side unit test of the serial
// displacement logic only (tests/test_core_host.py), by g++ with MHTE_HOST_ONLY defined.
#ifndef MHTE_CORE_H_
#define MHTE_CORE_H_

#include <stdint.h>

#if defined(MHTE_HOST_ONLY)
#include <math.h>
#define MHTE_HD
#else
#include <hip/hip_runtime.h>
#define MHTE_HD __host__ __device__ __forceinline__
#endif

namespace mhte {

constexpr int kSlots = 4;            // cuckoohash_config.hpp:26
constexpr int kMaxBfsPathLen = 5;    // cuckoohash_map.hpp:1432
constexpr int kMaxCuckooCount = 682; // 2*(4^5-1)/3, cuckoohash_map.hpp:1705-1709
constexpr int kMaxSegments = 8;
// The one int64 that cannot live in a bucket: it marks an empty slot.  FID v2 keeps bit 63 = 0
// (data/training_instance/cc/fid.h:61-68), so it never occurs in practice; it is still a legal
// key for the table API and is served from a side slot in Counters.
constexpr int64_t kEmptyKey = INT64_MIN;
constexpr uint32_t kNoRow = 0xFFFFFFFFu;

enum OptType : int32_t {
  kOptSgd = 0, kOptAdagrad = 1, kOptFtrl = 2,
  // (the fused training-step kernels serve these through their FULL instances; GroupAdaGrad — whole-segment —
  // through the op-level kernels, mhte_fused_optimize and the id-sharded step's owner side):
  kOptMomentum = 3, kOptAdadelta = 4, kOptRmsprop = 5, kOptRmspropV2 = 6, kOptAdam = 7, kOptAmsgrad = 8,
  kOptMovingAverage = 9, kOptBatchSoftmax = 10, kOptGroupAdagrad = 11,
  kOptCount = 12
};
enum InitType : int32_t { kInitZeros = 0, kInitOnes = 1, kInitConstant = 2, kInitRandomUniform = 3 };

// One 64-byte line per bucket: 4 keys, 4 row handles, 4 uint32 timestamps.  The reference's
// PACKED bucket is 4 x (int64 key + {u32 EntryAddress,u32 ts}) + 4 partial + 4 occupied = 72 B
// (bucket_container.hpp:72-122, entry_defs.h:24-39); here occupancy is "key != kEmptyKey" and the
// partial tag is recomputed from the key (ALU is free next to an HBM miss), which buys exact
// cache-line alignment.
struct alignas(64) Bucket {
  int64_t key[kSlots];
  uint32_t row[kSlots];
  uint32_t ts[kSlots];
};
static_assert(sizeof(Bucket) == 64, "bucket must be one 64-byte line");

struct SegDesc {
  int32_t dim;      // segment dim_size (embedding_hash_table.proto:23-43)
  int32_t w_off;    // float offset of this segment's weights inside the row
  int32_t st_off;   // float offset of this segment's optimizer ctx inside the row
  int32_t opt;      // OptType
  float p[8];       // adagrad:  {initial_accumulator_value, weight_decay_factor, avx form (adagrad_step_avx)}
                    // ftrl:     {initial_accumulator_value, beta, l1, l2}
                    // momentum: {momentum, weight_decay_factor, use_nesterov}
                    // adadelta: {averaging_ratio, epsilon, weight_decay_factor}
                    // rmsprop / rmspropv2: {momentum, weight_decay_factor, config learning_rate}
                    // adam / amsgrad: {beta1, beta2, epsilon, weight_decay_factor, use_nesterov}
  int32_t init;     // InitType
  float init_value; // constants initializer value; random uniform: minval
  float init_value2; // random uniform: maxval
  int32_t sr16;     // OptimizerConfig.stochastic_rounding_float16 (optimizer.proto:228)
};

// fmix64 (murmur3 finaliser).  Stands in for absl::Hash<int64_t> (cuckoohash_map.hpp:64-70), which
// is ASLR-seeded and not under /root/reference; oracle/ref_shim gives the reference map the same
// functor so physical placement is comparable.
MHTE_HD uint64_t hash_key(int64_t key) {
  uint64_t h = static_cast<uint64_t>(key);
  h ^= h >> 33;
  h *= 0xff51afd7ed558ccdULL;
  h ^= h >> 33;
  h *= 0xc4ceb9fe1a85ec53ULL;
  h ^= h >> 33;
  return h;
}
MHTE_HD uint32_t partial_key(uint64_t hash) {  // :860-869
  uint32_t h32 = static_cast<uint32_t>(hash) ^ static_cast<uint32_t>(hash >> 32);
  uint32_t h16 = (h32 ^ (h32 >> 16)) & 0xffffu;
  return (h16 ^ (h16 >> 8)) & 0xffu;
}
MHTE_HD uint64_t hash_mask(uint32_t hp) { return (uint64_t(1) << hp) - 1; }
MHTE_HD uint64_t index_hash(uint32_t hp, uint64_t hv) { return hv & hash_mask(hp); }  // :873-875
MHTE_HD uint64_t alt_index(uint32_t hp, uint32_t partial, uint64_t index) {          // :882-888
  uint64_t nonzero_tag = uint64_t(partial) + 1;
  return (index ^ (nonzero_tag * 0xc6a4a7935bd1e995ULL)) & hash_mask(hp);
}
MHTE_HD uint32_t reserve_calc(uint64_t n) {  // :2114-2121
  uint64_t buckets = (n + kSlots - 1) / kSlots;
  uint32_t blog2 = 0;
  while ((uint64_t(1) << blog2) < buckets) ++blog2;
  return blog2;
}

// ------------------------------------------------------------------------------------------
// Serial cuckoo displacement (slow path).  Runs in ONE lane of a single-workgroup kernel that has
// the table to itself (stream order), so no locks are needed; the search and move order follow
// the reference exactly, which is what makes sequential-insert placement bit-identical to the
// reference map (tests/test_parity_gpu.py::test_placement_matches_reference).
// ------------------------------------------------------------------------------------------
struct BfsSlot {
  uint64_t bucket;
  uint16_t pathcode;
  int8_t depth;
};
struct CuckooRecord {
  uint64_t bucket;
  int32_t slot;
  uint64_t hash;
  uint32_t partial;
};

MHTE_HD bool slot_occupied(const Bucket& b, int s) { return b.key[s] != kEmptyKey; }

// cuckoohash_map.hpp:1725-1762.  `q` must hold kMaxCuckooCount entries.
MHTE_HD BfsSlot slot_search(const Bucket* buckets, uint32_t hp, uint64_t i1, uint64_t i2,
                            BfsSlot* q) {
  int first = 0, last = 0;
  q[last++] = BfsSlot{i1, 0, 0};
  q[last++] = BfsSlot{i2, 1, 0};
  while (first != last) {
    BfsSlot x = q[first++];
    const Bucket& b = buckets[x.bucket];
    int starting_slot = x.pathcode % kSlots;
    for (int i = 0; i < kSlots; ++i) {
      uint16_t slot = static_cast<uint16_t>((starting_slot + i) % kSlots);
      if (!slot_occupied(b, slot)) {
        x.pathcode = static_cast<uint16_t>(x.pathcode * kSlots + slot);
        return x;
      }
      if (x.depth < kMaxBfsPathLen - 1) {
        uint32_t partial = partial_key(hash_key(b.key[slot]));
        BfsSlot y;
        y.bucket = alt_index(hp, partial, x.bucket);
        y.pathcode = static_cast<uint16_t>(x.pathcode * kSlots + slot);
        y.depth = static_cast<int8_t>(x.depth + 1);
        q[last++] = y;
      }
    }
  }
  return BfsSlot{0, 0, -1};
}

// cuckoohash_map.hpp:1508-1561
MHTE_HD int cuckoopath_search(const Bucket* buckets, uint32_t hp, CuckooRecord* path, uint64_t i1,
                              uint64_t i2, BfsSlot* q) {
  BfsSlot x = slot_search(buckets, hp, i1, i2, q);
  if (x.depth == -1) return -1;
  for (int i = x.depth; i >= 0; --i) {
    path[i].slot = x.pathcode % kSlots;
    x.pathcode = static_cast<uint16_t>(x.pathcode / kSlots);
  }
  path[0].bucket = (x.pathcode == 0) ? i1 : i2;
  {
    const Bucket& b = buckets[path[0].bucket];
    if (!slot_occupied(b, path[0].slot)) return 0;
    path[0].hash = hash_key(b.key[path[0].slot]);
    path[0].partial = partial_key(path[0].hash);
  }
  for (int i = 1; i <= x.depth; ++i) {
    path[i].bucket = alt_index(hp, path[i - 1].partial, path[i - 1].bucket);
    const Bucket& b = buckets[path[i].bucket];
    if (!slot_occupied(b, path[i].slot)) return i;
    path[i].hash = hash_key(b.key[path[i].slot]);
    path[i].partial = partial_key(path[i].hash);
  }
  return x.depth;
}

// cuckoohash_map.hpp:1569-1636 (single owner: the validity re-checks cannot fail, kept anyway)
MHTE_HD bool cuckoopath_move(Bucket* buckets, CuckooRecord* path, int depth) {
  if (depth == 0) return !slot_occupied(buckets[path[0].bucket], path[0].slot);
  while (depth > 0) {
    CuckooRecord& from = path[depth - 1];
    CuckooRecord& to = path[depth];
    Bucket& fb = buckets[from.bucket];
    Bucket& tb = buckets[to.bucket];
    if (slot_occupied(tb, to.slot) || !slot_occupied(fb, from.slot) ||
        hash_key(fb.key[from.slot]) != from.hash) {
      return false;
    }
    tb.row[to.slot] = fb.row[from.slot];
    tb.ts[to.slot] = fb.ts[from.slot];
    tb.key[to.slot] = fb.key[from.slot];
    fb.key[from.slot] = kEmptyKey;
    fb.row[from.slot] = kNoRow;   // invariant: an empty slot carries no row handle (a claim publishes
                                  // the key first and the handle after it — a reader that sees the
                                  // new key beside a stale handle would take another id's row)
    --depth;
  }
  return true;
}

// Finds (and reserves, by writing the key) a slot for `key`, which the caller has established is
// absent and whose two buckets are both full or contended.  Returns bucket*4+slot, or -1 when no
// displacement path of length <= 5 exists (the reference would double the table here,
// cuckoohash_map.hpp:1296-1299; the engine grows proactively on the host instead).
MHTE_HD int64_t serial_insert_slot(Bucket* buckets, uint32_t hp, int64_t key, BfsSlot* q,
                                   CuckooRecord* path) {
  uint64_t hv = hash_key(key);
  uint32_t partial = partial_key(hv);
  uint64_t i1 = index_hash(hp, hv);
  uint64_t i2 = alt_index(hp, partial, i1);
  // try_find_insert_bucket, :1398-1418 — last empty slot of b1, else of b2
  for (int pass = 0; pass < 2; ++pass) {
    uint64_t ib = pass == 0 ? i1 : i2;
    int found = -1;
    for (int s = 0; s < kSlots; ++s)
      if (!slot_occupied(buckets[ib], s)) found = s;
    if (found >= 0) {
      buckets[ib].key[found] = key;
      return static_cast<int64_t>(ib * kSlots + found);
    }
  }
  for (;;) {
    int depth = cuckoopath_search(buckets, hp, path, i1, i2, q);
    if (depth < 0) return -1;
    if (cuckoopath_move(buckets, path, depth)) break;
  }
  buckets[path[0].bucket].key[path[0].slot] = key;
  return static_cast<int64_t>(path[0].bucket * kSlots + path[0].slot);
}

// ------------------------------------------------------------------------------------------
// Per-element optimizer arithmetic.  Contraction is disabled so that results are bit-identical
// to the scalar reference path (avx_utils.h:29-38 BaselineAdagradOptimize; the AVX path differs
// by FMA rounding only, avx_test.cc:29-62 tolerates 1e-6).
// ------------------------------------------------------------------------------------------
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

MHTE_HD float sgd_step(float w, float g, float lr) {  // sgd_optimizer.cc:42-49
  float d = lr * g;
  return w - d;
}

MHTE_HD void adagrad_step(float& w, float& n, float grad, float lr, float wd) {  // avx_utils.h:29-38
  float t = wd * w;
  float g = grad + t;
  float g2 = g * g;
  n = n + g2;
  float eff = lr / sqrtf(n);
  float d = eff * g;
  w = w - d;
}

// The reference's AVX2 form of AdagradOptimize (avx_utils.h:96-119: what its .bazelrc:63-68 build
// runs), opt-in per segment (p[2] != 0; entry.AdagradOptimizer(avx_semantics=True)): fused
// multiply-adds, and inside the segment's blocks of 8 elements the weight step is taken with the RAW
// gradient (:112) — a different update from the baseline loop whenever weight_decay_factor != 0.  The
// len % 8 tail runs the baseline formula as that build contracts it.  Explicit fma: exempt from the
// contraction pragma above, bit-identical to the reference's AVX build (tests/test_oracle.py).
MHTE_HD void adagrad_step_avx(float& w, float& n, float grad, float lr, float wd, bool in_block) {
  const float ug = __builtin_fmaf(wd, w, grad);
  n = __builtin_fmaf(ug, ug, n);
  const float eff = lr / sqrtf(n);
  w = __builtin_fmaf(-eff, in_block ? grad : ug, w);
}
// (-DMHTE_NO_AVX_FORM compiles the opt-in out: A/B builds)
#ifdef MHTE_NO_AVX_FORM
#define MHTE_AVX_FORM(sd) false
#else
#define MHTE_AVX_FORM(sd) ((sd).p[2] != 0.f)
#endif
// elem: element index inside the segment of seg_dim elements
MHTE_HD void adagrad_any(float& w, float& n, float grad, float lr, float wd, float avx_flag,
                         uint32_t elem, uint32_t seg_dim) {
  if (avx_flag != 0.f) adagrad_step_avx(w, n, grad, lr, wd, elem < (seg_dim & ~7u));
  else adagrad_step(w, n, grad, lr, wd);
}

// ftrl_optimizer.cc:56-75, including its `std::signbit(z) * l1` term (1*l1 for negative z, else 0)
MHTE_HD void ftrl_step(float& w, float& n, float& z, float grad, float lr, float beta, float l1,
                       float l2) {
  float gg = grad * grad;
  float norm_new = n + gg;
  float sigma = (sqrtf(norm_new) - sqrtf(n)) / lr;
  float sw = sigma * w;
  z = z + (grad - sw);
  n = norm_new;
  if (fabsf(z) > l1) {
    float sl1 = (z < 0.f || (z == 0.f && (1.f / z) < 0.f)) ? l1 : 0.f;
    float num = lr * (sl1 - z);
    float l2lr = l2 * lr;
    float den = (sqrtf(n) + beta) + l2lr;
    w = num / den;
  } else {
    w = 0.f;
  }
}

// Optimizer context of a segment inside the row: `opt_vectors` state vectors of dim floats each,
// then (adam / amsgrad) the two running beta powers in a 4-float slot.  The reference keeps the
// vectors in the same order and the two scalars right behind them (adam_optimizer.cc:30-32,47-50;
// the padding to 4 floats is this engine's, for float4 alignment of the next segment).
MHTE_HD int opt_vectors(int opt) {
  switch (opt) {
    case kOptSgd: return 0;
    case kOptAdagrad: case kOptMomentum: case kOptRmsprop: case kOptRmspropV2: return 1;
    case kOptFtrl: case kOptAdadelta: case kOptAdam: return 2;
    case kOptAmsgrad: return 3;
    default: return 0;
  }
}
MHTE_HD int opt_scalars(int opt) { return (opt == kOptAdam || opt == kOptAmsgrad) ? 2 : 0; }
// batch softmax keeps one int64 (the global step of the id's last update, batch_softmax_optimizer.cc
// :33,52-63)
// in the first two words of the same 4-float slot; group adagrad one float (the running sum of the
// largest squared gradient of the segment, group_adagrad_optimizer.cc:31)
MHTE_HD bool opt_has_slot(int opt) {
  return opt_scalars(opt) != 0 || opt == kOptBatchSoftmax || opt == kOptGroupAdagrad;
}
MHTE_HD int opt_state_floats(int opt, int dim) {
  return opt_vectors(opt) * dim + (opt_has_slot(opt) ? 4 : 0);
}
// initial value of state vector k (optimizer Init()); scalars: adam/amsgrad {beta1, beta2}
MHTE_HD float opt_state_init(const SegDesc& s, int k) {
  if (s.opt == kOptAdagrad) return s.p[0];               // adagrad_optimizer.cc:47-52
  if (s.opt == kOptFtrl) return k == 0 ? s.p[0] : 0.f;   // ftrl_optimizer.cc:44-51: norm | zero
  return 0.f;                                            // momentum / adadelta / rmsprop / adam / amsgrad
}

// moving_average_optimizer.cc:44-52 (no state, no learning rate)
MHTE_HD float moving_average_step(float w, float grad, float mom) {
  float a = mom * w;
  float b = (1 - mom) * grad;
  return a + b;
}

// batch_softmax_optimizer.cc:52-63: B = (1 - alpha) B + alpha (global_step - A); A = global_step
MHTE_HD void batch_softmax_step(float& w, long long& last_step, float alpha, long long global_step) {
  float a = (1 - alpha) * w;
  float b = alpha * static_cast<float>(global_step - last_step);
  w = a + b;
  last_step = global_step;
}

// momentum_optimizer.cc:50-71
MHTE_HD void momentum_step(float& w, float& n, float grad, float lr, float mom, float wd,
                           bool nesterov) {
  float t = wd * w;
  float gg = grad + t;
  float dx = lr * gg;
  if (nesterov) {
    float prev_n = n;
    float mn = mom * n;
    n = mn - dx;
    float a = -mom * prev_n;
    float b = (1 + mom) * n;
    float d = a + b;
    w = w + d;
  } else {
    float mn = mom * n;
    n = mn - dx;
    w = w + n;
  }
}

// adadelta_optimizer.cc:51-72
MHTE_HD void adadelta_step(float& w, float& accum, float& accum_update, float grad, float lr,
                           float rho, float eps, float wd) {
  float t = wd * w;
  float g = grad + t;
  float a1 = accum * rho;
  float gg = g * g;
  float a2 = gg * (1 - rho);
  float new_accum = a1 + a2;
  float num = sqrtf(accum_update + eps);
  float den = sqrtf(new_accum + eps);
  float q = num / den;
  float update = q * g;
  float ul = update * lr;
  float u1 = accum_update * rho;
  float uu = update * update;
  float u2 = uu * (1 - rho);
  w = w - ul;
  accum = new_accum;
  accum_update = u1 + u2;
}

// rmsprop_optimizer.cc:54-72 (v1: the CONFIG's learning rate, double arithmetic) and :127-144 (v2)
MHTE_HD void rmsprop_step(float& w, float& n, float grad, double lr, float mom, float wd, bool v2) {
  double dx = grad + static_cast<double>(wd) * w;
  float new_n;
  if (v2) {
    new_n = static_cast<float>(static_cast<double>(mom) * n + dx * dx);
  } else {
    new_n = static_cast<float>(static_cast<double>(mom) * n + (1 - static_cast<double>(mom)) * dx * dx);
  }
  double eta = lr / (sqrtf(new_n) + 1);  // (std::sqrt of a float: float; the sum too)
  w = static_cast<float>(w - eta * dx);
  n = new_n;
}

// adam_optimizer.cc:56-86 / amsgrad_optimizer.cc; lr_eff = lr * sqrt(1 - beta2_power) / (1 - beta1_power).
// The reference's `sqrt` is unqualified after <cmath> alone, i.e. ::sqrt(double) (:64,74,76;
// amsgrad_optimizer.cc:66,77,79): lr_eff and every element's quotient are formed in double and
// rounded to float once, at the assignment.  Followed bit for bit (f64 sqrt / div are IEEE here).
MHTE_HD float adam_lr(float lr, float b1p, float b2p) {
  const double a = sqrt(static_cast<double>(1 - b2p));
  const double b = static_cast<double>(lr) * a;
  return static_cast<float>(b / static_cast<double>(1 - b1p));
}
MHTE_HD void adam_step(float& w, float& m, float& v, float* vhat, float grad, float lr_eff,
                       float beta1, float beta2, float eps, float wd, bool nesterov) {
  float t = wd * w;
  float g = grad + t;
  float dm = (g - m) * (1 - beta1);
  float new_m = m + dm;
  float gg = g * g;
  float dv = (gg - v) * (1 - beta2);
  float new_v = v + dv;
  float vv = new_v;
  if (vhat) {
    vv = (*vhat > new_v) ? *vhat : new_v;
    *vhat = vv;
  }
  const double den = sqrt(static_cast<double>(vv)) + static_cast<double>(eps);
  float numr;
  if (nesterov) {
    float a = g * (1 - beta1);
    float b = beta1 * new_m;
    numr = (a + b) * lr_eff;
  } else {
    numr = new_m * lr_eff;
  }
  const double q = static_cast<double>(numr) / den;
  w = static_cast<float>(static_cast<double>(w) - q);
  m = new_m;
  v = new_v;
}

// Initial value of the weight stored at `where` (initializer/{zeros,ones,constants,random_uniform}
// _initializer.cc).  RandomUniform: the reference draws from a thread-local mt19937
// (random_uniform_initializer.cc:31-37), i.e. values depend on which thread inserted what and are
// not reproducible; here the draw is a counter-based hash of the element's address — uniform in
// [minval, maxval), independent between elements, nothing to seed or to synchronise.
// ---- fp16 stochastic rounding (the StochasticRoundingFloat16OptimizerDecorator,
// optimizer/stochastic_rounding.h:27-59, OptimizerConfig.stochastic_rounding_float16): after every
// Optimize() the weights — not the optimizer's state — become one of their two binary16 neighbours,
// the upper one with probability (vf - down) / (up - down).  stochastic_round is the reference's
// function value for value (pinned through oracle/mhte_oracle.c to the reference's own, compiled in
// place).  Its draws come from a thread-local multiply-with-carry generator consumed in call order
// there — no sequence an engine with another execution order could reproduce — so the draw here is
// a counter-based hash of (the element's address, the unrounded value, update_time, the occurrence):
// a function of the table's state, the same on every run, unbiased over the updates of an element.
MHTE_HD uint32_t sr_float_bits(float f) {
  union { float f; uint32_t u; } c;
  c.f = f;
  return c.u;
}
MHTE_HD float sr_bits_float(uint32_t u) {
  union { float f; uint32_t u; } c;
  c.u = u;
  return c.f;
}
// float -> binary16 with a directed rounding (half.hpp float2half<round_toward_infinity /
// round_toward_neg_infinity>), returned as the float it stands for
MHTE_HD float half_neighbour(float f, bool toward_pos_inf) {
  const uint32_t x = sr_float_bits(f);
  const uint32_t sign = x >> 31, exp = (x >> 23) & 0xffu, man = x & 0x7fffffu;
  if (exp == 0xffu) return f;
  const bool mag_up = toward_pos_inf ? !sign : (sign != 0);
  const int e = int(exp) - 127;
  uint32_t h, rem;
  if (exp == 0) {
    h = 0;
    rem = man;
  } else if (e >= 16) {
    h = mag_up ? 0x7c00u : 0x7bffu;
    rem = 0;
  } else if (e >= -14) {
    h = (uint32_t(e + 15) << 10) | (man >> 13);
    rem = man & 0x1fffu;
  } else {
    const uint32_t full = 0x800000u | man;
    const int sh = 13 + (-14 - e);
    if (sh >= 32) {
      h = 0;
      rem = 1;
    } else {
      h = full >> sh;
      rem = full & ((1u << sh) - 1u);
    }
  }
  if (rem && mag_up) h += 1;
  // binary16 bits -> float
  const uint32_t hexp = (h >> 10) & 0x1fu, hman = h & 0x3ffu;
  uint32_t o;
  if (hexp == 0x1fu) {
    o = 0x7f800000u | (hman << 13);
  } else if (hexp) {
    o = ((hexp + 112u) << 23) | (hman << 13);
  } else if (hman) {
    int sft = 0;
    uint32_t m = hman;
    while (!(m & 0x400u)) {
      m <<= 1;
      ++sft;
    }
    o = (uint32_t(113 - sft) << 23) | ((m & 0x3ffu) << 13);
  } else {
    o = 0;
  }
  return sr_bits_float((sign << 31) | o);
}
MHTE_HD float stochastic_round(float vf, float p) {   // stochastic_rounding.h:27-40
  const float up = half_neighbour(vf, true), down = half_neighbour(vf, false);
  const float num = vf - down, den = up - down;
  const float frac = num / den;   // (0 / 0 = NaN for a representable vf: the comparison is false)
  return (p <= frac) ? up : down;
}
MHTE_HD float sr_draw(const float* where, float v, uint32_t ts, uint32_t occurrence) {
  const uint64_t h = hash_key(static_cast<int64_t>(reinterpret_cast<uintptr_t>(where)) ^
                              static_cast<int64_t>(uint64_t(sr_float_bits(v)) << 32) ^
                              static_cast<int64_t>(uint64_t(ts) * 0x9E3779B97F4A7C15ull) ^
                              static_cast<int64_t>(occurrence));
  return static_cast<float>(h >> 40) * (1.0f / 16777216.0f);   // 24 bits -> [0, 1)
}

MHTE_HD float init_weight(const SegDesc& s, const float* where) {
  if (s.init == kInitRandomUniform) {
    const uint64_t h = hash_key(static_cast<int64_t>(reinterpret_cast<uintptr_t>(where)) ^ 0x5851f42d4c957f2dLL);
    const float u = static_cast<float>(h >> 40) * (1.0f / 16777216.0f);  // 24 bits -> [0, 1)
    return s.init_value + (s.init_value2 - s.init_value) * u;
  }
  return s.init == kInitOnes ? 1.f : (s.init == kInitConstant ? s.init_value : 0.f);
}

}  // namespace mhte
#endif  // MHTE_CORE_H_
