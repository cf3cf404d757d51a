// TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.
//
// Thin C driver around the REFERENCE's own sources, compiled in place from /root/reference
// (nothing is copied into this repo):
//   * monolith/native_training/runtime/hash_table/cuckoohash/cuckoohash_map.hpp   (the map)
//   * monolith/native_training/runtime/hash_table/optimizer/avx_utils.h          (Adagrad math)
// with the three shadow headers in oracle/ref_shim/ (fixed fmix64 hash instead of the
// ASLR-seeded absl::Hash).  Output: oracle/_ref/libmonolith_ref.so.
//
// What is the reference's code here: bucket choice, slot choice, BFS displacement, doubling,
// partial_dump order, evict scan, AdagradOptimize/ReduceSum arithmetic.
// What is restated around it (no TF/absl/protobuf in this image), following
//   cuckoo_embedding_hash_table.cc:140-264,346-353 (Lookup/Assign/AssignAdd/Reinitialize/
//   Optimize/Evict/UpsertEntry), entry_accessor.cc:113-195 (row = float num[D] | opt ctx),
//   sgd_optimizer.cc:42-49, adagrad_optimizer.cc:47-60, entry_defs.h:24-39 (uint32 timestamp),
//   reader_util.h:36-38 (slot_id_v2), distributed_ps.py:282-329,489-514 (PS step).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "monolith/native_training/runtime/hash_table/optimizer/avx_utils.h"
#include "monolith/native_training/runtime/hash_table/cuckoohash/cuckoohash_map.hpp"

namespace {

constexpr int64_t kSecPerDay = 24 * 60 * 60;
constexpr int kRowsPerBlock = 4096;  // cf. allocator/block_allocator.h:114-131

enum OptType { kSgd = 0, kAdagrad = 1 };

struct RefTable;

// Restates PackedEntry (entry_defs.h:24-39): 32-bit row address + uint32 timestamp, wrapped
// like WithInitFn (cuckoo_embedding_hash_table.cc:40-48) so that the init+update runs inside
// the map's critical section when the key is new.
struct Entry {
  uint32_t row;
  uint32_t ts;
  Entry() : row(0), ts(0) {}
  Entry(RefTable* t, const std::function<void(Entry&)>& init_fn);
};

using Map = libcuckoo::cuckoohash_map<int64_t, Entry>;

struct RefTable {
  int dim;
  int opt;
  float init_acc;
  float wd;
  float init_value;
  int row_floats;
  Map m;
  std::mutex alloc_mu;
  std::vector<std::unique_ptr<float[]>> blocks;
  std::atomic<uint32_t> next_row{0};
  std::unordered_map<int64_t, int> slot_ttl_days;
  int64_t default_ttl_days = 36500;  // embedding_hash_table.proto:63

  RefTable(int dim_, int opt_, float init_acc_, float wd_, float init_value_, uint64_t cap)
      : dim(dim_), opt(opt_), init_acc(init_acc_), wd(wd_), init_value(init_value_),
        row_floats(dim_ + (opt_ == kAdagrad ? dim_ : 0)), m(cap) {}

  uint32_t Alloc() {
    uint32_t r = next_row.fetch_add(1);
    std::lock_guard<std::mutex> g(alloc_mu);
    while (blocks.size() * kRowsPerBlock <= r) {
      blocks.emplace_back(new float[size_t(kRowsPerBlock) * row_floats]);
    }
    return r;
  }
  float* Row(uint32_t r) {
    return blocks[r / kRowsPerBlock].get() + size_t(r % kRowsPerBlock) * row_floats;
  }
  const float* Row(uint32_t r) const {
    return blocks[r / kRowsPerBlock].get() + size_t(r % kRowsPerBlock) * row_floats;
  }
  // entry_accessor.cc:158-162 Init = initializer + optimizer Init.
  void InitRow(float* row) const {
    for (int i = 0; i < dim; ++i) row[i] = init_value;
    if (opt == kAdagrad) {
      for (int i = 0; i < dim; ++i) row[dim + i] = init_acc;  // adagrad_optimizer.cc:47-52
    }
  }
  // cuckoo_embedding_hash_table.cc:346-353
  bool UpsertEntry(int64_t id, const std::function<void(Entry&)>& upsert_fn) {
    std::function<void(Entry&)> init_fn = [&](Entry& e) {
      InitRow(Row(e.row));
      upsert_fn(e);
    };
    return m.upsert(id, upsert_fn, this, init_fn);
  }
  void OptimizeRow(float* row, const float* grad, float lr) const {
    if (opt == kSgd) {
      for (int i = 0; i < dim; ++i) row[i] -= lr * grad[i];  // sgd_optimizer.cc:42-49
    } else {
      monolith::hash_table::AdagradOptimize(row, row + dim, grad, dim, lr, wd);
    }
  }
};

Entry::Entry(RefTable* t, const std::function<void(Entry&)>& init_fn) : row(t->Alloc()), ts(0) {
  init_fn(*this);
}

inline int64_t slot_id_v2(int64_t fid) { return (fid >> 48) & 0x7fff; }  // reader_util.h:36-38

}  // namespace

extern "C" {

void* ref_table_new(int dim, int opt, float init_acc, float wd, float init_value,
                    uint64_t initial_capacity) {
  return new RefTable(dim, opt, init_acc, wd, init_value, initial_capacity);
}
void ref_table_free(void* h) { delete static_cast<RefTable*>(h); }

int64_t ref_size(void* h) { return static_cast<RefTable*>(h)->m.size(); }
int64_t ref_hashpower(void* h) { return static_cast<RefTable*>(h)->m.hashpower(); }
int ref_contains(void* h, int64_t id) { return static_cast<RefTable*>(h)->m.contains(id); }

// cuckoo_embedding_hash_table.cc:140-171
int64_t ref_lookup(void* h, const int64_t* ids, int64_t n, float* out) {
  RefTable* t = static_cast<RefTable*>(h);
  int64_t found = 0;
  for (int64_t i = 0; i < n; ++i) {
    float* dst = out + i * t->dim;
    bool hit = t->m.find_fn(ids[i], [&](const Entry& e) {
      std::memcpy(dst, t->Row(e.row), sizeof(float) * t->dim);
    });
    if (hit) {
      ++found;
    } else {
      std::memset(dst, 0, sizeof(float) * t->dim);
    }
  }
  return found;
}

// :186-203 (skip_zero_embedding=false)
void ref_assign(void* h, const int64_t* ids, int64_t n, const float* values, int64_t update_time) {
  RefTable* t = static_cast<RefTable*>(h);
  for (int64_t i = 0; i < n; ++i) {
    const float* v = values + i * t->dim;
    t->UpsertEntry(ids[i], [&](Entry& e) {
      e.ts = static_cast<uint32_t>(update_time);
      std::memcpy(t->Row(e.row), v, sizeof(float) * t->dim);
    });
  }
}

// :206-213
void ref_assign_add(void* h, const int64_t* ids, int64_t n, const float* values,
                    int64_t update_time) {
  RefTable* t = static_cast<RefTable*>(h);
  for (int64_t i = 0; i < n; ++i) {
    const float* v = values + i * t->dim;
    t->UpsertEntry(ids[i], [&](Entry& e) {
      e.ts = static_cast<uint32_t>(update_time);
      float* row = t->Row(e.row);
      for (int k = 0; k < t->dim; ++k) row[k] += v[k];
    });
  }
}

// :215-226 ; status 1 = existed, 0 = inserted.  `now` replaces absl::Now().
void ref_reinitialize(void* h, const int64_t* ids, int64_t n, int32_t* status, int64_t now) {
  RefTable* t = static_cast<RefTable*>(h);
  for (int64_t i = 0; i < n; ++i) {
    bool existed = !t->UpsertEntry(ids[i], [&](Entry& e) {
      e.ts = static_cast<uint32_t>(now);
      t->InitRow(t->Row(e.row));
    });
    status[i] = existed;
  }
}

// :229-247
void ref_optimize(void* h, const int64_t* ids, int64_t n, const float* grads, float lr,
                  int64_t update_time) {
  RefTable* t = static_cast<RefTable*>(h);
  for (int64_t i = 0; i < n; ++i) {
    const float* g = grads + i * t->dim;
    t->UpsertEntry(ids[i], [&](Entry& e) {
      e.ts = static_cast<uint32_t>(update_time);
      t->OptimizeRow(t->Row(e.row), g, lr);
    });
  }
}

void ref_set_ttl(void* h, int64_t default_days, int n, const int64_t* slots, const int32_t* days) {
  RefTable* t = static_cast<RefTable*>(h);
  t->default_ttl_days = default_days;
  t->slot_ttl_days.clear();
  for (int i = 0; i < n; ++i) t->slot_ttl_days[slots[i]] = days[i];
}

// :251-264
void ref_evict(void* h, int64_t max_update_time) {
  RefTable* t = static_cast<RefTable*>(h);
  t->m.evict([&](const int64_t& key, const Entry& e) {
    int64_t ttl = t->default_ttl_days;
    auto it = t->slot_ttl_days.find(slot_id_v2(key));
    if (it != t->slot_ttl_days.end()) ttl = it->second;
    return max_update_time - int64_t(e.ts) >= ttl * kSecPerDay;
  });
}

// Full dump in the reference's own partial_dump order (cuckoohash_map.hpp:740-773), one
// element per call so that iter->offset reveals the exact (bucket*4+slot) of every key.
// Returns the number of entries written (<= cap).  rows may be NULL.
int64_t ref_dump(void* h, int64_t cap, int64_t* ids, int64_t* positions, uint32_t* ts,
                 float* rows) {
  RefTable* t = static_cast<RefTable*>(h);
  monolith::hash_table::EmbeddingHashTableInterface::DumpShard shard{0, 1, 1};
  monolith::hash_table::EmbeddingHashTableInterface::DumpIterator it;
  int64_t n = 0;
  while (n < cap) {
    bool got = false;
    t->m.partial_dump(
        shard,
        [&](const int64_t& key, const Entry& e) {
          ids[n] = key;
          ts[n] = e.ts;
          if (rows) std::memcpy(rows + n * t->row_floats, t->Row(e.row),
                                sizeof(float) * t->row_floats);
          got = true;
          return true;
        },
        &it);
    if (!got) break;
    positions[n] = it.offset - 1;
    ++n;
  }
  return n;
}

// Direct access to the reference arithmetic (avx_utils.h:238-261).
void ref_adagrad_optimize(float* num, float* norm, const float* grad, int64_t len, float lr,
                          float wd) {
  monolith::hash_table::AdagradOptimize(num, norm, grad, len, lr, wd);
}
void ref_reduce_sum(const float* a, const float* b, float* out, int64_t len) {
  monolith::hash_table::ReduceSum(a, b, out, len);
}
int ref_built_with_avx(void) {
#if defined(_ENABLE_AVX) && defined(__AVX__)
  return 1;
#else
  return 0;
#endif
}

}  // extern "C"

// --------------------------------------------------------------------------------------------
// CPU-baseline step runner, both variants of SURVEY.md 8d:
//   (i)  "PS-style": P single-threaded table shards (shard = fid mod P, distributed_ps.py:289,497)
//   (ii) "shared table": ONE table, P threads over contiguous chunks of the distinct ids — the
//        Shard() loop of the fused ops (multi_hash_table_lookup_op.cc:191-196); the reference map
//        is concurrent (per-bucket spinlocks), the ids of a step are distinct
// around the same worker-side dedup in first-occurrence order (unique_mapping_ops.cc:82-114),
// BatchLookup, scatter to duplicates (:225-242), duplicate-gradient sum in occurrence order
// (:307-324), BatchOptimize.  No TF / grpc cost is included, which favours the reference.
// ref_ps_breakdown reports the seconds the last step spent in each phase.
// --------------------------------------------------------------------------------------------
namespace {

class Pool {
 public:
  explicit Pool(int n) : n_(n) {
    for (int i = 0; i < n; ++i) th_.emplace_back([this, i] { Loop(i); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> g(mu_);
      stop_ = true;
      ++gen_;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  void Run(const std::function<void(int)>& fn) {
    {
      std::lock_guard<std::mutex> g(mu_);
      fn_ = &fn;
      pending_ = n_;
      ++gen_;
    }
    cv_.notify_all();
    std::unique_lock<std::mutex> l(mu_);
    done_.wait(l, [this] { return pending_ == 0; });
  }

 private:
  void Loop(int i) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(int)>* fn;
      {
        std::unique_lock<std::mutex> l(mu_);
        cv_.wait(l, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        fn = fn_;
      }
      (*fn)(i);
      {
        std::lock_guard<std::mutex> g(mu_);
        if (--pending_ == 0) done_.notify_all();
      }
    }
  }
  int n_;
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* fn_ = nullptr;
  int pending_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

struct RefPs {
  int P, dim;
  bool shared = false;     // variant (ii): ONE table, P threads over contiguous id chunks
  double phase_s[6] = {0, 0, 0, 0, 0, 0};  // last step: dedup, partition, lookup, scatter, grad sum, optimize
  std::vector<std::unique_ptr<RefTable>> shards;
  std::unique_ptr<Pool> pool;
  // per-step scratch
  std::vector<int64_t> uniq;
  std::vector<int32_t> inverse;
  std::vector<std::vector<int32_t>> shard_u;  // unique indices per shard
  std::vector<float> emb_u, grad_u;
  std::vector<int32_t> seg_off, seg_pos;
};

}  // namespace

extern "C" {

void* ref_ps_new(int P, int dim, int opt, float init_acc, float wd, float init_value,
                 uint64_t initial_capacity_per_shard) {
  RefPs* ps = new RefPs;
  ps->P = P;
  ps->dim = dim;
  for (int i = 0; i < P; ++i) {
    ps->shards.emplace_back(
        new RefTable(dim, opt, init_acc, wd, init_value, initial_capacity_per_shard));
  }
  ps->pool.reset(new Pool(P));
  ps->shard_u.resize(P);
  return ps;
}
// variant (ii)
void* ref_ps_new_shared(int P, int dim, int opt, float init_acc, float wd, float init_value,
                        uint64_t initial_capacity) {
  RefPs* ps = new RefPs;
  ps->P = P;
  ps->dim = dim;
  ps->shared = true;
  ps->shards.emplace_back(new RefTable(dim, opt, init_acc, wd, init_value, initial_capacity));
  // (Row() reads the block list while other threads append to it: never let it reallocate)
  ps->shards[0]->blocks.reserve(size_t(1) << 20);
  ps->pool.reset(new Pool(P));
  ps->shard_u.resize(P);
  return ps;
}
void ref_ps_breakdown(void* h, double* out6) {
  RefPs* ps = static_cast<RefPs*>(h);
  for (int i = 0; i < 6; ++i) out6[i] = ps->phase_s[i];
}
void ref_ps_free(void* h) { delete static_cast<RefPs*>(h); }
int64_t ref_ps_hashpower(void* h, int shard) {
  return static_cast<RefPs*>(h)->shards[shard]->m.hashpower();
}
int64_t ref_ps_size(void* h) {
  RefPs* ps = static_cast<RefPs*>(h);
  int64_t s = 0;
  for (auto& t : ps->shards) s += t->m.size();
  return s;
}

// One training step on the reference semantics; returns the number of unique ids.  `emb_out`
// receives the [n, dim] looked-up rows (pre-update), so callers can check parity.
int64_t ref_ps_step(void* h, const int64_t* ids, int64_t n, const float* grads, float lr,
                    int64_t update_time, float* emb_out) {
  RefPs* ps = static_cast<RefPs*>(h);
  const int P = ps->P, D = ps->dim;
  using Clock = std::chrono::steady_clock;
  auto t_prev = Clock::now();
  auto lap = [&](int i) {
    auto now = Clock::now();
    ps->phase_s[i] = std::chrono::duration<double>(now - t_prev).count();
    t_prev = now;
  };
  // worker-side dedup, first-occurrence order
  ps->uniq.clear();
  ps->inverse.resize(n);
  {
    std::unordered_map<int64_t, int32_t> m;
    m.reserve(2 * n);
    for (int64_t i = 0; i < n; ++i) {
      auto it = m.find(ids[i]);
      if (it == m.end()) {
        int32_t u = static_cast<int32_t>(ps->uniq.size());
        m.emplace(ids[i], u);
        ps->uniq.push_back(ids[i]);
        ps->inverse[i] = u;
      } else {
        ps->inverse[i] = it->second;
      }
    }
  }
  const int64_t U = ps->uniq.size();
  lap(0);
  for (auto& v : ps->shard_u) v.clear();
  if (ps->shared) {
    for (int s = 0; s < P; ++s)
      for (int64_t u = U * s / P; u < U * (s + 1) / P; ++u) ps->shard_u[s].push_back(static_cast<int32_t>(u));
  } else {
    for (int64_t u = 0; u < U; ++u) {
      int64_t id = ps->uniq[u];
      int s = static_cast<int>(((id % P) + P) % P);  // floormod, distributed_ps.py:289
      ps->shard_u[s].push_back(static_cast<int32_t>(u));
    }
  }
  ps->emb_u.resize(U * D);
  lap(1);
  // per-shard BatchLookup
  ps->pool->Run([&](int s) {
    RefTable* t = ps->shards[ps->shared ? 0 : s].get();
    for (int32_t u : ps->shard_u[s]) {
      float* dst = ps->emb_u.data() + int64_t(u) * D;
      bool hit = t->m.find_fn(ps->uniq[u], [&](const Entry& e) {
        std::memcpy(dst, t->Row(e.row), sizeof(float) * D);
      });
      if (!hit) std::memset(dst, 0, sizeof(float) * D);
    }
  });
  lap(2);
  // fill_with_offset_map: scatter unique rows to every occurrence (parallel over occurrence
  // ranges; the reference op is single-threaded, so this favours the reference)
  if (emb_out) {
    ps->pool->Run([&](int s) {
      int64_t lo = n * s / P, hi = n * (s + 1) / P;
      for (int64_t i = lo; i < hi; ++i) {
        std::memcpy(emb_out + i * D, ps->emb_u.data() + int64_t(ps->inverse[i]) * D,
                    sizeof(float) * D);
      }
    });
  }
  lap(3);
  // fill_with_offset_map_gradient: sum duplicates in occurrence order.  Occurrence lists
  // (CSR) are built once; threads own disjoint unique ranges, each list summed in order.
  ps->grad_u.assign(U * D, 0.f);
  ps->seg_off.assign(U + 1, 0);
  for (int64_t i = 0; i < n; ++i) ps->seg_off[ps->inverse[i] + 1]++;
  for (int64_t u = 0; u < U; ++u) ps->seg_off[u + 1] += ps->seg_off[u];
  ps->seg_pos.resize(n);
  {
    std::vector<int32_t> cur(ps->seg_off.begin(), ps->seg_off.end() - 1);
    for (int64_t i = 0; i < n; ++i) ps->seg_pos[cur[ps->inverse[i]]++] = static_cast<int32_t>(i);
  }
  ps->pool->Run([&](int s) {
    int64_t lo = U * s / P, hi = U * (s + 1) / P;
    for (int64_t u = lo; u < hi; ++u) {
      float* dst = ps->grad_u.data() + u * D;
      for (int32_t q = ps->seg_off[u]; q < ps->seg_off[u + 1]; ++q) {
        const float* g = grads + int64_t(ps->seg_pos[q]) * D;
        for (int k = 0; k < D; ++k) dst[k] += g[k];
      }
    }
  });
  lap(4);
  // per-shard BatchOptimize
  ps->pool->Run([&](int s) {
    RefTable* t = ps->shards[ps->shared ? 0 : s].get();
    for (int32_t u : ps->shard_u[s]) {
      const float* g = ps->grad_u.data() + int64_t(u) * D;
      t->UpsertEntry(ps->uniq[u], [&](Entry& e) {
        e.ts = static_cast<uint32_t>(update_time);
        t->OptimizeRow(t->Row(e.row), g, lr);
      });
    }
  });
  lap(5);
  return U;
}

int64_t ref_ps_lookup(void* h, const int64_t* ids, int64_t n, float* out) {
  RefPs* ps = static_cast<RefPs*>(h);
  int64_t found = 0;
  for (int64_t i = 0; i < n; ++i) {
    int s = ps->shared ? 0 : static_cast<int>(((ids[i] % ps->P) + ps->P) % ps->P);
    found += ref_lookup(ps->shards[s].get(), ids + i, 1, out + i * ps->dim);
  }
  return found;
}

}  // extern "C"
