// CDNA4 (gfx950) kernels of the embedding-table engine.  Included by mhte.hip only.
//
// Execution shape shared by the probe kernels: a GROUP of G lanes (G = 8..64, power of two, G | 64)
// serves one feature id.  Lanes 0..7 of the group each own one candidate slot (lanes 0-3: the four
// slots of bucket i1, lanes 4-7: bucket i2) so a probe is ONE round of 8-byte key loads that the
// coalescer merges into one 32-byte request per bucket line; a wave ballot finds the matching
// lane; then all G lanes move the row as float4 (G*16 B contiguous per id).  With B = 65 536 ids a
// launch has B*G/64 wavefronts (16 384 at dim 64) instead of B/64, which is what hides the
// id -> bucket -> row dependent-miss chain on a 256-CU part.
#ifndef MHTE_KERNELS_H_
#define MHTE_KERNELS_H_

#include "mhte_core.h"

// Development builds (-DMHTE_DEV_FAST, scripts/dev_build.sh) instantiate the device-side lane-group
// switches for G = 16 only (dim 64): the other cases do nothing.
#ifdef MHTE_DEV_FAST
#define MHTE_OTHER_G(...) ((void)0)
#else
#define MHTE_OTHER_G(...) __VA_ARGS__
#endif

namespace mhte {

struct Counters {
  // One word, one atomic per wavefront that inserts: low 32 bits = next row handle (bump
  // allocator), high 32 bits = live keys in buckets (side-slot key not included).  Same-address
  // device atomics cost ~12 ns each on MI355X, so the two counters share a single returning add.
  unsigned long long alloc;
  unsigned long long hits;     // lookup hits since last reset
  unsigned int reserved;       // keys counted in `alloc` for row reservations that no update has consumed yet
                               // (ProbeOut: the build role's table probe); live keys = alloc >> 32 minus this
  unsigned int n_pending;      // ids whose two buckets were full in the fast path
  unsigned int error;          // bit0: displacement failed (id dropped)
  unsigned int n_dropped;
  unsigned int special_state;  // 1 when kEmptyKey itself is stored
  unsigned int special_row;
  unsigned int special_ts;
  unsigned int n_evicted;
  unsigned int scan_count;     // generic scan output counter
  unsigned int pad;
};

struct TableView {
  Bucket* buckets;
  float* chunk0;          // first row slab (rows [0, 1<<chunk_shift)) — no pointer-table load
  float* const* chunks;   // device array of slab pointers
  Counters* ctr;
  uint32_t hp;
  uint32_t chunk_shift;
  uint32_t row_floats;
  uint32_t dim;
  uint32_t nseg;
  SegDesc seg[kMaxSegments];
  // admission (hash filter, runtime/hash_filter/hash_filter.h): null = no filter attached
  uint32_t* flt_slots;            // [flt_nsplit][flt_stride] signature << 4 | count (0 = empty)
  uint64_t flt_total;             // hash range of one split (its slots + kFilterMaxStep of overrun)
  uint32_t* flt_state;            // FilterState: head, head_increment, ..., elements per split
  uint32_t flt_nsplit;            // splits of the sliding window (>= 5)
  uint32_t flt_stride;            // words per split = flt_total + kFilterMaxStep
  uint32_t flt_cap;               // elements a split takes before the window moves on
  int32_t occ_default;            // SlotOccurrenceThresholdConfig.default_occurrence_threshold
  int32_t occ_n;
  const int64_t* occ_slots;       // device: per-feature-slot overrides
  const int32_t* occ_thr;
  // measurement aid (mhte_trace_begin): when non-null, every wavefront of a step kernel records
  // {begin, end, role, marks} at trace[kTraceWords * global wave index]; 100 MHz wall clock.
  unsigned long long* trace;
};

// Per-wavefront timeline record of the step kernels (null trace pointer: wave-uniform branches).
// Record = kTraceWords uint64: begin, end, role, then up to 5 intermediate marks.
constexpr int kTraceWords = 8;
struct WaveTrace {
  unsigned long long* rec;
  unsigned long long t0;
  // (the record pointer is the same for the whole wavefront: computed from the scalar wave index so
  // that it lives in SGPRs — as a VGPR pair it gets spilled in the tight kernels, and a spill
  // reload waits for EVERY outstanding global access of the wavefront)
  __device__ __forceinline__ WaveTrace(unsigned long long* trace) : rec(nullptr), t0(0) {
    if (trace) {
      const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
      // (multi-table launches: blockIdx.y = table; 0 elsewhere)
      rec = trace + uint64_t(kTraceWords) *
                        ((uint64_t(blockIdx.y) * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + wave);
      t0 = wall_clock64();
    }
  }
  __device__ __forceinline__ void mark(int i) {
    if (rec && (threadIdx.x & 63) == 0) rec[3 + i] = wall_clock64();
  }
  __device__ __forceinline__ void end(uint32_t role) {
    if (rec && (threadIdx.x & 63) == 0) {
      rec[0] = t0;
      rec[1] = wall_clock64();
      rec[2] = role;
    }
  }
};

enum ApplyOp : int { kOpAssign = 0, kOpAssignAdd = 1, kOpOptimize = 2, kOpReinit = 3 };
// (checkpoint restore is its own small kernel: restore_rows_kernel)

struct ApplyArgs {
  float lr[kMaxSegments];  // one per segment (SliceSize), multi_hash_table_update_op.cc:73-77
  uint32_t ts;             // (uint32)update_time, entry_defs.h:36-38
  int32_t filter_mode;     // admission (filter_consult): 0 off, 1 guarded by Contains, 3 unguarded
  long long global_step;   // batch softmax only
  int32_t sum_dups;        // 1: duplicates' values are added first, one optimizer step
                           //    (enable_grad_accumulation, tf_bridge.cc:270-310)
                           // 0: one optimizer step per occurrence, in order
                           //    (cuckoo_embedding_hash_table.cc:229-236)
};

// Tells the compiler that p points into global memory (HBM).  A pointer it cannot trace back to a
// kernel argument — the slab table's entries are loaded from memory — is otherwise accessed with
// FLAT instructions, which count on the LDS/scalar counter as well: every LDS wait, scalar load
// and LDS-only barrier of the wavefront then also waits for the HBM access to come back.
template <typename T>
__device__ __forceinline__ T* assume_global(T* p) {
  return (T*)(__attribute__((address_space(1))) T*)(p);
}
// (the cast pair above does not always survive to instruction selection; the hot accesses below
// go through pointers that are global BY TYPE)
#define MHTE_GLOBAL __attribute__((address_space(1)))
typedef MHTE_GLOBAL Bucket GBucket;
__device__ __forceinline__ GBucket* global_bucket(Bucket* b) { return (GBucket*)b; }
// 64-bit CAS on a bucket's key word (what atomicCAS does, on a global-typed pointer); returns the old value
__device__ __forceinline__ unsigned long long cas_key(MHTE_GLOBAL int64_t* p, int64_t expect,
                                                      int64_t desired) {
  unsigned long long e = static_cast<unsigned long long>(expect);
  __hip_atomic_compare_exchange_strong((MHTE_GLOBAL unsigned long long*)p, &e,
                                       static_cast<unsigned long long>(desired), __ATOMIC_RELAXED,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return e;
}

// Segment descriptor of row element e (k: its index).  One-segment tables — the usual case — read
// it with scalar loads at a constant offset; indexing tv.seg with a per-lane k is a vector load
// from the kernel-argument buffer, i.e. one more dependent round trip in front of the row's
// optimizer state.
// ONESEG is a compile-time fact of the caller's code path (the step kernels branch on tv.nseg once,
// at the top): mixing the two forms under one result keeps the descriptor in 14 vector registers.
template <bool ONESEG>
__device__ __forceinline__ SegDesc seg_of(const TableView& tv, uint32_t e, uint32_t& k) {
  k = 0;
  if (ONESEG) return tv.seg[0];
  while (k + 1 < tv.nseg && e >= uint32_t(tv.seg[k + 1].w_off)) ++k;
  return tv.seg[k];
}

__device__ __forceinline__ float* row_ptr(const TableView& tv, uint32_t r) {
  const uint32_t c = r >> tv.chunk_shift;
  typedef float* slab_ptr;
  float* base = (c == 0) ? tv.chunk0 : *(const MHTE_GLOBAL slab_ptr*)(tv.chunks + c);
  return assume_global(base + size_t(r & ((1u << tv.chunk_shift) - 1u)) * tv.row_floats);
}

template <int G>
__device__ __forceinline__ uint64_t group_mask_of(uint64_t wave_mask, int gbase) {
  if (G == 64) return wave_mask;
  return (wave_mask >> gbase) & ((uint64_t(1) << G) - 1);
}

template <int VEC>
struct Vec;
template <>
struct Vec<4> {
  float v[4];
  // (rows, gradients and outputs all live in HBM: global loads / stores by pointer type)
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  __device__ __forceinline__ void load(const float* p) {
    const f32x4 t = *(const MHTE_GLOBAL f32x4*)(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ __forceinline__ void store(float* p) const {
    f32x4 t;
    t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3];
    *(MHTE_GLOBAL f32x4*)(p) = t;
  }
};
template <>
struct Vec<1> {
  float v[1];
  __device__ __forceinline__ void load(const float* p) { v[0] = *(const MHTE_GLOBAL float*)(p); }
  __device__ __forceinline__ void store(float* p) const { *(MHTE_GLOBAL float*)(p) = v[0]; }
};

// =============================================================================================
// Admission filter: the reference's SlidingHashFilter (runtime/hash_filter/sliding_hash_filter.cc,
// .h; one split = HashFilter<uint16_t>, hash_filter.h:33-214; created by HashFilterOp,
// ops/hash_filter_op.cc:47-81; bridge call sites ops/embedding_hash_table_tf_bridge.cc
// :182-185,208-211,230-232,300-321).  An id that is not in the table yet is dropped while the
// number of times it has been seen is below its feature slot's occurrence threshold.
//   * `nsplit` (>= 5) counting filters of capacity / (nsplit - 1) elements each, fill rate 1.2,
//     4-bit saturating counts; slot word = signature << 4 | count, signature = (fid >> 17 | fid << 15)
//     & 0xfff (the reference's uint16 word, kept in 32 bits here); linear probing, 16 probes
//     (SlidingHashFilter::MAX_STEP), no wrap: the split has 64 words of overrun like the reference's
//     map_ (total_size + MAX_STEP).
//   * add (:56-91): look FORWARD from `head` over 2 splits for a usable slot (the id's, or an empty
//     one); the id's own -> count there.  An empty one -> look BACKWARD over up to
//     min(head_increment, nsplit - 2) older splits for the id's last count and start the new slot
//     from it.  No usable slot -> failure, counts as "seen max_count times".
//   * the window moves on (head advances, the split two ahead is cleared) when the head split is full.
//     The reference checks after every add; here the check runs between launches
//     (filter_advance_kernel) — inside one launch all adds see one window, as they would under any
//     one of the reference's thread interleavings in which the split fills at the batch's end.
// Physical placement uses the engine's fixed hash (the reference's absl::Hash is seeded per process).
// Checked against the reference's own sources compiled with that hash (oracle/_ref, through the
// restatement oracle/mhte_filter_oracle.c): counts, admission decisions, the split words.
// =============================================================================================
constexpr uint32_t kFilterMaxCount = 15;   // count_bit = 4, filter.h:56-57
constexpr int kFilterMaxStep = 64;         // words of overrun behind a split (hash_filter.h:190)
constexpr int kFilterProbe = 16;           // SlidingHashFilter::MAX_STEP
constexpr int kFilterForward = 2;          // max_forward_step_
constexpr int kFilterMaxSplits = 64;
constexpr int kFilterWays = 32;
constexpr int kFilterWayStride = 32;      // words between two partial counts of a split (a 128-byte line each)

struct FilterState {           // device words behind TableView::flt_state
  uint32_t head;
  uint32_t head_increment;
  uint32_t clear_req;          // split to clear + 1 (set by filter_advance_kernel)
  uint32_t head_elements;      // elements of the head split as filter_advance_kernel last saw them (what the
                               // host's FilterBudget fetches: these 16 bytes, not the 256 KB of counts below)
  unsigned long long failure_count;
  // elements per split, kept as kFilterWays partial counts (an id adds to way home % kFilterWays):
  // a step that starts 40 000 slots would otherwise queue 40 000 adds on ONE word.  Round 5: each
  // partial count in a 128-byte line of its own — atomics on one LINE are served one after the other
  // like atomics on one word (≈ 5 ns each), and the 32 counts of a split sat in one line: the 6 000 new
  // slots of a cold step were 30 µs of queued adds, the whole difference between the filtered and the
  // unfiltered step.
  uint32_t num_elements[kFilterMaxSplits][kFilterWays * kFilterWayStride];
};
__device__ __host__ inline uint32_t filter_split_elements(const FilterState& fs, uint32_t sp) {
  uint32_t n = 0;
  for (int w = 0; w < kFilterWays; ++w) n += fs.num_elements[sp][w * kFilterWayStride];
  return n;
}

__device__ __forceinline__ int32_t occurrence_threshold(const TableView& tv, int64_t id) {
  const int64_t slot = (id >> 48) & 0x7fff;  // slot_id_v2
  int32_t thr = tv.occ_default;
  for (int i = 0; i < tv.occ_n; ++i)
    if (tv.occ_slots[i] == slot) thr = tv.occ_thr[i];
  return thr;
}

// HashFilter<uint16_t>::signature (hash_filter.h:151): 12 bits — sign_mask = 0xffff >> count_bit.
// A slot word is the reference's uint16 value (signature << 4 | count) in a 32-bit word (the CAS
// unit), so a dump is readable by the reference ((uint16_t)data, hash_filter.cc:75) and two ids
// alias exactly when they would there: the reference's own test expects 0.9 % of the counts to be
// off at its load (sliding_hash_filter_test.cc:95-99).
__device__ __forceinline__ uint32_t filter_sign(int64_t id) {
  const uint64_t fid = uint64_t(id);
  return uint32_t((fid >> 17) | (fid << 15)) & 0x0fffu;
}
__device__ __forceinline__ uint64_t filter_home(int64_t id, uint64_t total) {
  return hash_key(id ^ 0x5bd1e995) % total;
}

// HashFilter::find (hash_filter.h:118-134) on one split: the first of kFilterProbe slots that is
// empty or carries `sign`; with `nonempty` only a slot that carries it.  Returns the slot index or
// -1; *v = the word seen there.
__device__ __forceinline__ long long filter_find(uint32_t* split, uint64_t home, uint32_t sign,
                                                 bool nonempty, uint32_t* v) {
  for (int t = 0; t < kFilterProbe; ++t) {
    const uint32_t w = __hip_atomic_load(&split[home + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (w == 0u) {
      if (nonempty) return -1;   // (the id would have been put here or before)
      *v = 0u;
      return static_cast<long long>(home + t);
    }
    if ((w >> 4) == sign) {
      *v = w;
      return static_cast<long long>(home + t);
    }
  }
  return -1;
}

// The decision for ONE id whose k occurrences arrive in order (one lane calls it).
//   mode 1 (Assign / AssignAdd / BatchOptimize without dedup): occurrence i is dropped iff the id is
//          still absent and the count seen before it is < thr; the first admitted occurrence inserts
//          the id, the rest no longer consult the filter.  Returns the index of the first admitted
//          occurrence (k: all dropped).
//   mode 2 (BatchOptimize with dedup, tf_bridge.cc:300-310): one consultation with count k.
//   mode 3 (multi-table AssignAdd = AssignAdd2, :230-232): no Contains guard — every occurrence
//          consults the filter; returns the first admitted index (later ones are admitted too, as
//          the count only grows).
// `contained`: the id is in the table already (modes 1, 2: filter untouched, everything admitted).
// ProbabilisticFilter (runtime/hash_filter/probabilistic_filter.{h,cc}; op MonolithProbabilisticFilter,
// ops/hash_filter_op.cc:81-110): stateless — an id that is NOT in the table is admitted with
// probability count / threshold per consultation (:24-28: Rand32 * threshold < UINT32_MAX * count) or,
// equal_probability, 1 - (1 - p)^count with p = 1 - 0.05^(1 / threshold) (:30-38); an id the table
// holds is never filtered (:43).  The reference draws from a thread-local xorshift seeded with
// time(0): there is no reproducible sequence to match, so the device draws from a counter-based
// generator — fmix64 of (seed, id, launch number, occurrence) — and parity is the admission RATE.
// A probabilistic filter is a view with flt_nsplit == 0: flt_total = 1 for equal_probability,
// FilterState::failure_count holds the seed, head_increment the launch number.
__device__ __forceinline__ uint32_t prob_rand32(const FilterState* fs, int64_t id, uint32_t occurrence) {
  uint64_t h = fs->failure_count ^ (uint64_t(id) * 0x9E3779B97F4A7C15ull) ^
               (uint64_t(fs->head_increment) << 32) ^ uint64_t(occurrence);
  h ^= h >> 33;
  h *= 0xff51afd7ed558ccdULL;
  h ^= h >> 33;
  h *= 0xc4ceb9fe1a85ec53ULL;
  h ^= h >> 33;
  return uint32_t(h >> 16);
}
__device__ __forceinline__ bool prob_admit(uint32_t r, uint32_t count, uint32_t thr, bool equal) {
  if (!equal) return uint64_t(r) * uint64_t(thr) < uint64_t(0xffffffffu) * uint64_t(count);
  const float p = 1.f - powf(0.05f, 1.f / float(thr));
  return float(r) < float(0xffffffffu) * (1.f - powf(1.f - p, float(count)));
}
__device__ __forceinline__ uint32_t prob_consult(const TableView& tv, int64_t id, uint32_t k, int mode,
                                                 bool contained, uint32_t thr) {
  if (contained) return 0u;                          // `table && !table->Contains(fid)` (:43)
  const FilterState* fs = reinterpret_cast<const FilterState*>(tv.flt_state);
  const bool equal = tv.flt_total == 1;
  if (mode == 2) return prob_admit(prob_rand32(fs, id, 0u), k, thr, equal) ? 0u : k;
  // one draw per occurrence while the id is absent; the first admitted one inserts it
  for (uint32_t i = 0; i < k; ++i)
    if (prob_admit(prob_rand32(fs, id, i), 1u, thr, equal)) return i;
  return k;
}

__device__ __forceinline__ uint32_t filter_consult(const TableView& tv, int64_t id, uint32_t k,
                                                   int mode, bool contained) {
  const int32_t thr_i = occurrence_threshold(tv, id);
  if (tv.flt_nsplit == 0u) {                         // probabilistic: a threshold of 0 admits (0 < count)
    if (k == 0) return 0u;
    return prob_consult(tv, id, k, mode, contained, uint32_t(thr_i < 0 ? 0 : thr_i));
  }
  if (thr_i <= 0 || k == 0) return 0u;              // ShouldBeFiltered: threshold <= 0 disables
  if (contained && mode != 3) return 0u;
  const uint32_t thr = uint32_t(thr_i);
  const uint32_t sign = filter_sign(id);
  const uint64_t home = filter_home(id, tv.flt_total);
  FilterState* fs = reinterpret_cast<FilterState*>(tv.flt_state);
  const uint32_t S = tv.flt_nsplit;
  const uint32_t head = fs->head, hinc = fs->head_increment;
  // first / adds for a count c0 seen before this batch's occurrences
  auto decide = [&](uint32_t c0, uint32_t* first, uint32_t* adds) {
    if (mode == 2) {
      *first = (c0 < thr) ? k : 0u;
      *adds = k;
    } else if (mode == 3) {
      *first = (c0 >= thr) ? 0u : min(k, thr - c0);
      *adds = k;
    } else {
      *first = (c0 >= thr) ? 0u : (thr - c0);
      *adds = min(k, *first + 1u);                  // the admitted occurrence is the last to ask
      *first = min(*first, k);
    }
  };
  for (int attempt = 0; attempt < 64; ++attempt) {
    // ---- look forward: a slot in the head split or the one after it
    long long pos = -1;
    uint32_t sp = head, v = 0;
    for (int f = 0; f < kFilterForward && pos < 0; ++f) {
      sp = (head + uint32_t(f)) % S;
      pos = filter_find(tv.flt_slots + size_t(sp) * tv.flt_stride, home, sign, false, &v);
    }
    if (pos < 0) {                                  // :64-67: failure, "seen max_count times"
      atomicAdd(&fs->failure_count, 1ull);
      uint32_t first, adds;
      decide(kFilterMaxCount, &first, &adds);
      return first;
    }
    uint32_t* slot = tv.flt_slots + size_t(sp) * tv.flt_stride + pos;
    if (v != 0u) {                                  // the id's slot: count there
      for (;;) {
        uint32_t first, adds;
        decide(v & kFilterMaxCount, &first, &adds);
        const uint32_t c1 = min(kFilterMaxCount, (v & kFilterMaxCount) + min(adds, kFilterMaxCount));
        const uint32_t old = atomicCAS(slot, v, (sign << 4) | c1);
        if (old == v) return first;
        v = old;                                    // (same signature: another lane counted)
      }
    }
    // ---- an empty slot: the id's last count from the older splits, then start the new slot
    uint32_t old_count = 0;
    {
      const uint32_t nb = min(hinc, S - uint32_t(kFilterForward));
      uint32_t sb = head;
      for (uint32_t i = 0; i < nb; ++i) {
        sb = (sb == 0u) ? S - 1u : sb - 1u;
        uint32_t w = 0;
        if (filter_find(tv.flt_slots + size_t(sb) * tv.flt_stride, home, sign, true, &w) >= 0) {
          old_count = w & kFilterMaxCount;
          break;
        }
      }
    }
    uint32_t first, adds;
    decide(old_count, &first, &adds);
    const uint32_t c1 = min(kFilterMaxCount, old_count + min(adds, kFilterMaxCount));
    if (atomicCAS(slot, 0u, (sign << 4) | c1) == 0u) {
      atomicAdd(&fs->num_elements[sp][(home & uint64_t(kFilterWays - 1)) * kFilterWayStride], 1u);
      return first;
    }
    // the slot went to another id meanwhile: look again
  }
  return 0u;
}

// The same consultation by a lane GROUP (round 5; the fused step's id-major groups).  filter_consult
// walks up to 2 + (nsplit - 2) splits of 16 slots one dependent load after the other on ONE lane — six
// round trips in front of the insert of every id the table does not hold yet, which is what made the
// filtered step 2.2x the unfiltered one.  Here lanes 0-15 of the group each fetch one slot of the
// probe window of every split at once (filter_probe_issue, issued beside the table probe of the same
// trip), the group's ballots replay HashFilter::find's "first slot that is empty or carries the
// signature" on the loaded words, and lane 0 makes the one CAS: one round trip behind the loads.
// Anything the loaded window cannot settle — a CAS that lost its slot, an older split beyond
// kFilterBack — goes to the serial form, which starts over (as its own retry does).
constexpr int kFilterBack = 6;   // older splits fetched ahead (nsplit <= 8: all of them)
template <int G>
struct FilterProbe {
  static constexpr int PERW = (kFilterProbe + G - 1) / G;   // window slots per lane (2 with 8 lanes)
  uint32_t w[(kFilterForward + kFilterBack) * PERW];
};
// (back0: the first of the older splits to fetch — 0 = the one right behind the head; a filter with more than
// kFilterBack older splits is walked kFilterBack at a time)
template <int G>
__device__ __forceinline__ FilterProbe<G> filter_probe_issue(const TableView& tv, int64_t id, bool active, int j,
                                                             uint32_t head, uint32_t hinc, uint32_t back0 = 0u) {
  FilterProbe<G> fp;
  constexpr int PERW = FilterProbe<G>::PERW;
  const uint32_t S = tv.flt_nsplit;
  const uint64_t home = active ? filter_home(id, tv.flt_total) : 0ull;
  const uint32_t nb_all = min(hinc, S - uint32_t(kFilterForward));
  const uint32_t nb = nb_all > back0 ? min(nb_all - back0, uint32_t(kFilterBack)) : 0u;
#pragma unroll
  for (int s = 0; s < kFilterForward; ++s) {
    const uint32_t* split = tv.flt_slots + size_t((head + uint32_t(s)) % S) * tv.flt_stride + home;
#pragma unroll
    for (int q = 0; q < PERW; ++q)
      fp.w[s * PERW + q] =
          __hip_atomic_load(&split[(j + q * G) & (kFilterProbe - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // older splits: nothing writes them while they are behind the head (plain, cached loads), and only the
  // `nb` the window has moved over exist (a filter that has not moved yet fetches none: the six fetches
  // were 2.6 us of a 37-us filtered step, profiles/r05/filter_step.md)
#pragma unroll
  for (int i = 0; i < kFilterBack; ++i) {
    if (uint32_t(i) < nb) {
      const uint32_t sp = (head + S - 1u - ((back0 + uint32_t(i)) % S)) % S;
      const uint32_t* split = tv.flt_slots + size_t(sp) * tv.flt_stride + home;
#pragma unroll
      for (int q = 0; q < PERW; ++q)
        fp.w[(kFilterForward + i) * PERW + q] = split[(j + q * G) & (kFilterProbe - 1)];
    }   // (else: left as it is — the consultation does not look at it; a value merged in here would
        //  make the compiler wait for the loads where the branch ends)
  }
  return fp;
}
// first slot of split s's window that is empty or carries `sign` (-1: none) and the word there
template <int G>
__device__ __forceinline__ int filter_window_first(const FilterProbe<G>& fp, int s, uint32_t sign, int j, int gbase,
                                                   uint32_t* word) {
  constexpr int PERW = FilterProbe<G>::PERW;
  uint64_t m = 0;
#pragma unroll
  for (int q = 0; q < PERW; ++q) {
    const uint32_t w = fp.w[s * PERW + q];
    const bool hit = (j + q * G) < kFilterProbe && (w == 0u || (w >> 4) == sign);
    m |= group_mask_of<G>(__ballot(hit), gbase) << (q * G);
  }
  m &= (1ull << kFilterProbe) - 1ull;
  const int p = m ? __ffsll(static_cast<long long>(m)) - 1 : -1;
  uint32_t v = 0;
#pragma unroll
  for (int q = 0; q < PERW; ++q) {
    const uint32_t x = __shfl(fp.w[s * PERW + q], gbase + ((p < 0 ? 0 : p) & (G - 1)));
    if (p >= q * G && p < (q + 1) * G) v = x;
  }
  *word = v;
  return p;
}
// Every lane of the group calls this (group-uniform arguments; `act`: the group has an id to ask about — the
// ballots inside run whatever `act` says; `have`: `fp` holds this id's windows, fetched ahead — otherwise (a
// hint that failed after the fetch, a caller that did not fetch) the first pass fetches them).  A CAS that lost
// its slot, and older splits beyond the kFilterBack fetched at a time, go round the loop again with fresh
// windows (HashFilter's own retry; no second, serial copy of the walk in the kernel).  Returns what
// filter_consult returns, in every lane of the group.
template <int G>
__device__ __forceinline__ uint32_t filter_consult_group(const TableView& tv, int64_t id, uint32_t k, int mode,
                                                         bool contained, bool act, bool have,
                                                         FilterProbe<G> fp, int j, int gbase, uint32_t head,
                                                         uint32_t hinc) {
  const int32_t thr_i = occurrence_threshold(tv, id);
  if (tv.flt_nsplit == 0u) {   // probabilistic filter (stateless): lane 0 draws
    uint32_t f0 = 0;
    if (act && j == 0 && k != 0) f0 = prob_consult(tv, id, k, mode, contained, uint32_t(thr_i < 0 ? 0 : thr_i));
    return __shfl(f0, gbase);
  }
  const uint32_t S = tv.flt_nsplit;
  const uint32_t sign = filter_sign(id);
  const uint32_t thr = uint32_t(thr_i > 0 ? thr_i : 0);
  const uint32_t nb_all = min(hinc, S - uint32_t(kFilterForward));
  bool pending = act && k != 0 && thr_i > 0 && !(contained && mode != 3);   // (group-uniform)
  uint32_t first = 0, back0 = 0;
  auto decide = [&](uint32_t c0, uint32_t* f1, uint32_t* adds) {
    if (mode == 2) {
      *f1 = (c0 < thr) ? k : 0u;
      *adds = k;
    } else if (mode == 3) {
      *f1 = (c0 >= thr) ? 0u : min(k, thr - c0);
      *adds = k;
    } else {
      *f1 = (c0 >= thr) ? 0u : (thr - c0);
      *adds = min(k, *f1 + 1u);
      *f1 = min(*f1, k);
    }
  };
#pragma unroll 1
  for (int attempt = 0; attempt < 256 && __any(pending); ++attempt) {
    if (__any(pending && !have)) {
      const FilterProbe<G> f2 = filter_probe_issue<G>(tv, id, pending && !have, j, head, hinc, back0);
      if (pending && !have) fp = f2;
    }
    have = false;   // (whatever happens below, a further pass needs fresh windows)
    // the windows, replayed on the loaded words (ballots: every lane of the wavefront)
    uint32_t v0, v1;
    const int p0 = filter_window_first<G>(fp, 0, sign, j, gbase, &v0);
    const int p1 = filter_window_first<G>(fp, 1, sign, j, gbase, &v1);
    const uint32_t nb = nb_all > back0 ? min(nb_all - back0, uint32_t(kFilterBack)) : 0u;
    uint32_t old_count = 0;
    bool older_found = false;
#pragma unroll
    for (int i = 0; i < kFilterBack; ++i) {
      uint32_t wv;
      const int p = filter_window_first<G>(fp, kFilterForward + i, sign, j, gbase, &wv);
      if (uint32_t(i) < nb && !older_found && p >= 0 && wv != 0u) {   // (an empty slot first: not in this split)
        old_count = wv & kFilterMaxCount;
        older_found = true;
      }
    }
    // lane 0 of a pending group acts; status: 0 = settled, 1 = look again from the head, 2 = the next older splits
    uint32_t status = 0, f1 = 0;
    if (pending && j == 0) {
      FilterState* fs = reinterpret_cast<FilterState*>(tv.flt_state);
      const uint64_t home = filter_home(id, tv.flt_total);
      const int pos = p0 >= 0 ? p0 : p1;
      const uint32_t sp = (head + (p0 >= 0 ? 0u : 1u)) % S;
      const uint32_t v = p0 >= 0 ? v0 : v1;
      uint32_t adds;
      if (pos < 0) {                                   // no usable slot: "seen max_count times"
        atomicAdd(&fs->failure_count, 1ull);
        decide(kFilterMaxCount, &f1, &adds);
      } else if (v == 0u && !older_found && back0 + nb < nb_all) {
        status = 2;                                    // older splits beyond the fetched ones
      } else {
        uint32_t* slot = tv.flt_slots + size_t(sp) * tv.flt_stride + home + uint32_t(pos);
        const uint32_t c0 = v != 0u ? (v & kFilterMaxCount) : old_count;
        decide(c0, &f1, &adds);
        const uint32_t c1 = min(kFilterMaxCount, c0 + min(adds, kFilterMaxCount));
        if (atomicCAS(slot, v, (sign << 4) | c1) == v) {
          if (v == 0u) atomicAdd(&fs->num_elements[sp][(home & uint64_t(kFilterWays - 1)) * kFilterWayStride], 1u);
        } else {
          status = 1;                                  // the slot changed meanwhile: look again
        }
      }
    }
    status = __shfl(status, gbase);
    f1 = __shfl(f1, gbase);
    if (pending) {
      if (status == 0u) {
        first = f1;
        pending = false;
      } else if (status == 1u) {
        back0 = 0;
      } else {
        back0 += uint32_t(kFilterBack);
      }
    }
  }
  return first;
}

// SlidingHashFilter::get (:93-114)
__global__ __launch_bounds__(256) void filter_get_kernel(TableView tv, const int64_t* __restrict__ ids,
                                                         int64_t n, uint32_t* __restrict__ out) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (tv.flt_nsplit == 0u) {                          // ProbabilisticFilter::get: max_count (:31-33)
    out[i] = kFilterMaxCount;                         // (first: such a filter has flt_total == 0, and
    return;                                           // filter_home takes the hash modulo it)
  }
  const int64_t id = ids[i];
  const uint32_t sign = filter_sign(id);
  const uint64_t home = filter_home(id, tv.flt_total);
  const FilterState* fs = reinterpret_cast<const FilterState*>(tv.flt_state);
  const uint32_t S = tv.flt_nsplit, head = fs->head;
  long long pos = -1;
  uint32_t v = 0;
  for (int f = 0; f < kFilterForward && pos < 0; ++f)
    pos = filter_find(tv.flt_slots + size_t((head + uint32_t(f)) % S) * tv.flt_stride, home, sign, false, &v);
  uint32_t c = kFilterMaxCount;                     // no usable slot: max_count
  if (pos >= 0) {
    c = v & kFilterMaxCount;
    if (v == 0u) {
      const uint32_t nb = min(fs->head_increment, S - uint32_t(kFilterForward));
      uint32_t sb = head;
      for (uint32_t k = 0; k < nb; ++k) {
        sb = (sb == 0u) ? S - 1u : sb - 1u;
        uint32_t w = 0;
        if (filter_find(tv.flt_slots + size_t(sb) * tv.flt_stride, home, sign, true, &w) >= 0) {
          c = w & kFilterMaxCount;
          break;
        }
      }
    }
  }
  out[i] = c;
}

// Between launches: the window moves on when the head split is full (:85-89); the split that
// becomes the look-ahead one is cleared by filter_clear_kernel, launched right behind.
__global__ void filter_advance_kernel(TableView tv) {
  FilterState* fs = reinterpret_cast<FilterState*>(tv.flt_state);
  if (threadIdx.x != 0) return;
  fs->clear_req = 0;
  if (tv.flt_nsplit == 0u) {     // probabilistic: a new launch number for the next update's draws
    fs->head_increment += 1u;
    return;
  }
  if (filter_split_elements(*fs, fs->head) + 1u >= tv.flt_cap) {   // HashFilter::full(): >= capacity - 1
    fs->head = (fs->head + 1u) % tv.flt_nsplit;
    fs->head_increment += 1u;
    const uint32_t c = (fs->head + uint32_t(kFilterForward) - 1u) % tv.flt_nsplit;
    for (int w = 0; w < kFilterWays; ++w) fs->num_elements[c][w * kFilterWayStride] = 0;
    fs->clear_req = c + 1u;
  }
  fs->head_elements = filter_split_elements(*fs, fs->head);
}
__global__ __launch_bounds__(256) void filter_clear_kernel(TableView tv) {
  const FilterState* fs = reinterpret_cast<const FilterState*>(tv.flt_state);
  const uint32_t req = fs->clear_req;
  if (req == 0u) return;
  uint32_t* split = tv.flt_slots + size_t(req - 1u) * tv.flt_stride;
  for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < tv.flt_stride;
       i += uint64_t(gridDim.x) * blockDim.x)
    split[i] = 0u;
}

// =============================================================================================
// Lookup: ids[n] -> out[n, dim].  Absent id -> zeros, never inserts
// (cuckoo_embedding_hash_table.cc:161-171).  Algorithmic bytes per id: 8 (id) + 36..72 (probe)
// + 4*dim (row) + 4*dim (output).
// =============================================================================================
template <int G, int VEC>
__device__ __forceinline__ void lookup_role(const TableView& tv, const int64_t* __restrict__ ids,
                                            int64_t n, const uint32_t* __restrict__ n_dev,
                                            float* __restrict__ out, int count_hits, uint32_t bid,
                                            int gate = 0) {
  const int lane = threadIdx.x & 63;
  const int j = lane & (G - 1);
  const int gbase = lane & ~(G - 1);
  const int64_t g = (int64_t(bid) * blockDim.x + threadIdx.x) / G;
  if (n_dev) n = min(n, int64_t(*n_dev));
  const bool valid = g < n;
  const int64_t id = valid ? ids[g] : 0;
  if (gate) {
    // A displacement pass for the previous update runs in another workgroup of this launch
    // (step_ka_kernel).  No table word is read before it has finished: n_pending drops to 0 only
    // after its stores were written back (release), and every load below is control-dependent on
    // having seen the 0, so no stale bucket or row line can be in this XCD's L2.
    if (__hip_atomic_load(&tv.ctr->n_pending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
      while (__hip_atomic_load(&tv.ctr->n_pending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
        __builtin_amdgcn_s_sleep(8);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  }
  const uint64_t hv = hash_key(id);
  const uint64_t i1 = index_hash(tv.hp, hv);
  const uint64_t i2 = alt_index(tv.hp, partial_key(hv), i1);
  bool match = false;
  uint32_t row = kNoRow;
  if (valid && j < 8 && id != kEmptyKey) {
    const Bucket* b = assume_global(tv.buckets + ((j < 4) ? i1 : i2));
    const int s = j & 3;
    const int64_t k = b->key[s];
    row = b->row[s];
    match = (k == id);
  }
  const uint64_t m = group_mask_of<G>(__ballot(match), gbase);
  bool found = m != 0;
  const int src = found ? (__ffsll(static_cast<long long>(m)) - 1) : 0;
  uint32_t r = __shfl(row, gbase + src);
  if (valid && id == kEmptyKey) {
    found = tv.ctr->special_state == 1;
    r = tv.ctr->special_row;
  }
  if (valid) {
    const float* rp = found ? row_ptr(tv, r) : nullptr;
    float* op = out + g * int64_t(tv.dim);
    for (uint32_t e = j * VEC; e < tv.dim; e += G * VEC) {
      Vec<VEC> v;
      if (found) {
        v.load(rp + e);
      } else {
#pragma unroll
        for (int c = 0; c < VEC; ++c) v.v[c] = 0.f;
      }
      v.store(op + e);
    }
  }
  if (count_hits) {
    const uint64_t hm = __ballot(valid && found && j == 0);
    if (lane == 0 && hm) atomicAdd(&tv.ctr->hits, (unsigned long long)__popcll(hm));
  }
}
// Unrolled form: a G-lane group serves UNR consecutive ids with all UNR probes, then all UNR row
// loads in flight at once.  A launch is bound by how many wavefronts the chip can start and hold
// (profiles/r01/e_wave_timeline_2launch.md), not by bytes, so fewer, fatter wavefronts win:
// B = 65 536 ids at dim 64 are 16 384 / UNR wavefronts.
template <int VEC>
__device__ __forceinline__ void store_stream(float* p, const Vec<VEC>& v) {
#pragma unroll
  for (int c = 0; c < VEC; ++c) __builtin_nontemporal_store(v.v[c], p + c);
}
// Output rows of the per-occurrence lookup (B rows, written once, read by the next consumer from
// HBM): write-through (sc1) 16-byte stores — the rows leave the XCD's L2 while the launch runs
// instead of sitting there dirty until its end, when the next launch's start waits for them to drain
// (21.7 MB per launch at the bench's shape: step_fwd 13.9 -> 13.4 us, same-box A/B r03f).  NOT for
// the multi-table forward's scatter (one id's row to many positions): there the same store form
// costs 160 -> 204 us.
template <int VEC>
__device__ __forceinline__ void store_out_rows(float* p, const Vec<VEC>& v) {
  if constexpr (VEC == 4) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 t;
    t.x = v.v[0]; t.y = v.v[1]; t.z = v.v[2]; t.w = v.v[3];
    // (s_nop: the data registers must not be overwritten before the store has read them,
    // cdna_hip_programming.md 5.7)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"((MHTE_GLOBAL float*)p), "v"(t) : "memory");
  } else {
    store_stream<VEC>(p, v);
  }
}
// NT: 0 plain stores, 1 streaming (nontemporal), 2 write-through (store_out_rows)
template <int G, int VEC, int UNR, int NT>
__device__ __forceinline__ void lookup_role_u(const TableView& tv, const int64_t* __restrict__ ids,
                                              int64_t n, const uint32_t* __restrict__ n_dev,
                                              float* __restrict__ out, int count_hits,
                                              int64_t group, const int64_t* pre = nullptr) {
  const int lane = threadIdx.x & 63;
  const int j = lane & (G - 1);
  const int gbase = lane & ~(G - 1);
  if (n_dev) n = min(n, int64_t(*n_dev));
  const int64_t g0 = group * UNR;
  if (g0 >= n) return;
  // one 8-byte load per lane for the group's ids, then broadcast
  // (pre: the lane's id, fetched by the caller ahead of a wait of its own — step_fwd's gate)
  const int64_t myid = pre ? *pre : ((j < UNR && g0 + j < n) ? ids[g0 + j] : 0);
  int64_t id[UNR];
  bool valid[UNR], match[UNR];
  uint32_t row[UNR];
  // (every lane loads a slot of one of the id's two buckets — lines that are fetched anyway; an
  // index past the end probes the buckets of id 0 — and the result is masked afterwards: probes
  // under `if (valid ...)` are waited for where the branch ends, one id after the other)
  int64_t kk[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    id[u] = __shfl(myid, gbase + (u & (G - 1)));
    valid[u] = g0 + u < n;
    const uint64_t hv = hash_key(id[u]);
    const uint64_t i1 = index_hash(tv.hp, hv);
    const uint64_t i2 = alt_index(tv.hp, partial_key(hv), i1);
    const GBucket* b = global_bucket(tv.buckets + ((j & 4) ? i2 : i1));
    kk[u] = b->key[j & 3];
    row[u] = b->row[j & 3];
  }
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const bool use = valid[u] && j < 8 && id[u] != kEmptyKey;
    match[u] = use && kk[u] == id[u];
    row[u] = use ? row[u] : kNoRow;
  }
  bool found[UNR];
  const float* rp[UNR];
  uint64_t hits = 0;
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const uint64_t m = group_mask_of<G>(__ballot(match[u]), gbase);
    found[u] = m != 0;
    const int src = found[u] ? (__ffsll(static_cast<long long>(m)) - 1) : 0;
    uint32_t r = __shfl(row[u], gbase + src);
    if (valid[u] && id[u] == kEmptyKey) {
      found[u] = tv.ctr->special_state == 1;
      r = tv.ctr->special_row;
    }
    found[u] = found[u] && valid[u];
    rp[u] = found[u] ? row_ptr(tv, r) : nullptr;
    if (count_hits) hits += __popcll(__ballot(found[u] && j == 0));
  }
  for (uint32_t e = j * VEC; e < tv.dim; e += G * VEC) {
    Vec<VEC> v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (found[u]) {
        v[u].load(rp[u] + e);
      } else {
#pragma unroll
        for (int c = 0; c < VEC; ++c) v[u].v[c] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (valid[u]) {
        float* op = out + (g0 + u) * int64_t(tv.dim) + e;
        if (NT == 2) store_out_rows<VEC>(op, v[u]);
        else if (NT == 1) store_stream<VEC>(op, v[u]);
        else v[u].store(op);
      }
    }
  }
  if (count_hits && hits && lane == __ffsll(static_cast<long long>(__ballot(1))) - 1)
    atomicAdd(&tv.ctr->hits, (unsigned long long)hits);
}
template <int G, int VEC, int UNR, int NT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void lookup_kernel_u(TableView tv,
                                                         const int64_t* __restrict__ ids, int64_t n,
                                                         const uint32_t* __restrict__ n_dev,
                                                         float* __restrict__ out, int count_hits) {
  WaveTrace wt(tv.trace);
  lookup_role_u<G, VEC, UNR, NT>(tv, ids, n, n_dev, out, count_hits,
                                 (int64_t(blockIdx.x) * BLOCK + threadIdx.x) / G);
  wt.end(5u);
}

template <int G, int VEC>
__global__ __launch_bounds__(256) void lookup_kernel(TableView tv, const int64_t* __restrict__ ids,
                                                     int64_t n, const uint32_t* __restrict__ n_dev,
                                                     float* __restrict__ out, int count_hits) {
  WaveTrace wt(tv.trace);
  lookup_role<G, VEC>(tv, ids, n, n_dev, out, count_hits, blockIdx.x);
  wt.end(5u);
}


// =============================================================================================
// Row update shared by the fast path and the slow path.  The G lanes of a group own elements
// e = j*VEC + k*G*VEC of the row; weights and optimizer state stay in registers across all
// occurrences of the id, so a duplicated id costs one row read + one row write.
//   rp        row pointer
//   is_new    row was just allocated: start from initializer + optimizer Init
//             (entry_accessor.cc:158-162) instead of reading HBM
//   values    [*, dim] value rows (gradients / assigned values), indexed by occurrence position
//   seg_pos   optional occurrence list (positions in occurrence order), [q0, q1)
// =============================================================================================
// GroupAdaGrad (group_adagrad_optimizer.cc:50-93) on segment k of a row: the one optimizer whose
// step needs the whole segment — the largest squared (decayed) gradient feeds the accumulator,
// and the group-lasso shrinkage needs the norm of the intermediate vector.  The G lanes of the
// group cooperate: max by butterfly, the norm as ONE chain in element order (every lane adds the
// same broadcast values), so it is the reference's sequential sum bit for bit.  The vector is
// written in place between the passes, as the reference does.  params: {initial_accumulator_value,
// beta, l2_regularization_strength, weight_decay_factor}.
template <int G, int VEC>
__device__ __forceinline__ void group_adagrad_segment(const TableView& tv, float* rp, bool is_new,
                                                      int j, const float* __restrict__ values,
                                                      const uint32_t* __restrict__ seg_pos,
                                                      uint32_t q0, uint32_t q1, int64_t self_pos,
                                                      const ApplyArgs& a, uint32_t k) {
  const SegDesc sd = tv.seg[k];
  const int gbase = (threadIdx.x & 63) & ~(G - 1);
  const int64_t dim = tv.dim;
  const uint32_t lo = uint32_t(sd.w_off), hi = uint32_t(sd.w_off + sd.dim);
  const uint32_t t0 = lo / (G * VEC), t1 = (hi + G * VEC - 1) / (G * VEC);  // trips that touch it
  const float beta = sd.p[1], l2 = sd.p[2], wd = sd.p[3], lr0 = a.lr[k];
  float* sc = rp + sd.st_off;
  float gss = is_new ? sd.p[0] : sc[0];
  if (is_new) {
    for (uint32_t t = t0; t < t1; ++t) {
      const uint32_t e = uint32_t(j) * VEC + t * G * VEC;
      if (e >= lo && e < hi) {
        Vec<VEC> w;
#pragma unroll
        for (int c = 0; c < VEC; ++c) w.v[c] = init_weight(sd, rp + e + c);
        w.store(rp + e);
      }
    }
  }
  const uint32_t nq = seg_pos ? (q1 - q0) : 1u;
  const uint32_t steps = a.sum_dups ? 1u : nq;
  for (uint32_t s = 0; s < steps; ++s) {
    // gradient of this step for the chunk at e: one occurrence, or the occurrences added in order
    auto grad_of = [&](uint32_t e, Vec<VEC>& g) {
      if (a.sum_dups) {
        vec_zero(g);
        for (uint32_t q = 0; q < nq; ++q) {
          const int64_t pos = seg_pos ? int64_t(seg_pos[q0 + q]) : self_pos;
          Vec<VEC> v;
          v.load(values + pos * dim + e);
#pragma unroll
          for (int c = 0; c < VEC; ++c) g.v[c] = g.v[c] + v.v[c];
        }
      } else {
        const int64_t pos = seg_pos ? int64_t(seg_pos[q0 + s]) : self_pos;
        g.load(values + pos * dim + e);
      }
    };
    // pass 1: largest squared decayed gradient
    float mx = 0.f;
    for (uint32_t t = t0; t < t1; ++t) {
      const uint32_t e = uint32_t(j) * VEC + t * G * VEC;
      if (e >= lo && e < hi) {
        Vec<VEC> w, g;
        w.load(rp + e);
        grad_of(e, g);
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
          const float t2 = wd * w.v[c];
          const float gd = g.v[c] + t2;
          const float sq = gd * gd;
          if (sq > mx) mx = sq;
        }
      }
    }
#pragma unroll
    for (int o = G / 2; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    gss = gss + mx;
    const float lr = lr0 / (beta + sqrtf(gss));
    // pass 2: z = g_decayed - w / lr in place; ||z||^2 as one chain in element order
    float zn = 0.f;
    for (uint32_t t = t0; t < t1; ++t) {
      const uint32_t e = uint32_t(j) * VEC + t * G * VEC;
      const bool in = e >= lo && e < hi;
      Vec<VEC> z;
      vec_zero(z);
      if (in) {
        Vec<VEC> w, g;
        w.load(rp + e);
        grad_of(e, g);
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
          const float t2 = wd * w.v[c];
          const float gd = g.v[c] + t2;
          const float q = w.v[c] / lr;
          z.v[c] = gd - q;
        }
        z.store(rp + e);
      }
      for (int j2 = 0; j2 < G; ++j2) {
        const uint32_t e2 = uint32_t(j2) * VEC + t * G * VEC;
        const bool in2 = e2 >= lo && e2 < hi;  // group-uniform
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
          const float zz = __shfl(z.v[c], gbase + j2);
          if (in2) {
            const float sq = zz * zz;
            zn = zn + sq;
          }
        }
      }
    }
    const float z_norm = sqrtf(zn);
    const bool zero = z_norm < l2;
    const float num = -lr * (z_norm - l2);
    const float coeff = num / z_norm;
    // pass 3: shrink
    for (uint32_t t = t0; t < t1; ++t) {
      const uint32_t e = uint32_t(j) * VEC + t * G * VEC;
      if (e >= lo && e < hi) {
        Vec<VEC> z;
        z.load(rp + e);
#pragma unroll
        for (int c = 0; c < VEC; ++c) z.v[c] = zero ? 0.f : coeff * z.v[c];
        z.store(rp + e);
      }
    }
  }
  if (j == 0) {
    sc[0] = gss;
    sc[1] = 0.f;
    sc[2] = 0.f;
    sc[3] = 0.f;
  }
}

// BASIC: the table is known to use SGD / Adagrad / FTRL only (the fused training-step kernels:
// Table::fusable) — the other optimizers are compiled out, which is what keeps the displacement
// role of step_fwd inside that kernel's register budget.
// GROUP = false: the table has no GroupAdaGrad segment (the host picks the instantiation): the
// whole-segment pass is compiled out — with it, the compiler keeps the table descriptor in scratch
// memory and every other optimizer's update runs three times slower.
template <int G, int VEC, int OP, bool BASIC = false, bool GROUP = true>
__device__ __forceinline__ void apply_row(const TableView& tv, float* rp, bool is_new, int j,
                                          const float* __restrict__ values,
                                          const uint32_t* __restrict__ seg_pos, uint32_t q0,
                                          uint32_t q1, int64_t self_pos, const ApplyArgs& a) {
  const int64_t dim = tv.dim;
  if (OP == kOpOptimize && !BASIC && GROUP) {  // (group-uniform: every lane walks the segments)
    for (uint32_t k = 0; k < tv.nseg; ++k)
      if (tv.seg[k].opt == kOptGroupAdagrad)
        group_adagrad_segment<G, VEC>(tv, rp, is_new, j, values, seg_pos, q0, q1, self_pos, a, k);
  }
  for (uint32_t e = j * VEC; e < tv.dim; e += G * VEC) {
    uint32_t k = 0;
    const SegDesc sd = seg_of<false>(tv, e, k);
    const uint32_t le = e - sd.w_off;  // element index inside the segment
    const bool gag = !BASIC && GROUP && sd.opt == kOptGroupAdagrad;
    if (gag && OP == kOpOptimize) continue;  // done above, by the whole group
    const float lr = a.lr[k];
    const int nv = BASIC ? (sd.opt == kOptFtrl ? 2 : (sd.opt == kOptAdagrad ? 1 : 0)) : opt_vectors(sd.opt);
    const bool scal = !BASIC && opt_scalars(sd.opt) != 0;
    Vec<VEC> w, s1, s2, s3;
    float* st1 = rp + sd.st_off + le;
    float* st2 = st1 + sd.dim;
    float* st3 = st2 + sd.dim;
    float* sc = rp + sd.st_off + nv * sd.dim;  // adam / amsgrad: {beta1_power, beta2_power}
    float c1 = 0.f, c2 = 0.f;
    const bool bsm = !BASIC && sd.opt == kOptBatchSoftmax;  // the slot holds the id's last global step
    long long last_step = 0;
    if (is_new || OP == kOpReinit) {
#pragma unroll
      for (int c = 0; c < VEC; ++c) {
        w.v[c] = init_weight(sd, rp + e + c);
        s1.v[c] = opt_state_init(sd, 0);
        s2.v[c] = opt_state_init(sd, 1);
        s3.v[c] = opt_state_init(sd, 2);
      }
      c1 = sd.p[0];  // adam_optimizer.cc:52-53: the powers start at beta1, beta2
      c2 = sd.p[1];
    } else {
      if (OP != kOpAssign) w.load(rp + e);
      if (OP == kOpOptimize) {
        if (nv > 0) s1.load(st1);
        if (nv > 1) s2.load(st2);
        if (nv > 2) s3.load(st3);
        if (scal) {
          c1 = sc[0];
          c2 = sc[1];
        }
        if (bsm)
          last_step = static_cast<long long>(
              (static_cast<unsigned long long>(__float_as_uint(sc[1])) << 32) | __float_as_uint(sc[0]));
      }
    }
    if (OP == kOpAssign) {
      // sequential memcpy per occurrence: the last one wins (cuckoo_embedding_hash_table.cc:186-203)
      const int64_t pos = seg_pos ? int64_t(seg_pos[q1 - 1]) : self_pos;
      w.load(values + pos * dim + e);
    } else if (OP == kOpAssignAdd || OP == kOpOptimize) {
      Vec<VEC> acc;
#pragma unroll
      for (int c = 0; c < VEC; ++c) acc.v[c] = 0.f;
      const uint32_t nq = seg_pos ? (q1 - q0) : 1u;
      for (uint32_t t = 0; t < nq; ++t) {
        const int64_t pos = seg_pos ? int64_t(seg_pos[q0 + t]) : self_pos;
        Vec<VEC> v;
        v.load(values + pos * dim + e);
        if (OP == kOpOptimize && a.sum_dups) {
#pragma unroll
          for (int c = 0; c < VEC; ++c) acc.v[c] = acc.v[c] + v.v[c];
          if (t + 1 < nq) continue;
          v = acc;
        }
        const float lr_eff = scal ? adam_lr(lr, c1, c2) : lr;
        // (the AVX form of Adagrad is decided once per row, not per element: a branch per element
        // took the segment kernels from 128 to 131 VGPRs = 4 -> 3 wavefronts per SIMD)
        if (OP != kOpAssignAdd && sd.opt == kOptAdagrad && MHTE_AVX_FORM(sd)) {
#pragma unroll
          for (int c = 0; c < VEC; ++c)
            adagrad_step_avx(w.v[c], s1.v[c], v.v[c], lr, sd.p[1], uint32_t(le + c) < (uint32_t(sd.dim) & ~7u));
        } else
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
          if (OP == kOpAssignAdd) {
            w.v[c] = w.v[c] + v.v[c];  // entry_accessor.cc:179-185
          } else {
            if (BASIC) {
              if (sd.opt == kOptSgd) w.v[c] = sgd_step(w.v[c], v.v[c], lr);
              else if (sd.opt == kOptAdagrad) adagrad_step(w.v[c], s1.v[c], v.v[c], lr, sd.p[1]);
              else ftrl_step(w.v[c], s1.v[c], s2.v[c], v.v[c], lr, sd.p[1], sd.p[2], sd.p[3]);
            } else
            switch (sd.opt) {
              case kOptSgd: w.v[c] = sgd_step(w.v[c], v.v[c], lr); break;
              case kOptAdagrad: adagrad_step(w.v[c], s1.v[c], v.v[c], lr, sd.p[1]); break;
              case kOptFtrl:
                ftrl_step(w.v[c], s1.v[c], s2.v[c], v.v[c], lr, sd.p[1], sd.p[2], sd.p[3]);
                break;
              case kOptMomentum:
                momentum_step(w.v[c], s1.v[c], v.v[c], lr, sd.p[0], sd.p[1], sd.p[2] != 0.f);
                break;
              case kOptAdadelta:
                adadelta_step(w.v[c], s1.v[c], s2.v[c], v.v[c], lr, sd.p[0], sd.p[1], sd.p[2]);
                break;
              case kOptRmsprop:
                rmsprop_step(w.v[c], s1.v[c], v.v[c], double(sd.p[2]), sd.p[0], sd.p[1], false);
                break;
              case kOptRmspropV2:
                rmsprop_step(w.v[c], s1.v[c], v.v[c], double(lr), sd.p[0], sd.p[1], true);
                break;
              case kOptAdam:
                adam_step(w.v[c], s1.v[c], s2.v[c], nullptr, v.v[c], lr_eff, sd.p[0], sd.p[1],
                          sd.p[2], sd.p[3], sd.p[4] != 0.f);
                break;
              case kOptMovingAverage: w.v[c] = moving_average_step(w.v[c], v.v[c], sd.p[0]); break;
              case kOptBatchSoftmax: batch_softmax_step(w.v[c], last_step, lr, a.global_step); break;
              default:  // kOptAmsgrad
                adam_step(w.v[c], s1.v[c], s2.v[c], &s3.v[c], v.v[c], lr_eff, sd.p[0], sd.p[1],
                          sd.p[2], sd.p[3], sd.p[4] != 0.f);
                break;
            }
          }
        }
        if (OP == kOpOptimize && scal) {  // one Optimize() call done: the powers move on
          c1 = c1 * sd.p[0];
          c2 = c2 * sd.p[1];
        }
        if (OP == kOpOptimize && !BASIC && sd.sr16) {   // the fp16 stochastic-rounding decorator
#pragma unroll
          for (int c = 0; c < VEC; ++c)
            w.v[c] = stochastic_round(w.v[c], sr_draw(rp + e + c, w.v[c], a.ts, t));
        }
      }
    }
    w.store(rp + e);
    if (is_new || OP == kOpReinit || OP == kOpOptimize) {
      if (nv > 0) s1.store(st1);
      if (nv > 1) s2.store(st2);
      if (nv > 2) s3.store(st3);
      if (scal && le == 0) {  // (every lane of the segment computed the same powers)
        sc[0] = c1;
        sc[1] = c2;
        sc[2] = 0.f;
        sc[3] = 0.f;
      }
      if (gag && le == 0 && (is_new || OP == kOpReinit)) {  // GroupAdaGrad Init()
        sc[0] = sd.p[0];
        sc[1] = 0.f;
        sc[2] = 0.f;
        sc[3] = 0.f;
      }
      if (bsm && le == 0) {
        sc[0] = __uint_as_float(static_cast<uint32_t>(static_cast<unsigned long long>(last_step)));
        sc[1] = __uint_as_float(static_cast<uint32_t>(static_cast<unsigned long long>(last_step) >> 32));
        sc[2] = 0.f;
        sc[3] = 0.f;
      }
    }
  }
}

// =============================================================================================
// Upsert + apply, fast path.  Precondition: ids[0..n) are pairwise distinct (callers dedup first,
// or go through the in-op grouping that supplies seg_off/seg_pos).  Phase-separated from lookups by
// stream order, so the only concurrency is insert vs insert: a slot is claimed with one 64-bit
// CAS kEmptyKey -> id on the key word.  An id whose two buckets are full is appended to `pending`
// and finished by slowpath_kernel (displacement must run alone).
// Algorithmic bytes per unique id: 8 (id) + 36..72 (probe) + 4 (ts) + 4*dim (value)
//   + 2 * 4 * row_floats (row read-modify-write).
// =============================================================================================
// Probe + insert for ONE id per G-lane group; every lane of the wavefront must call it (it uses
// wave ballots).  The caller has already issued the slot loads: lane j < 8 of the group holds
// key `k` / handle `row` of slot (j & 3) of bucket i1 (j < 4) or i2 (j >= 4), `b` points at that
// bucket.  On return: `r` row handle, `is_new` (row just allocated: start from the initializer),
// `deferred` (both buckets full: caller queues the id for slowpath_kernel).  Timestamps and new
// handles are written here.
struct SlotResult {
  uint32_t r;
  bool is_new;
  bool deferred;
};

// a wave-uniform 32-bit value as a store operand, copied to its VGPR where it is used (a copy made
// in front of a loop gets spilled there in the 96-VGPR kernels and reloaded — behind a wait for every
// load in flight — at the store)
__device__ __forceinline__ uint32_t vgpr_copy_of_uniform(uint32_t u) {
#ifndef MHTE_NO_ANTIHOIST
  uint32_t v;
  asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(__builtin_amdgcn_readfirstlane(u)));
  return v;
#else
  return u;
#endif
}
template <int G>
__device__ __forceinline__ SlotResult upsert_resolve(const TableView& tv, Bucket* b_generic, int64_t id,
                                                     bool valid, int64_t k, uint32_t row, int lane,
                                                     uint32_t ts, uint32_t reserved = kNoRow) {
  const int j = lane & (G - 1);
  const int gbase = lane & ~(G - 1);
  const int s = j & 3;
  GBucket* b = global_bucket(b_generic);
  const bool special = valid && id == kEmptyKey;
  const bool prober = valid && !special && j < 8;
  uint64_t m = group_mask_of<G>(__ballot(prober && k == id), gbase);
  bool found = m != 0;
  bool is_new = false;
  bool deferred = false;
  int owner = found ? (__ffsll(static_cast<long long>(m)) - 1) : -1;  // lane (in group) of the slot

  // ---- insert: claim the LAST empty slot of b1, else of b2 (cuckoohash_map.hpp:1398-1418) ----
  bool need = valid && !special && !found;
  while (__any(need)) {
    const uint64_t em = group_mask_of<G>(__ballot(prober && k == kEmptyKey), gbase) & 0xffull;
    int pick = -1;
    if (need) {
      const uint32_t m1 = uint32_t(em) & 0xfu, m2 = (uint32_t(em) >> 4) & 0xfu;
      if (m1) pick = 31 - __clz(m1);
      else if (m2) pick = 4 + (31 - __clz(m2));
      if (pick < 0) {  // both buckets full -> slow path
        deferred = true;
        need = false;
      }
    }
    bool won = false;
    if (need && j == pick) {
      const unsigned long long old = cas_key(&b->key[s], kEmptyKey, id);
      won = (static_cast<int64_t>(old) == kEmptyKey);
      k = won ? id : static_cast<int64_t>(old);
    }
    const uint64_t wm = group_mask_of<G>(__ballot(won), gbase);
    if (need && wm) {
      owner = pick;
      is_new = true;
      need = false;
    }
  }

  // ---- side slot for the one key that cannot live in a bucket ----
  if (special) {
    unsigned int st = 0;
    if (j == 0) st = atomicExch(&tv.ctr->special_state, 1u);
    st = __shfl(st, gbase);
    found = true;
    is_new = (st == 0);
  }

  // ---- row handles + live-key count for new ids: ONE atomic per wave ----
  // (reserved != kNoRow: the row was allocated — and the key counted — ahead, by the build role (ProbeOut);
  // such a group needs nothing from the counter and returns its key if it did not insert)
  const bool has_res = reserved != kNoRow;
  const bool leader_new = is_new && j == 0 && !has_res;
  const uint64_t newm = __ballot(leader_new);
  const uint64_t keym = __ballot(leader_new && !special);
  {
    const uint64_t back = __ballot(valid && has_res && j == 0 && !is_new);
    if (back && lane == __ffsll(static_cast<long long>(back)) - 1)
      atomicAdd(&tv.ctr->alloc, ~((static_cast<unsigned long long>(__popcll(back)) << 32) - 1ull));
  }
  const int first_new = newm ? (__ffsll(static_cast<long long>(newm)) - 1) : 0;
  uint32_t base_row = 0;
  if (newm && lane == first_new) {
    const unsigned long long add =
        (static_cast<unsigned long long>(__popcll(keym)) << 32) | (unsigned long long)__popcll(newm);
    base_row = static_cast<uint32_t>(atomicAdd(&tv.ctr->alloc, add));
  }
  base_row = __shfl(base_row, first_new);
  const uint32_t found_row = __shfl(row, gbase + (owner < 0 ? 0 : owner));
  uint32_t r;
  if (is_new) {
    const uint32_t rank = __popcll(newm & ((uint64_t(1) << gbase) - 1));
    r = has_res ? reserved : base_row + rank;
  } else {
    r = found_row;
  }
  if (special) {
    if (is_new) {
      if (j == 0) tv.ctr->special_row = r;
    } else {
      r = tv.ctr->special_row;  // written by an earlier kernel
    }
    if (j == 0) tv.ctr->special_ts = ts;
  } else if (valid && !deferred && j == owner) {
    if (is_new) b->row[s] = r;
    b->ts[s] = ts;  // SetTimestamp(update_time), cuckoo_embedding_hash_table.cc:242-246
                    // (per lane: a restore passes every id's own timestamp)
  }
  SlotResult out;
  out.r = r;
  out.is_new = is_new;
  out.deferred = deferred;
  return out;
}

template <int G, int VEC, int OP, bool GROUP = false>
__global__ __launch_bounds__(256) void upsert_kernel(TableView tv, const int64_t* __restrict__ ids,
                                                     int64_t n, const uint32_t* __restrict__ n_dev,
                                                     const float* __restrict__ values,
                                                     const uint32_t* __restrict__ seg_off,
                                                     const uint32_t* __restrict__ seg_pos,
                                                     ApplyArgs a, int32_t* __restrict__ status,
                                                     uint32_t* __restrict__ pending,
                                                     uint32_t* __restrict__ skip) {
  const int lane = threadIdx.x & 63;
  const int j = lane & (G - 1);
  const int64_t g = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  if (n_dev) n = min(n, int64_t(*n_dev));
  bool valid = g < n;
  const int64_t id = valid ? ids[g] : 0;
  const uint64_t hv = hash_key(id);
  const uint64_t i1 = index_hash(tv.hp, hv);
  const uint64_t i2 = alt_index(tv.hp, partial_key(hv), i1);

  Bucket* b = assume_global(tv.buckets + ((j < 4) ? i1 : i2));
  int64_t k = kEmptyKey;
  uint32_t row = kNoRow;
  if (valid && id != kEmptyKey && j < 8) {
    k = b->key[j & 3];
    row = b->row[j & 3];
  }
  uint32_t q0 = (valid && seg_off) ? seg_off[g] : 0u;
  const uint32_t q1 = (valid && seg_off) ? seg_off[g + 1] : 1u;
  // ---- admission filter: may drop the first occurrences of an id that is not in the table yet
  if (tv.flt_slots && a.filter_mode) {
    const int gbase = lane & ~(G - 1);
    bool contained = group_mask_of<G>(__ballot(valid && id != kEmptyKey && j < 8 && k == id), gbase) != 0;
    if (valid && id == kEmptyKey) contained = tv.ctr->special_state == 1;
    uint32_t first = 0;
    if (valid && j == 0) {
      const int mode = (a.filter_mode == 1 && a.sum_dups) ? 2 : a.filter_mode;
      first = filter_consult(tv, id, q1 - q0, mode, contained);
    }
    first = __shfl(first, gbase);
    if (first >= q1 - q0) valid = false;  // every occurrence dropped: the id is not inserted
    q0 += first;
  }
  const SlotResult sr = upsert_resolve<G>(tv, b, id, valid, k, row, lane, a.ts);
  // ---- defer to the slow path ----
  if (sr.deferred && j == 0) {
    const uint32_t slot = atomicAdd(&tv.ctr->n_pending, 1u);
    pending[slot] = static_cast<uint32_t>(g);
    if (skip) skip[g] = q0;
  }
  // ---- apply ----
  if (valid && !sr.deferred) {
    apply_row<G, VEC, OP, false, GROUP>(tv, row_ptr(tv, sr.r), sr.is_new, j, values, seg_off ? seg_pos : nullptr,
                          q0, q1, g, a);
    if (OP == kOpReinit && j == 0) {
      // status: 0 inserted, 1 existed (cuckoo_embedding_hash_table.cc:215-226); later duplicates
      // of an id always see it existing.
      if (seg_off) {
        for (uint32_t q = q0; q < q1; ++q) status[seg_pos[q]] = (q == q0 && sr.is_new) ? 0 : 1;
      } else {
        status[g] = sr.is_new ? 0 : 1;
      }
    }
  }
}

// =============================================================================================
// Slow path: ONE wavefront finishes the ids the fast path deferred.  Lane 0 runs the reference's
// BFS displacement serially (mhte_core.h), then all 64 lanes apply the row update.  Rare by
// construction: the host keeps the load factor <= max_load_factor (default 0.5), where both
// 4-slot buckets of a fresh id are full with probability ~1e-4.
// =============================================================================================
// SOLO: the role is the whole (64-thread) workgroup; otherwise it is wave 0 of a larger one
// and must not use workgroup barriers (lane 0 alone reads and writes q/path and the buckets).
// Wave-parallel form of slot_search (mhte_core.h; cuckoohash_map.hpp:1725-1762) with the SAME
// result: the BFS queue is processed level by level, 64 queue entries per round, every lane
// loading one candidate bucket.  The serial search returns at the first queue entry (in queue
// order) that has an empty slot; entries of one depth are independent of each other and, when
// none of a round has an empty slot, each pushes exactly its four children in slot order — so
// child positions are arithmetic and the first hit in lane order is the serial answer.  A
// displacement then costs ~one memory round trip per BFS level instead of one per bucket looked
// at (each is a random HBM line: ~3 us on a table of this size).
__device__ __forceinline__ BfsSlot slot_search_wave(const Bucket* buckets, uint32_t hp, uint64_t i1,
                                                    uint64_t i2, BfsSlot* q, int lane) {
  if (lane == 0) {
    q[0] = BfsSlot{i1, 0, 0};
    q[1] = BfsSlot{i2, 1, 0};
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  int first = 0, last = 2;
  for (int depth = 0; depth < kMaxBfsPathLen; ++depth) {
    const int next0 = last;  // children of this level start here
    for (int c = first; c < last; c += 64) {
      const int idx = c + lane;
      const bool has = idx < last;
      BfsSlot x = has ? q[idx] : BfsSlot{0, 0, 0};
      int64_t key[kSlots];
#pragma unroll
      for (int s = 0; s < kSlots; ++s) key[s] = has ? buckets[x.bucket].key[s] : 0;
      const int starting_slot = x.pathcode % kSlots;
      int empty_i = -1;  // first empty slot in scan order
#pragma unroll
      for (int i = kSlots - 1; i >= 0; --i) {
        const int slot = (starting_slot + i) % kSlots;
        int64_t k = key[0];
#pragma unroll
        for (int s = 1; s < kSlots; ++s) k = (slot == s) ? key[s] : k;
        if (k == kEmptyKey) empty_i = i;
      }
      const uint64_t fm = __ballot(has && empty_i >= 0);
      if (fm) {
        const int src = __ffsll(static_cast<long long>(fm)) - 1;
        const int slot = (starting_slot + empty_i) % kSlots;
        BfsSlot r = x;
        r.pathcode = static_cast<uint16_t>(x.pathcode * kSlots + slot);
        BfsSlot out;
        out.bucket = __shfl(r.bucket, src);
        out.pathcode = static_cast<uint16_t>(__shfl(int(r.pathcode), src));
        out.depth = static_cast<int8_t>(__shfl(int(r.depth), src));
        return out;
      }
      if (has && depth < kMaxBfsPathLen - 1) {
#pragma unroll
        for (int i = 0; i < kSlots; ++i) {
          const int slot = (starting_slot + i) % kSlots;
          int64_t k = key[0];
#pragma unroll
          for (int s = 1; s < kSlots; ++s) k = (slot == s) ? key[s] : k;
          BfsSlot y;
          y.bucket = alt_index(hp, partial_key(hash_key(k)), x.bucket);
          y.pathcode = static_cast<uint16_t>(x.pathcode * kSlots + slot);
          y.depth = static_cast<int8_t>(depth + 1);
          q[next0 + (idx - first) * kSlots + i] = y;
        }
      }
    }
    if (depth == kMaxBfsPathLen - 1) break;
    const int n = last - first;
    first = last;
    last = next0 + n * kSlots;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  return BfsSlot{0, 0, -1};
}

// serial_insert_slot (mhte_core.h) with the path search done by the whole wavefront; every lane
// calls it, lane 0 alone writes the buckets.  Same placement as the serial form.
__device__ __forceinline__ long long wave_insert_slot(Bucket* buckets, uint32_t hp, int64_t key,
                                                      BfsSlot* q, CuckooRecord* path, int lane) {
  const uint64_t hv = hash_key(key);
  const uint32_t partial = partial_key(hv);
  const uint64_t i1 = index_hash(hp, hv);
  const uint64_t i2 = alt_index(hp, partial, i1);
  long long pos = -2;
  if (lane == 0) {  // try_find_insert_bucket, :1398-1418 — last empty slot of b1, else of b2
    for (int pass = 0; pass < 2 && pos == -2; ++pass) {
      const uint64_t ib = pass == 0 ? i1 : i2;
      int found = -1;
      for (int s = 0; s < kSlots; ++s)
        if (!slot_occupied(buckets[ib], s)) found = s;
      if (found >= 0) {
        buckets[ib].key[found] = key;
        pos = static_cast<long long>(ib * kSlots + found);
      }
    }
  }
  pos = __shfl(pos, 0);
  if (pos != -2) return pos;
  for (int attempt = 0; attempt < 64; ++attempt) {  // (single owner: the first path always moves)
    BfsSlot x = slot_search_wave(buckets, hp, i1, i2, q, lane);
    int done = 0;  // 1 placed, -1 no path
    if (lane == 0) {
      if (x.depth == -1) {
        done = -1;
      } else {  // cuckoopath_search's path reconstruction, :1508-1561
        const int d0 = x.depth;
        for (int i = x.depth; i >= 0; --i) {
          path[i].slot = x.pathcode % kSlots;
          x.pathcode = static_cast<uint16_t>(x.pathcode / kSlots);
        }
        path[0].bucket = (x.pathcode == 0) ? i1 : i2;
        int depth = d0;
        {
          const Bucket& b = buckets[path[0].bucket];
          if (!slot_occupied(b, path[0].slot)) {
            depth = 0;
          } else {
            path[0].hash = hash_key(b.key[path[0].slot]);
            path[0].partial = partial_key(path[0].hash);
            for (int i = 1; i <= d0; ++i) {
              path[i].bucket = alt_index(hp, path[i - 1].partial, path[i - 1].bucket);
              const Bucket& bi = buckets[path[i].bucket];
              if (!slot_occupied(bi, path[i].slot)) {
                depth = i;
                break;
              }
              path[i].hash = hash_key(bi.key[path[i].slot]);
              path[i].partial = partial_key(path[i].hash);
            }
          }
        }
        if (cuckoopath_move(buckets, path, depth)) {
          buckets[path[0].bucket].key[path[0].slot] = key;
          pos = static_cast<long long>(path[0].bucket * kSlots + path[0].slot);
          done = 1;
        }
      }
    }
    done = __shfl(done, 0);
    if (done == -1) return -1;
    if (done == 1) return __shfl(pos, 0);
  }
  return -1;
}

// GATED: other workgroups of the same launch wait for n_pending == 0 before they touch the table
// (lookup_role's gate): the pass ends with an agent-scope release (its bucket and row stores are
// written back from this XCD's L2) followed by an agent-scope store of the 0.
// BASIC (gated launches = the fused step kernels): the update code of SGD / Adagrad / FTRL only.
template <int VEC, int OP, bool SOLO, bool GATED = false, bool BASIC = GATED>
__device__ __forceinline__ void slowpath_role(const TableView& tv, const int64_t* __restrict__ ids,
                                              const float* __restrict__ values,
                                              const uint32_t* __restrict__ seg_off,
                                              const uint32_t* __restrict__ seg_pos,
                                              const ApplyArgs& a, int32_t* __restrict__ status,
                                              const uint32_t* __restrict__ pending,
                                              BfsSlot* q, CuckooRecord* path,
                                              const uint32_t* __restrict__ skip = nullptr) {
  const int lane = threadIdx.x;
  // (the count and the first entry of the list are fetched together: the other workgroups of a
  // gated launch wait for this pass, every dependent round trip in it is paid by all of them)
  const uint32_t first = pending[0];
  const uint32_t np = tv.ctr->n_pending;
  if (np == 0) return;
  for (uint32_t i = 0; i < np; ++i) {
    const uint32_t g = i == 0 ? first : pending[i];
    const int64_t id = ids[g];
    // the row handle is allocated while the slot search runs (a search that fails — the table is
    // over its load limit — gives the key back and strands the handle)
    uint32_t r;  // (only lane 0's value is read; merging it with a constant here would make the
                 // compiler wait for the atomic before the search instead of after it)
    if (lane == 0) r = static_cast<uint32_t>(atomicAdd(&tv.ctr->alloc, (1ull << 32) | 1ull));
    long long pos = wave_insert_slot(tv.buckets, tv.hp, id, q, path, lane);
    if (lane == 0) {
      if (pos >= 0) {
        Bucket* b = tv.buckets + (pos >> 2);
        b->row[pos & 3] = r;
        b->ts[pos & 3] = a.ts;
      } else {
        atomicAdd(&tv.ctr->alloc, ~((1ull << 32) - 1ull));  // - (1 << 32): not a live key
        atomicOr(&tv.ctr->error, 1u);
        atomicAdd(&tv.ctr->n_dropped, 1u);
      }
    }
    r = __shfl(r, 0);
    if (pos >= 0) {
      // (skip: first occurrence the admission filter let through, upsert_kernel)
      const uint32_t q0 = (skip && seg_off) ? skip[g] : (seg_off ? seg_off[g] : 0u);
      const uint32_t q1 = seg_off ? seg_off[g + 1] : 1u;
      apply_row<64, VEC, OP, BASIC, (GATED ? false : true)>(tv, row_ptr(tv, r), true, lane, values,
                                                            seg_off ? seg_pos : nullptr, q0, q1, g, a);
      if (OP == kOpReinit && lane == 0) {
        if (seg_off) {
          for (uint32_t t = q0; t < q1; ++t) status[seg_pos[t]] = (t == q0) ? 0 : 1;
        } else {
          status[g] = 0;
        }
      }
    }
    if (SOLO) __syncthreads();  // (lane 0 alone touches q, path and the buckets)
  }
  if (GATED) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0)
      __hip_atomic_store(&tv.ctr->n_pending, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    if (lane == 0) tv.ctr->n_pending = 0;
  }
}
// ---------------------------------------------------------------------------------------------
// The displacement pass with W wavefronts of ONE workgroup working on W pending ids at a time
// (the multi-table step: a table defers 1-5 ids per step and a displacement is ~10 dependent round
// trips, so one wavefront per table was 24-59 us of a 390 us step).  Per round of W ids:
//   A  every wavefront searches a slot for its id WITHOUT writing — the home buckets' empty slot
//      (try_find_insert_bucket, cuckoohash_map.hpp:1398-1418), else the BFS path (:1725-1762) — and
//      publishes the buckets it will write (the path's, <= kMaxBfsPathLen) in LDS;
//   B  a wavefront none of whose buckets is on the list of a LOWER wavefront of the round moves its
//      path and writes its key: the buckets written in one sub-round are pairwise disjoint, and what
//      a path's validity depends on is the content of its own buckets, which nobody has touched
//      since the search.  The others search again after the barrier (the lowest unfinished
//      wavefront is always clear: every sub-round finishes at least one id).
// All wavefronts of a workgroup share one CU's L1, so the barrier (which drains the stores) is all
// the visibility the rounds need.  The result is a valid cuckoo placement of the same key set; which
// of several valid placements depends on the round structure, as the reference's does on its
// threads' interleaving.  Row handles come from the table's counter as they do in the serial pass.
// ---------------------------------------------------------------------------------------------
template <int W>
struct SlowParLds {
  BfsSlot q[W][kMaxCuckooCount];
  CuckooRecord path[W][kMaxBfsPathLen];
  unsigned long long wb[W][8];   // buckets wavefront w is going to write (sub-round)
  int nwb[W];                    // 0: nothing to place in this sub-round
  int again[2];
};

// phase A for one id: 0 = no slot (table over its load limit), 1 = slot found.  path[0..depth] is
// the displacement path (depth 0: path[0] is an empty slot, of a home bucket or at the end of a
// search whose first hop has since become free)
__device__ __forceinline__ int wave_plan_slot(const Bucket* buckets, uint32_t hp, int64_t key,
                                              BfsSlot* q, CuckooRecord* path, int lane, int* depth_out) {
  const uint64_t hv = hash_key(key);
  const uint64_t i1 = index_hash(hp, hv);
  const uint64_t i2 = alt_index(hp, partial_key(hv), i1);
  int found = -1;
  if (lane < 8) {
    const int64_t k = buckets[lane < 4 ? i1 : i2].key[lane & 3];
    found = (k == kEmptyKey) ? 1 : 0;
  }
  const uint64_t fm = __ballot(found == 1) & 0xffull;
  if (fm) {   // last empty slot of b1, else of b2
    const int s = (fm & 0xfull) ? 63 - __builtin_clzll(fm & 0xfull) : (63 - __builtin_clzll(fm)) - 4;
    if (lane == 0) {
      path[0].bucket = (fm & 0xfull) ? i1 : i2;
      path[0].slot = s;
    }
    *depth_out = 0;
    return 1;
  }
  BfsSlot x = slot_search_wave(buckets, hp, i1, i2, q, lane);
  if (x.depth == -1) return 0;
  int depth = 0;
  if (lane == 0) {  // cuckoopath_search's path reconstruction, :1508-1561
    const int d0 = x.depth;
    for (int i = x.depth; i >= 0; --i) {
      path[i].slot = x.pathcode % kSlots;
      x.pathcode = static_cast<uint16_t>(x.pathcode / kSlots);
    }
    path[0].bucket = (x.pathcode == 0) ? i1 : i2;
    depth = d0;
    const Bucket& b = buckets[path[0].bucket];
    if (!slot_occupied(b, path[0].slot)) {
      depth = 0;
    } else {
      path[0].hash = hash_key(b.key[path[0].slot]);
      path[0].partial = partial_key(path[0].hash);
      for (int i = 1; i <= d0; ++i) {
        path[i].bucket = alt_index(hp, path[i - 1].partial, path[i - 1].bucket);
        const Bucket& bi = buckets[path[i].bucket];
        if (!slot_occupied(bi, path[i].slot)) {
          depth = i;
          break;
        }
        path[i].hash = hash_key(bi.key[path[i].slot]);
        path[i].partial = partial_key(path[i].hash);
      }
    }
  }
  *depth_out = __shfl(depth, 0);
  return 1;
}

template <int VEC, int OP, int W, bool BASIC>
__device__ __forceinline__ void slowpath_par_role(const TableView& tv, const int64_t* __restrict__ ids,
                                                  const float* __restrict__ values, const ApplyArgs& a,
                                                  const uint32_t* __restrict__ pending,
                                                  SlowParLds<W>& L) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t np = tv.ctr->n_pending;
  if (np == 0) return;   // (uniform: every thread reads the same word)
  if (threadIdx.x < 2) L.again[threadIdx.x] = 0;
  uint32_t k = 0;        // sub-round counter (parity of the `again` word in use)
  for (uint32_t base = 0; base < np; base += W) {
    const uint32_t i = base + uint32_t(w);
    const bool active = i < np;
    const uint32_t g = active ? pending[i] : 0u;
    const int64_t id = active ? ids[g] : 0;
    uint32_t r;   // (lane 0's value; left uninitialised on the other path: see slowpath_role)
    if (active && lane == 0) r = static_cast<uint32_t>(atomicAdd(&tv.ctr->alloc, (1ull << 32) | 1ull));
    int state = active ? 0 : 2;   // 0 to place, 1 placed, 2 nothing to do / failed
    int attempts = 0;
    for (;; ++k) {
      int depth = 0, have = 0;
      if (state == 0) {
        have = (++attempts <= 64) ? wave_plan_slot(tv.buckets, tv.hp, id, L.q[w], L.path[w], lane, &depth) : 0;
        if (!have) {   // no path of length <= 5: the key is not inserted (reported at the next call)
          state = 2;
          if (lane == 0) {
            atomicAdd(&tv.ctr->alloc, ~((1ull << 32) - 1ull));
            atomicOr(&tv.ctr->error, 1u);
            atomicAdd(&tv.ctr->n_dropped, 1u);
          }
        }
      }
      if (lane == 0) {
        L.nwb[w] = have ? depth + 1 : 0;
        for (int e = 0; e <= depth && have; ++e) L.wb[w][e] = L.path[w][e].bucket;
      }
      __syncthreads();
      if (state == 0) {
        // lane -> (lower wavefront, entry of its list) against this wavefront's buckets
        const int ow = lane >> 3, oe = lane & 7;
        bool hit = false;
        if (ow < w && ow < W && oe < L.nwb[ow]) {
          const unsigned long long ob = L.wb[ow][oe];
          for (int e = 0; e <= depth; ++e) hit |= (L.wb[w][e] == ob);
        }
        const bool clear = __ballot(hit) == 0ull;
        int ok = 0;
        if (clear && lane == 0) {
          CuckooRecord* path = L.path[w];
          if (cuckoopath_move(tv.buckets, path, depth)) {
            Bucket* b = tv.buckets + path[0].bucket;
            b->row[path[0].slot] = r;
            b->ts[path[0].slot] = a.ts;
            b->key[path[0].slot] = id;
            ok = 1;
          }
        }
        ok = __shfl(ok, 0);
        if (ok) state = 1;
        else if (lane == 0) L.again[k & 1u] = 1;
      }
      __syncthreads();
      const int more = L.again[k & 1u];
      if (threadIdx.x == 0) L.again[(k + 1u) & 1u] = 0;
      if (!more) { ++k; break; }
    }
    if (state == 1) {
      r = __shfl(r, 0);
      apply_row<64, VEC, OP, BASIC, true>(tv, row_ptr(tv, r), true, lane, values, nullptr, 0u, 1u, g, a);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) tv.ctr->n_pending = 0;
}

template <int VEC, int OP>
__global__ __launch_bounds__(64) void slowpath_kernel(TableView tv, const int64_t* __restrict__ ids,
                                                      const float* __restrict__ values,
                                                      const uint32_t* __restrict__ seg_off,
                                                      const uint32_t* __restrict__ seg_pos,
                                                      ApplyArgs a, int32_t* __restrict__ status,
                                                      const uint32_t* __restrict__ pending,
                                                      const uint32_t* __restrict__ skip) {
  __shared__ BfsSlot q[kMaxCuckooCount];
  __shared__ CuckooRecord path[kMaxBfsPathLen];
  slowpath_role<VEC, OP, true>(tv, ids, values, seg_off, seg_pos, a, status, pending, q, path, skip);
}


// =============================================================================================
// Checkpoint restore (cuckoo_embedding_hash_table.cc:299-320; EntryAccessor::Restore,
// entry_accessor.cc:228-239): upsert of distinct ids, the WHOLE row (weights and optimizer state)
// and the row's own timestamp taken from the checkpoint.  values [n, row_floats], ts [n].
// =============================================================================================
template <int G>
__global__ __launch_bounds__(256) void restore_rows_kernel(TableView tv,
                                                           const int64_t* __restrict__ ids,
                                                           int64_t n,
                                                           const float* __restrict__ values,
                                                           const uint32_t* __restrict__ ts,
                                                           uint32_t* __restrict__ pending) {
  const int lane = threadIdx.x & 63;
  const int j = lane & (G - 1);
  const int64_t g = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  const bool valid = g < n;
  const int64_t id = valid ? ids[g] : 0;
  const uint64_t hv = hash_key(id);
  const uint64_t i1 = index_hash(tv.hp, hv);
  const uint64_t i2 = alt_index(tv.hp, partial_key(hv), i1);
  Bucket* b = assume_global(tv.buckets + ((j < 4) ? i1 : i2));
  int64_t k = kEmptyKey;
  uint32_t row = kNoRow;
  if (valid && id != kEmptyKey && j < 8) {
    k = b->key[j & 3];
    row = b->row[j & 3];
  }
  const uint32_t t = valid ? ts[g] : 0u;
  const SlotResult sr = upsert_resolve<G>(tv, b, id, valid, k, row, lane, t);
  if (sr.deferred && j == 0) pending[atomicAdd(&tv.ctr->n_pending, 1u)] = static_cast<uint32_t>(g);
  if (valid && !sr.deferred) {
    float* rp = row_ptr(tv, sr.r);
    const float* vp = values + g * int64_t(tv.row_floats);
    for (uint32_t e = j; e < tv.row_floats; e += G) rp[e] = vp[e];
  }
}
// displacement pass of a restore: ids[pending[i]] get a slot, then the whole row
__global__ __launch_bounds__(64) void restore_slowpath_kernel(TableView tv,
                                                              const int64_t* __restrict__ ids,
                                                              const float* __restrict__ values,
                                                              const uint32_t* __restrict__ ts,
                                                              const uint32_t* __restrict__ pending) {
  __shared__ BfsSlot q[kMaxCuckooCount];
  __shared__ CuckooRecord path[kMaxBfsPathLen];
  const int lane = threadIdx.x;
  const uint32_t np = tv.ctr->n_pending;
  for (uint32_t i = 0; i < np; ++i) {
    const uint32_t g = pending[i];
    const long long pos = wave_insert_slot(tv.buckets, tv.hp, ids[g], q, path, lane);
    uint32_t r = kNoRow;
    if (lane == 0) {
      if (pos >= 0) {
        r = static_cast<uint32_t>(atomicAdd(&tv.ctr->alloc, (1ull << 32) | 1ull));
        Bucket* b = tv.buckets + (pos >> 2);
        b->row[pos & 3] = r;
        b->ts[pos & 3] = ts[g];
      } else {
        atomicOr(&tv.ctr->error, 1u);
        atomicAdd(&tv.ctr->n_dropped, 1u);
      }
    }
    r = __shfl(r, 0);
    if (pos >= 0) {
      float* rp = row_ptr(tv, r);
      const float* vp = values + int64_t(g) * tv.row_floats;
      for (uint32_t e = lane; e < tv.row_floats; e += 64) rp[e] = vp[e];
    }
    __syncthreads();
  }
  if (lane == 0) tv.ctr->n_pending = 0;
}

// =============================================================================================
// Doubling (cuckoo_fast_double / move_bucket, cuckoohash_map.hpp:1768-1894): index_hash and
// alt_index gain one top bit, so every key of old bucket i lands in new bucket i (same slot) or
// i + 2^hp (compacted from slot 0).  Pure streaming kernel: reads 64 B, writes 128 B per bucket.
// =============================================================================================
__global__ __launch_bounds__(256) void split_kernel(const Bucket* __restrict__ oldb,
                                                    Bucket* __restrict__ newb, uint32_t old_hp) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const uint64_t n_old = uint64_t(1) << old_hp;
  if (i >= n_old) return;
  const Bucket ob = oldb[i];
  const uint32_t new_hp = old_hp + 1;
  const uint64_t new_ind = i + n_old;
  // moves[s]: slot s goes to the new bucket; rank[s]: its compacted slot there.  Everything is
  // indexed by compile-time constants so the buckets stay in registers.
  bool occ[kSlots], moves[kSlots];
  int rank[kSlots];
  int nm = 0;
#pragma unroll
  for (int s = 0; s < kSlots; ++s) {
    occ[s] = ob.key[s] != kEmptyKey;
    const uint64_t hv = hash_key(ob.key[s]);
    const uint32_t p = partial_key(hv);
    const uint64_t old_i = index_hash(old_hp, hv);
    const uint64_t old_a = alt_index(old_hp, p, old_i);
    const uint64_t new_i = index_hash(new_hp, hv);
    const uint64_t new_a = alt_index(new_hp, p, new_i);
    moves[s] = occ[s] && ((i == old_i && new_i == new_ind) || (i == old_a && new_a == new_ind));
    rank[s] = nm;
    nm += moves[s] ? 1 : 0;
  }
  Bucket lo, hi;
#pragma unroll
  for (int d = 0; d < kSlots; ++d) {
    const bool stay = occ[d] && !moves[d];
    lo.key[d] = stay ? ob.key[d] : kEmptyKey;
    lo.row[d] = stay ? ob.row[d] : kNoRow;
    lo.ts[d] = stay ? ob.ts[d] : 0u;
    int64_t hk = kEmptyKey;
    uint32_t hr = kNoRow, ht = 0u;
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
      const bool sel = moves[s] && rank[s] == d;
      hk = sel ? ob.key[s] : hk;
      hr = sel ? ob.row[s] : hr;
      ht = sel ? ob.ts[s] : ht;
    }
    hi.key[d] = hk;
    hi.row[d] = hr;
    hi.ts[d] = ht;
  }
  newb[i] = lo;
  newb[new_ind] = hi;
}

__global__ __launch_bounds__(256) void clear_buckets_kernel(Bucket* __restrict__ b, uint64_t n) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Bucket e;
#pragma unroll
  for (int s = 0; s < kSlots; ++s) { e.key[s] = kEmptyKey; e.row[s] = kNoRow; e.ts[s] = 0; }
  b[i] = e;
}

// =============================================================================================
// Full-table scans.
//   evict_kernel: cuckoo_embedding_hash_table.cc:251-264 — drop a key when
//     max_update_time - ts >= ttl_days(slot_id_v2(key)) * 86400; slot_id_v2 = (fid >> 48) & 0x7fff
//     (data/training_instance/cc/reader_util.h:36-38).
//   dump_kernel:  bucket-major, slot-minor enumeration (partial_dump, cuckoohash_map.hpp:740-773).
// =============================================================================================
struct TtlConfig {
  int64_t default_days;
  int32_t n;
  const int64_t* slots;  // device
  const int32_t* days;   // device
};

// Persistent workgroups, grid-stride over the slots (4 threads per 64-B bucket: coalesced), the
// eviction count kept in registers and added to the table's counters ONCE per workgroup: with every
// row expired a per-wavefront add is 2 atomics x nslots / 64 on two addresses — they queue behind
// each other for milliseconds (r1: 6.1 ms for 4 M rows against 45 us when nothing expires).
__global__ __launch_bounds__(256) void evict_kernel(TableView tv, int64_t max_update_time,
                                                    TtlConfig ttl) {
  __shared__ uint32_t wsum[4];
  const uint64_t nslots = (uint64_t(1) << tv.hp) * kSlots;
  const int64_t def_secs = ttl.default_days * int64_t(86400);
  uint32_t mine = 0;   // (lane 0 of each wavefront: evictions of the wavefront)
#pragma unroll 1
  for (uint64_t t0 = uint64_t(blockIdx.x) * 256; t0 < nslots; t0 += uint64_t(gridDim.x) * 256) {
    const uint64_t t = t0 + threadIdx.x;
    bool ev = false;
    if (t < nslots) {
      Bucket* b = tv.buckets + (t >> 2);
      const int s = t & 3;
      const int64_t key = b->key[s];
      if (key != kEmptyKey) {
        int64_t secs = def_secs;
        if (ttl.n) {
          const int64_t slot = (key >> 48) & 0x7fff;
          for (int i = 0; i < ttl.n; ++i)
            if (ttl.slots[i] == slot) secs = int64_t(ttl.days[i]) * int64_t(86400);
        }
        if (max_update_time - int64_t(b->ts[s]) >= secs) {
          b->key[s] = kEmptyKey;
          b->row[s] = kNoRow;
          ev = true;
        }
      }
    }
    mine += uint32_t(__popcll(__ballot(ev)));
  }
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long c = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (c) {
      atomicAdd(&tv.ctr->alloc, ~(c << 32) + 1ull);  // live keys -= c
      atomicAdd(&tv.ctr->n_evicted, (unsigned int)c);
    }
  }
}

// count occupied slots per block of 1024 slots -> block_counts; then dump with offsets
// (slot range [slot0, slot1): Save's bucket-range shards and chunks)
__global__ __launch_bounds__(256) void dump_count_kernel(TableView tv, uint64_t slot0,
                                                         uint64_t slot1,
                                                         uint32_t* __restrict__ bc) {
  __shared__ uint32_t wsum[4];
  const uint64_t nslots = slot1;
  const uint64_t base = slot0 + uint64_t(blockIdx.x) * 1024;
  uint32_t c = 0;
  for (int k = 0; k < 4; ++k) {
    const uint64_t t = base + uint64_t(threadIdx.x) * 4 + k;
    if (t < nslots && tv.buckets[t >> 2].key[t & 3] != kEmptyKey) ++c;
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) bc[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// block_offsets = exclusive scan of block_counts (done on host for dumps: it is a cold path)
__global__ __launch_bounds__(256) void dump_emit_kernel(TableView tv, uint64_t slot0,
                                                        uint64_t slot1,
                                                        const uint64_t* __restrict__ block_off,
                                                        int64_t* __restrict__ ids,
                                                        int64_t* __restrict__ positions,
                                                        uint32_t* __restrict__ ts,
                                                        float* __restrict__ rows) {
  __shared__ uint32_t wsum[4];
  const uint64_t nslots = slot1;
  const uint64_t base = slot0 + uint64_t(blockIdx.x) * 1024;
  uint32_t occ[4];
  uint32_t c = 0;
  for (int k = 0; k < 4; ++k) {
    const uint64_t t = base + uint64_t(threadIdx.x) * 4 + k;
    occ[k] = (t < nslots && tv.buckets[t >> 2].key[t & 3] != kEmptyKey) ? 1u : 0u;
    c += occ[k];
  }
  // exclusive scan of c over the block (thread order == slot order)
  uint32_t incl = c;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  uint32_t woff = 0;
  for (int i = 0; i < w; ++i) woff += wsum[i];
  uint64_t o = block_off[blockIdx.x] + woff + (incl - c);
  for (int k = 0; k < 4; ++k) {
    if (!occ[k]) continue;
    const uint64_t t = base + uint64_t(threadIdx.x) * 4 + k;
    const Bucket* b = tv.buckets + (t >> 2);
    const int s = t & 3;
    ids[o] = b->key[s];
    positions[o] = static_cast<int64_t>(t);
    ts[o] = b->ts[s];
    if (rows) {
      const float* rp = row_ptr(tv, b->row[s]);
      for (uint32_t e = 0; e < tv.row_floats; ++e) rows[o * tv.row_floats + e] = rp[e];
    }
    ++o;
  }
}

// =============================================================================================
// Batch dedup in first-occurrence order with per-key occurrence lists (CSR), the device form of
// UniqueKeyWithValueAndOffset / FusedReorderByIndices' per-table dedup
// (ops/unique_mapping_ops.cc:82-114, ops/fused_reorder_by_indices.cc:52-60).
//   dd_clear   scratch hash set (capacity C = pow2 >= 2n) <- empty
//   dd_insert  position p claims/joins the slot of ids[p]; atomicMin first position, count
//   dd_tile    per 1024-position tile: (#first occurrences, sum of their counts)
//   dd_emit    two-accumulator exclusive scan over positions: unique index u of each first
//              occurrence and the start seg_off[u] of its occurrence list; writes uids, U
//   dd_place   inverse[p] = u; appends p to list u (unordered, atomic cursor)
//   dd_order   lists of 2..32 positions: in-thread insertion sort; longer: queued as heavy
//   dd_heavy   one workgroup per heavy key: ordered stream compaction over inverse[]
// All counts stay on the device (U is read by downstream kernels through n_dev).
// =============================================================================================
struct DedupView {
  int64_t* hkey;      // [C+1]  (+1 = side slot for kEmptyKey)
  uint32_t* hmin;     // [C+1]
  uint32_t* hcnt;     // [C+1]
  uint32_t* huidx;    // [C+1]
  uint32_t* hcur;     // [C+1]
  uint32_t* slot_of;  // [n]
  uint32_t* seg_tmp;  // [n] unordered occurrence lists (ordered copy goes to seg_pos)
  uint32_t* work;     // [2 * (n/kChunk + n/33 + 2)] (unique index, chunk) items over the heavy lists
  uint32_t* tile_a;   // [ntiles] first-occurrence counts
  uint32_t* tile_b;   // [ntiles] occurrence-count sums
  uint32_t* heavy;    // [n/33 + 1]
  uint32_t* heavy_n;  // [4]: [0] heavy-list length, [1] unique counter, [2] list-space cursor,
                      //      [3] number of work items
  uint32_t* hstart;   // [C+1] list start of the slot's id (unordered dedup), kUnset when unclaimed
  uint32_t cap_mask;  // C-1
};

constexpr int kDdTile = 1024;   // positions per tile (256 threads x 4)
constexpr int kLightMax = 32;   // longest list sorted in-thread
constexpr uint32_t kChunk = 256;  // entries per work item of the fused backward (sum_apply_kernel)

__global__ __launch_bounds__(256) void dd_clear_kernel(DedupView d) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= d.cap_mask + 1u) {
    d.hkey[i] = kEmptyKey;
    d.hmin[i] = 0xffffffffu;
    d.hcnt[i] = 0;
    d.hcur[i] = 0;
    d.hstart[i] = 0xffffffffu;
  }
  if (i < 4) d.heavy_n[i] = 0;
}

// LDS-side pre-aggregation: the 256 positions of a block are first deduplicated in a 512-entry LDS
// hash set, so a Zipf head key with ~12 000 occurrences costs one global atomic pair per BLOCK
// (256 per launch) instead of one per occurrence (same-address L2 atomics run at ~12 ns each).
constexpr int kDdBlock = 1024;  // positions per workgroup in dd_insert / dd_place
constexpr int kDdLds = 2048;   // LDS hash entries (2x the positions)

__device__ __forceinline__ uint32_t dd_global_slot(const DedupView& d, int64_t id) {
  if (id == kEmptyKey) return d.cap_mask + 1u;
  uint32_t s = uint32_t(hash_key(id)) & d.cap_mask;
  for (;;) {
    int64_t k = d.hkey[s];
    if (k == kEmptyKey) {
      k = static_cast<int64_t>(atomicCAS(reinterpret_cast<unsigned long long*>(&d.hkey[s]),
                                         static_cast<unsigned long long>(kEmptyKey),
                                         static_cast<unsigned long long>(id)));
      if (k == kEmptyKey) return s;
    }
    if (k == id) return s;
    s = (s + 1u) & d.cap_mask;
  }
}

__global__ __launch_bounds__(kDdBlock) void dd_insert_kernel(DedupView d,
                                                             const int64_t* __restrict__ ids,
                                                             uint32_t n) {
  __shared__ unsigned long long lkey[kDdLds + 1];
  __shared__ uint32_t lmin[kDdLds + 1], lcnt[kDdLds + 1], lslot[kDdLds + 1];
  for (int i = threadIdx.x; i <= kDdLds; i += kDdBlock) {
    lkey[i] = static_cast<unsigned long long>(kEmptyKey);
    lmin[i] = 0xffffffffu;
    lcnt[i] = 0;
  }
  __syncthreads();
  const uint32_t p = blockIdx.x * kDdBlock + threadIdx.x;
  if (p == 0) {  // dd_emit / dd_finish (later kernels) refill these
    d.heavy_n[0] = 0;
    d.heavy_n[3] = 0;
  }
  const bool valid = p < n;
  uint32_t ls = 0;
  int64_t id = 0;
  if (valid) {
    id = ids[p];
    if (id == kEmptyKey) {
      ls = kDdLds;
    } else {
      ls = uint32_t(hash_key(id) >> 40) & (kDdLds - 1);
      for (;;) {
        unsigned long long k = lkey[ls];
        if (k == static_cast<unsigned long long>(kEmptyKey)) {
          k = atomicCAS(&lkey[ls], static_cast<unsigned long long>(kEmptyKey),
                        static_cast<unsigned long long>(id));
          if (k == static_cast<unsigned long long>(kEmptyKey)) break;
        }
        if (k == static_cast<unsigned long long>(id)) break;
        ls = (ls + 1u) & (kDdLds - 1);
      }
    }
    atomicMin(&lmin[ls], p);
    atomicAdd(&lcnt[ls], 1u);
  }
  __syncthreads();
  if (valid && lmin[ls] == p) {  // the block's first occurrence of this id speaks for all of them
    const uint32_t gs = dd_global_slot(d, id);
    atomicMin(&d.hmin[gs], p);
    atomicAdd(&d.hcnt[gs], lcnt[ls]);
    lslot[ls] = gs;
  }
  __syncthreads();
  if (valid) d.slot_of[p] = lslot[ls];
}

__device__ __forceinline__ void block_reduce2(uint32_t& a, uint32_t& b, uint32_t* sh) {
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_down(a, o);
    b += __shfl_down(b, o);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sh[w] = a;
    sh[4 + w] = b;
  }
  __syncthreads();
  a = sh[0] + sh[1] + sh[2] + sh[3];
  b = sh[4] + sh[5] + sh[6] + sh[7];
  __syncthreads();
}

__global__ __launch_bounds__(256) void dd_tile_kernel(DedupView d, uint32_t n) {
  __shared__ uint32_t sh[8];
  uint32_t a = 0, b = 0;
  const uint32_t base = blockIdx.x * kDdTile + threadIdx.x * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t p = base + k;
    if (p < n) {
      const uint32_t s = d.slot_of[p];
      if (d.hmin[s] == p) {
        ++a;
        b += d.hcnt[s];
      }
    }
  }
  block_reduce2(a, b, sh);
  if (threadIdx.x == 0) {
    d.tile_a[blockIdx.x] = a;
    d.tile_b[blockIdx.x] = b;
  }
}

__global__ __launch_bounds__(256) void dd_emit_kernel(DedupView d, const int64_t* __restrict__ ids,
                                                      uint32_t n, int64_t* __restrict__ uids,
                                                      uint32_t* __restrict__ seg_off,
                                                      uint32_t* __restrict__ n_unique) {
  __shared__ uint32_t sh[8];
  __shared__ uint32_t wa[4], wb[4];
  // offsets of this tile = sums over the tiles before it (each block recomputes: ntiles is small)
  uint32_t pa = 0, pb = 0;
  for (uint32_t t = threadIdx.x; t < blockIdx.x; t += blockDim.x) {
    pa += d.tile_a[t];
    pb += d.tile_b[t];
  }
  block_reduce2(pa, pb, sh);
  const uint32_t base = blockIdx.x * kDdTile + threadIdx.x * 4;
  uint32_t fa[4], fb[4], slot[4];
  uint32_t a = 0, b = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t p = base + k;
    fa[k] = 0;
    fb[k] = 0;
    slot[k] = 0;
    if (p < n) {
      slot[k] = d.slot_of[p];
      if (d.hmin[slot[k]] == p) {
        fa[k] = 1;
        fb[k] = d.hcnt[slot[k]];
      }
    }
    a += fa[k];
    b += fb[k];
  }
  uint32_t ia = a, ib = b;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t va = __shfl_up(ia, o), vb = __shfl_up(ib, o);
    if (lane >= o) {
      ia += va;
      ib += vb;
    }
  }
  if (lane == 63) {
    wa[w] = ia;
    wb[w] = ib;
  }
  __syncthreads();
  uint32_t oa = pa + (ia - a), ob = pb + (ib - b);
  for (int i = 0; i < w; ++i) {
    oa += wa[i];
    ob += wb[i];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (fa[k]) {
      const uint32_t p = base + k;
      uids[oa] = ids[p];
      seg_off[oa] = ob;
      d.huidx[slot[k]] = oa;
      if (fb[k] > kLightMax) d.heavy[atomicAdd(d.heavy_n, 1u)] = oa;
      ++oa;
      ob += fb[k];
    }
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == blockDim.x - 1) {
    *n_unique = oa;   // after the last position: total number of unique ids
    seg_off[oa] = n;  // == ob
  }
}

__global__ __launch_bounds__(kDdBlock) void dd_place_kernel(DedupView d, uint32_t n,
                                                            const uint32_t* __restrict__ seg_off,
                                                            uint32_t* __restrict__ inverse) {
  // same LDS pre-aggregation, keyed by the global slot: in-block rank from an LDS counter, one
  // global cursor bump per distinct id per block
  __shared__ uint32_t lkey[kDdLds], lcnt[kDdLds], lbase[kDdLds];
  for (int i = threadIdx.x; i < kDdLds; i += kDdBlock) {
    lkey[i] = 0xffffffffu;
    lcnt[i] = 0;
  }
  __syncthreads();
  const uint32_t p = blockIdx.x * kDdBlock + threadIdx.x;
  const bool valid = p < n;
  uint32_t s = 0, ls = 0, rank = 0;
  if (valid) {
    s = d.slot_of[p];
    ls = (s * 2654435761u >> 16) & (kDdLds - 1);
    for (;;) {
      uint32_t k = lkey[ls];
      if (k == 0xffffffffu) {
        k = atomicCAS(&lkey[ls], 0xffffffffu, s);
        if (k == 0xffffffffu) break;
      }
      if (k == s) break;
      ls = (ls + 1u) & (kDdLds - 1);
    }
    rank = atomicAdd(&lcnt[ls], 1u);
  }
  __syncthreads();
  if (valid && rank == 0) lbase[ls] = atomicAdd(&d.hcur[s], lcnt[ls]);
  __syncthreads();
  if (valid) {
    const uint32_t u = d.huidx[s];
    inverse[p] = u;
    d.seg_tmp[seg_off[u] + lbase[ls] + rank] = p;
  }
}

// ---------------------------------------------------------------------------------------------
// Unordered dedup for the fused training step (mhte_unique_unordered): the step only needs SOME
// numbering of the distinct ids and their occurrence lists, not the first-occurrence numbering
// the reference op emits, which costs a two-kernel scan over the batch.  Three launches:
//   dd_insert_fast  as dd_insert, and the thread that claims a scratch slot numbers the id
//                   (one LDS counter per block, one global atomic per block)
//   dd_place_fast   the first block that touches an id carves its list [start, start+count) out
//                   of [0, n) (block-aggregated cursor), publishes it; other blocks wait for the
//                   publication, then positions are appended as in dd_place.  Light lists go
//                   straight to seg_pos (the consumer sorts <= kLightMax positions in registers),
//                   heavy lists to seg_tmp for the bitmap ordering pass of dd_finish.
//   dd_finish       (order_light = 0) heavy lists only + scratch reset.
// Unique numbering and list placement depend on atomic arrival order; every VALUE computed from
// them downstream (sums in position order, optimizer steps) does not.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kUnset = 0xffffffffu;
constexpr uint32_t kBusy = 0xfffffffeu;

__device__ __forceinline__ uint32_t dd_global_slot_claim(const DedupView& d, int64_t id,
                                                         bool* claimed) {
  *claimed = false;
  if (id == kEmptyKey) {
    const uint32_t s = d.cap_mask + 1u;
    const unsigned long long old =
        atomicCAS(reinterpret_cast<unsigned long long*>(&d.hkey[s]),
                  static_cast<unsigned long long>(kEmptyKey), 0ull);
    *claimed = static_cast<int64_t>(old) == kEmptyKey;
    return s;
  }
  uint32_t s = uint32_t(hash_key(id)) & d.cap_mask;
  for (;;) {
    int64_t k = d.hkey[s];
    if (k == kEmptyKey) {
      k = static_cast<int64_t>(atomicCAS(reinterpret_cast<unsigned long long*>(&d.hkey[s]),
                                         static_cast<unsigned long long>(kEmptyKey),
                                         static_cast<unsigned long long>(id)));
      if (k == kEmptyKey) {
        *claimed = true;
        return s;
      }
    }
    if (k == id) return s;
    s = (s + 1u) & d.cap_mask;
  }
}

__device__ __forceinline__ void dd_insert_fast_role(const DedupView& d,
                                                    const int64_t* __restrict__ ids, uint32_t n,
                                                    int64_t* __restrict__ uids, uint32_t bid) {
  __shared__ unsigned long long lkey[kDdLds + 1];
  __shared__ uint32_t lmin[kDdLds + 1], lcnt[kDdLds + 1], lslot[kDdLds + 1];
  __shared__ uint32_t l_nclaim, l_base;
  for (int i = threadIdx.x; i <= kDdLds; i += kDdBlock) {
    lkey[i] = static_cast<unsigned long long>(kEmptyKey);
    lmin[i] = 0xffffffffu;
    lcnt[i] = 0;
  }
  if (threadIdx.x == 0) l_nclaim = 0;
  __syncthreads();
  const uint32_t p = bid * kDdBlock + threadIdx.x;
  if (p == 0) {  // dd_place_fast / dd_finish (later kernels) refill these
    d.heavy_n[0] = 0;
    d.heavy_n[3] = 0;
  }
  const bool valid = p < n;
  uint32_t ls = 0;
  int64_t id = 0;
  if (valid) {
    id = ids[p];
    if (id == kEmptyKey) {
      ls = kDdLds;
    } else {
      ls = uint32_t(hash_key(id) >> 40) & (kDdLds - 1);
      for (;;) {
        unsigned long long k = lkey[ls];
        if (k == static_cast<unsigned long long>(kEmptyKey)) {
          k = atomicCAS(&lkey[ls], static_cast<unsigned long long>(kEmptyKey),
                        static_cast<unsigned long long>(id));
          if (k == static_cast<unsigned long long>(kEmptyKey)) break;
        }
        if (k == static_cast<unsigned long long>(id)) break;
        ls = (ls + 1u) & (kDdLds - 1);
      }
    }
    atomicMin(&lmin[ls], p);
    atomicAdd(&lcnt[ls], 1u);
  }
  __syncthreads();
  bool claimed = false;
  uint32_t gs = 0, crank = 0;
  if (valid && lmin[ls] == p) {  // the block's first occurrence of this id speaks for all of them
    gs = dd_global_slot_claim(d, id, &claimed);
    atomicAdd(&d.hcnt[gs], lcnt[ls]);
    lslot[ls] = gs;
    if (claimed) crank = atomicAdd(&l_nclaim, 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0 && l_nclaim) l_base = atomicAdd(&d.heavy_n[1], l_nclaim);
  __syncthreads();
  if (claimed) {
    const uint32_t u = l_base + crank;
    d.huidx[gs] = u;
    uids[u] = id;
  }
  if (valid) d.slot_of[p] = lslot[ls];
}
__global__ __launch_bounds__(kDdBlock) void dd_insert_fast_kernel(DedupView d,
                                                                  const int64_t* __restrict__ ids,
                                                                  uint32_t n,
                                                                  int64_t* __restrict__ uids) {
  dd_insert_fast_role(d, ids, n, uids, blockIdx.x);
}


template <int BLOCK>  // threads per workgroup; every workgroup still covers kDdBlock positions
__device__ __forceinline__ void dd_place_fast_role(
    const DedupView& d, uint32_t n, uint32_t* __restrict__ inverse,
    uint32_t* __restrict__ lst_start, uint32_t* __restrict__ lst_end,
    uint32_t* __restrict__ seg_pos, uint32_t bid) {
  constexpr int PPT = kDdBlock / BLOCK;  // positions per thread
  __shared__ uint32_t lkey[kDdLds], lcnt[kDdLds], lbase[kDdLds], lstart[kDdLds];
  __shared__ uint32_t l_total, l_gbase;
  for (int i = threadIdx.x; i < kDdLds; i += BLOCK) {
    lkey[i] = 0xffffffffu;
    lcnt[i] = 0;
  }
  if (threadIdx.x == 0) l_total = 0;
  __syncthreads();
  uint32_t p[PPT], s[PPT], ls[PPT], rank[PPT];
  bool valid[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    p[k] = bid * kDdBlock + k * BLOCK + threadIdx.x;
    valid[k] = p[k] < n;
    s[k] = valid[k] ? d.slot_of[p[k]] : 0u;
    ls[k] = 0;
    rank[k] = 0;
  }
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    if (valid[k]) {
      uint32_t l = (s[k] * 2654435761u >> 16) & (kDdLds - 1);
      for (;;) {
        uint32_t kk = lkey[l];
        if (kk == 0xffffffffu) {
          kk = atomicCAS(&lkey[l], 0xffffffffu, s[k]);
          if (kk == 0xffffffffu) break;
        }
        if (kk == s[k]) break;
        l = (l + 1u) & (kDdLds - 1);
      }
      ls[k] = l;
      rank[k] = atomicAdd(&lcnt[l], 1u);
    }
  }
  __syncthreads();
  // ---- phase 1 (never waits): the first block to reach an id allocates its list
  bool rep[PPT], won[PPT];
  uint32_t cnt[PPT], loff[PPT], st[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    rep[k] = valid[k] && rank[k] == 0;
    won[k] = false;
    cnt[k] = 0;
    loff[k] = 0;
    st[k] = 0;
    if (rep[k]) won[k] = atomicCAS(&d.hstart[s[k]], kUnset, kBusy) == kUnset;
  }
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    if (won[k]) {
      cnt[k] = d.hcnt[s[k]];
      loff[k] = atomicAdd(&l_total, cnt[k]);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && l_total) l_gbase = atomicAdd(&d.heavy_n[2], l_total);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    if (won[k]) {
      st[k] = l_gbase + loff[k];
      const uint32_t u = d.huidx[s[k]];
      lst_start[u] = st[k];
      lst_end[u] = st[k] + cnt[k];
      if (cnt[k] > kLightMax) d.heavy[atomicAdd(&d.heavy_n[0], 1u)] = u;
      __hip_atomic_store(&d.hstart[s[k]], st[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // ---- phase 2: everybody else waits for the publication (every allocator of this workgroup is
  // past its last barrier and has published, so a waiter can only depend on another workgroup's
  // allocator, which in turn waits for nobody before publishing)
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    if (rep[k]) {
      if (!won[k]) {
        for (;;) {
          st[k] = __hip_atomic_load(&d.hstart[s[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (st[k] < kBusy) break;
          __builtin_amdgcn_s_sleep(1);
        }
      }
      lstart[ls[k]] = st[k];
      lbase[ls[k]] = atomicAdd(&d.hcur[s[k]], lcnt[ls[k]]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    if (valid[k]) {
      const uint32_t u = d.huidx[s[k]];
      const uint32_t c = d.hcnt[s[k]];
      const uint32_t q = lstart[ls[k]] + lbase[ls[k]] + rank[k];
      inverse[p[k]] = u;
      if (c > kLightMax) {
        d.seg_tmp[q] = p[k];
      } else {
        seg_pos[q] = p[k];
      }
    }
  }
}
__global__ __launch_bounds__(kDdBlock) void dd_place_fast_kernel(
    DedupView d, uint32_t n, uint32_t* __restrict__ inverse, uint32_t* __restrict__ lst_start,
    uint32_t* __restrict__ lst_end, uint32_t* __restrict__ seg_pos) {
  dd_place_fast_role<kDdBlock>(d, n, inverse, lst_start, lst_end, seg_pos, blockIdx.x);
}


// Final dedup kernel, two roles in one launch (1024-thread workgroups):
//  * blocks [0, nb_rank): one thread per position p.  Lists of <= kLightMax occurrences are put in
//    position order by rank-counting: p's rank is the number of smaller positions in its
//    (unordered) list, so p itself writes seg_pos[list_start + rank].  The same threads reset the
//    scratch hash slots they used (clean-after-use: the next dedup needs no clear pass).
//  * blocks [nb_rank, grid): one workgroup per heavy key (> kLightMax occurrences, ~130 under
//    Zipf(1.2) at B = 65 536).  Positions are distinct integers below n, so ordering the list is a
//    bitmap problem: the workgroup sets one LDS bit per position of the (unordered) list, then
//    emits the set bits in ascending order with one popcount scan — O(len/1024 + n/32768) per
//    thread instead of a scan over the whole inverse[] array.  n > kBmBits is handled in chunks.
// Thread 0 of a heavy workgroup also appends the list's (unique index, chunk) work items for the
// fused backward (sum_apply_kernel).
constexpr int kBmWords = 8192;               // 32 KiB of LDS: 262 144 positions per chunk
constexpr uint32_t kBmBits = kBmWords * 32u;

//   lst_start / lst_end: list bounds per unique index (ordered dedup: seg_off and seg_off + 1).
//   order_light = 0 (unordered dedup): light lists are already in seg_pos, unordered; only the
//   scratch reset remains for the per-position blocks, and block 0 publishes the unique count.
__device__ __forceinline__ void dd_finish_role(const DedupView& d, uint32_t n, uint32_t nb_rank,
                                               const uint32_t* __restrict__ inverse,
                                               const uint32_t* __restrict__ lst_start,
                                               const uint32_t* __restrict__ lst_end,
                                               uint32_t* __restrict__ seg_pos, int order_light,
                                               uint32_t* __restrict__ n_unique_out, uint32_t bid,
                                               uint32_t nblocks, uint32_t* bm /* LDS [kBmWords + 16] */) {
  if (bid < nb_rank) {
    const uint32_t p = bid * 1024 + threadIdx.x;
    if (!order_light && p == 0) {
      *n_unique_out = d.heavy_n[1];
      d.heavy_n[1] = 0;
      d.heavy_n[2] = 0;
    }
    if (p >= n) return;
    const uint32_t s = d.slot_of[p];
    if (order_light) {
      const uint32_t u = inverse[p];
      const uint32_t q0 = lst_start[u], len = lst_end[u] - q0;
      if (len <= kLightMax) {
      uint32_t r = 0;
      if (len > 1) {
        uint32_t t[kLightMax];
#pragma unroll
        for (int k = 0; k < kLightMax; ++k)
          t[k] = (uint32_t(k) < len) ? d.seg_tmp[q0 + k] : 0xffffffffu;  // all loads in flight
#pragma unroll
        for (int k = 0; k < kLightMax; ++k) r += (t[k] < p) ? 1u : 0u;
      }
      seg_pos[q0 + r] = p;
      }
    }
    d.hkey[s] = kEmptyKey;
    d.hmin[s] = 0xffffffffu;
    d.hcnt[s] = 0;
    d.hcur[s] = 0;
    d.hstart[s] = kUnset;
    return;
  }
  uint32_t* wcnt = bm + kBmWords;
  const uint32_t nh = *d.heavy_n;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (uint32_t h = bid - nb_rank; h < nh; h += nblocks - nb_rank) {
    const uint32_t u = d.heavy[h];
    const uint32_t q0 = lst_start[u], len = lst_end[u] - q0;
    uint32_t* outp = seg_pos + q0;
    if (threadIdx.x == 0) {  // work items of the fused backward: consecutive per list
      const uint32_t nc = (len + kChunk - 1) / kChunk;
      const uint32_t w0 = atomicAdd(&d.heavy_n[3], nc);
      for (uint32_t k = 0; k < nc; ++k) {
        d.work[2 * (w0 + k)] = u;
        d.work[2 * (w0 + k) + 1] = k;
      }
    }
    uint32_t running = 0;
    for (uint32_t c0 = 0; c0 < n; c0 += kBmBits) {
      const uint32_t nbits = min(kBmBits, n - c0);
      const uint32_t nw = (nbits + 31u) >> 5;
      const uint32_t wpt = (nw + 1023u) >> 10;  // words per thread (contiguous)
      for (uint32_t i = threadIdx.x; i < nw; i += 1024) bm[i] = 0;
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < len; i += 1024) {
        const uint32_t rel = d.seg_tmp[q0 + i] - c0;
        if (rel < nbits) atomicOr(&bm[rel >> 5], 1u << (rel & 31u));
      }
      __syncthreads();
      const uint32_t w0 = threadIdx.x * wpt;
      uint32_t cnt = 0;
      for (uint32_t i = 0; i < wpt; ++i)
        if (w0 + i < nw) cnt += __popc(bm[w0 + i]);
      uint32_t incl = cnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
      }
      if (lane == 63) wcnt[w] = incl;
      __syncthreads();
      uint32_t off = running + (incl - cnt), total = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        off += (i < w) ? wcnt[i] : 0u;
        total += wcnt[i];
      }
      for (uint32_t i = 0; i < wpt; ++i) {
        if (w0 + i >= nw) break;
        uint32_t bits = bm[w0 + i];
        const uint32_t pbase = c0 + ((w0 + i) << 5);
        while (bits) {
          const uint32_t bit = __ffs(bits) - 1;
          bits &= bits - 1;
          outp[off] = pbase + bit;
          ++off;
        }
      }
      running += total;
      __syncthreads();  // bm / wcnt are reused
    }
  }
}
__global__ __launch_bounds__(1024) void dd_finish_kernel(DedupView d, uint32_t n, uint32_t nb_rank,
                                                         const uint32_t* __restrict__ inverse,
                                                         const uint32_t* __restrict__ lst_start,
                                                         const uint32_t* __restrict__ lst_end,
                                                         uint32_t* __restrict__ seg_pos,
                                                         int order_light,
                                                         uint32_t* __restrict__ n_unique_out) {
  __shared__ uint32_t bm[kBmWords + 16];
  dd_finish_role(d, n, nb_rank, inverse, lst_start, lst_end, seg_pos, order_light, n_unique_out,
                 blockIdx.x, gridDim.x, bm);
}


// =============================================================================================
// Forward scatter of unique rows to every occurrence: out[p] = src[inverse[p]]
// (FillWithOffsetMap, ops/unique_mapping_ops.cc:225-242, in gather form).
// =============================================================================================
template <int G, int VEC>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src,
                                                          const uint32_t* __restrict__ index,
                                                          int64_t n, uint32_t dim,
                                                          float* __restrict__ out) {
  const int j = threadIdx.x & (G - 1);
  const int64_t p = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  if (p >= n) return;
  const float* sp = src + int64_t(index[p]) * dim;
  float* op = out + p * int64_t(dim);
  for (uint32_t e = j * VEC; e < dim; e += G * VEC) {
    Vec<VEC> v;
    v.load(sp + e);
    v.store(op + e);
  }
}

// =============================================================================================
// Duplicate-gradient sum in occurrence order (FillWithOffsetMapGradient,
// ops/unique_mapping_ops.cc:307-324): out[u] = 0 + g[p0] + g[p1] + ...  over list u.
//
// Under Zipf(1.2) a 65 536-id batch has a head key with ~12 000 occurrences and ~60 % of all
// occurrences belong to <100 keys, so the reduction must be spread over the chip by OCCURRENCE,
// not by key.  The flat CSR position array is cut into windows of WIN = min(G,16) entries, one
// G-lane group per window: lane t preloads entry t (position, list id, list bounds) and all WIN
// gradient rows are in flight at once.  A run that covers its whole list is written straight to
// out[u] — bit-identical to the reference's sequential sum.  A list that crosses a window
// boundary leaves per-window partials:
//   part[2*w+0] : the run that entered window w from the previous window
//   part[2*w+1] : the run that starts inside window w and leaves it unfinished (last_u[w] = list)
// which segsum_combine_kernel adds in window order with a fixed association (deterministic; it
// differs from the sequential sum only by fp32 re-association, well inside the 1e-5 bar).
// segsum_exact_kernel keeps the strictly sequential order for parity runs.
// =============================================================================================
constexpr uint32_t kNone = 0xffffffffu;

template <int G, int VEC>
__global__ __launch_bounds__(256) void segsum_window_kernel(
    const float* __restrict__ grads, const uint32_t* __restrict__ inverse,
    const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ seg_pos, uint32_t n,
    uint32_t dim, float* __restrict__ out, float* __restrict__ part,
    uint32_t* __restrict__ last_u) {
  constexpr int WIN = G < 16 ? G : 16;
  const int lane = threadIdx.x & 63;
  const int j = lane & (G - 1);
  const int gbase = lane & ~(G - 1);
  const int64_t w = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  const int64_t qb = w * WIN;
  const bool live = qb < int64_t(n);  // group-uniform
  const bool has = live && j < WIN && qb + j < int64_t(n);
  const uint32_t p = has ? seg_pos[qb + j] : 0u;
  const uint32_t u = has ? inverse[p] : kNone;
  const uint32_t s0 = has ? seg_off[u] : 0u;
  const uint32_t s1 = has ? seg_off[u + 1] : 0u;
  if (live && j == 0) last_u[w] = kNone;
  uint32_t pt[WIN], ut[WIN + 1], a0[WIN], a1[WIN];
#pragma unroll
  for (int t = 0; t < WIN; ++t) {
    pt[t] = __shfl(p, gbase + t);
    ut[t] = __shfl(u, gbase + t);
    a0[t] = __shfl(s0, gbase + t);
    a1[t] = __shfl(s1, gbase + t);
  }
  ut[WIN] = kNone;
  if (!live) return;
  for (uint32_t e = j * VEC; e < dim; e += G * VEC) {
    Vec<VEC> v[WIN];
#pragma unroll
    for (int t = 0; t < WIN; ++t)
      if (ut[t] != kNone) v[t].load(grads + int64_t(pt[t]) * dim + e);
    Vec<VEC> acc;
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc.v[c] = 0.f;
    uint32_t cur = ut[0], c0 = a0[0], c1 = a1[0];
    uint32_t run_start = uint32_t(qb);
#pragma unroll
    for (int t = 0; t <= WIN; ++t) {
      if (ut[t] != cur) {
        const uint32_t run_end = uint32_t(qb) + t;
        float* dst;
        if (run_start == c0 && run_end == c1) {
          dst = out + int64_t(cur) * dim + e;
        } else if (run_start != c0) {
          dst = part + (w * 2 + 0) * int64_t(dim) + e;
        } else {
          dst = part + (w * 2 + 1) * int64_t(dim) + e;
          if (j == 0) last_u[w] = cur;
        }
        acc.store(dst);
        if (ut[t] == kNone) break;
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc.v[c] = 0.f;
        cur = ut[t];
        c0 = a0[t < WIN ? t : 0];
        c1 = a1[t < WIN ? t : 0];
        run_start = run_end;
      }
      if (t < WIN) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc.v[c] = acc.v[c] + v[t].v[c];
      }
    }
  }
}

// One workgroup per window that holds the head of a boundary-crossing list.
template <int G, int VEC>
__global__ __launch_bounds__(256) void segsum_combine_kernel(
    const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ last_u, uint32_t dim,
    float* __restrict__ out, const float* __restrict__ part) {
  constexpr int WIN = G < 16 ? G : 16;
  constexpr int NG = 256 / G;
  __shared__ float sh[NG * G * VEC];
  const int64_t w = blockIdx.x;
  const uint32_t u = last_u[w];
  if (u == kNone) return;  // block-uniform
  const int j = threadIdx.x & (G - 1);
  const int gi = threadIdx.x / G;
  const uint32_t s1 = seg_off[u + 1];
  const int64_t w1 = int64_t(s1 - 1) / WIN;
  const int64_t items = (w1 - w) + 1;  // item 0 = part[2w+1], item i>0 = part[2(w+i)+0]
  const int64_t per = (items + NG - 1) / NG;
  const int64_t i0 = gi * per;
  const int64_t i1 = (i0 + per < items) ? (i0 + per) : items;
  for (uint32_t e = j * VEC; e < dim; e += G * VEC) {
    Vec<VEC> acc;
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc.v[c] = 0.f;
    for (int64_t i = i0; i < i1; i += 8) {
      Vec<VEC> v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int64_t it = i + t;
        if (it < i1) {
          const int64_t row = (it == 0) ? (w * 2 + 1) : ((w + it) * 2);
          v[t].load(part + row * int64_t(dim) + e);
        }
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (i + t < i1) {
#pragma unroll
          for (int c = 0; c < VEC; ++c) acc.v[c] = acc.v[c] + v[t].v[c];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < VEC; ++c) sh[(gi * G + j) * VEC + c] = acc.v[c];
    __syncthreads();
    if (gi == 0) {
      Vec<VEC> tot;
#pragma unroll
      for (int c = 0; c < VEC; ++c) tot.v[c] = sh[j * VEC + c];
      for (int g2 = 1; g2 < NG; ++g2) {
        if (int64_t(g2) * per >= items) break;
#pragma unroll
        for (int c = 0; c < VEC; ++c) tot.v[c] = tot.v[c] + sh[(g2 * G + j) * VEC + c];
      }
      tot.store(out + int64_t(u) * dim + e);
    }
    __syncthreads();
  }
}

// Strictly sequential per-list sum (bit-exact with the reference; slow for Zipf head keys).
template <int G, int VEC>
__global__ __launch_bounds__(256) void segsum_exact_kernel(
    const float* __restrict__ grads, const uint32_t* __restrict__ n_unique,
    const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ seg_pos, uint32_t dim,
    float* __restrict__ out) {
  const int j = threadIdx.x & (G - 1);
  const uint32_t u = uint32_t((int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G);
  if (u >= *n_unique) return;
  const uint32_t q0 = seg_off[u], q1 = seg_off[u + 1];
  for (uint32_t e = j * VEC; e < dim; e += G * VEC) {
    Vec<VEC> acc;
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc.v[c] = 0.f;
    for (uint32_t q = q0; q < q1; ++q) {
      Vec<VEC> v;
      v.load(grads + int64_t(seg_pos[q]) * dim + e);
#pragma unroll
      for (int c = 0; c < VEC; ++c) acc.v[c] = acc.v[c] + v.v[c];
    }
    acc.store(out + int64_t(u) * dim + e);
  }
}

// =============================================================================================
// Fused backward of the sparse step: duplicate-gradient sum (FillWithOffsetMapGradient,
// ops/unique_mapping_ops.cc:307-324) + upsert + optimizer apply (multi_hash_table_update_op.cc
// :47-100) in ONE launch, for rows of dim <= G*VEC (one element vector per lane).
//
//   blocks [nblk_b, grid)  "id-major": one G-lane group per unique id u.  Lists of <= light_max
//       occurrences are summed by the group itself in occurrence order (bit-identical to the
//       reference's sequential sum) while the group's probe of the table is in flight, then the
//       optimizer is applied from registers.  Longer lists are left to the window blocks.
//   blocks [0, nblk_b)     "chunk blocks": block b takes work item b = (heavy list u, chunk k of
//       kChunk = 256 entries) from the list dd_finish left behind (blocks past the item count
//       exit).  The block's groups sum windows of 16 gradient rows each, LDS adds the groups in
//       order.  A one-chunk list is applied on the spot; otherwise the chunk sum is handed over
//       (write-through partial row + arrival counter) and the block that arrives last adds the
//       chunk sums in chunk order and applies the optimizer.  Chunks are cut relative to the
//       list's start, so the association — and with it every bit of the result — depends on the
//       list only (deterministic), and differs from the sequential sum by fp32 re-association.
//       arrive[] is left zeroed (clean-after-use).
//
// With exact order requested the host passes light_max = 0xffffffff and nblk_b = 0.
// Deferred ids (both buckets full) get their summed gradient stored to grad_u[u] and are finished by
// slowpath_kernel<VEC, kOpOptimize>(values = grad_u) in stream order.
// =============================================================================================

template <int VEC>
__device__ __forceinline__ void vec_zero(Vec<VEC>& v) {
#pragma unroll
  for (int c = 0; c < VEC; ++c) v.v[c] = 0.f;
}
template <int VEC>
__device__ __forceinline__ void vec_add(Vec<VEC>& a, const Vec<VEC>& b) {
#pragma unroll
  for (int c = 0; c < VEC; ++c) a.v[c] = a.v[c] + b.v[c];
}

// Cross-workgroup hand-off of the block partial rows (cdna_hip_programming.md §6 Guideline 16, R1):
// the payload is stored write-through (agent-scope relaxed atomic stores = sc1), every storing
// wave drains its stores, ONE lane bumps the arrival counter; the last arriver reads the payload
// with agent-scope loads.  No release/acquire fence: on gfx950 an agent-scope fence writes back /
// invalidates the whole per-XCD L2, which costs far more than the rows being handed over.
template <int VEC>
__device__ __forceinline__ void store_wt(float* p, const Vec<VEC>& v) {
#pragma unroll
  for (int c = 0; c < VEC; ++c)
    __hip_atomic_store(p + c, v.v[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int VEC>
__device__ __forceinline__ void load_agent(const float* p, Vec<VEC>& v) {
#pragma unroll
  for (int c = 0; c < VEC; ++c)
    v.v[c] = __hip_atomic_load(p + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Store of an updated row's vector in the fused step kernels.  (Measured and dropped: write-through
// `global_store_dwordx4 ... sc1` stores here, so that the rows leave the XCD's L2 while the launch
// runs instead of being drained at its end — step and mstep times unchanged within noise, r03e.)
template <int VEC>
__device__ __forceinline__ void row_store(float* p, const Vec<VEC>& v) {
  v.store(p);
}

// Rare paths of the fused step kernels (a NEW id's initializer, FTRL's constants): what they compute
// from the table descriptor is loop-invariant, so the compiler hoists it in front of the trip loop
// and — at 96 VGPRs — spills it there: scratch stores at every loop entry for values most trips
// never read.  A descriptor whose fields pass through an empty volatile asm stays in the branch.
__device__ __forceinline__ float opaque_f(float x) {
#ifndef MHTE_NO_ANTIHOIST
  asm volatile("" : "+v"(x));
#endif
  return x;
}
__device__ __forceinline__ SegDesc seg_for_init(const SegDesc& sd) {
  SegDesc si = sd;
  si.init_value = opaque_f(sd.init_value);
  si.init_value2 = opaque_f(sd.init_value2);
  return si;
}

// one optimizer step on the lane's element vector, gradient in registers (kOpOptimize only)
template <int VEC, bool ONESEG = false>
__device__ __forceinline__ void optimize_row_reg(const TableView& tv, float* rp, bool is_new,
                                                 uint32_t e, const Vec<VEC>& g, const ApplyArgs& a) {
  if (e >= tv.dim) return;
  uint32_t k = 0;
  const SegDesc sd = seg_of<ONESEG>(tv, e, k);
  const uint32_t le = e - sd.w_off;
  const float lr = a.lr[k];
  Vec<VEC> w, s1, s2;
  float* st1 = rp + sd.st_off + le;
  float* st2 = st1 + sd.dim;
  const bool has1 = sd.opt == kOptAdagrad || sd.opt == kOptFtrl;
  const bool has2 = sd.opt == kOptFtrl;
  if (is_new) {
    const SegDesc si = seg_for_init(sd);
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      w.v[c] = init_weight(si, rp + e + c);
      s1.v[c] = sd.p[0];
      s2.v[c] = 0.f;
    }
  } else {
    w.load(rp + e);
    if (has1) s1.load(st1);
    if (has2) s2.load(st2);
  }
  if (sd.opt == kOptSgd) {
    const float slr = opaque_f(lr);
#pragma unroll
    for (int c = 0; c < VEC; ++c) w.v[c] = sgd_step(w.v[c], g.v[c], slr);
  } else if (sd.opt == kOptAdagrad) {
    const float alr = opaque_f(lr), wd = opaque_f(sd.p[1]);
    if (MHTE_AVX_FORM(sd)) {   // the reference's AVX2 form (adagrad_step_avx), opt-in
#pragma unroll
      for (int c = 0; c < VEC; ++c)
        adagrad_step_avx(w.v[c], s1.v[c], g.v[c], alr, wd, uint32_t(le + c) < (uint32_t(sd.dim) & ~7u));
    } else {
#pragma unroll
      for (int c = 0; c < VEC; ++c) adagrad_step(w.v[c], s1.v[c], g.v[c], alr, wd);
    }
  } else {
    const float flr = opaque_f(lr), beta = opaque_f(sd.p[1]), l1 = opaque_f(sd.p[2]), l2 = opaque_f(sd.p[3]);
#pragma unroll
    for (int c = 0; c < VEC; ++c) ftrl_step(w.v[c], s1.v[c], s2.v[c], g.v[c], flr, beta, l1, l2);
  }
  row_store<VEC>(rp + e, w);
  if (has1) row_store<VEC>(st1, s1);
  if (has2) row_store<VEC>(st2, s2);
}

// The same for a table that uses ANY per-element optimizer (Momentum, Adadelta, RMSProp v1 / v2, Adam,
// AMSGrad, MovingAverage, BatchSoftmax beside SGD / Adagrad / FTRL; not the whole-segment GroupAdaGrad):
// the fused step kernels' FULL instantiations.  Kept apart from optimize_row_reg so that the kernels
// of SGD / Adagrad / FTRL tables keep their register budget (the twelve-way switch costs ~20 VGPRs).
// Arithmetic: the one source of apply_row (optimizer steps of mhte_core.h), one Optimize() call.
// the row of a one-segment table whose optimizer is a compile-time fact (OPTK): weights, its state vectors
// and (adam / amsgrad) the two running powers, fetched while the gradient chain is in flight
template <int VEC>
struct RowRegsF {
  Vec<VEC> w, s1, s2, s3;
  float c1, c2;
};
template <int VEC, int OPTK>
__device__ __forceinline__ void row_prefetch_full(const TableView& tv, const float* rp, uint32_t e,
                                                  RowRegsF<VEC>& r) {
  if (e >= tv.dim) return;
  const int nv = opt_vectors(OPTK);
  const uint32_t st = uint32_t(tv.seg[0].st_off), sdim = uint32_t(tv.seg[0].dim);
  r.w.load(rp + e);
  if (nv > 0) r.s1.load(rp + st + e);
  if (nv > 1) r.s2.load(rp + st + sdim + e);
  if (nv > 2) r.s3.load(rp + st + 2u * sdim + e);
  if (opt_scalars(OPTK) != 0) {
    const float* sc = rp + st + uint32_t(nv) * sdim;
    r.c1 = *(const MHTE_GLOBAL float*)(sc);
    r.c2 = *(const MHTE_GLOBAL float*)(sc + 1);
  }
}

// OPTK >= 0: the (one-segment) table's optimizer is a compile-time fact of the caller's instance — only its
// update rule, state vectors and hyper-parameters are compiled in (the eleven-way switch with everything
// every rule needs live around it is what spills 54-114 registers in the step kernels' FULL instances).
// pre: the row as row_prefetch_full fetched it (used unless is_new)
template <int VEC, bool ONESEG = false, int OPTK = -1>
__device__ __forceinline__ void optimize_row_reg_full(const TableView& tv, float* rp, bool is_new,
                                                      uint32_t e, const Vec<VEC>& g, const ApplyArgs& a,
                                                      const RowRegsF<VEC>* pre = nullptr) {
  if (e >= tv.dim) return;
  uint32_t k = 0;
  SegDesc sd = seg_of<ONESEG>(tv, e, k);
  if (OPTK >= 0) sd.opt = OPTK;
  const uint32_t le = e - sd.w_off;
  const float lr = a.lr[k];
  const int nv = opt_vectors(sd.opt);
  const bool scal = opt_scalars(sd.opt) != 0;
  const bool bsm = sd.opt == kOptBatchSoftmax;
  Vec<VEC> w, s1, s2, s3;
  float* st1 = rp + sd.st_off + le;
  float* st2 = st1 + sd.dim;
  float* st3 = st2 + sd.dim;
  float* sc = rp + sd.st_off + nv * sd.dim;
  float c1 = 0.f, c2 = 0.f;
  long long last_step = 0;
  if (is_new) {
    SegDesc si = seg_for_init(sd);
    if (ONESEG) {
#pragma unroll
      for (int i = 0; i < 8; ++i) si.p[i] = opaque_f(sd.p[i]);
    }
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      w.v[c] = init_weight(si, rp + e + c);
      s1.v[c] = opt_state_init(si, 0);
      s2.v[c] = opt_state_init(si, 1);
      s3.v[c] = opt_state_init(si, 2);
    }
    c1 = si.p[0];
    c2 = si.p[1];
  } else if (pre) {
    w = pre->w;
    if (nv > 0) s1 = pre->s1;
    if (nv > 1) s2 = pre->s2;
    if (nv > 2) s3 = pre->s3;
    if (scal) {
      c1 = pre->c1;
      c2 = pre->c2;
    }
  } else {
    w.load(rp + e);
    if (nv > 0) s1.load(st1);
    if (nv > 1) s2.load(st2);
    if (nv > 2) s3.load(st3);
    if (scal) {
      c1 = sc[0];
      c2 = sc[1];
    }
    if (bsm)
      last_step = static_cast<long long>(
          (static_cast<unsigned long long>(__float_as_uint(sc[1])) << 32) | __float_as_uint(sc[0]));
  }
  // (ONESEG: the descriptor is uniform, and what the eleven update rules derive from it — VGPR copies
  // of the hyper-parameters, their double forms, 1 - beta — would be hoisted in front of the caller's
  // trip loop and spilled there: 100 B per lane at every loop entry, reloaded inside the rules.
  // Copies the compiler cannot see through keep all of that in the trip; see opaque_f.)
  float hp[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) hp[i] = ONESEG ? opaque_f(sd.p[i]) : sd.p[i];
  const float lrv = ONESEG ? opaque_f(lr) : lr;
  const float lr_eff = scal ? adam_lr(lrv, c1, c2) : lrv;
#pragma unroll
  for (int c = 0; c < VEC; ++c) {
    switch (sd.opt) {
      case kOptSgd: w.v[c] = sgd_step(w.v[c], g.v[c], lrv); break;
      case kOptAdagrad: adagrad_any(w.v[c], s1.v[c], g.v[c], lrv, hp[1], hp[2], le + c, sd.dim); break;
      case kOptFtrl: ftrl_step(w.v[c], s1.v[c], s2.v[c], g.v[c], lrv, hp[1], hp[2], hp[3]); break;
      case kOptMomentum: momentum_step(w.v[c], s1.v[c], g.v[c], lrv, hp[0], hp[1], hp[2] != 0.f); break;
      case kOptAdadelta: adadelta_step(w.v[c], s1.v[c], s2.v[c], g.v[c], lrv, hp[0], hp[1], hp[2]); break;
      case kOptRmsprop: rmsprop_step(w.v[c], s1.v[c], g.v[c], double(hp[2]), hp[0], hp[1], false); break;
      case kOptRmspropV2: rmsprop_step(w.v[c], s1.v[c], g.v[c], double(lrv), hp[0], hp[1], true); break;
      case kOptAdam:
        adam_step(w.v[c], s1.v[c], s2.v[c], nullptr, g.v[c], lr_eff, hp[0], hp[1], hp[2], hp[3], hp[4] != 0.f);
        break;
      case kOptMovingAverage: w.v[c] = moving_average_step(w.v[c], g.v[c], hp[0]); break;
      case kOptBatchSoftmax: batch_softmax_step(w.v[c], last_step, lrv, a.global_step); break;
      default:  // kOptAmsgrad
        adam_step(w.v[c], s1.v[c], s2.v[c], &s3.v[c], g.v[c], lr_eff, hp[0], hp[1], hp[2], hp[3], hp[4] != 0.f);
        break;
    }
  }
  if (scal) {
    c1 = c1 * hp[0];
    c2 = c2 * hp[1];
  }
  if (sd.sr16) {   // the fp16 stochastic-rounding decorator (one Optimize() per step here)
#pragma unroll
    for (int c = 0; c < VEC; ++c) w.v[c] = stochastic_round(w.v[c], sr_draw(rp + e + c, w.v[c], a.ts, 0u));
  }
  row_store<VEC>(rp + e, w);
  if (nv > 0) row_store<VEC>(st1, s1);
  if (nv > 1) row_store<VEC>(st2, s2);
  if (nv > 2) row_store<VEC>(st3, s3);
  if (scal && le == 0) {
    sc[0] = c1;
    sc[1] = c2;
    sc[2] = 0.f;
    sc[3] = 0.f;
  }
  if (bsm && le == 0) {
    const unsigned long long gs = static_cast<unsigned long long>(last_step);
    sc[0] = __uint_as_float(uint32_t(gs));
    sc[1] = __uint_as_float(uint32_t(gs >> 32));
    sc[2] = 0.f;
    sc[3] = 0.f;
  }
}

// probe + insert + apply for the id of unique index u, gradient vector in registers.  Wave-uniform
// call (every lane of the wavefront), `valid` per group.
template <int G, int VEC>
__device__ __forceinline__ void upsert_reg(const TableView& tv, const int64_t* __restrict__ uids,
                                           uint32_t u, bool valid, const Vec<VEC>& g, int lane,
                                           const ApplyArgs& a, float* __restrict__ grad_u,
                                           uint32_t* __restrict__ pending) {
  const int j = lane & (G - 1);
  const int64_t id = valid ? uids[u] : 0;
  const uint64_t hv = hash_key(id);
  const uint64_t i1 = index_hash(tv.hp, hv);
  const uint64_t i2 = alt_index(tv.hp, partial_key(hv), i1);
  Bucket* b = assume_global(tv.buckets + ((j < 4) ? i1 : i2));
  int64_t k = kEmptyKey;
  uint32_t row = kNoRow;
  if (valid && id != kEmptyKey && j < 8) {
    k = b->key[j & 3];
    row = b->row[j & 3];
  }
  const SlotResult sr = upsert_resolve<G>(tv, b, id, valid, k, row, lane, a.ts);
  const uint32_t e = uint32_t(j) * VEC;
  if (sr.deferred) {
    if (e < tv.dim) g.store(grad_u + int64_t(u) * tv.dim + e);
    if (j == 0) pending[atomicAdd(&tv.ctr->n_pending, 1u)] = u;
  } else if (valid) {
    optimize_row_reg<VEC>(tv, row_ptr(tv, sr.r), sr.is_new, e, g, a);
  }
}

template <int G, int VEC, int BLOCK>
__device__ __forceinline__ void sum_apply_role(
    const TableView& tv, const int64_t* __restrict__ uids, const uint32_t* __restrict__ n_unique,
    int64_t n_max, const float* __restrict__ grads, const uint32_t* __restrict__ lst_start,
    const uint32_t* __restrict__ lst_end, const uint32_t* __restrict__ seg_pos,
    const uint32_t* __restrict__ work, const uint32_t* __restrict__ n_work,
    uint32_t nblk_b, uint32_t light_max, float* part, uint32_t* arrive,
    float* __restrict__ grad_u, const ApplyArgs& a, uint32_t* __restrict__ pending, uint32_t bid) {
  constexpr int WIN = G < 16 ? G : 16;
  constexpr int NG = 256 / G;    // groups of the chunk role (its first 256 threads)
  constexpr int NGA = BLOCK / G;  // groups of the id-major role
  const int lane = threadIdx.x & 63;
  const int j = lane & (G - 1);
  const int gbase = lane & ~(G - 1);
  const uint32_t dim = tv.dim;
  const uint32_t e = uint32_t(j) * VEC;
  const bool ev = e < dim;

  if (bid >= nblk_b) {
    // ------------------------------------------------------------------ id-major part
    const int64_t g = (int64_t(bid - nblk_b) * BLOCK + threadIdx.x) / G;
    const int64_t nu = min(n_max, int64_t(*n_unique));
    bool valid = g < nu;
    const int64_t id = valid ? uids[g] : 0;
    __shared__ uint32_t sh_sorted[NGA][kLightMax];
    const int grp = threadIdx.x / G;
    const uint32_t q0 = valid ? lst_start[g] : 0u;
    const uint32_t q1 = valid ? lst_end[g] : 0u;
    if (q1 - q0 > light_max) valid = false;  // heavy list: the window blocks own it
    const uint32_t len = valid ? q1 - q0 : 0u;
    // lists of <= kLightMax positions may arrive unordered (mhte_unique_unordered): rank them in
    // registers (positions are distinct, so the ranks are a permutation) and read them back in
    // position order from LDS.  Longer lists are always stored ordered (dd_finish).
    const bool small = len <= uint32_t(kLightMax);
    constexpr int PER = (kLightMax + G - 1) / G;
    uint32_t x[PER], xr[PER];
#pragma unroll
    for (int c = 0; c < PER; ++c) {
      const uint32_t idx = uint32_t(j) + uint32_t(c) * G;
      x[c] = (small && idx < len) ? seg_pos[q0 + idx] : 0xffffffffu;
      xr[c] = 0;
    }
    const uint64_t hv = hash_key(id);
    const uint64_t i1 = index_hash(tv.hp, hv);
    const uint64_t i2 = alt_index(tv.hp, partial_key(hv), i1);
    Bucket* b = assume_global(tv.buckets + ((j < 4) ? i1 : i2));
    int64_t k = kEmptyKey;
    uint32_t row = kNoRow;
    if (valid && id != kEmptyKey && j < 8) {  // probe loads go out first ...
      k = b->key[j & 3];
      row = b->row[j & 3];
    }
#pragma unroll
    for (int t = 0; t < kLightMax; ++t) {
      const uint32_t y = __shfl(x[t / G], gbase + (t % G));
#pragma unroll
      for (int c = 0; c < PER; ++c) xr[c] += (y < x[c]) ? 1u : 0u;
    }
#pragma unroll
    for (int c = 0; c < PER; ++c)
      if (x[c] != 0xffffffffu) sh_sorted[grp][xr[c]] = x[c];
    __syncthreads();
    Vec<VEC> acc;  // ... and the gradient chain seg_pos -> grads overlaps them
    vec_zero(acc);
    if (valid) {
      for (uint32_t q = q0; q < q1; q += 8) {
        uint32_t pos[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
          pos[t] = (q + t < q1) ? (small ? sh_sorted[grp][q - q0 + t] : seg_pos[q + t]) : 0u;
        Vec<VEC> v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (q + t < q1 && ev) v[t].load(grads + int64_t(pos[t]) * dim + e);
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (q + t < q1 && ev) vec_add(acc, v[t]);
      }
    }
    const SlotResult sr = upsert_resolve<G>(tv, b, id, valid, k, row, lane, a.ts);
    if (sr.deferred) {
      if (ev) acc.store(grad_u + g * int64_t(dim) + e);
      if (j == 0) pending[atomicAdd(&tv.ctr->n_pending, 1u)] = uint32_t(g);
    } else if (valid) {
      optimize_row_reg<VEC>(tv, row_ptr(tv, sr.r), sr.is_new, e, acc, a);
    }
    return;
  }

  // -------------------------------------------------------------------- chunk blocks
  // Block b takes work item b of the list dd_finish built: chunk k (kChunk entries) of heavy list u.
  // Chunks are cut relative to the list's own start, so the association of the sum is a function
  // of the list alone, wherever the dedup happened to place it.
  __shared__ float sh_sum[NG][G * VEC];
  __shared__ uint32_t sh_last;
  if (bid >= *n_work) return;
  const bool act = BLOCK == 256 || threadIdx.x < 256;  // wider workgroups: the rest only sync
  const int wl = act ? threadIdx.x / G : 0;
  const uint32_t u = work[2 * bid], kc = work[2 * bid + 1];
  const uint32_t st = lst_start[u], en = lst_end[u];
  const uint32_t c0 = st + kc * kChunk, c1 = min(en, c0 + kChunk);
  const uint32_t nchunk = (en - st + kChunk - 1) / kChunk;
  Vec<VEC> acc;
  vec_zero(acc);
#pragma unroll 1
  for (uint32_t qb = act ? c0 + wl * WIN : c1; qb < c1; qb += NG * WIN) {  // this group's windows
    const uint32_t p = (j < WIN && qb + j < c1) ? seg_pos[qb + j] : kNone;
    uint32_t pt[WIN];
#pragma unroll
    for (int t = 0; t < WIN; ++t) pt[t] = __shfl(p, gbase + t);
    Vec<VEC> v[WIN];
#pragma unroll
    for (int t = 0; t < WIN; ++t)
      if (pt[t] != kNone && ev) v[t].load(grads + int64_t(pt[t]) * dim + e);
#pragma unroll
    for (int t = 0; t < WIN; ++t)
      if (pt[t] != kNone && ev) vec_add(acc, v[t]);
  }
  if (act) {
#pragma unroll
    for (int c = 0; c < VEC; ++c) sh_sum[wl][j * VEC + c] = acc.v[c];
  }
  __syncthreads();
  // groups in order -> the chunk's sum (wave 0 holds it; group 0 uses it)
  Vec<VEC> tot;
  vec_zero(tot);
  if (threadIdx.x < 64) {
#pragma unroll 1
    for (int g2 = 0; g2 < NG; ++g2) {
#pragma unroll
      for (int c = 0; c < VEC; ++c) tot.v[c] = tot.v[c] + sh_sum[g2][j * VEC + c];
    }
  }
  if (nchunk == 1) {  // the whole list: apply on the spot
    if (threadIdx.x < 64)
      upsert_reg<G, VEC>(tv, uids, u, threadIdx.x < G, tot, lane, a, grad_u, pending);
    return;
  }
  // one partial row per chunk, stored where the work items of the list sit (they are consecutive)
  const uint32_t b0 = bid - kc;
  if (threadIdx.x < G && ev) store_wt<VEC>(part + int64_t(bid) * dim + e, tot);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the storing wave drains its partial row
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t last = (atomicAdd(&arrive[u], 1u) == nchunk - 1) ? 1u : 0u;
    if (last) arrive[u] = 0;  // clean-after-use
    sh_last = last;
  }
  __syncthreads();
  if (!sh_last) return;  // block-uniform
  // last arriver: add the chunk sums in chunk order (fixed association), then apply
  const uint32_t per = (nchunk + NG - 1) / NG;
  const uint32_t k0 = act ? min(nchunk, uint32_t(wl) * per) : nchunk, k1 = min(nchunk, k0 + per);
  Vec<VEC> sacc;
  vec_zero(sacc);
  for (uint32_t kk = k0; kk < k1; kk += 8) {
    Vec<VEC> r[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
      if (kk + t < k1 && ev) load_agent<VEC>(part + int64_t(b0 + kk + t) * dim + e, r[t]);
#pragma unroll
    for (int t = 0; t < 8; ++t)
      if (kk + t < k1 && ev) vec_add(sacc, r[t]);
  }
  __syncthreads();  // sh_sum is reused
  if (act) {
#pragma unroll
    for (int c = 0; c < VEC; ++c) sh_sum[wl][j * VEC + c] = sacc.v[c];
  }
  __syncthreads();
  if (threadIdx.x < 64) {  // wave 0; group 0 applies
    Vec<VEC> fin;
    vec_zero(fin);
    for (int g2 = 0; g2 < NG && uint32_t(g2) * per < nchunk; ++g2) {
#pragma unroll
      for (int c = 0; c < VEC; ++c) fin.v[c] = fin.v[c] + sh_sum[g2][j * VEC + c];
    }
    upsert_reg<G, VEC>(tv, uids, u, threadIdx.x < G, fin, lane, a, grad_u, pending);
  }
}
template <int G, int VEC>
__global__ __launch_bounds__(256) void sum_apply_kernel(
    TableView tv, const int64_t* __restrict__ uids, const uint32_t* __restrict__ n_unique,
    int64_t n_max, const float* __restrict__ grads, const uint32_t* __restrict__ lst_start,
    const uint32_t* __restrict__ lst_end, const uint32_t* __restrict__ seg_pos,
    const uint32_t* __restrict__ work, const uint32_t* __restrict__ n_work,
    uint32_t nblk_b, uint32_t light_max, float* part, uint32_t* arrive,
    float* __restrict__ grad_u, ApplyArgs a, uint32_t* __restrict__ pending) {
  WaveTrace wt(tv.trace);
  sum_apply_role<G, VEC, 256>(tv, uids, n_unique, n_max, grads, lst_start, lst_end, seg_pos, work,
                              n_work, nblk_b, light_max, part, arrive, grad_u, a, pending,
                              blockIdx.x);
  wt.end(blockIdx.x < nblk_b ? 7u : 8u);
}


// value_offset[q] = base + (seg_pos[q] * dim)   (the float offsets the reference op emits)
__global__ __launch_bounds__(256) void offsets_from_positions_kernel(
    const uint32_t* __restrict__ seg_pos, uint32_t n, int64_t base, int64_t dim,
    int64_t* __restrict__ value_offset) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) value_offset[q] = base + int64_t(seg_pos[q]) * dim;
}
__global__ __launch_bounds__(256) void widen_offsets_kernel(const uint32_t* __restrict__ seg_off,
                                                            const uint32_t* __restrict__ n_unique,
                                                            int64_t base,
                                                            int64_t* __restrict__ out) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u <= *n_unique) out[u] = base + int64_t(seg_off[u]);
}

// =============================================================================================
// General (op-level) forms of FillWithOffsetMap / FillWithOffsetMapGradient for one table
// (ops/unique_mapping_ops.cc:204-329): `pos[i]` names a unique key, whose float offsets into the
// flat buffer are offset_map[split[pos[i]] .. split[pos[i]+1]).
// =============================================================================================
template <int G, int VEC>
__global__ __launch_bounds__(256) void scatter_offsets_kernel(
    const int64_t* __restrict__ pos, int64_t n, const float* __restrict__ value,
    const int64_t* __restrict__ offset_map, const int64_t* __restrict__ offset_split, uint32_t dim,
    float* __restrict__ buffer) {
  const int j = threadIdx.x & (G - 1);
  const int64_t i = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  if (i >= n) return;
  const int64_t p = pos[i];
  const int64_t q0 = offset_split[p], q1 = offset_split[p + 1];
  for (uint32_t e = j * VEC; e < dim; e += G * VEC) {
    Vec<VEC> v;
    v.load(value + i * int64_t(dim) + e);
    for (int64_t q = q0; q < q1; ++q) v.store(buffer + offset_map[q] + e);
  }
}

template <int G, int VEC>
__global__ __launch_bounds__(256) void gather_sum_offsets_kernel(
    const int64_t* __restrict__ pos, int64_t n, const float* __restrict__ grad,
    const int64_t* __restrict__ offset_map, const int64_t* __restrict__ offset_split, uint32_t dim,
    float* __restrict__ out) {
  const int j = threadIdx.x & (G - 1);
  const int64_t i = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  if (i >= n) return;
  const int64_t p = pos[i];
  const int64_t q0 = offset_split[p], q1 = offset_split[p + 1];
  for (uint32_t e = j * VEC; e < dim; e += G * VEC) {
    Vec<VEC> acc;
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc.v[c] = 0.f;
    for (int64_t q = q0; q < q1; ++q) {
      Vec<VEC> v;
      v.load(grad + offset_map[q] + e);
#pragma unroll
      for (int c = 0; c < VEC; ++c) acc.v[c] = acc.v[c] + v.v[c];
    }
    acc.store(out + i * int64_t(dim) + e);
  }
}

}  // namespace mhte
#endif  // MHTE_KERNELS_H_
