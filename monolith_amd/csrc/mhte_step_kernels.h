// Pipelined training step of one table (mhte_table_step_forward / _backward): two launches per
// step, every wavefront of which has work.  Included by mhte.hip after mhte_kernels.h.
//
// What bounds a launch at B = 65 536 ids is not bytes but (a) the number of dependent memory round
// trips on the longest chain and (b) how many wavefronts must be started and held
// (profiles/r01/e_wave_timeline_2launch.md: the list-building dedup needed three dependent
// phases — insert, place with cross-workgroup waits, heavy-list ordering — of 9-20 us each, and the
// backward launched 21 000 wavefronts that found no work).  Hence:
//
//  * RUN DEDUP (rd_dedup_role) — ONE phase instead of three.  Workgroup b owns positions
//    [1024 b, 1024 b + 1024).  It deduplicates them in an LDS hash set, claims each distinct id's slot
//    in a global scratch hash with ONE CAS, ORs bit b into the id's 64-bit workgroup mask and adds
//    its occurrence count (fire-and-forget atomics); the positions are grouped by id INSIDE the
//    workgroup's own 1024-entry region ("runs", each sorted ascending), and the LDS table
//    (id -> run offset, length, first position) is dumped as is.  No list space is allocated
//    globally, nothing is numbered, nobody waits for anybody, and since workgroups are position
//    ranges, "runs in workgroup order" IS occurrence order — nothing is left to sort.
//  * BUILD (rd_build_role) — off the critical path, in the previous step's backward launch: a
//    coalesced scan of the scratch compacts the occupied slots into dense arrays (unique id, count,
//    workgroup mask, a position) — the unique numbering — resets the scratch, and cuts ids with
//    > kLightMax occurrences into work items of ~256 entries (a power-of-two range of workgroups
//    each) with their run descriptors copied next to the item.
//  * APPLY (rd_apply_role) — fixed-size grid with grid-stride loops.  Id-major groups: everything
//    about the id in one round trip; then table probe | gradient of a lone occurrence | run tables
//    of a short list; then the row; optimizer.  Item workgroups: 16 gradient rows in flight per
//    group, LDS adds the groups in order, multi-item ids hand over through write-through partial
//    rows.
//
// Summation order: every id's gradients are added in ascending position order inside a light list
// (bit-identical to the reference's sequential sum, unique_mapping_ops.cc:307-324); a heavy list is
// a fixed tree over (item, group, window) that depends on the positions only — deterministic, fp32
// re-association only.  MHTE_EXACT_ORDER sums every list strictly sequentially.
#ifndef MHTE_STEP_KERNELS_H_
#define MHTE_STEP_KERNELS_H_

#include "mhte_kernels.h"

namespace mhte {

constexpr int kRdBlock = 1024;          // positions per dedup workgroup (one per thread)
constexpr int kRdLds = 2048;            // LDS hash entries per workgroup
constexpr int kRdStride = kRdLds + 1;   // + side entry for kEmptyKey
constexpr int kRdMaxBlocks = 64;        // bits of the workgroup mask: n <= 65 536 positions
constexpr int kLongRun = kLightMax;  // (a run that may belong to a light list keeps its positions in LDS)
// Lists of up to this many occurrences are summed by ONE lane group, 8 gradient rows in flight at
// a time; longer ones go to the item workgroups.  (16 instead: +4 us on step_bwd — the extra
// ~70 items land on the critical path of the launch.)
constexpr int kStepLightMax = 32;
constexpr int kMaxLongRuns = 16;
constexpr uint32_t kItemTarget = 256;   // entries per heavy work item (expected)
constexpr uint32_t kSpecSlackRows = 32; // row handles an update may strand (upsert_issue), per op
#ifndef MHTE_BWD_OCC
#define MHTE_BWD_OCC 5
#endif
constexpr int kBwdBlocksPerCu = MHTE_BWD_OCC;  // 256-thread workgroups of step_bwd resident per CU (5: <= 96 VGPRs)

// Workgroup barrier for LDS traffic only.  __syncthreads() also drains every outstanding global
// store and atomic of the wavefront (s_waitcnt vmcnt(0)), which costs a full memory round trip per
// barrier on these latency-bound chains; the roles below only exchange data through LDS, and
// global results are consumed by the NEXT launch.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// Same for lanes of ONE wavefront exchanging data through LDS (a G-lane group never spans
// wavefronts): LDS operations of a wavefront execute in order, so only the compiler and the
// operation counter have to be told.
__device__ __forceinline__ void lds_wave_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// run descriptor: first local position | offset in the workgroup's region | length (0..1024)
__device__ __forceinline__ uint32_t run_pack(uint32_t first, uint32_t off, uint32_t cnt) {
  return (first << 21) | (off << 11) | cnt;
}
__device__ __forceinline__ uint32_t run_cnt(uint32_t v) { return v & 0x7ffu; }
__device__ __forceinline__ uint32_t run_off(uint32_t v) { return (v >> 11) & 0x3ffu; }
__device__ __forceinline__ uint32_t run_first(uint32_t v) { return v >> 21; }

struct __attribute__((aligned(16))) ItemHdr {   // one heavy work item: workgroups [b0, b0 + nbk) of list u
  int64_t id;
  uint32_t u;      // unique index
  uint32_t meta;   // b0 | nbk << 8 | k << 16 | nitems << 24   (nbk 1..64, k < nitems <= 64)
  // what the build role's table probe found for the id (ProbeOut; kNoRow / kNoRow / 0 without it)
  uint32_t row;
  uint32_t spec;
  unsigned long long loc;
};

// One slot of the global scratch hash: key, count and a position share 16 bytes, so a run's CAS,
// count bump and position store touch ONE line (three arrays cost three lines per run and two
// partial-line write-backs per slot at reset: PMC, profiles/r02), and the build role's scan / reset
// is one 16-byte load / store per slot.
struct __attribute__((aligned(16))) RdSlot {
  int64_t key;
  unsigned long long cp;        // low half: occurrences of the slot's id in the batch; high half:
                                // the sum of one position per run — THE position when the id
                                // occurs once (one 64-bit add per run sets both)
};

struct RunView {
  // global scratch hash, capacity cap_mask + 1 (+1 side slot); all-empty between uses
  RdSlot* hs;
  uint32_t* hlist;              // [slots][kLightMax] the positions of a light list (<= kLightMax
                                //     occurrences), runs in arrival order; never scanned or reset
  uint32_t cap_mask;
  // per batch
  uint32_t* uslot;              // [n] dense copies made by the build role: scratch slot,
  uint32_t* ucnt;               //     occurrences,
  uint32_t* upos;               //     a position (the only one when ucnt == 1) of unique index u
  int64_t* btab_key;            // [nblk][kRdStride] dumped LDS tables
  uint32_t* btab_val;           // [nblk][kRdStride] run_pack
  uint16_t* seg;                // [nblk * 1024] local positions grouped by run, ascending in a run
  ItemHdr* item_hdr;            // heavy work items
  uint32_t* item_runs;          // [items][64] run_pack of workgroup b0 + t (0: no run)
  uint32_t* ctr;                // [0] unique counter, [1] build waves done, [2] number of items,
                                // [3] rows the build role's probe reserved for the batch (ProbeOut)
  const int64_t* ids;
  uint32_t n;
  uint32_t nblk;                // ceil(n / 1024) <= 64; 0 = nothing to do
  uint32_t item_target;         // expected entries per heavy work item
  int64_t* uids;                // out (build role): unique ids, unspecified order
  uint32_t* n_unique;           // out (build role)
};

__global__ __launch_bounds__(256) void rd_clear_kernel(RunView d) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= d.cap_mask + 1u) {
    d.hs[i] = RdSlot{kEmptyKey, 0ull};
  }
  if (i < 4) d.ctr[i] = 0;
}

// LDS of the dedup role; the caller declares it (and may alias it with other roles' scratch)
struct RdLds {
  unsigned long long key[kRdStride];
  uint32_t cnt[kRdStride];
  uint32_t off[kRdStride];
  uint32_t first[kRdStride];
  uint16_t pos[kRdBlock];
  uint32_t wtot[16];
  // runs of > kLongRun positions are ordered through a 1024-bit bitmap each (rank = set bits
  // below), the first kMaxLongRuns of a workgroup; shorter ones by counting smaller entries
  uint32_t bm[kMaxLongRuns][32];
  uint32_t bmpre[kMaxLongRuns][32];
  uint8_t lslot[kRdStride + 3];
  uint32_t nlong;
};

// probe start of an id in a workgroup's dumped table (the side entry for kEmptyKey)
__device__ __forceinline__ uint32_t rd_home(int64_t id) {
  return uint32_t(hash_key(id) >> 40) & (kRdLds - 1);
}

__device__ __forceinline__ void rd_dedup_role(const RunView& d, uint32_t bid, RdLds& L,
                                              WaveTrace& wt) {
  const uint32_t t = threadIdx.x;
  for (uint32_t i = t; i < uint32_t(kRdStride); i += kRdBlock) {
    L.key[i] = static_cast<unsigned long long>(kEmptyKey);
    L.cnt[i] = 0;
    L.lslot[i] = 0xff;
  }
  if (t < kMaxLongRuns * 32) (&L.bm[0][0])[t] = 0;
  if (t == 0) L.nlong = 0;
  if (bid == 0 && t < 4) d.ctr[t] = 0;  // counters of the build role, which runs after this launch
  const uint32_t p = bid * kRdBlock + t;
  const bool valid = p < d.n;
  const int64_t id = valid ? d.ids[p] : 0;   // round trip 1
  lds_barrier();
  uint32_t ls = 0, arr = 0;
  if (valid) {
    if (id == kEmptyKey) {
      ls = kRdLds;
    } else {
      ls = rd_home(id);
      for (;;) {
        unsigned long long k = L.key[ls];
        if (k == static_cast<unsigned long long>(kEmptyKey)) {
          k = atomicCAS(&L.key[ls], static_cast<unsigned long long>(kEmptyKey),
                        static_cast<unsigned long long>(id));
          if (k == static_cast<unsigned long long>(kEmptyKey)) break;
        }
        if (k == static_cast<unsigned long long>(id)) break;
        ls = (ls + 1u) & (kRdLds - 1);
      }
    }
    arr = atomicAdd(&L.cnt[ls], 1u);
    if (arr == 0) L.first[ls] = t;
  }
  lds_barrier();
  wt.mark(0);
  // ---- the run's first arrival speaks for the whole run in the global scratch.  Round trip 2 is
  // its CAS on the id's home slot; the result is only looked at after the LDS work below, which
  // therefore overlaps the atomic's latency (and nobody waits at a barrier for the slowest CAS).
  const bool speaker = valid && arr == 0;
  uint32_t gs = 0;
  int64_t cas_old = kEmptyKey;
  if (speaker) {
    if (id == kEmptyKey) {
      gs = d.cap_mask + 1u;
      cas_old = static_cast<int64_t>(atomicCAS(reinterpret_cast<unsigned long long*>(&d.hs[gs].key),
                                               static_cast<unsigned long long>(kEmptyKey), 0ull));
    } else {
      gs = uint32_t(hash_key(id)) & d.cap_mask;
      cas_old = static_cast<int64_t>(atomicCAS(reinterpret_cast<unsigned long long*>(&d.hs[gs].key),
                                               static_cast<unsigned long long>(kEmptyKey),
                                               static_cast<unsigned long long>(id)));
    }
    if (L.cnt[ls] > uint32_t(kLongRun)) {
      const uint32_t sl = atomicAdd(&L.nlong, 1u);
      if (sl < uint32_t(kMaxLongRuns)) L.lslot[ls] = uint8_t(sl);
    }
  }
  wt.mark(1);
  // ---- run offsets: exclusive scan of the LDS counts (two entries per thread, side entry last)
  {
    const uint32_t a = L.cnt[2 * t], b = L.cnt[2 * t + 1];
    uint32_t incl = a + b;
    const int lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);  // (scalar: see WaveTrace)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    if (lane == 63) L.wtot[w] = incl;
    lds_barrier();
    uint32_t woff = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) woff += (i < w) ? L.wtot[i] : 0u;
    const uint32_t excl = woff + incl - (a + b);
    L.off[2 * t] = excl;
    L.off[2 * t + 1] = excl + a;
    if (t == kRdBlock - 1) L.off[kRdLds] = woff + incl;
  }
  lds_barrier();
  wt.mark(2);
  const uint32_t lsl = valid ? uint32_t(L.lslot[ls]) : 0xffu;
  if (valid) {
    if (lsl != 0xffu) atomicOr(&L.bm[lsl][t >> 5], 1u << (t & 31u));
    else L.pos[L.off[ls] + arr] = uint16_t(t);  // grouped by run, arrival order
  }
  lds_barrier();
  if (t < kMaxLongRuns * 32) {  // exclusive prefix of the set-bit counts over each bitmap's 32 words
    const uint32_t w = (&L.bm[0][0])[t];
    uint32_t incl = __popc(w);
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
      const uint32_t v = __shfl_up(incl, o, 32);
      if ((t & 31u) >= uint32_t(o)) incl += v;
    }
    (&L.bmpre[0][0])[t] = incl - __popc(w);
  }
  lds_barrier();
  // ---- ascending order inside the run: rank = number of smaller positions in it (8 LDS reads in
  // flight: a Zipf head key has runs of ~200)
  if (valid) {
    const uint32_t c = L.cnt[ls], o = L.off[ls];
    uint32_t rank = 0;
    if (lsl != 0xffu) {
      rank = L.bmpre[lsl][t >> 5] + __popc(L.bm[lsl][t >> 5] & ((1u << (t & 31u)) - 1u));
    } else if (c > 1) {
      uint32_t i = 0;
      for (; i + 8 <= c; i += 8) {
        uint32_t x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = L.pos[o + i + q];
#pragma unroll
        for (int q = 0; q < 8; ++q) rank += (x[q] < t) ? 1u : 0u;
      }
      for (; i < c; ++i) rank += (uint32_t(L.pos[o + i]) < t) ? 1u : 0u;
    }
    d.seg[bid * kRdBlock + o + rank] = uint16_t(t);
  }
  wt.mark(3);
  // ---- now the CAS: the home slot was free or already held the id (usual), else linear probing.
  // The count bump returns the number of occurrences other workgroups have registered so far =
  // where this run goes in the id's position list; it is in flight during the table dump.
  // (deliberately not initialised: only speakers read it, and a value merged with a constant at
  // the end of the branch would make the compiler wait for the atomic right there instead of
  // after the table dump below)
  uint32_t lbase;
  if (speaker) {
    if (id != kEmptyKey) {
      while (cas_old != kEmptyKey && cas_old != id) {
        gs = (gs + 1u) & d.cap_mask;
        cas_old = static_cast<int64_t>(atomicCAS(reinterpret_cast<unsigned long long*>(&d.hs[gs].key),
                                                 static_cast<unsigned long long>(kEmptyKey),
                                                 static_cast<unsigned long long>(id)));
      }
    }
    // (which workgroups hold a run of the id is not recorded: device-scope atomics are the scarce
    // resource of this role — ~36 G/s on MI355X, scripts/mstep_probe.py — and the few heavy ids
    // find their runs by probing every workgroup's directory, rd_find_run_opt)
    lbase = uint32_t(atomicAdd(&d.hs[gs].cp, (static_cast<unsigned long long>(p) << 32) |
                                                 static_cast<unsigned long long>(L.cnt[ls])));
  }
  // ---- the LDS table is the workgroup's run directory
  for (uint32_t i = t; i < uint32_t(kRdStride); i += kRdBlock) {
    d.btab_key[size_t(bid) * kRdStride + i] = static_cast<int64_t>(L.key[i]);
    d.btab_val[size_t(bid) * kRdStride + i] = run_pack(L.first[i] & 0x3ffu, L.off[i] & 0x3ffu, L.cnt[i]);
  }
  // ---- position list of a (so far) light id: this run's positions behind those already there.
  // A list that outgrows kLightMax is simply left incomplete — its id is heavy and served by runs.
  if (speaker) {
    const uint32_t c = L.cnt[ls];
    if (lbase + c <= uint32_t(kLightMax)) {
      const uint32_t o = L.off[ls];
      for (uint32_t i = 0; i < c; ++i)
        d.hlist[size_t(gs) * kLightMax + lbase + i] = bid * kRdBlock + uint32_t(L.pos[o + i]);
    }
  }
  wt.mark(4);
}

__global__ __launch_bounds__(kRdBlock) void rd_dedup_kernel(RunView d) {
  __shared__ RdLds L;
  WaveTrace wt(nullptr);
  rd_dedup_role(d, blockIdx.x, L, wt);
}

// ---------------------------------------------------------------------------------------------
// The same run dedup for a 256-THREAD workgroup, four positions per thread — so that it can ride in
// step_bwd (256-thread workgroups, five resident per CU): the dedup of batch s + 2 beside the
// update of batch s, two batches of look-ahead, and the forward launch is left with the lookups
// alone.  Same outputs, bit for bit the same format (scratch slots, position lists,
// run directory, `seg`): the consumers do not know which role produced them.
// LDS: five such workgroups must fit a CU's 160 KB beside nothing else, so the set's three
// per-entry words (count, run offset, long-run slot) share ONE 32-bit word and `first` is read
// back from `pos` when the table is dumped (it only matters for runs of one): 30.8 KB.
//   word = count (bits 0-10, <= 1024) | offset << 11 (bits 11-21) | long-run slot << 22 (31: none)
// ---------------------------------------------------------------------------------------------
constexpr int kRd4Threads = 256;
constexpr int kRd4PerThread = kRdBlock / kRd4Threads;   // 4
struct __attribute__((aligned(16))) RdLds4 {
  unsigned long long key[kRdStride];
  uint32_t co[kRdStride + 3];
  uint16_t pos[kRdBlock];
  uint32_t bm[kMaxLongRuns][32];
  uint32_t bmpre[kMaxLongRuns][32];
  uint32_t wtot[4];
  uint32_t nlong;
};
static_assert(sizeof(RdLds4) <= 31 * 1024, "five dedup workgroups (+ the other roles' statics) per CU");

__device__ __forceinline__ void rd_dedup4_role(const RunView& d, uint32_t bid, RdLds4& L) {
  // (four wavefronts that issue mostly LDS instructions, among sixteen that wait for memory: with
  // the default priority they get a fifth of their SIMDs' issue slots and the role's chain outlasts
  // the update's)
  __builtin_amdgcn_s_setprio(3);
  constexpr int Q = kRd4PerThread;
  constexpr uint32_t kNoLong = 31u;
  const uint32_t t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  for (uint32_t i = t; i < uint32_t(kRdStride); i += kRd4Threads) {
    L.key[i] = static_cast<unsigned long long>(kEmptyKey);
    L.co[i] = kNoLong << 22;
  }
  (&L.bm[0][0])[t] = 0;
  (&L.bm[0][0])[t + 256] = 0;
  if (t == 0) L.nlong = 0;
  if (bid == 0 && t < 4) d.ctr[t] = 0;  // counters of the build role, which runs after this launch
  int64_t id[Q];
  bool valid[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const uint32_t p = bid * kRdBlock + uint32_t(q) * kRd4Threads + t;
    valid[q] = p < d.n;
    id[q] = d.ids[valid[q] ? p : 0u];   // round trip 1 (four coalesced loads, masked below)
  }
  lds_barrier();
  uint32_t ls[Q], arr[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    ls[q] = 0;
    arr[q] = 1;   // (not a speaker)
    if (valid[q]) {
      if (id[q] == kEmptyKey) {
        ls[q] = kRdLds;
      } else {
        uint32_t h = rd_home(id[q]);
        for (;;) {
          unsigned long long k = L.key[h];
          if (k == static_cast<unsigned long long>(kEmptyKey)) {
            k = atomicCAS(&L.key[h], static_cast<unsigned long long>(kEmptyKey),
                          static_cast<unsigned long long>(id[q]));
            if (k == static_cast<unsigned long long>(kEmptyKey)) break;
          }
          if (k == static_cast<unsigned long long>(id[q])) break;
          h = (h + 1u) & (kRdLds - 1);
        }
        ls[q] = h;
      }
      arr[q] = atomicAdd(&L.co[ls[q]], 1u) & 0x7ffu;
    }
  }
  lds_barrier();
  // ---- the run's first arrival speaks for the whole run in the global scratch: its CAS is in
  // flight during the LDS work below
  uint32_t gs[Q];
  int64_t cas_old[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    gs[q] = 0;
    cas_old[q] = kEmptyKey;
    if (valid[q] && arr[q] == 0) {
      if (id[q] == kEmptyKey) {
        gs[q] = d.cap_mask + 1u;
        cas_old[q] = static_cast<int64_t>(atomicCAS(reinterpret_cast<unsigned long long*>(&d.hs[gs[q]].key),
                                                    static_cast<unsigned long long>(kEmptyKey), 0ull));
      } else {
        gs[q] = uint32_t(hash_key(id[q])) & d.cap_mask;
        cas_old[q] = static_cast<int64_t>(atomicCAS(reinterpret_cast<unsigned long long*>(&d.hs[gs[q]].key),
                                                    static_cast<unsigned long long>(kEmptyKey),
                                                    static_cast<unsigned long long>(id[q])));
      }
      if ((L.co[ls[q]] & 0x7ffu) > uint32_t(kLongRun)) {
        const uint32_t sl = atomicAdd(&L.nlong, 1u);
        if (sl < uint32_t(kMaxLongRuns)) atomicAnd(&L.co[ls[q]], ~((kNoLong ^ sl) << 22));
      }
    }
  }
  // ---- run offsets: exclusive scan of the counts, eight entries per thread (+ the side entry)
  {
    uint32_t c8[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      c8[k] = L.co[8u * t + uint32_t(k)] & 0x7ffu;
      sum += c8[k];
    }
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    if (lane == 63) L.wtot[wave] = incl;
    lds_barrier();
    uint32_t run = incl - sum;
#pragma unroll
    for (int i = 0; i < 4; ++i) run += (i < wave) ? L.wtot[i] : 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (run) atomicOr(&L.co[8u * t + uint32_t(k)], (run & 0x7ffu) << 11);   // (offsets <= 1024)
      run += c8[k];
    }
    if (t == kRd4Threads - 1 && run) atomicOr(&L.co[kRdLds], (run & 0x7ffu) << 11);
  }
  lds_barrier();
  uint32_t lsl[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    lsl[q] = kNoLong;
    if (valid[q]) {
      const uint32_t w = L.co[ls[q]];
      const uint32_t tq = uint32_t(q) * kRd4Threads + t;
      lsl[q] = (w >> 22) & 31u;
      if (lsl[q] != kNoLong) atomicOr(&L.bm[lsl[q]][tq >> 5], 1u << (tq & 31u));
      else L.pos[((w >> 11) & 0x7ffu) + arr[q]] = uint16_t(tq);  // grouped by run, arrival order
    }
  }
  lds_barrier();
#pragma unroll
  for (int h = 0; h < 2; ++h) {  // exclusive prefix of the set-bit counts over each bitmap's 32 words
    const uint32_t i = t + uint32_t(h) * 256u;
    const uint32_t w = (&L.bm[0][0])[i];
    uint32_t incl = __popc(w);
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
      const uint32_t v = __shfl_up(incl, o, 32);
      if ((i & 31u) >= uint32_t(o)) incl += v;
    }
    (&L.bmpre[0][0])[i] = incl - __popc(w);
  }
  lds_barrier();
  // ---- ascending order inside the run: rank = number of smaller positions in it
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    if (valid[q]) {
      const uint32_t w = L.co[ls[q]];
      const uint32_t c = w & 0x7ffu, o = (w >> 11) & 0x7ffu;
      const uint32_t tq = uint32_t(q) * kRd4Threads + t;
      uint32_t rank = 0;
      if (lsl[q] != kNoLong) {
        rank = L.bmpre[lsl[q]][tq >> 5] + __popc(L.bm[lsl[q]][tq >> 5] & ((1u << (tq & 31u)) - 1u));
      } else if (c > 1) {
        uint32_t i = 0;
        for (; i + 8 <= c; i += 8) {
          uint32_t x[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) x[k] = L.pos[o + i + k];
#pragma unroll
          for (int k = 0; k < 8; ++k) rank += (x[k] < tq) ? 1u : 0u;
        }
        for (; i < c; ++i) rank += (uint32_t(L.pos[o + i]) < tq) ? 1u : 0u;
      }
      d.seg[bid * kRdBlock + o + rank] = uint16_t(tq);
    }
  }
  // ---- now the CAS results: the home slot was free or already held the id (usual), else linear
  // probing; the count bump returns where this run goes in the id's position list
  uint32_t lbase[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    lbase[q] = 0;
    if (valid[q] && arr[q] == 0) {
      if (id[q] != kEmptyKey) {
        while (cas_old[q] != kEmptyKey && cas_old[q] != id[q]) {
          gs[q] = (gs[q] + 1u) & d.cap_mask;
          cas_old[q] = static_cast<int64_t>(atomicCAS(reinterpret_cast<unsigned long long*>(&d.hs[gs[q]].key),
                                                      static_cast<unsigned long long>(kEmptyKey),
                                                      static_cast<unsigned long long>(id[q])));
        }
      }
      const uint32_t p = bid * kRdBlock + uint32_t(q) * kRd4Threads + t;
      lbase[q] = uint32_t(atomicAdd(&d.hs[gs[q]].cp, (static_cast<unsigned long long>(p) << 32) |
                                                        static_cast<unsigned long long>(L.co[ls[q]] & 0x7ffu)));
    }
  }
  // ---- the LDS table is the workgroup's run directory (first position: only read for runs of one,
  // whose single position sits at pos[offset])
  for (uint32_t i = t; i < uint32_t(kRdStride); i += kRd4Threads) {
    const uint32_t w = L.co[i];
    const uint32_t c = w & 0x7ffu, o = (w >> 11) & 0x7ffu;
    d.btab_key[size_t(bid) * kRdStride + i] = static_cast<int64_t>(L.key[i]);
    d.btab_val[size_t(bid) * kRdStride + i] = run_pack(c == 1 ? uint32_t(L.pos[o & 0x3ffu]) : 0u, o & 0x3ffu, c);
  }
  // ---- position list of a (so far) light id: this run's positions behind those already there
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    if (valid[q] && arr[q] == 0) {
      const uint32_t w = L.co[ls[q]];
      const uint32_t c = w & 0x7ffu, o = (w >> 11) & 0x7ffu;
      if (lbase[q] + c <= uint32_t(kLightMax)) {
        for (uint32_t i = 0; i < c; ++i)
          d.hlist[size_t(gs[q]) * kLightMax + lbase[q] + i] = bid * kRdBlock + uint32_t(L.pos[o + i]);
      }
    }
  }
}

// run descriptor of `id` in workgroup b's table (the id is known to have a run there).  Key and
// descriptor of the home entry are fetched together: one round trip unless the LDS set had a
// collision there.
__device__ __forceinline__ uint32_t rd_find_run(const RunView& d, uint32_t b, int64_t id) {
  const int64_t* kt = d.btab_key + size_t(b) * kRdStride;
  const uint32_t* vt = d.btab_val + size_t(b) * kRdStride;
  if (id == kEmptyKey) return vt[kRdLds];
  uint32_t ls = rd_home(id);
  int64_t k = kt[ls];
  uint32_t v = vt[ls];
  int left = kRdLds;  // (bounded: a directory that lacks the id would mean a corrupted dedup)
  while (k != id) {
    if (--left == 0) return 0u;
    ls = (ls + 1u) & (kRdLds - 1);
    k = kt[ls];
    v = vt[ls];
  }
  return v;
}

// the same for an id that may have no run in workgroup b: 0 when the probe sequence ends at an empty
// entry (run_pack of a present run is never 0: its count is >= 1)
__device__ __forceinline__ uint32_t rd_find_run_opt(const RunView& d, uint32_t b, int64_t id) {
  const int64_t* kt = d.btab_key + size_t(b) * kRdStride;
  const uint32_t* vt = d.btab_val + size_t(b) * kRdStride;
  if (id == kEmptyKey) return run_cnt(vt[kRdLds]) ? vt[kRdLds] : 0u;
  uint32_t ls = rd_home(id);
  int64_t k = kt[ls];
  uint32_t v = vt[ls];
  int left = kRdLds;
  while (k != id) {
    if (k == kEmptyKey || --left == 0) return 0u;
    ls = (ls + 1u) & (kRdLds - 1);
    k = kt[ls];
    v = vt[ls];
  }
  return v;
}

// workgroups per item for a list of c occurrences: power of two, ~kItemTarget entries expected
// A list that fits one item (<= 2 * target entries) stays whole; a longer one is cut into items of
// target / 2 .. target expected entries: its items are followed by a hand-off (partial rows, arrival
// counter, last arriver) and are the longest chain of the launch, so each of them gets one window
// round instead of two.
__device__ __forceinline__ uint32_t rd_item_blocks(uint32_t c, uint32_t target) {
  uint32_t nbk = 64;
  const uint64_t lim = uint64_t(target) * 64 * (c > 2 * target ? 1 : 2);
  while (nbk > 1 && uint64_t(c) * nbk > lim) nbk >>= 1;
  return nbk;
}

// ---------------------------------------------------------------------------------------------
// Build: unique numbering + heavy work list of a deduplicated batch, and the scratch reset.
// Every wavefront scans 256 scratch slots per trip (four coalesced 64-slot reads of each array in
// flight), compacts the occupied ones into the dense arrays with ONE counter bump, and handles the
// heavy ids it found with the whole wavefront (lane = workgroup index of a potential run).
// Slot order is hash order, so hot ids are spread evenly over the wavefronts.
// ---------------------------------------------------------------------------------------------
constexpr int kBuildSlotsPerLane = 4;

// Optional second output of the numbering (the id-sharded step, mhte_shard_kernels.h): every
// distinct id also gets a slot in the block of its owner floormod(id, world) — FusedReorderByIndices'
// shard-major packing (RT/ops/fused_reorder_by_indices.cc:75-123) — through one LDS histogram per
// workgroup trip and one global add per (trip, owner) on the block's header word.
constexpr int kMaxShards = 64;
__device__ __forceinline__ uint32_t shard_of_id(int64_t id, uint32_t nshards) {
  const int64_t m = id % int64_t(nshards);
  return uint32_t(m < 0 ? m + int64_t(nshards) : m);
}
struct PackCtl {
  int64_t* send_ids;     // [world][ids_block]
  // direct peer stores (mhte_shard_host.h): the id goes straight into its owner's window — peer_win[owner]
  // + peer_ids_off (this rank's block of the slot's id buffer there); the counts stay in send_ids' headers
  const unsigned long long* peer_win;   // [world] device array of window addresses; nullptr: send_ids
  unsigned long long peer_ids_off;
  uint32_t* slot_off;    // [n_max] of this table: float offset of unique index u's row slot
  uint32_t* flags;
  uint32_t world, ids_block, rows_block;
  uint32_t hdr_word;     // the table's count word in a block's header
  uint32_t cap, id_off, row_off, dim;
};

// Work items of ONE heavy id (a whole wavefront; lane b looks the id up in dedup workgroup b's run
// directory): a power-of-two range of workgroups per item, the item's run descriptors next to it.
// hrow / hspec / hloc: what a table probe of the id found (ProbeOut; kNoRow / kNoRow / 0 without).
__device__ __forceinline__ void rd_emit_items(const RunView& d, int64_t id, uint32_t hu, uint32_t c,
                                              uint32_t hrow, uint32_t hspec, unsigned long long hloc,
                                              int lane) {
  const uint32_t val = (uint32_t(lane) < d.nblk) ? rd_find_run_opt(d, uint32_t(lane), id) : 0u;
  const unsigned long long bm = __ballot(val != 0u);
  const uint32_t nbk = rd_item_blocks(c, d.item_target);
  const uint32_t b0 = uint32_t(lane) & ~(nbk - 1u);
  const unsigned long long rmask = (nbk == 64 ? ~0ull : ((1ull << nbk) - 1ull)) << b0;
  const bool leader = (uint32_t(lane) == b0) && (bm & rmask) != 0ull;
  const unsigned long long lm = __ballot(leader);
  const uint32_t nitems = __popcll(lm);
  uint32_t w0 = 0;
  if (lane == 0) w0 = atomicAdd(&d.ctr[2], nitems);
  w0 = __shfl(w0, 0);
  // item index of this lane's range = rank of its leader among the leaders
  const uint32_t k = __popcll(lm & ((1ull << b0) - 1ull));
  if (leader) {
    ItemHdr hd;
    hd.id = id;
    hd.u = hu;
    hd.meta = b0 | (nbk << 8) | (k << 16) | (nitems << 24);
    hd.row = hrow;
    hd.spec = hspec;
    hd.loc = hloc;
    d.item_hdr[w0 + k] = hd;
  }
  if ((bm & rmask) != 0ull) d.item_runs[size_t(w0 + k) * 64 + (uint32_t(lane) - b0)] = val;
}

// Optional third output of the numbering (PROBE; the single-table training step): the table is
// probed for every distinct id while its dense index is being assigned — a launch (and with the
// pipelined step: a whole update) before the batch's own update needs the answer.
//   found    -> urow = row handle, uloc = bucket * 4 + slot.  The update reaches the row without
//               reading a bucket line; it VERIFIES the hint (the key at uloc must still be the id:
//               displacement, doubling, eviction, clear / restore may have moved or removed the entry
//               since) and probes as before when it does not hold.  Row handles are never recycled, so
//               a hint that verifies is exact.
//   missing  -> spec = a row handle reserved for the id now, ONE bump of the table's allocation
//               counter per workgroup trip here instead of one per wavefront inside the update —
//               that single address takes every allocation of a launch and same-address atomics are
//               served one after the other (~18 ns each): on the update's critical path they were
//               its longest wait.  An id that turns out to exist by then (the update running beside
//               this probe inserted it) gives the key back (upsert_complete: lostm); the handle
//               is stranded (kSpecSlackRows).
// The probe may run BESIDE an update of the same table (step_bwd: apply of batch s | build of
// batch s + 1).  A claim publishes the key first and the row handle after it, and an empty slot
// carries kNoRow (cuckoopath_move, evict, split): a key seen beside kNoRow is an insert in flight —
// treated as "not found, nothing reserved", the update's own probe will find it.
constexpr unsigned long long kNoLoc = ~0ull;
struct __attribute__((aligned(16))) URec {   // everything the update needs about unique index u: ONE 32-byte load
  int64_t id;
  uint32_t cnt;              // occurrences in the batch
  uint32_t pos;              // a position (the only one when cnt == 1)
  uint32_t slot;             // scratch slot: where the id's position list lives
  uint32_t row;              // loc != kNoLoc: the id's row handle; else a row reserved for it (kNoRow: none)
  unsigned long long loc;    // bucket * 4 + slot of a found id, or kNoLoc
};
struct ProbeOut {
  URec* urec;                // [n_max]
  uint32_t reserve;          // 0: hints only — a table with an admission filter may not insert a
                             // missing id at all, its update allocates for itself
};

template <bool PACK = false, bool PROBE = false>
__device__ __forceinline__ void rd_build_role(const RunView& d, uint32_t light_max, uint32_t bid,
                                              uint32_t nblocks, const PackCtl* pc = nullptr,
                                              const TableView* tv = nullptr, ProbeOut po = ProbeOut{}) {
  constexpr int Q = kBuildSlotsPerLane;
  __shared__ uint32_t sh_tot[4];
  __shared__ uint32_t sh_base;
  __shared__ uint32_t sh_mtot[PROBE ? 4 : 1];
  __shared__ uint32_t sh_mbase;
  __shared__ uint8_t sh_stage[PROBE ? 4 * 256 : 4];   // per wavefront: (q, lane) of the trip's r-th id
  __shared__ uint32_t sh_pc[PACK ? kMaxShards : 1], sh_pb[PACK ? kMaxShards : 1];
  const uint32_t t = threadIdx.x;
  const int lane = t & 63;
  const uint32_t nslots = d.cap_mask + 2u;  // + the side slot of kEmptyKey
  // (256-thread workgroups; the trip count is the same for the four wavefronts of a workgroup)
  for (uint32_t bb = bid * (256u * Q); bb < nslots; bb += nblocks * (256u * Q)) {
    const uint32_t base = bb + (t >> 6) * (64u * Q);
    int64_t key[Q];
    uint32_t cnt[Q], pos[Q];
    unsigned long long occ[Q];
    uint32_t total = 0;
    if (PACK) {
      if (t < pc->world) sh_pc[t] = 0;
      lds_barrier();
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const uint32_t sl = base + q * 64 + lane;
      const bool in = sl < nslots;
      const RdSlot sv = d.hs[in ? sl : 0u];   // (one 16-byte load from a safe index, masked below)
      key[q] = in ? sv.key : kEmptyKey;
      cnt[q] = in ? uint32_t(sv.cp) : 0u;
      pos[q] = in ? uint32_t(sv.cp >> 32) : 0u;
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const uint32_t sl = base + q * 64 + lane;
      const bool o = key[q] != kEmptyKey;
      if (o) {  // clean-after-use: the scratch is all-empty again after this launch
        d.hs[sl] = RdSlot{kEmptyKey, 0ull};   // (one 16-byte store)
        if (sl == d.cap_mask + 1u) key[q] = kEmptyKey;  // the side slot stands for that id itself
      }
      occ[q] = __ballot(o);
      total += uint32_t(__popcll(occ[q]));
    }
    uint32_t pown[Q], prank[Q];
    if (PACK) {
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        pown[q] = prank[q] = 0;
        if ((occ[q] >> lane) & 1ull) {
          pown[q] = shard_of_id(key[q], pc->world);
          prank[q] = atomicAdd(&sh_pc[pown[q]], 1u);
        }
      }
    }
    // one counter bump per WORKGROUP and trip (hundreds of wavefronts bumping one address would
    // queue behind each other for longer than the whole launch should take)
    if (lane == 0) sh_tot[t >> 6] = total;
    lds_barrier();
    if (t == 0) {
      const uint32_t all = sh_tot[0] + sh_tot[1] + sh_tot[2] + sh_tot[3];
      sh_base = all ? atomicAdd(&d.ctr[0], all) : 0u;
    }
    if (PACK && t < pc->world) {
      const uint32_t c = sh_pc[t];
      sh_pb[t] = c ? uint32_t(atomicAdd(reinterpret_cast<unsigned long long*>(
                                            pc->send_ids + size_t(t) * pc->ids_block + pc->hdr_word),
                                        static_cast<unsigned long long>(c)))
                   : 0u;
    }
    lds_barrier();
    uint32_t k0 = sh_base;
    for (uint32_t w2 = 0; w2 < (t >> 6); ++w2) k0 += sh_tot[w2];
    const uint32_t kw0 = k0;   // dense index of this wavefront's first id of the trip
    const uint32_t wg_max = max(max(sh_tot[0], sh_tot[1]), max(sh_tot[2], sh_tot[3]));
    lds_barrier();  // (sh_tot / sh_base are rewritten by the next trip)
    uint32_t kq[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      kq[q] = k0 + uint32_t(__popcll(occ[q] & ((1ull << lane) - 1ull)));
      k0 += uint32_t(__popcll(occ[q]));
      if ((occ[q] >> lane) & 1ull) {
        d.uids[kq[q]] = key[q];
        d.uslot[kq[q]] = base + q * 64 + lane;
        d.ucnt[kq[q]] = cnt[q];
        d.upos[kq[q]] = pos[q];
        if (PACK) {
          const uint32_t sl = sh_pb[pown[q]] + prank[q];
          if (sl < pc->cap) {
            if (pc->peer_win)
              reinterpret_cast<int64_t*>(pc->peer_win[pown[q]] + pc->peer_ids_off)[pc->id_off + sl] = key[q];
            else
              pc->send_ids[size_t(pown[q]) * pc->ids_block + pc->id_off + sl] = key[q];
            pc->slot_off[kq[q]] = pown[q] * pc->rows_block + pc->row_off + sl * pc->dim;
          } else {   // no room in the owner's block: zero row back, gradient dropped, flagged
            pc->slot_off[kq[q]] = 0xffffffffu;
            atomicOr(pc->flags, 1u);
          }
        }
      }
    }
    if (!PROBE) {
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        unsigned long long hm = occ[q] & __ballot(cnt[q] > light_max);
        while (hm) {
          const int src = __ffsll(static_cast<long long>(hm)) - 1;
          hm &= hm - 1ull;
          rd_emit_items(d, __shfl(key[q], src), __shfl(kq[q], src), __shfl(cnt[q], src), kNoRow, kNoRow, 0ull,
                        lane);
        }
      }
    } else {
      // ---- the trip's ids, one per lane in dense-index order (a wavefront's 256 slots hold ~25 ids at
      // the scratch's load: one round), probed in the table; then the work items of the heavy ones
      typedef long long i64x2 __attribute__((ext_vector_type(2)));
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      uint8_t* const stg = sh_stage + (t >> 6) * 256u;
      uint32_t run = 0;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        if ((occ[q] >> lane) & 1ull) stg[run + uint32_t(__popcll(occ[q] & ((1ull << lane) - 1ull)))] = uint8_t(q * 64 + lane);
        run += uint32_t(__popcll(occ[q]));
      }
      lds_wave_sync();
#pragma unroll 1
      for (uint32_t r0 = 0; r0 < wg_max; r0 += 64) {   // (workgroup-uniform bound: barriers inside)
        const uint32_t r = r0 + uint32_t(lane);
        const bool have = r < run;
        const uint32_t src = have ? uint32_t(stg[r]) : 0u;
        const int sl = int(src & 63u);
        int64_t kid = 0;
        uint32_t kc = 0, kp = 0;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const int64_t kk = __shfl(key[q], sl);
          const uint32_t cc = __shfl(cnt[q], sl), pp = __shfl(pos[q], sl);
          if (int(src >> 6) == q) {
            kid = kk;
            kc = cc;
            kp = pp;
          }
        }
        const bool act = have && kid != kEmptyKey;   // (the side slot's key: the update's own path)
        const int64_t pid = act ? kid : 0;           // (lanes without an id probe id 0's lines)
        const uint64_t hv = hash_key(pid);
        const uint64_t i1 = index_hash(tv->hp, hv);
        const uint64_t i2 = alt_index(tv->hp, partial_key(hv), i1);
        const GBucket* b1 = global_bucket(tv->buckets + i1);
        const GBucket* b2 = global_bucket(tv->buckets + i2);
        const i64x2 k0a = *(const MHTE_GLOBAL i64x2*)(&b1->key[0]);
        const i64x2 k0b = *(const MHTE_GLOBAL i64x2*)(&b1->key[2]);
        const i64x2 k1a = *(const MHTE_GLOBAL i64x2*)(&b2->key[0]);
        const i64x2 k1b = *(const MHTE_GLOBAL i64x2*)(&b2->key[2]);
        const u32x4 r0v = *(const MHTE_GLOBAL u32x4*)(&b1->row[0]);
        const u32x4 r1v = *(const MHTE_GLOBAL u32x4*)(&b2->row[0]);
        int fs = -1;
        uint32_t fr = kNoRow;
        if (k1b.y == pid) { fs = 7; fr = r1v.w; }
        if (k1b.x == pid) { fs = 6; fr = r1v.z; }
        if (k1a.y == pid) { fs = 5; fr = r1v.y; }
        if (k1a.x == pid) { fs = 4; fr = r1v.x; }
        if (k0b.y == pid) { fs = 3; fr = r0v.w; }
        if (k0b.x == pid) { fs = 2; fr = r0v.z; }
        if (k0a.y == pid) { fs = 1; fr = r0v.y; }
        if (k0a.x == pid) { fs = 0; fr = r0v.x; }
        const bool found = act && fs >= 0 && fr != kNoRow;   // (a key beside kNoRow: an insert in flight)
        const bool miss = act && fs < 0;
        uint32_t spec = kNoRow;
        if (po.reserve) {   // rows for the ids the table lacks: one bump per workgroup and round
          const unsigned long long mm = __ballot(miss);
          if (lane == 0) sh_mtot[t >> 6] = uint32_t(__popcll(mm));
          lds_barrier();
          if (t == 0) {
            const unsigned long long all = sh_mtot[0] + sh_mtot[1] + sh_mtot[2] + sh_mtot[3];
            sh_mbase = all ? uint32_t(atomicAdd(&tv->ctr->alloc, (all << 32) | all)) : 0u;
            if (all) {   // (the keys are counted now, inserted by the batch's update: Counters::reserved)
              atomicAdd(&tv->ctr->reserved, uint32_t(all));
              atomicAdd(&d.ctr[3], uint32_t(all));
            }
          }
          lds_barrier();
          uint32_t m0 = sh_mbase;
          for (uint32_t w2 = 0; w2 < (t >> 6); ++w2) m0 += sh_mtot[w2];
          lds_barrier();  // (sh_mtot / sh_mbase are rewritten by the next round)
          if (miss) spec = m0 + uint32_t(__popcll(mm & ((1ull << lane) - 1ull)));
        }
        const unsigned long long floc = found ? (((fs & 4) ? i2 : i1) << 2) | uint64_t(fs & 3) : kNoLoc;
        const uint32_t frow = found ? fr : spec;
        if (have) {
          URec rec;
          rec.id = kid;
          rec.cnt = kc;
          rec.pos = kp;
          rec.slot = base + (src >> 6) * 64u + (src & 63u);
          rec.row = frow;
          rec.loc = floc;
          po.urec[kw0 + r] = rec;
        }
        unsigned long long hm = __ballot(have && kc > light_max);
        while (hm) {
          const int s2 = __ffsll(static_cast<long long>(hm)) - 1;
          hm &= hm - 1ull;
          const unsigned long long hl = __shfl(floc, s2);
          rd_emit_items(d, __shfl(kid, s2), kw0 + r0 + uint32_t(s2), __shfl(kc, s2),
                        hl != kNoLoc ? __shfl(frow, s2) : kNoRow, hl != kNoLoc ? kNoRow : __shfl(frow, s2),
                        hl != kNoLoc ? hl : 0ull, lane);
        }
      }
    }
  }
  // ---- the last workgroup to finish publishes the unique count (thread 0 made every bump of
  // ctr[0] itself and they have returned, so they are ordered before its arrival here)
  if (t == 0) {
    if (atomicAdd(&d.ctr[1], 1u) == nblocks - 1) *d.n_unique = atomicAdd(&d.ctr[0], 0u);
  }
}

__global__ __launch_bounds__(256) void rd_build_kernel(RunView d, uint32_t light_max) {
  rd_build_role(d, light_max, blockIdx.x, gridDim.x);
}
// ... with the table probe (ProbeOut): the first batch of a pipeline, the unpipelined step
__global__ __launch_bounds__(256) void rd_build_probe_kernel(RunView d, uint32_t light_max, TableView tv,
                                                             ProbeOut po) {
  rd_build_role<false, true>(d, light_max, blockIdx.x, gridDim.x, nullptr, &tv, po);
}
// the table probe on its own, for a batch that was numbered without it (mhte_step_dedup): one lane
// per distinct id over the dense arrays
// (light_max: ids with longer lists are applied by the item workgroups from their work-item headers,
// which were written without a probe — no row is reserved for those here, they allocate their own)
__global__ __launch_bounds__(256) void rd_probe_kernel(RunView d, TableView tv, ProbeOut po, uint32_t n_max,
                                                       uint32_t light_max) {
  typedef long long i64x2 __attribute__((ext_vector_type(2)));
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  __shared__ uint32_t sh_mtot[4];
  __shared__ uint32_t sh_mbase;
  const uint32_t t = threadIdx.x;
  const int lane = t & 63;
  const uint32_t nu = min(n_max, d.ctr[0]);
#pragma unroll 1
  for (uint32_t u0 = blockIdx.x * 256u; u0 < nu; u0 += gridDim.x * 256u) {   // (workgroup-uniform)
    const uint32_t u = u0 + t;
    const bool have = u < nu;
    const uint32_t us = have ? u : 0u;
    const int64_t kid = d.uids[us];
    const bool act = have && kid != kEmptyKey;
    const int64_t pid = act ? kid : 0;
    const uint64_t hv = hash_key(pid);
    const uint64_t i1 = index_hash(tv.hp, hv);
    const uint64_t i2 = alt_index(tv.hp, partial_key(hv), i1);
    const GBucket* b1 = global_bucket(tv.buckets + i1);
    const GBucket* b2 = global_bucket(tv.buckets + i2);
    const i64x2 k0a = *(const MHTE_GLOBAL i64x2*)(&b1->key[0]);
    const i64x2 k0b = *(const MHTE_GLOBAL i64x2*)(&b1->key[2]);
    const i64x2 k1a = *(const MHTE_GLOBAL i64x2*)(&b2->key[0]);
    const i64x2 k1b = *(const MHTE_GLOBAL i64x2*)(&b2->key[2]);
    const u32x4 r0v = *(const MHTE_GLOBAL u32x4*)(&b1->row[0]);
    const u32x4 r1v = *(const MHTE_GLOBAL u32x4*)(&b2->row[0]);
    URec rec;
    rec.id = kid;
    rec.cnt = d.ucnt[us];
    rec.pos = d.upos[us];
    rec.slot = d.uslot[us];
    int fs = -1;
    uint32_t fr = kNoRow;
    if (k1b.y == pid) { fs = 7; fr = r1v.w; }
    if (k1b.x == pid) { fs = 6; fr = r1v.z; }
    if (k1a.y == pid) { fs = 5; fr = r1v.y; }
    if (k1a.x == pid) { fs = 4; fr = r1v.x; }
    if (k0b.y == pid) { fs = 3; fr = r0v.w; }
    if (k0b.x == pid) { fs = 2; fr = r0v.z; }
    if (k0a.y == pid) { fs = 1; fr = r0v.y; }
    if (k0a.x == pid) { fs = 0; fr = r0v.x; }
    const bool found = act && fs >= 0 && fr != kNoRow;
    const bool miss = act && fs < 0 && rec.cnt <= light_max;
    uint32_t spec = kNoRow;
    if (po.reserve) {
      const unsigned long long mm = __ballot(miss);
      if (lane == 0) sh_mtot[t >> 6] = uint32_t(__popcll(mm));
      __syncthreads();
      if (t == 0) {
        const unsigned long long all = sh_mtot[0] + sh_mtot[1] + sh_mtot[2] + sh_mtot[3];
        sh_mbase = all ? uint32_t(atomicAdd(&tv.ctr->alloc, (all << 32) | all)) : 0u;
        if (all) {
          atomicAdd(&tv.ctr->reserved, uint32_t(all));
          atomicAdd(&d.ctr[3], uint32_t(all));
        }
      }
      __syncthreads();
      uint32_t m0 = sh_mbase;
      for (uint32_t w2 = 0; w2 < (t >> 6); ++w2) m0 += sh_mtot[w2];
      __syncthreads();
      if (miss) spec = m0 + uint32_t(__popcll(mm & ((1ull << lane) - 1ull)));
    }
    rec.row = found ? fr : spec;
    rec.loc = found ? (((fs & 4) ? i2 : i1) << 2) | uint64_t(fs & 3) : kNoLoc;
    if (have) po.urec[u] = rec;
  }
}
// a batch whose probe is void (its table was cleared — restored — after it was numbered): the heavy
// work items forget what that probe left in their headers; rd_probe_kernel then probes the light ids
// again and the items take the update's own probe / allocation path
__global__ __launch_bounds__(256) void rd_items_unhint_kernel(RunView d) {
  const uint32_t n = d.ctr[2];
  for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < n; w += gridDim.x * blockDim.x) {
    d.item_hdr[w].row = kNoRow;
    d.item_hdr[w].spec = kNoRow;
    d.item_hdr[w].loc = 0ull;
  }
}
// keys of reservations that will never be used (a numbered batch that is dropped instead of applied)
// go back to the table's live-key count; the row handles stay stranded (rows are not recycled)
__global__ __launch_bounds__(256) void rd_unreserve_kernel(const URec* __restrict__ urec,
                                                           const uint32_t* __restrict__ n_unique,
                                                           uint32_t n_max, Counters* ctr) {
  const uint32_t n = min(n_max, *n_unique);
  uint32_t mine = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    mine += (urec[i].loc == kNoLoc && urec[i].row != kNoRow) ? 1u : 0u;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mine += __shfl_xor(mine, o);
  if ((threadIdx.x & 63) == 0 && mine) {
    atomicAdd(&ctr->alloc, ~((static_cast<unsigned long long>(mine) << 32) - 1ull));
    atomicSub(&ctr->reserved, mine);
  }
}

// ---------------------------------------------------------------------------------------------
// Apply: duplicate-gradient sum + upsert + optimizer of a deduplicated batch.
// ---------------------------------------------------------------------------------------------
struct ApplyCtl {
  const float* grads;
  float* grad_u;          // [n_max, dim] summed gradients of ids deferred to the displacement pass
  uint32_t* pending;
  float* part;            // [items, dim] partial rows of multi-item lists
  uint32_t* arrive;       // [n] arrival counters, kept zeroed
  int64_t n_max;
  uint32_t light_max;     // 0xffffffff: every list strictly sequential (MHTE_EXACT_ORDER)
  uint32_t nblk_items;    // workgroups [0, nblk_items) take work items, the rest ids
  uint32_t nblk_ids;
  // (multi-table step, HINT) what the forward launch's lookup found for unique index u: row handle
  // (kNoRow: absent) and bucket * 4 + slot; nullptr: probe as usual
  const uint32_t* urow;
  const unsigned long long* uloc;
  const uint32_t* uts;          // timestamp the forward launch saw in the id's slot (or nullptr)
  const URec* urec;             // (single-table step) the build role's packed record per unique index:
                                // replaces the dense arrays, urow / uloc in ONE load
  uint32_t trusted;             // 1: nothing has touched the table since urow / uloc were written
                                // (multi-table step, Table::mut_epoch): the hints need no check
  uint32_t pre_summed;          // 1 (MHTE_EXACT_ORDER): every heavy list's strictly sequential sum is already in
                                // part[its first item] (rd_exact_sum_kernel): the item workgroups only apply
};

// row of a found id, fetched while the gradient chain is in flight
template <int VEC>
struct RowRegs {
  Vec<VEC> w, s1;   // (FTRL's second state vector is read by optimize_row_pre itself: keeping
                    // it here as well costs four registers of every SGD / Adagrad step)
};

template <int VEC, bool ONESEG>
__device__ __forceinline__ void row_prefetch(const TableView& tv, const float* rp, uint32_t e,
                                             RowRegs<VEC>& r) {
  if (e >= tv.dim) return;
  uint32_t k = 0;
  const SegDesc sd = seg_of<ONESEG>(tv, e, k);
  const uint32_t le = e - sd.w_off;
  r.w.load(rp + e);
  if (sd.opt == kOptAdagrad || sd.opt == kOptFtrl) r.s1.load(rp + sd.st_off + le);
}

// one optimizer step with the row already in registers (is_new: start from the initializer)
template <int VEC, bool ONESEG>
__device__ __forceinline__ void optimize_row_pre(const TableView& tv, float* rp, bool is_new,
                                                 uint32_t e, const Vec<VEC>& g, const ApplyArgs& a,
                                                 RowRegs<VEC>& r) {
  if (e >= tv.dim) return;
  uint32_t k = 0;
  const SegDesc sd = seg_of<ONESEG>(tv, e, k);
  const uint32_t le = e - sd.w_off;
  const float lr = a.lr[k];
  float* st1 = rp + sd.st_off + le;
  float* st2 = st1 + sd.dim;
  const bool has1 = sd.opt == kOptAdagrad || sd.opt == kOptFtrl;
  const bool has2 = sd.opt == kOptFtrl;
  Vec<VEC> s2;
  vec_zero(s2);
  if (is_new) {
    const SegDesc si = seg_for_init(sd);
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      r.w.v[c] = init_weight(si, rp + e + c);
      r.s1.v[c] = sd.p[0];
    }
  } else if (has2) {
    s2.load(st2);
  }
  if (sd.opt == kOptSgd) {
    const float slr = opaque_f(lr);
#pragma unroll
    for (int c = 0; c < VEC; ++c) r.w.v[c] = sgd_step(r.w.v[c], g.v[c], slr);
  } else if (sd.opt == kOptAdagrad) {
    const float alr = opaque_f(lr), wd = opaque_f(sd.p[1]);
    if (MHTE_AVX_FORM(sd)) {   // the reference's AVX2 form (adagrad_step_avx), opt-in
#pragma unroll
      for (int c = 0; c < VEC; ++c)
        adagrad_step_avx(r.w.v[c], r.s1.v[c], g.v[c], alr, wd, uint32_t(le + c) < (uint32_t(sd.dim) & ~7u));
    } else {
#pragma unroll
      for (int c = 0; c < VEC; ++c) adagrad_step(r.w.v[c], r.s1.v[c], g.v[c], alr, wd);
    }
  } else {
    const float flr = opaque_f(lr), beta = opaque_f(sd.p[1]), l1 = opaque_f(sd.p[2]), l2 = opaque_f(sd.p[3]);
#pragma unroll
    for (int c = 0; c < VEC; ++c) ftrl_step(r.w.v[c], r.s1.v[c], s2.v[c], g.v[c], flr, beta, l1, l2);
  }
  row_store<VEC>(rp + e, r.w);
  if (has1) row_store<VEC>(st1, r.s1);
  if (has2) row_store<VEC>(st2, s2);
}

// an id whose two buckets are full goes to the displacement pass's list (rare: the counter's address
// is formed here, not in front of the trip loop — see opaque_f)
__device__ __forceinline__ void defer_id(const TableView& tv, uint32_t* pending, uint32_t u) {
  auto* ctr = tv.ctr;
#ifndef MHTE_NO_ANTIHOIST
  asm volatile("" : "+v"(ctr));
#endif
  pending[atomicAdd(&ctr->n_pending, 1u)] = u;
}

// Probe state of one id per G-lane group, split in two so the caller can put work between the
// bucket loads and their use.
template <int G>
struct Probe {
  GBucket* b;
  int64_t k;
  uint32_t row;
};
template <int G>
__device__ __forceinline__ Probe<G> probe_issue(const TableView& tv, int64_t id, bool valid, int j) {
  const uint64_t hv = hash_key(id);
  const uint64_t i1 = index_hash(tv.hp, hv);
  const uint64_t i2 = alt_index(tv.hp, partial_key(hv), i1);
  Probe<G> p;
  p.b = global_bucket(tv.buckets + ((j < 4) ? i1 : i2));
  p.k = kEmptyKey;
  p.row = kNoRow;
  if (valid && id != kEmptyKey && j < 8) {
    p.k = p.b->key[j & 3];
    p.row = p.b->row[j & 3];
  }
  return p;
}

// Sum of grads[pos[q]] for q in [0, c) in list order, positions in LDS, 8 rows in flight.
template <int VEC>
__device__ __forceinline__ void sum_list_lds(const float* __restrict__ grads, uint32_t dim,
                                             uint32_t e, bool ev, const uint32_t* pos, uint32_t c,
                                             Vec<VEC>& acc) {
#pragma unroll 1
  for (uint32_t q = 0; q < c; q += 8) {
    uint32_t ps[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) ps[t] = (q + t < c) ? pos[q + t] : 0u;
    Vec<VEC> v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) vec_zero(v[t]);
#pragma unroll
    for (int t = 0; t < 8; ++t)
      if (q + t < c && ev) v[t].load(grads + int64_t(ps[t]) * dim + e);
#pragma unroll
    for (int t = 0; t < 8; ++t)
      if (q + t < c && ev) vec_add(acc, v[t]);
  }
}

// upsert_resolve (mhte_kernels.h) in two halves.  A new id costs two returning atomics — the slot
// claim (CAS on the key word) and the row allocation — of 2-3 us each under load, and a wavefront
// holds several ids: done one after the other they also hold back the row loads of the FOUND ids
// next to them.  upsert_issue puts the claims in flight and says which groups need a row; the
// caller allocates for them speculatively (ONE bump of the table's counter per workgroup: that
// address takes every allocation of the launch, and same-address atomics are served one after the
// other) and issues its row / gradient loads; upsert_complete consumes the results.  A claim lost
// to another id falls back to upsert_resolve's loop (the row is kept); an id that then finds both
// buckets full is deferred: its speculative row handle is never used (the host's row accounting
// carries slack for these, kSpecSlackRows) and its key count is taken back.
template <int G>
struct UpsertFlight {
  unsigned long long cas_old;
  uint64_t specm;     // leader lanes of the groups that get a speculative row
  int owner, pick;
  bool found, need, special, deferred;
};

template <int G>
__device__ __forceinline__ UpsertFlight<G> upsert_issue(const TableView& tv, GBucket* b, int64_t id,
                                                        bool valid, int64_t k, int lane,
                                                        uint32_t reserved) {
  const int j = lane & (G - 1);
  const int gbase = lane & ~(G - 1);
  UpsertFlight<G> f;
  f.special = valid && id == kEmptyKey;
  const bool prober = valid && !f.special && j < 8;
  const uint64_t m = group_mask_of<G>(__ballot(prober && k == id), gbase);
  f.found = m != 0;
  f.owner = f.found ? (__ffsll(static_cast<long long>(m)) - 1) : -1;
  f.need = valid && !f.special && !f.found;
  f.deferred = false;
  const uint64_t em = group_mask_of<G>(__ballot(prober && k == kEmptyKey), gbase) & 0xffull;
  const uint32_t m1 = uint32_t(em) & 0xfu, m2 = (uint32_t(em) >> 4) & 0xfu;
  f.pick = -1;
  if (m1) f.pick = 31 - __clz(m1);
  else if (m2) f.pick = 4 + (31 - __clz(m2));
  if (f.need && f.pick < 0) {  // both buckets full -> displacement pass
    f.deferred = true;
    f.need = false;
  }
  f.cas_old = 0ull;
  if (f.need && j == f.pick)
    f.cas_old = cas_key(&b->key[j & 3], kEmptyKey, id);
  f.specm = __ballot(f.need && j == 0 && reserved == kNoRow);
  return f;
}

template <int G>
__device__ __forceinline__ SlotResult upsert_complete(const TableView& tv, GBucket* b, int64_t id,
                                                      bool valid, uint32_t row, int lane, uint32_t ts,
                                                      const UpsertFlight<G>& f, uint32_t base_row,
                                                      uint32_t reserved) {
  const int j = lane & (G - 1);
  const int gbase = lane & ~(G - 1);
  const int s = j & 3;
  const bool prober = valid && !f.special && j < 8;
  bool need = f.need, deferred = f.deferred, found = f.found, is_new = false;
  int owner = f.owner;
  int64_t k = 0;  // (only read by lanes that lost a claim)
  bool won = false;
  if (need && j == f.pick) {
    won = static_cast<int64_t>(f.cas_old) == kEmptyKey;
    k = won ? id : static_cast<int64_t>(f.cas_old);
  }
  {
    const uint64_t wm = group_mask_of<G>(__ballot(won), gbase);
    if (need && wm) {
      owner = f.pick;
      is_new = true;
      need = false;
    }
  }
  if (__any(need)) {
    // the slot went to another id in the meantime: re-read the two buckets and claim again
    // (cuckoohash_map.hpp:1398-1418, as upsert_resolve does)
    if (prober) k = __hip_atomic_load(&b->key[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__any(need)) {
      const uint64_t em = group_mask_of<G>(__ballot(prober && k == kEmptyKey), gbase) & 0xffull;
      int pick = -1;
      if (need) {
        const uint32_t m1 = uint32_t(em) & 0xfu, m2 = (uint32_t(em) >> 4) & 0xfu;
        if (m1) pick = 31 - __clz(m1);
        else if (m2) pick = 4 + (31 - __clz(m2));
        if (pick < 0) {
          deferred = true;
          need = false;
        }
      }
      bool w2 = false;
      if (need && j == pick) {
        const unsigned long long old = cas_key(&b->key[s], kEmptyKey, id);
        w2 = (static_cast<int64_t>(old) == kEmptyKey);
        k = w2 ? id : static_cast<int64_t>(old);
      }
      const uint64_t wm = group_mask_of<G>(__ballot(w2), gbase);
      if (need && wm) {
        owner = pick;
        is_new = true;
        need = false;
      }
    }
  }
  if (f.special) {  // side slot of the one key that cannot live in a bucket
    unsigned int st = 0;
    if (j == 0) st = atomicExch(&tv.ctr->special_state, 1u);
    st = __shfl(st, gbase);
    found = true;
    is_new = (st == 0);
  }
  (void)found;
  // rows: the speculative handles (base_row = this wavefront's first; the bump counted them as
  // live keys too); the side slot's row (rare) with its own bump; speculated keys that ended up
  // deferred are taken back
  // (reserved: the forward launch allocated the row — and counted the key — when it found the id
  // missing; such a group needs nothing from the counter, and gives its key back if it does not
  // insert after all)
  const bool leader_new = is_new && j == 0;
  const uint64_t newm = __ballot(leader_new);
  const uint64_t resm = __ballot(valid && reserved != kNoRow && j == 0);
  const uint64_t extra = newm & ~f.specm & ~resm;
  const uint64_t lostm = (f.specm | resm) & ~newm;
  uint32_t base2 = 0;
  if (extra) {
    const int first2 = __ffsll(static_cast<long long>(extra)) - 1;
    if (lane == first2)
      base2 = static_cast<uint32_t>(
          atomicAdd(&tv.ctr->alloc, static_cast<unsigned long long>(__popcll(extra))));
    base2 = __shfl(base2, first2);
  }
  if (lostm && lane == __ffsll(static_cast<long long>(lostm)) - 1)
    atomicAdd(&tv.ctr->alloc, ~((static_cast<unsigned long long>(__popcll(lostm)) << 32) - 1ull));
  const uint32_t found_row = __shfl(row, gbase + (owner < 0 ? 0 : owner));
  // (computed from a value the compiler cannot trace back to the lane id: hoisted out of the
  // caller's loop, this mask gets spilled, and a spill reload waits for every load in flight)
  int gb = gbase;
  asm volatile("" : "+v"(gb));
  const uint64_t below = (uint64_t(1) << gb) - 1;
  uint32_t r;
  if (is_new) {
    r = (reserved != kNoRow) ? reserved
        : ((f.specm >> gbase) & 1ull) ? base_row + uint32_t(__popcll(f.specm & below))
                                      : base2 + uint32_t(__popcll(extra & below));
  } else {
    r = found_row;
  }
  if (f.special) {
    auto* ctr = tv.ctr;   // (kept inside the branch: see opaque_f)
#ifndef MHTE_NO_ANTIHOIST
    asm volatile("" : "+v"(ctr));
#endif
    if (is_new) {
      if (j == 0) ctr->special_row = r;
    } else {
      r = ctr->special_row;  // written by an earlier kernel
    }
    if (j == 0) ctr->special_ts = vgpr_copy_of_uniform(ts);
  } else if (valid && !deferred && j == owner) {
    if (is_new) b->row[s] = r;
    // (the uniform's VGPR copy, made in front of the trip loop, was spilled there and reloaded here
    // behind a wait for every load in flight: the copy is made in place instead — see opaque_f)
    b->ts[s] = vgpr_copy_of_uniform(ts);  // SetTimestamp(update_time), cuckoo_embedding_hash_table.cc:242-246
  }
  SlotResult out;
  out.r = r;
  out.is_new = is_new;
  out.deferred = deferred;
  return out;
}

// LDS of the apply role, declared by the kernel: a kernel that instantiates the role for several
// lane-group widths (mstep_bwd_kernel) would otherwise get one copy of the role's scratch per
// instantiation.  Sized for G = 8 (32 groups per workgroup).
struct ApplyLds {
  uint32_t pos[(256 / 8) * kStepLightMax];  // [group][kStepLightMax] positions of a short list
  uint32_t rstart[65];                      // first flat entry of run t of the item
  uint32_t rval[64];
  float sum[256 * 4];                       // [group][G * VEC]
  uint32_t last;
};

// FULL: the table uses per-element optimizers beyond SGD / Adagrad / FTRL (optimize_row_reg_full; no
// row prefetch: the state layout is the optimizer's); the host picks the instantiation.
// (HINT: kept for the call sites' sake — every launch takes hints when ApplyCtl carries them)
// FILT: 0 = the table has no admission filter (no filter code in the instance), 1 = it has one and the
// lane group consults it (filter_consult_group), -1 = decided at run time, serial form (the multi-table
// launch, whose tables differ).
template <int G, int VEC, bool ONESEG, bool HINT = false, bool FULL = false, int OPTK = -1, int FILT = -1>
__device__ __forceinline__ void rd_apply_role(const TableView& tv, const RunView& d,
                                              const ApplyCtl& c, const ApplyArgs& a, uint32_t bid,
                                              WaveTrace& wt, ApplyLds& L) {
  constexpr int WIN = G < 8 ? G : 8;   // gradient rows in flight per group of an item workgroup
  constexpr int NG = 256 / G;  // groups per workgroup
  const int lane = threadIdx.x & 63;
  const int j = lane & (G - 1);
  const int gbase = lane & ~(G - 1);
  const int grp = threadIdx.x / G;
  const uint32_t dim = tv.dim;
  const uint32_t e = uint32_t(j) * VEC;
  const bool ev = e < dim;

  if (bid >= c.nblk_items) {
    // ------------------------------------------------------------------ id-major groups
    // Group `grp` of workgroup k takes the unique indices u = it * stride + grp * nblk_ids + k:
    // consecutive indices (the claim order puts the hot ids first) land in different workgroups.
    // A group lives inside one wavefront and nothing in a trip is shared between wavefronts (row
    // handles of new ids were reserved by the build role's probe, ProbeOut; the few that were not
    // take one bump per wavefront): no workgroup barrier in the loop, the four wavefronts run free.
    // (32-bit index arithmetic: n_max is a batch size; the 64-bit form kept a hoisted per-lane offset
    // in a register pair that was spilled at the loop entry and reloaded — with a wait for every
    // load in flight — in front of each trip's first loads)
    const uint32_t stride = c.nblk_ids * uint32_t(NG);
    const uint32_t k = bid - c.nblk_items;
    const uint32_t wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar)
    uint32_t nu = c.n_max;  // refined below, once the count has arrived with the first trip's loads
#pragma unroll 1
    for (uint32_t it = 0; uint64_t(it) * stride < nu; ++it) {
      // (a group makes one or two trips: anything per lane that lives across the loop — the lane's
      // index in its group, the group's index, element offsets, addresses of the form base + e — is
      // not worth its registers: the compiler spilled them, and a spill reload waits for EVERY load
      // the wavefront has in flight.  All of it is derived inside the trip from the lane id, which is
      // recomputed from an operand the compiler cannot see through.)
      uint32_t zero_ = 0;
      asm volatile("" : "+v"(zero_));
      const int lane = int(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero_)));
      const int j = lane & (G - 1);
      const int gbase = lane & ~(G - 1);
      const uint32_t e = uint32_t(j) * VEC;
      const bool ev = e < dim;
      const uint32_t grp_ = (uint32_t(wave_s) * 64u + uint32_t(lane)) / uint32_t(G);
      uint32_t* const sh_pos = L.pos + grp_ * kStepLightMax;  // this group's slice
      const uint32_t g = it * stride + grp_ * c.nblk_ids + k;
      // round trip 1: everything about unique index g (the build role's dense arrays; an index past
      // the count reads stale entries of the preallocated arrays and is dropped)
      const bool inb = g < c.n_max;
      const uint32_t n_unique = d.ctr[0];  // (issued with the rest of the trip: not behind its wait)
      // the id was resolved a launch ago (single table: the build role's probe, URec — one
      // unconditional 32-byte load from a safe index, masked afterwards: a load under a branch is
      // waited for where the branch ends; multi-table step: the forward launch's lookup): its row
      // handle and bucket slot arrive with the rest of the trip
      int64_t id;
      uint32_t cnt, hp, gs, reserved = kNoRow, hrow = kNoRow, huts = 0;
      unsigned long long hloc = 0;
      if (!HINT) {
        const URec rec = c.urec[inb ? g : 0u];
        id = rec.id;
        cnt = inb ? rec.cnt : 0u;
        hp = rec.pos;
        gs = rec.slot;
        const bool fnd = rec.loc != kNoLoc;
        hrow = fnd ? rec.row : kNoRow;
        hloc = fnd ? rec.loc : 0ull;
        reserved = fnd ? kNoRow : rec.row;
      } else {
        reserved = kNoRow;
        id = inb ? d.uids[g] : 0;
        cnt = inb ? d.ucnt[g] : 0u;
        hp = inb ? d.upos[g] : 0u;
        gs = inb ? d.uslot[g] : 0u;
        if (c.urow) {
          hrow = inb ? c.urow[g] : kNoRow;
          hloc = inb ? c.uloc[g] : 0ull;
          if (c.uts) huts = inb ? c.uts[g] : 0u;
        }
      }
      // sliding admission filter: the window's position, fetched with the trip (it only moves between
      // launches) — the group fetches the filter's slots beside the table probe below
      const bool fgroup = FILT == 1 && tv.flt_slots != nullptr && tv.flt_nsplit != 0u;
      uint32_t fhead = 0, fhinc = 0;
      if (fgroup) {
        const FilterState* fs0 = reinterpret_cast<const FilterState*>(tv.flt_state);
        fhead = fs0->head;
        fhinc = fs0->head_increment;
      }
      // (every reservation of the batch is consumed by this launch — inserted, or its key given back:
      // one thread takes them off the table's count of outstanding ones)
      if (!HINT && it == 0 && k == 0 && threadIdx.x == 0) {
        const uint32_t nres = d.ctr[3];
        if (nres) atomicSub(&tv.ctr->reserved, nres);
      }
      nu = min(uint32_t(c.n_max), n_unique);
      bool valid = g < nu;
      if (!valid) cnt = 0;
      if (cnt > c.light_max) valid = false;  // heavy list: the item workgroups own it
      bool hinted = valid && hrow != kNoRow;
      // round trip 2: hinted id: the key in its slot (the hint's check) | its row;  other ids: the
      // table probe;  both: gradient of a lone occurrence | position list of a short list
      const bool ts_known = c.uts != nullptr && huts == a.ts;   // (multi-table step: the slot already
                                                                // carries this second — no store, no check)
      Probe<G> pr = probe_issue<G>(tv, id, valid && !hinted, j);
      const bool fact = valid && !hinted;   // (the filter windows below are this id's)
      FilterProbe<G> fp;                    // (not initialised without a sliding filter: never read then)
      if constexpr (FILT == 1) {
        if (fgroup) fp = filter_probe_issue<G>(tv, id, fact, j, fhead, fhinc);
      }
      // (a group without a hint — or an index past the count, whose record is stale — checks bucket
      // 0's first key: every lane loads, unconditionally, from an address that exists)
      // (the bucket pointer is formed here and again at the timestamp store below: kept live across the
      // gradient sums it cost the trip a spilled register pair — 8 bytes per lane stored on every trip)
      int64_t vraw = id;
      if (!HINT) {
        const unsigned long long sloc = hinted ? hloc : 0ull;
        vraw = global_bucket(tv.buckets + (sloc >> 2))->key[sloc & 3ull];
      }
      RowRegs<VEC> rr;
      vec_zero(rr.w);
      vec_zero(rr.s1);
      // (FULL with the optimizer known at compile time, OPTK: its row is fetched ahead too — the FULL form
      // otherwise pays a dependent round trip for the row behind everything else)
#ifdef MHTE_NO_FULL_PREFETCH
      constexpr bool PF = false;
#else
      constexpr bool PF = FULL && OPTK >= 0 && OPTK != kOptAmsgrad;   // (AMSGrad's three state vectors: 49 spilled)
#endif
      RowRegsF<VEC> rf;
      if constexpr (PF) {
        vec_zero(rf.w);
        vec_zero(rf.s1);
        vec_zero(rf.s2);
        vec_zero(rf.s3);
        rf.c1 = rf.c2 = 0.f;
      }
      // the row of a hinted id.  Rows of the first slab — all of them in a table created with
      // reserve_rows — are addressed without reading the slab table: row_ptr's load of the slab
      // pointer sits under a branch, and the wait where that branch ends would hold the row loads
      // back behind everything issued above (a whole round trip).  Later slabs: the slow way.
      const bool slab0 = (hrow >> tv.chunk_shift) == 0u;
      if (hinted && slab0 && !FULL)
        row_prefetch<VEC, ONESEG>(tv, assume_global(tv.chunk0 + size_t(hrow) * tv.row_floats), e, rr);
      if constexpr (PF) {
        if (hinted && slab0)
          row_prefetch_full<VEC, OPTK>(tv, assume_global(tv.chunk0 + size_t(hrow) * tv.row_floats), e, rf);
      }
      if (__any(hinted && !slab0)) {
        if (hinted && !slab0 && !FULL) row_prefetch<VEC, ONESEG>(tv, row_ptr(tv, hrow), e, rr);
        if constexpr (PF) {
          if (hinted && !slab0) row_prefetch_full<VEC, OPTK>(tv, row_ptr(tv, hrow), e, rf);
        }
      }
      if (it == 0) wt.mark(0);
      const bool single = valid && cnt == 1;
      const bool big = valid && cnt > uint32_t(kStepLightMax);  // (exact order only)
      const bool flat = valid && !single && !big;
      Vec<VEC> acc;  // (a lone occurrence's gradient is loaded straight into the accumulator)
      vec_zero(acc);
      if (single && ev) acc.load(c.grads + int64_t(hp) * dim + e);
      // a short list (2..kLightMax occurrences): its positions from the dedup's per-id list, runs in
      // arrival order -> ranked in registers (positions are distinct, so the ranks are a
      // permutation) and handed over in position order through LDS
      constexpr int PER = (kStepLightMax + G - 1) / G;
      uint32_t x[PER];
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const uint32_t idx = uint32_t(j) + uint32_t(q) * G;
        x[q] = (flat && idx < cnt) ? d.hlist[size_t(gs) * kLightMax + idx] : 0xffffffffu;
      }
      if (it == 0) wt.mark(1);
      // the hint holds when the slot still carries the id; otherwise (rare: the entry was moved by a
      // displacement pass or a doubling, or evicted, since the probe) the group probes now
      {
        const bool hok = HINT || c.trusted || vraw == id;   // (every lane of the group read the same word)
        if (__any(hinted && !hok)) {
          const bool redo = hinted && !hok;
          const Probe<G> p2 = probe_issue<G>(tv, id, redo, j);
          if (redo) {
            pr = p2;
            hinted = false;
            vec_zero(rr.w);
            vec_zero(rr.s1);
          }
        }
      }
      // admission filter (one consultation with the occurrence count: BatchOptimize with dedup,
      // tf_bridge.cc:300-310): an id that is not in the table yet and has not been seen often
      // enough is dropped — no insert, no update
      if constexpr (FILT != 0) {
        if (tv.flt_slots) {
          bool contained = group_mask_of<G>(__ballot(valid && id != kEmptyKey && j < 8 && pr.k == id), gbase) != 0;
          if (valid && id == kEmptyKey) contained = tv.ctr->special_state == 1;
          if (hinted) contained = true;
          uint32_t first = 0;
          if constexpr (FILT == 1) {
            // (a group whose hint failed fetched no windows: the consultation fetches them itself)
            first = filter_consult_group<G>(tv, id, cnt, 2, contained, valid, fgroup && fact, fp, j, gbase, fhead,
                                            fhinc);
          } else {
            if (valid && j == 0) first = filter_consult(tv, id, cnt, 2, contained);
            first = __shfl(first, gbase);
          }
          if (first != 0u) valid = false;
          hinted = hinted && valid;
        }
      }
      // round trip 3 (ids without a hint): slot claim + row handle of a new id | the row of a found
      // one | (below) the gradients of a list — all in flight together
      const UpsertFlight<G> uf = upsert_issue<G>(tv, pr.b, id, valid && !hinted, pr.k, lane, reserved);
      const bool pre = valid && !hinted && uf.found;
      if (__any(pre)) {
        const uint32_t frow = __shfl(pr.row, gbase + (uf.owner < 0 ? 0 : uf.owner));
        if (pre && !FULL) row_prefetch<VEC, ONESEG>(tv, row_ptr(tv, frow), e, rr);
        if constexpr (PF) {
          if (pre) row_prefetch_full<VEC, OPTK>(tv, row_ptr(tv, frow), e, rf);
        }
      }
      if (__any(flat)) {
        uint32_t xr[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) xr[q] = 0;
#pragma unroll
        for (int q2 = 0; q2 < PER; ++q2) {
#pragma unroll 4
          for (int t2 = 0; t2 < G; ++t2) {
            const uint32_t y = __shfl(x[q2], gbase + t2);
#pragma unroll
            for (int q = 0; q < PER; ++q) xr[q] += (y < x[q]) ? 1u : 0u;
          }
        }
#pragma unroll
        for (int q = 0; q < PER; ++q)
          if (x[q] != 0xffffffffu) sh_pos[xr[q]] = x[q];
      }
      // rows for new ids that have no reservation (the first step of a pipeline, an id the probe saw
      // mid-insert and that was evicted since, the multi-table step): one bump per wavefront, placed
      // behind the row loads — the compiler waits for a result produced under a branch where the
      // branch ends, and here that wait is shared with loads this wavefront needs next anyway
      uint32_t rows0;  // (not initialised on purpose: see lbase in rd_dedup_role; the low half — the
                       // first row handle — is all that is kept: the pair was a spilled register pair)
      bool bumped = false;
      {
        const unsigned long long tot = __popcll(uf.specm);
        if (tot && lane == 0) {
          rows0 = uint32_t(atomicAdd(&tv.ctr->alloc, (tot << 32) | tot));
          bumped = true;
        }
      }
      lds_wave_sync();
      if (it == 0) wt.mark(2);
      if (single) {  // 0 + g, as the sequential sum starts
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc.v[q] = 0.f + acc.v[q];
      }
      if (flat) sum_list_lds<VEC>(c.grads, dim, e, ev, sh_pos, cnt, acc);
      if (big) {
        // strictly sequential sum of a long list (MHTE_EXACT_ORDER): run after run.  Which dedup
        // workgroups hold a run: the group's lanes probe G directories per round.
        unsigned long long rest = 0ull;
#pragma unroll 1
        for (uint32_t b1 = 0; b1 < d.nblk; b1 += G) {
          const uint32_t bq = b1 + uint32_t(j);
          const uint32_t vq = (bq < d.nblk) ? rd_find_run_opt(d, bq, id) : 0u;
          rest |= group_mask_of<G>(__ballot(vq != 0u), gbase) << b1;
        }
#pragma unroll 1
        while (rest) {
          const uint32_t b = uint32_t(__ffsll(static_cast<long long>(rest)) - 1);
          rest &= rest - 1ull;
          const uint32_t val = rd_find_run(d, b, id);
          const uint32_t cb = run_cnt(val);
          const uint16_t* sp = d.seg + b * kRdBlock + run_off(val);
#pragma unroll 1
          for (uint32_t i = 0; i < cb; i += 8) {
            Vec<VEC> v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) vec_zero(v[t]);
#pragma unroll
            for (int t = 0; t < 8; ++t)
              if (i + t < cb && ev)
                v[t].load(c.grads + int64_t(b * kRdBlock + sp[i + t]) * dim + e);
#pragma unroll
            for (int t = 0; t < 8; ++t)
              if (i + t < cb && ev) vec_add(acc, v[t]);
          }
        }
      }
      uint32_t base_row;
      {
        uint32_t br = bumped ? rows0 : 0u;  // (lane 0 of a wavefront that needed rows)
        base_row = __shfl(br, 0);
      }
      SlotResult sr =
          upsert_complete<G>(tv, pr.b, id, valid && !hinted, pr.row, lane, a.ts, uf, base_row, reserved);
      if (hinted) {  // resident id: nothing to claim; the timestamp goes to the slot the probe found
        sr.r = hrow;
        sr.is_new = false;
        sr.deferred = false;
        // (a 4-byte store into a line this launch otherwise never touches costs a 128-byte fetch and
        // a write-back: 21 of mstep_bwd's 200 us at 26 x 65 536 ids.  update_time has the
        // resolution of a second; an id updated again within the second already carries it.)
        if (valid && j == 0 && !ts_known)
          global_bucket(tv.buckets + (hloc >> 2))->ts[hloc & 3ull] = vgpr_copy_of_uniform(a.ts);
      }
      float* rp = nullptr;
      if (valid && !sr.deferred) {
        rp = row_ptr(tv, sr.r);
        if (!FULL && !sr.is_new && !pre && !hinted) row_prefetch<VEC, ONESEG>(tv, rp, e, rr);  // (the side slot's row)
        if constexpr (PF) {
          if (!sr.is_new && !pre && !hinted) row_prefetch_full<VEC, OPTK>(tv, rp, e, rf);
        }
      }
      if (it == 0) wt.mark(3);
      if (sr.deferred) {
        if (ev) acc.store(c.grad_u + g * int64_t(dim) + e);
        if (j == 0) defer_id(tv, c.pending, uint32_t(g));
      } else if (valid) {
        if constexpr (PF) optimize_row_reg_full<VEC, ONESEG, OPTK>(tv, rp, sr.is_new, e, acc, a, &rf);
        else if (FULL) optimize_row_reg_full<VEC, ONESEG, OPTK>(tv, rp, sr.is_new, e, acc, a);
        else optimize_row_pre<VEC, ONESEG>(tv, rp, sr.is_new, e, acc, a, rr);
      }
      if (it == 0) wt.mark(4);
      lds_wave_sync();  // (sh_pos is the group's own)
    }
    return;
  }

  // -------------------------------------------------------------------- item workgroups
  // (the longest chain of the launch: its wavefronts go first when a SIMD has a choice)
  __builtin_amdgcn_s_setprio(3);
  uint32_t* const sh_rstart = L.rstart;
  uint32_t* const sh_rval = L.rval;
  float (*const sh_sum)[G * VEC] = reinterpret_cast<float (*)[G * VEC]>(L.sum);
  uint32_t& sh_last = L.last;
#pragma unroll 1
  for (uint32_t w = bid;; w += c.nblk_items) {  // block-uniform
    // round trip 1: item count, header and run descriptors together (an index past the count reads
    // stale entries of the preallocated list and is dropped)
    const uint32_t nitems_all = d.ctr[2];
    const ItemHdr hd = d.item_hdr[w];
    const uint32_t rval = d.item_runs[size_t(w) * 64 + lane];  // (every wavefront: no branch, no queueing)
    // (pins the three loads above in front of the exit test: sunk below it, they would wait for
    // the count's round trip first)
    asm volatile("" ::"v"(hd.meta), "v"(rval), "v"(nitems_all));
    if (w >= nitems_all) break;
    wt.mark(0);
    const uint32_t b0 = hd.meta & 0xffu, nbk = (hd.meta >> 8) & 0xffu, kk = (hd.meta >> 16) & 0xffu,
                   nitems = hd.meta >> 24;
    // round trip 2: the table probe of the id (used by whoever applies: wave 0, group 0) | positions
    // (the build role's probe left the id's row handle and slot in the header: no bucket read for a
    // resident id, only the hint's check — the key in its slot, fetched beside the positions)
    const bool ihint = hd.row != kNoRow;   // block-uniform
    GBucket* const hb = global_bucket(tv.buckets + (hd.loc >> 2));
    int64_t vkey = hd.id;
    if (ihint && !c.trusted && threadIdx.x == 0) vkey = hb->key[hd.loc & 3ull];
    Probe<G> pr = probe_issue<G>(tv, hd.id, threadIdx.x < G && !ihint, j);
    const uint32_t reserved = threadIdx.x < G ? hd.spec : kNoRow;
    if (threadIdx.x < 64) {
      const uint32_t val = (uint32_t(lane) < nbk) ? rval : 0u;
      uint32_t incl = run_cnt(val);
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
      }
      sh_rval[lane] = val;
      sh_rstart[lane + 1] = incl;
      if (lane == 0) sh_rstart[0] = 0;
    }
    lds_barrier();
    wt.mark(1);
    const uint32_t E = c.pre_summed ? 0u : sh_rstart[64];   // (summed ahead: no window to walk)
    Vec<VEC> acc;
    vec_zero(acc);
    // this group's windows: qb = (grp + k * NG) * WIN.  The positions of kPre windows are fetched
    // together (one round trip), then each window's WIN gradient rows
    constexpr int kPre = 2;
#pragma unroll 1
    for (uint32_t qb0 = uint32_t(grp) * WIN; qb0 < E; qb0 += kPre * NG * WIN) {
      // positions of the group's next kPre windows: kPre * WIN entries, PPL per lane (entry
      // idx = r * G + j belongs to window idx / WIN).  Every lane loads — from entry 0 where it has
      // nothing to fetch — and all loads are issued before any is used: a load under a branch makes
      // the compiler wait for it where the branch ends, and the fetches would queue up one round
      // trip after the other.
      constexpr int PPL = (kPre * WIN + G - 1) / G;
      uint32_t pp[PPL], pbv[PPL];
      uint16_t ps[PPL];
#pragma unroll
      for (int r = 0; r < PPL; ++r) {
        const uint32_t idx = uint32_t(r) * G + uint32_t(j);
        const uint32_t q = qb0 + (idx / WIN) * (NG * WIN) + (idx % WIN);
        const bool has = idx < uint32_t(kPre * WIN) && q < E;
        const uint32_t qq = has ? q : 0u;
        uint32_t lo = 0, hi = 63;  // run r with rstart[r] <= q < rstart[r+1]
#pragma unroll
        for (int it2 = 0; it2 < 6; ++it2) {
          const uint32_t mid = (lo + hi + 1) >> 1;
          const bool le = sh_rstart[mid] <= qq;
          lo = le ? mid : lo;
          hi = le ? hi : mid - 1;
        }
        const uint32_t val = has ? sh_rval[lo] : 0u;
        const uint32_t base = (b0 + lo) * kRdBlock;
        ps[r] = d.seg[has ? base + run_off(val) + (qq - sh_rstart[lo]) : 0u];
        // (base and the run's first position in one word: base < 2^16, first < 2^10, count flag)
        pbv[r] = base | (run_first(val) << 16) | ((run_cnt(val) == 1 ? 1u : 0u) << 31);
      }
#pragma unroll
      for (int r = 0; r < PPL; ++r)
        pp[r] = (pbv[r] & 0xffffu) + ((pbv[r] >> 31) ? ((pbv[r] >> 16) & 0x3ffu) : uint32_t(ps[r]));
#pragma unroll
      for (int w2 = 0; w2 < kPre; ++w2) {
        const uint32_t qb = qb0 + uint32_t(w2) * (NG * WIN);
        if (qb >= E) break;  // group-uniform
        // WIN gradient rows in flight
        Vec<VEC> v[WIN];
#pragma unroll
        for (int t = 0; t < WIN; ++t) vec_zero(v[t]);
#pragma unroll
        for (int t = 0; t < WIN; ++t) {
          const int idx = w2 * WIN + t;  // (compile-time: which lane and register hold the position)
          const uint32_t pt = __shfl(pp[idx / G], gbase + (idx % G));
          if (qb + t < E && ev) v[t].load(c.grads + int64_t(pt) * dim + e);
        }
#pragma unroll
        for (int t = 0; t < WIN; ++t)
          if (qb + t < E && ev) vec_add(acc, v[t]);
      }
    }
#pragma unroll
    for (int cc = 0; cc < VEC; ++cc) sh_sum[grp][j * VEC + cc] = acc.v[cc];
    wt.mark(2);
    lds_barrier();
    Vec<VEC> tot;  // groups in order -> the item's sum (wave 0 holds it; group 0 uses it)
    vec_zero(tot);
    if (threadIdx.x < 64) {
#pragma unroll 1
      for (int g2 = 0; g2 < NG; ++g2) {
#pragma unroll
        for (int cc = 0; cc < VEC; ++cc) tot.v[cc] = tot.v[cc] + sh_sum[g2][j * VEC + cc];
      }
    }
    wt.mark(3);
    bool apply = nitems == 1;  // block-uniform
    if (c.pre_summed) {   // the list's first item applies the sum rd_exact_sum_kernel left for it
      apply = kk == 0;
      if (apply && threadIdx.x < G && ev) tot.load(c.part + int64_t(w) * dim + e);
    } else if (nitems > 1) {
      // partial row per item (the items of a list are consecutive), write-through hand-off
      if (threadIdx.x < G && ev) store_wt<VEC>(c.part + int64_t(w) * dim + e, tot);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
      if (threadIdx.x == 0) {
        const uint32_t last = (atomicAdd(&c.arrive[hd.u], 1u) == nitems - 1) ? 1u : 0u;
        if (last) c.arrive[hd.u] = 0;  // clean-after-use
        sh_last = last;
      }
      lds_barrier();
      if (sh_last) {  // add the item sums in item order (fixed association), then apply
        apply = true;
        const uint32_t w0 = w - kk;
        const uint32_t per = (nitems + NG - 1) / NG;
        const uint32_t k0 = min(nitems, uint32_t(grp) * per), k1 = min(nitems, k0 + per);
        Vec<VEC> sacc;
        vec_zero(sacc);
#pragma unroll 1
        for (uint32_t q = k0; q < k1; q += 8) {
          Vec<VEC> r[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) vec_zero(r[t]);
#pragma unroll
          for (int t = 0; t < 8; ++t)
            if (q + t < k1 && ev) load_agent<VEC>(c.part + int64_t(w0 + q + t) * dim + e, r[t]);
#pragma unroll
          for (int t = 0; t < 8; ++t)
            if (q + t < k1 && ev) vec_add(sacc, r[t]);
        }
        lds_barrier();  // sh_sum is reused
#pragma unroll
        for (int cc = 0; cc < VEC; ++cc) sh_sum[grp][j * VEC + cc] = sacc.v[cc];
        lds_barrier();
        vec_zero(tot);
        if (threadIdx.x < 64) {
          for (int g2 = 0; g2 < NG && uint32_t(g2) * per < nitems; ++g2) {
#pragma unroll
            for (int cc = 0; cc < VEC; ++cc) tot.v[cc] = tot.v[cc] + sh_sum[g2][j * VEC + cc];
          }
        }
      }
    }
    if (apply && threadIdx.x < 64) {
      bool valid = threadIdx.x < G;
      bool hok = ihint && __shfl(vkey == hd.id ? 1 : 0, 0) != 0;
      if (ihint && !hok) pr = probe_issue<G>(tv, hd.id, valid, j);   // (rare: the entry moved or left)
      if constexpr (FILT != 0) {
        if (tv.flt_slots) {  // (as in the id-major groups; the list's length is its count)
          bool contained =
              group_mask_of<G>(__ballot(valid && hd.id != kEmptyKey && j < 8 && pr.k == hd.id), gbase) != 0;
          if (valid && hd.id == kEmptyKey) contained = tv.ctr->special_state == 1;
          if (hok) contained = true;
          uint32_t first = 0;
          if constexpr (FILT == 1) {   // (the lane group's form: no second, serial copy of the walk in the kernel)
            uint32_t ih = 0, ii = 0;
            if (tv.flt_nsplit != 0u) {
              const FilterState* fs0 = reinterpret_cast<const FilterState*>(tv.flt_state);
              ih = fs0->head;
              ii = fs0->head_increment;
            }
            FilterProbe<G> fp0;   // (not fetched ahead here: the consultation's first pass does it)
            first = filter_consult_group<G>(tv, hd.id, d.ucnt[hd.u], 2, contained, valid, false, fp0, j, gbase, ih, ii);
          } else {
            if (valid && j == 0) first = filter_consult(tv, hd.id, d.ucnt[hd.u], 2, contained);
            first = __shfl(first, gbase);
          }
          if (first != 0u) valid = false;
        }
      }
      SlotResult sr;
      if (hok) {
        sr.r = hd.row;
        sr.is_new = false;
        sr.deferred = false;
        if (valid && threadIdx.x == 0) hb->ts[hd.loc & 3ull] = vgpr_copy_of_uniform(a.ts);
      } else {
        // (the uniform's VGPR copy is made here, not in front of the item loop: see vgpr_copy_of_uniform)
        sr = upsert_resolve<G>(tv, (Bucket*)pr.b, hd.id, valid, pr.k, pr.row, lane, vgpr_copy_of_uniform(a.ts),
                               reserved);
      }
      if (sr.deferred) {
        uint32_t ed = e;   // (rare path: its address arithmetic stays here, see opaque_f)
#ifndef MHTE_NO_ANTIHOIST
        asm volatile("" : "+v"(ed));
#endif
        if (ev) tot.store(c.grad_u + int64_t(hd.u) * dim + ed);
        if (j == 0) defer_id(tv, c.pending, hd.u);
      } else if (valid) {
        if (FULL) optimize_row_reg_full<VEC, ONESEG, OPTK>(tv, row_ptr(tv, sr.r), sr.is_new, e, tot, a);
        else optimize_row_reg<VEC, ONESEG>(tv, row_ptr(tv, sr.r), sr.is_new, e, tot, a);
      }
    }
    wt.mark(4);
    lds_barrier();  // LDS is reused by the next item
  }
}

// ---------------------------------------------------------------------------------------------
// MHTE_EXACT_ORDER: the strictly sequential sum of every HEAVY list (> kStepLightMax occurrences), one
// launch in front of step_bwd (the reference sums an id's gradients in occurrence order,
// RT/ops/unique_mapping_ops.cc:284-329; lists of <= kStepLightMax are summed that way by the id-major groups
// in every mode).  The adds of a list are a dependent chain — its LOADS are not: one 512-thread workgroup
// per list, seven wavefronts stream the list's gradient rows into a double-buffered LDS chunk (64 KB each;
// the positions of the chunk after next are fetched in the same phase, so a phase is one memory round
// trip), wavefront 0 adds the chunk before it in occurrence order, a lane per float of the row — one
// v_add per row on its chain.  Round 5 walked a list with one 16-lane group, two dependent round trips per
// 8 rows: 4.8 ms per step at Zipf(1.2), where one id holds ~12 000 of the 65 536 positions.
// The list's runs are found in the dedup workgroups' directories (lane b probes workgroup b's), so the
// kernel does not depend on how the build role cut the list into items: the item with k = 0 stands for
// the list, and its sum goes to part[that item] — where step_bwd's item workgroup picks it up
// (ApplyCtl::pre_summed).
// ---------------------------------------------------------------------------------------------
constexpr int kExactThreads = 512;
constexpr int kExactBufFloats = 16384;   // 64 KB per chunk buffer
constexpr int kExactMaxRows = 512;       // rows per chunk (<= one position per thread)
struct __attribute__((aligned(16))) ExactLds {
  float buf[2][kExactBufFloats];
  uint32_t pos[3][kExactMaxRows];
  uint32_t rstart[65];
  uint32_t rbase[64];
};

// wg / nwg: this workgroup's index among the nwg that share the batch's heavy lists.  dst_of(item, header) = where
// the list's sum goes (dim floats; nullptr: nowhere)
template <int VEC, class DST>
__device__ __forceinline__ void rd_exact_sum_role(const RunView& d, const float* __restrict__ grads, uint32_t dim,
                                                  DST dst_of, uint32_t wg, uint32_t nwg, ExactLds& L) {
  constexpr uint32_t NL = kExactThreads - 64;   // loader lanes (wavefronts 1-7)
  const uint32_t t = threadIdx.x, lane = t & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const uint32_t C = min(uint32_t(kExactMaxRows), uint32_t(kExactBufFloats) / dim);   // rows per chunk
  const uint32_t upr = dim / VEC;                                                       // float4s (floats) per row
  const uint32_t nitems_all = d.ctr[2];
#pragma unroll 1
  for (uint32_t w = wg; w < nitems_all; w += nwg) {
    const ItemHdr hd = d.item_hdr[w];
    if (((hd.meta >> 16) & 0xffu) != 0u) continue;   // (workgroup-uniform: not the list's first item)
    __syncthreads();   // (the previous list's tables are no longer read)
    if (wave == 0) {
      const uint32_t val = (lane < d.nblk) ? rd_find_run_opt(d, lane, hd.id) : 0u;
      uint32_t incl = run_cnt(val);
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (int(lane) >= o) incl += v;
      }
      L.rstart[lane + 1] = incl;
      if (lane == 0) L.rstart[0] = 0;
      L.rbase[lane] = lane * uint32_t(kRdBlock) + run_off(val);
    }
    __syncthreads();
    const uint32_t E = L.rstart[64];
    const uint32_t nch = (E + C - 1) / C;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};   // wavefront 0: floats lane, lane + 64, ... of the row
    // phase k: positions of chunk k + 2 | rows of chunk k + 1 | sum of chunk k
#pragma unroll 1
    for (int k = -2; k < int(nch); ++k) {
      if (wave != 0) {
        const uint32_t lt = t - 64u;
        // ---- positions of chunk k + 2 (entry q of the list: run r with rstart[r] <= q < rstart[r + 1])
        const uint32_t c2 = uint32_t(k + 2);
        uint32_t pv[2] = {0u, 0u};
        uint32_t pb[2] = {0u, 0u};
        if (c2 < nch) {
#pragma unroll
          for (int x = 0; x < 2; ++x) {
            const uint32_t i = lt + uint32_t(x) * NL;
            const uint32_t q = c2 * C + i;
            const bool has = i < C && q < E;
            const uint32_t qq = has ? q : 0u;
            uint32_t lo = 0, hi = 63;
#pragma unroll
            for (int s2 = 0; s2 < 6; ++s2) {
              const uint32_t mid = (lo + hi + 1) >> 1;
              const bool le = L.rstart[mid] <= qq;
              lo = le ? mid : lo;
              hi = le ? hi : mid - 1;
            }
            pb[x] = lo * uint32_t(kRdBlock);
            pv[x] = d.seg[L.rbase[lo] + (qq - L.rstart[lo])];   // (every lane loads, from an entry that exists)
          }
        }
        // ---- rows of chunk k + 1 -> buf[(k + 1) & 1]: element v of the chunk = float4 (float) v of its rows
        const uint32_t c1 = uint32_t(k + 1);
        if (k + 1 >= 0 && c1 < nch) {
          const uint32_t rows = min(C, E - c1 * C);
          const uint32_t nvec = rows * upr;
          const uint32_t* ps = L.pos[c1 % 3u];
          float* dst = L.buf[c1 & 1u];
          constexpr int U = VEC == 4 ? 10 : 8;
#pragma unroll 1
          for (uint32_t v0 = lt; v0 < nvec; v0 += NL * U) {
            Vec<VEC> r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const uint32_t v = v0 + uint32_t(u) * NL;
              const uint32_t vv = v < nvec ? v : 0u;
              const uint32_t row = vv / upr, col = vv - row * upr;
              r[u].load(grads + int64_t(ps[row]) * dim + col * VEC);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const uint32_t v = v0 + uint32_t(u) * NL;
              if (v < nvec) {
#pragma unroll
                for (int cc = 0; cc < VEC; ++cc) dst[v * VEC + cc] = r[u].v[cc];
              }
            }
          }
        }
        if (c2 < nch) {
#pragma unroll
          for (int x = 0; x < 2; ++x) {
            const uint32_t i = lt + uint32_t(x) * NL;
            if (i < C) L.pos[c2 % 3u][i] = pb[x] + pv[x];
          }
        }
      } else if (k >= 0) {
        // ---- chunk k, row after row: acc = acc + g, the reference's loop
        const uint32_t rows = min(C, E - uint32_t(k) * C);
        const float* src = L.buf[k & 1];
#pragma unroll 1
        for (uint32_t cc = 0; cc * 64u < dim; ++cc) {
          const uint32_t f = lane + cc * 64u;
          const bool on = f < dim;
          float a = acc[0];
          if (cc == 1) a = acc[1];
          if (cc == 2) a = acc[2];
          if (cc == 3) a = acc[3];
          uint32_t r0 = 0;
#pragma unroll 1
          for (; r0 + 8 <= rows; r0 += 8) {
            float g8[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) g8[x] = src[(r0 + uint32_t(x)) * dim + (on ? f : 0u)];
#pragma unroll
            for (int x = 0; x < 8; ++x) a = a + g8[x];
          }
          for (; r0 < rows; ++r0) a = a + src[r0 * dim + (on ? f : 0u)];
          if (cc == 0) acc[0] = a;
          if (cc == 1) acc[1] = a;
          if (cc == 2) acc[2] = a;
          if (cc == 3) acc[3] = a;
        }
      }
      __syncthreads();
    }
    if (wave == 0) {
      float* const dst = dst_of(w, hd);
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const uint32_t f = lane + uint32_t(cc) * 64u;
        if (dst && f < dim) dst[f] = acc[cc];
      }
    }
  }
}
// the list's sum into part[its first item]: where the apply role's item workgroup picks it up
struct ExactToPart {
  float* part;
  uint32_t dim;
  __device__ __forceinline__ float* operator()(uint32_t w, const ItemHdr&) const { return part + int64_t(w) * dim; }
};
template <int VEC>
__global__ __launch_bounds__(kExactThreads) void rd_exact_sum_kernel(RunView d, const float* __restrict__ grads,
                                                                     uint32_t dim, float* __restrict__ part) {
  __shared__ ExactLds L;
  rd_exact_sum_role<VEC>(d, grads, dim, ExactToPart{part, dim}, blockIdx.x, gridDim.x, L);
}

// ---------------------------------------------------------------------------------------------
// The two launches.
// ---------------------------------------------------------------------------------------------
struct SlowArgs {  // slowpath_role's arguments; enabled = 0: no displacement pass outstanding
  const int64_t* uids;
  const float* grad_u;
  const uint32_t* pending;
  ApplyArgs a;
  int32_t enabled;
};

// ids per group of the forward lookup: 2 is the fastest shape (profiles/r01/f_lookup_sweep.jsonl); the host
// picks 3 or 4 when that is what it takes to have every workgroup of the launch resident at once

// step_fwd:  run dedup of the NEXT batch | displacement pass of the previous update (one wavefront,
//            the lookup workgroups gate on it) | lookup of this batch
// (<= 80 SGPRs: with more, the hardware admits 7 wavefronts per SIMD and only ONE of these
// 16-wavefront workgroups per CU instead of two — MI355X_MICROARCH.md, residency)
// BASIC: the table's optimizers are SGD / Adagrad / FTRL (the displacement role is compiled for those
// three: 17 spilled VGPRs with all twelve); BASIC = false: any per-element optimizer.
template <int G, int VEC, int UNR, bool BASIC = true>
__global__ __launch_bounds__(kRdBlock, 8) __attribute__((amdgpu_num_sgpr(80))) void step_fwd_kernel(RunView nxt, TableView tv,
                                                            const int64_t* __restrict__ ids,
                                                            int64_t n, float* __restrict__ out,
                                                            int count_hits, SlowArgs sp,
                                                            uint32_t nblk_l) {
  __shared__ __attribute__((aligned(16))) RdLds L;
  static_assert(sizeof(BfsSlot) * kMaxCuckooCount + sizeof(CuckooRecord) * kMaxBfsPathLen <=
                    sizeof(RdLds), "BFS scratch must fit the dedup's LDS");
  WaveTrace wt(tv.trace);
  uint32_t bid = blockIdx.x;
  if (bid < nxt.nblk) {
    rd_dedup_role(nxt, bid, L, wt);
    wt.end(3u);
    return;
  }
  bid -= nxt.nblk;
  if (sp.enabled) {
    if (bid == 0) {
      if (threadIdx.x < 64) {
        BfsSlot* q = reinterpret_cast<BfsSlot*>(&L);
        CuckooRecord* path = reinterpret_cast<CuckooRecord*>(q + kMaxCuckooCount);
        slowpath_role<VEC, kOpOptimize, false, true, BASIC>(tv, sp.uids, sp.grad_u, nullptr, nullptr, sp.a,
                                                            nullptr, sp.pending, q, path);
      }
      wt.end(4u);
      return;
    }
    bid -= 1;
  }
  // lookup workgroups: as many as are resident beside the other roles (nblk_l), grid-stride.
  // The first trip's ids are fetched BEFORE the gate below: the gate's poll and this load share one
  // round trip (with the run dedup out of this launch — two batches of look-ahead — the lookups'
  // own chain ids -> buckets -> rows -> stores is what the launch lasts).
  const int64_t ngroups = (n + UNR - 1) / UNR;
  const uint32_t lbid = bid;
  const int64_t g_first = (int64_t(lbid) * kRdBlock + threadIdx.x) / G;
  int64_t id_first;
  {
    const int j = int(threadIdx.x & 63u) & (G - 1);
    const int64_t p0 = g_first * UNR + j;
    id_first = ids[(j < UNR && p0 < n) ? p0 : 0];   // (unconditional, from a safe index)
    if (!(j < UNR && p0 < n)) id_first = 0;
  }
  if (sp.enabled) {
    // A displacement pass for the previous update runs in another workgroup of this launch.  No
    // table word is read before it has finished: ONE lane polls n_pending (it drops to 0 only
    // after the pass's stores were written back: agent-scope release), then ONE agent-scope
    // acquire for the workgroup (cdna_hip_programming.md G16: per-wavefront acquires multiply the
    // cost), then plain loads.  Usually the first poll already reads 0.
    if (threadIdx.x == 0) {
      if (__hip_atomic_load(&tv.ctr->n_pending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        while (__hip_atomic_load(&tv.ctr->n_pending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
          __builtin_amdgcn_s_sleep(16);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    }
    __syncthreads();
  }
  if (g_first < ngroups) lookup_role_u<G, VEC, UNR, 2>(tv, ids, n, nullptr, out, count_hits, g_first, &id_first);
#pragma unroll 1
  for (int64_t g = g_first + int64_t(nblk_l) * kRdBlock / G; g < ngroups;
       g += int64_t(nblk_l) * kRdBlock / G)
    lookup_role_u<G, VEC, UNR, 2>(tv, ids, n, nullptr, out, count_hits, g);
  wt.end(5u);
}

// what the dedup role needs of a workspace (the run dedup of batch s + 2 inside step_bwd: a compact
// argument instead of a third RunView)
struct DedupArgs {
  RdSlot* hs;
  uint32_t* hlist;
  int64_t* btab_key;
  uint32_t* btab_val;
  uint16_t* seg;
  uint32_t* ctr;
  const int64_t* ids;
  uint32_t cap_mask, n;
  uint32_t nblk;        // workgroups of the role (0: none)
};

// step_bwd:  run dedup of the batch TWO ahead | numbering + table probe of the NEXT batch | apply of
//            this batch
// ONESEG: the table has one segment (the host picks the instantiation: see seg_of)
union __attribute__((aligned(16))) StepBwdLds {
  ApplyLds apply;
  RdLds4 dedup;
};
// FILT: the table consults an admission filter (its own instances: the others carry no filter code)
template <int G, int VEC, bool ONESEG, bool FULL = false, int OPTK = -1, bool FILT = false>
__global__ __launch_bounds__(256, kBwdBlocksPerCu) void step_bwd_kernel(RunView nxt, uint32_t nblk_build,
                                                       TableView tv, RunView cur, ApplyCtl c,
                                                       ApplyArgs a, ProbeOut po, DedupArgs da) {
  __shared__ StepBwdLds L;
  WaveTrace wt(tv.trace);
  uint32_t bid = blockIdx.x;
  if (bid < da.nblk) {
    // (first in the grid: its chain — ids, LDS set, scratch CAS, count bump, position lists — is as
    // long as the update's, and it needs nothing the other roles produce)
    RunView d{};
    d.hs = da.hs;
    d.hlist = da.hlist;
    d.btab_key = da.btab_key;
    d.btab_val = da.btab_val;
    d.seg = da.seg;
    d.ctr = da.ctr;
    d.ids = da.ids;
    d.cap_mask = da.cap_mask;
    d.n = da.n;
    d.nblk = da.nblk;
    rd_dedup4_role(d, bid, L.dedup);
    wt.end(3u);
    return;
  }
  bid -= da.nblk;
  if (bid < nblk_build) {
    // numbering of the next batch + its table probe (hints and row reservations for ITS update)
    rd_build_role<false, true>(nxt, uint32_t(kStepLightMax), bid, nblk_build, nullptr, &tv, po);
    wt.end(6u);
    return;
  }
  bid -= nblk_build;
  rd_apply_role<G, VEC, ONESEG, false, FULL, OPTK, FILT ? 1 : 0>(tv, cur, c, a, bid, wt, L.apply);
  wt.end(bid < c.nblk_items ? 7u : 8u);
}

// ---------------------------------------------------------------------------------------------
// Sender side of the id-sharded step (monolith_amd/distributed_ps_sync.py): the two ops of the
// reference's worker that sit between its dedup and the all-to-all, on the run format of a
// deduplicated + built batch:
//   SCATTER  out[p, :]   = rows[idx(u), :]  for every occurrence p of unique index u
//            (MonolithFillWithOffsetMap, RT/ops/unique_mapping_ops.cc:204-268)
//   SUM      out[idx(u), :] = sum over the occurrences p of u of grads[p, :], occurrence order
//            (MonolithFillWithOffsetMapGradient, :284-329)
// idx(u) = index ? index[u] : u — the position of u in the shard-major send buffer
// (rd_partition below), so rows come back / gradients leave in send order with no extra permute.
// Same traversal as rd_apply_role: id-major groups for lists of <= kLightMax occurrences, item
// workgroups for the heavy ones; same summation order, so a sharded run adds up exactly as the
// single-GPU step would have.
// ---------------------------------------------------------------------------------------------
struct GatherCtl {
  const float* in;        // SCATTER: rows [*, dim]; SUM: gradients [n, dim]
  float* out;             // SCATTER: [n, dim];      SUM: [*, dim]
  const uint32_t* index;  // optional indirection of the unique index
  float* part;            // SUM: [items, dim] partial rows of multi-item lists
  uint32_t* arrive;       // SUM: arrival counters, kept zeroed
  int64_t n_max;          // capacity of the dense arrays
  uint32_t dim;
  uint32_t nblk_items;
  uint32_t nblk_ids;
  uint32_t index_is_offset;  // index[u] is a float offset into the rows (0xffffffff: no row — the
                             // id found no room in its peer block) instead of a row number
  // SUM with direct peer stores: offset ix = owner * rows_block + r goes to peer_win[owner] + peer_out_off
  // (this rank's block of the owner's gradient buffer) + r floats instead of out + ix
  const unsigned long long* peer_win;   // nullptr: out
  unsigned long long peer_out_off;
  uint32_t rows_block;
  uint32_t pre_summed;   // SUM, 1 (exact order): the heavy lists' strictly sequential sums are already where
                         // they belong (shard_exact_sum_kernel, launched in front): the item workgroups leave
};

// LDS of the gather role, for NG = 256 / G lane groups; the caller declares it
template <int G, int VEC>
struct GatherLds {
  uint32_t pos[256 / G][kStepLightMax];
  uint32_t rstart[65];
  uint32_t rval[64];
  float sum[256 / G][G * VEC];
  uint32_t last;
};

__device__ __forceinline__ int64_t gather_row_off(const GatherCtl& c, uint32_t ix) {
  return c.index_is_offset ? int64_t(ix) : int64_t(ix) * c.dim;
}
// where the sum of the row with offset / index ix is stored
__device__ __forceinline__ float* gather_out_ptr(const GatherCtl& c, uint32_t ix) {
  if (c.peer_win) {
    const uint32_t owner = ix / c.rows_block;
    return reinterpret_cast<float*>(c.peer_win[owner] + c.peer_out_off) + (ix - owner * c.rows_block);
  }
  return c.out + gather_row_off(c, ix);
}

template <int G, int VEC, bool SCATTER>
__device__ __forceinline__ void rd_gather_role(const RunView& d, const GatherCtl& c, uint32_t bid,
                                               GatherLds<G, VEC>& L) {
  constexpr int WIN = G < 8 ? G : 8;
  constexpr int NG = 256 / G;
  const int lane = threadIdx.x & 63;
  const int j = lane & (G - 1);
  const int gbase = lane & ~(G - 1);
  const int grp = threadIdx.x / G;
  const uint32_t dim = c.dim;
  const int64_t n_max = c.n_max;
  const uint32_t e = uint32_t(j) * VEC;
  const bool ev = e < dim;

  if (bid >= c.nblk_items) {
    auto& sh_pos = L.pos;
    const int64_t stride = int64_t(c.nblk_ids) * NG;
    const int64_t k = bid - c.nblk_items;
    int64_t nu = n_max;
#pragma unroll 1
    for (int64_t it = 0; it * stride < nu; ++it) {
      const int64_t g = it * stride + int64_t(grp) * c.nblk_ids + k;
      const bool inb = g < n_max;
      uint32_t cnt = inb ? d.ucnt[g] : 0u;
      const uint32_t hp = inb ? d.upos[g] : 0u;
      const uint32_t gs = inb ? d.uslot[g] : 0u;
      const uint32_t ix = inb ? (c.index ? c.index[g] : uint32_t(g)) : 0u;
      if (it == 0) nu = min(n_max, int64_t(d.ctr[0]));
      const bool inrange = g < nu && cnt <= uint32_t(kStepLightMax);
      if (!inrange) cnt = 0;
      const bool valid = inrange && ix != 0xffffffffu;  // (no row: zeros out, gradient dropped)
      constexpr int PER = (kStepLightMax + G - 1) / G;
      uint32_t x[PER];
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const uint32_t idx = uint32_t(j) + uint32_t(q) * G;
        x[q] = (cnt > 1 && idx < cnt) ? d.hlist[size_t(gs) * kLightMax + idx] : 0xffffffffu;
      }
      if (SCATTER) {
        Vec<VEC> row;
        vec_zero(row);
        if (valid && ev) row.load(c.in + gather_row_off(c, ix) + e);
        if (cnt == 1) {
          if (ev) row.store(c.out + int64_t(hp) * dim + e);
        } else if (cnt > 1) {
#pragma unroll
          for (int q = 0; q < PER; ++q) {
#pragma unroll 4
            for (int t2 = 0; t2 < G; ++t2) {
              const uint32_t p = __shfl(x[q], gbase + t2);
              if (p != 0xffffffffu && ev) row.store(c.out + int64_t(p) * dim + e);
            }
          }
        }
      } else {
        Vec<VEC> acc;
        vec_zero(acc);
        if (cnt == 1) {
          Vec<VEC> g1;
          vec_zero(g1);
          if (ev) g1.load(c.in + int64_t(hp) * dim + e);
          vec_add(acc, g1);
        }
        if (__any(cnt > 1)) {  // rank the positions in registers, hand over in position order
          uint32_t xr[PER];
#pragma unroll
          for (int q = 0; q < PER; ++q) xr[q] = 0;
#pragma unroll
          for (int q2 = 0; q2 < PER; ++q2) {
#pragma unroll 4
            for (int t2 = 0; t2 < G; ++t2) {
              const uint32_t y = __shfl(x[q2], gbase + t2);
#pragma unroll
              for (int q = 0; q < PER; ++q) xr[q] += (y < x[q]) ? 1u : 0u;
            }
          }
#pragma unroll
          for (int q = 0; q < PER; ++q)
            if (x[q] != 0xffffffffu) sh_pos[grp][xr[q]] = x[q];
          lds_wave_sync();
          if (cnt > 1) sum_list_lds<VEC>(c.in, dim, e, ev, sh_pos[grp], cnt, acc);
          lds_wave_sync();
        }
        if (valid && ev) acc.store(gather_out_ptr(c, ix) + e);
      }
    }
    return;
  }

  // ---- item workgroups
  auto& sh_rstart = L.rstart;
  auto& sh_rval = L.rval;
  auto& sh_sum = L.sum;
  uint32_t& sh_last = L.last;
#pragma unroll 1
  for (uint32_t w = bid;; w += c.nblk_items) {
    const uint32_t nitems_all = d.ctr[2];
    const ItemHdr hd = d.item_hdr[w];
    const uint32_t rval = d.item_runs[size_t(w) * 64 + lane];  // (every wavefront: no branch, no queueing)
    // (pins the three loads above in front of the exit test: sunk below it, they would wait for
    // the count's round trip first)
    asm volatile("" ::"v"(hd.meta), "v"(rval), "v"(nitems_all));
    if (w >= nitems_all) break;
    const uint32_t b0 = hd.meta & 0xffu, nbk = (hd.meta >> 8) & 0xffu, kk = (hd.meta >> 16) & 0xffu,
                   nitems = hd.meta >> 24;
    const uint32_t ix = c.index ? c.index[hd.u] : hd.u;
    if (threadIdx.x < 64) {
      const uint32_t val = (uint32_t(lane) < nbk) ? rval : 0u;
      uint32_t incl = run_cnt(val);
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
      }
      sh_rval[lane] = val;
      sh_rstart[lane + 1] = incl;
      if (lane == 0) sh_rstart[0] = 0;
    }
    lds_barrier();
    if (!SCATTER && c.pre_summed) {   // (workgroup-uniform: nothing to sum, nothing to hand over)
      lds_barrier();
      continue;
    }
    const uint32_t E = sh_rstart[64];
    Vec<VEC> acc, row;
    vec_zero(acc);
    vec_zero(row);
    if (SCATTER && ev && ix != 0xffffffffu) row.load(c.in + gather_row_off(c, ix) + e);
#pragma unroll 1
    for (uint32_t qb = uint32_t(grp) * WIN; qb < E; qb += NG * WIN) {
      uint32_t p = 0;
      if (j < WIN && qb + j < E) {
        const uint32_t q = qb + j;
        uint32_t lo = 0, hi = 63;
        while (lo < hi) {
          const uint32_t mid = (lo + hi + 1) >> 1;
          if (sh_rstart[mid] <= q) lo = mid; else hi = mid - 1;
        }
        const uint32_t val = sh_rval[lo];
        const uint32_t i = q - sh_rstart[lo];
        const uint32_t b = b0 + lo;
        p = b * kRdBlock + ((run_cnt(val) == 1) ? run_first(val)
                                                 : uint32_t(d.seg[b * kRdBlock + run_off(val) + i]));
      }
      if (SCATTER) {
#pragma unroll
        for (int t = 0; t < WIN; ++t) {
          const uint32_t pt = __shfl(p, gbase + t);
          if (qb + t < E && ev) row.store(c.out + int64_t(pt) * dim + e);
        }
      } else {
        Vec<VEC> v[WIN];
#pragma unroll
        for (int t = 0; t < WIN; ++t) vec_zero(v[t]);
#pragma unroll
        for (int t = 0; t < WIN; ++t) {
          const uint32_t pt = __shfl(p, gbase + t);
          if (qb + t < E && ev) v[t].load(c.in + int64_t(pt) * dim + e);
        }
#pragma unroll
        for (int t = 0; t < WIN; ++t)
          if (qb + t < E && ev) vec_add(acc, v[t]);
      }
    }
    if (!SCATTER) {
#pragma unroll
      for (int cc = 0; cc < VEC; ++cc) sh_sum[grp][j * VEC + cc] = acc.v[cc];
      lds_barrier();
      Vec<VEC> tot;
      vec_zero(tot);
      if (threadIdx.x < 64) {
#pragma unroll 1
        for (int g2 = 0; g2 < NG; ++g2) {
#pragma unroll
          for (int cc = 0; cc < VEC; ++cc) tot.v[cc] = tot.v[cc] + sh_sum[g2][j * VEC + cc];
        }
      }
      bool fin = nitems == 1;
      if (nitems > 1) {
        if (threadIdx.x < G && ev) store_wt<VEC>(c.part + int64_t(w) * dim + e, tot);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        if (threadIdx.x == 0) {
          const uint32_t last = (atomicAdd(&c.arrive[hd.u], 1u) == nitems - 1) ? 1u : 0u;
          if (last) c.arrive[hd.u] = 0;
          sh_last = last;
        }
        lds_barrier();
        if (sh_last) {
          fin = true;
          const uint32_t w0 = w - kk;
          const uint32_t per = (nitems + NG - 1) / NG;
          const uint32_t k0 = min(nitems, uint32_t(grp) * per), k1 = min(nitems, k0 + per);
          Vec<VEC> sacc;
          vec_zero(sacc);
#pragma unroll 1
          for (uint32_t q = k0; q < k1; q += 8) {
            Vec<VEC> r[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) vec_zero(r[t]);
#pragma unroll
            for (int t = 0; t < 8; ++t)
              if (q + t < k1 && ev) load_agent<VEC>(c.part + int64_t(w0 + q + t) * dim + e, r[t]);
#pragma unroll
            for (int t = 0; t < 8; ++t)
              if (q + t < k1 && ev) vec_add(sacc, r[t]);
          }
          lds_barrier();
#pragma unroll
          for (int cc = 0; cc < VEC; ++cc) sh_sum[grp][j * VEC + cc] = sacc.v[cc];
          lds_barrier();
          vec_zero(tot);
          if (threadIdx.x < 64) {
            for (int g2 = 0; g2 < NG && uint32_t(g2) * per < nitems; ++g2) {
#pragma unroll
              for (int cc = 0; cc < VEC; ++cc) tot.v[cc] = tot.v[cc] + sh_sum[g2][j * VEC + cc];
            }
          }
        }
      }
      if (fin && threadIdx.x < G && ev && ix != 0xffffffffu) tot.store(gather_out_ptr(c, ix) + e);
    }
    lds_barrier();
  }
}

template <int G, int VEC, bool SCATTER>
__global__ __launch_bounds__(256) void rd_gather_kernel(RunView d, GatherCtl c) {
  __shared__ GatherLds<G, VEC> L;
  rd_gather_role<G, VEC, SCATTER>(d, c, blockIdx.x, L);
}

// ---------------------------------------------------------------------------------------------
// Shard packing of the unique ids of a batch: FusedReorderByIndices' shard-major layout
// (RT/ops/fused_reorder_by_indices.cc:75-123; shard = floormod(id, N), NT/distributed_ps.py:289)
// for one table, on ids that are already unique.  Two launches:
//   rd_shard_count   counts[s] = number of ids of shard s (LDS histogram, one global add per
//                    workgroup and shard)
//   rd_shard_place   send_pos[u] = position of unique index u in the shard-major buffer,
//                    send_ids[send_pos[u]] = id.  The order inside a shard is arrival order:
//                    every consumer goes through send_pos, nothing depends on it.
// n comes from device memory (the build role's unique count).
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(1024) void rd_shard_count_kernel(const int64_t* __restrict__ ids,
                                                              const uint32_t* __restrict__ n_dev,
                                                              int64_t n_max, uint32_t nshards,
                                                              uint32_t* __restrict__ counts) {
  __shared__ uint32_t h[kMaxShards];
  if (threadIdx.x < kMaxShards) h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t n = min(n_max, int64_t(*n_dev));
  const int64_t u = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u < n) atomicAdd(&h[shard_of_id(ids[u], nshards)], 1u);
  __syncthreads();
  if (threadIdx.x < nshards && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], h[threadIdx.x]);
}

__global__ __launch_bounds__(1024) void rd_shard_place_kernel(const int64_t* __restrict__ ids,
                                                              const uint32_t* __restrict__ n_dev,
                                                              int64_t n_max, uint32_t nshards,
                                                              const uint32_t* __restrict__ counts,
                                                              uint32_t* __restrict__ cursor,
                                                              int64_t* __restrict__ send_ids,
                                                              uint32_t* __restrict__ send_pos) {
  __shared__ uint32_t h[kMaxShards], base[kMaxShards], off[kMaxShards];
  if (threadIdx.x < kMaxShards) h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t n = min(n_max, int64_t(*n_dev));
  const int64_t u = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  uint32_t sh = 0, rank = 0;
  int64_t id = 0;
  if (u < n) {
    id = ids[u];
    sh = shard_of_id(id, nshards);
    rank = atomicAdd(&h[sh], 1u);
  }
  __syncthreads();
  if (threadIdx.x < nshards) {
    uint32_t o = 0;
    for (uint32_t s2 = 0; s2 < threadIdx.x; ++s2) o += counts[s2];
    off[threadIdx.x] = o;
    base[threadIdx.x] = h[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], h[threadIdx.x]) : 0u;
  }
  __syncthreads();
  if (u < n) {
    const uint32_t q = off[sh] + base[sh] + rank;
    send_pos[u] = q;
    send_ids[q] = id;
  }
}

}  // namespace mhte
#endif  // MHTE_STEP_KERNELS_H_
