// Multi-table forms of the training step and of the fused ops: ONE launch covers every table of a
// MultiHashTable (blockIdx.y = table or segment), reading the tables' descriptors from a device
// array instead of the kernel-argument buffer.  Included by mhte.hip after mhte_step_kernels.h.
//
// Reference shape being served (paths relative to /root/reference/monolith/native_training/):
//   runtime/ops/multi_hash_table_lookup_op.cc:33-89     Lookup over T tables (id, id_split)
//   runtime/ops/multi_hash_table_update_op.cc:47-100    Optimize over T tables
//   runtime/ops/multi_hash_table_lookup_op.cc:128-197   FusedLookup: [shard][table] segments
//   runtime/ops/multi_hash_table_update_op.cc:247-308   FusedOptimize
//   multi_type_hash_table.py:253-303                    merged multi-table layout of a model
// The reference loops over the tables (and Shard()s over the shards); one table at B = 65 536 ids
// has too few bytes per dependent phase to fill an MI355X, 26 of them side by side
// do.
//
// What lives where:
//   device memory, rewritten by the host only when it changes
//       TableView[T]     the tables (buckets, slabs, counters, segments; grows on doubling)
//       MStepStatic[T]   the step's two run-dedup workspaces per table and its scratch
//   kernel arguments, per launch (<= 4 KB: kMaxStepTables tables per launch, the host chunks)
//       where each table's slice of the ragged batch starts, grid shares, learning rates.
// The roles are the single-table step's own (mhte_step_kernels.h); the lane-group width G of a
// table is a workgroup-uniform switch.
#ifndef MHTE_MSTEP_KERNELS_H_
#define MHTE_MSTEP_KERNELS_H_

#include "mhte_step_kernels.h"

namespace mhte {

constexpr int kMaxStepTables = 32;   // tables per launch (kernel-argument budget)
constexpr int kMaxSegs = 160;        // [shard][table] segments per fused-op launch

// Descriptors are read through the constant address space: uniform loads become scalar loads that
// no store of the kernel can be thought to clobber.
#define MHTE_CONST __attribute__((address_space(4)))
typedef const MHTE_CONST TableView* ConstViews;

// Lane-group shape of a table in a launch: lanes per id (8 / 16 / 32 / 64), | 1 when a lane moves ONE
// float instead of a float4 — rows that are not whole float4s (a dim-1 FTRL bias slice in front of
// the vector, NT/feature.py:117-120: dims 17 / 33 ...; dim <= 64 then) or whose slice of the flat
// embedding / gradient buffer does not start on a 16-byte boundary.  CALL(G, VEC) is expanded for the
// table's shape; -DMHTE_DEV_FAST keeps (16, 4) only.
// One kernel INSTANCE per lane width (VW = 4 / 1): a launch serves the tables whose shape code has
// that width, the other tables' workgroups leave at once and the host launches the other instance
// only when the model has such tables.  (All eight shapes in one kernel made the register allocation
// of the float4 paths worse: mstep_bwd 178 -> 192 us on 26 float4 tables.)
#define MHTE_SWITCH_G(VW, code, CALL)                      \
  if constexpr ((VW) == 4) {                               \
    switch (code) {                                        \
      case 8: MHTE_OTHER_G(CALL(8, 4)); break;             \
      case 16: CALL(16, 4); break;                         \
      case 32: MHTE_OTHER_G(CALL(32, 4)); break;           \
      case 64: MHTE_OTHER_G(CALL(64, 4)); break;           \
      default: break;                                      \
    }                                                      \
  } else {                                                 \
    switch (code) {                                        \
      case 9: MHTE_OTHER_G(CALL(8, 1)); break;             \
      case 17: MHTE_OTHER_G(CALL(16, 1)); break;           \
      case 33: MHTE_OTHER_G(CALL(32, 1)); break;           \
      case 65: MHTE_OTHER_G(CALL(64, 1)); break;           \
      default: break;                                      \
    }                                                      \
  }
// true when shape code `code` belongs to the instance of lane width VW
#define MHTE_SHAPE_IS(VW, code) ((((code) & 1u) != 0u) == ((VW) == 1))
// (segment kernels only) bit 1 of a shape code: the table has a whole-segment optimizer (GroupAdaGrad,
// group_adagrad_segment) — served by the GROUP instance of seg_upsert_kernel / shard_upsert_kernel
constexpr uint32_t kShapeGroupBit = 2u;

struct MStepStatic {
  RunView rv[2];           // run-dedup workspaces: slot s holds the batch deduplicated into it
                           // (ids / n / nblk are per launch: kernel arguments)
  float* grad_u;           // [n_max, dim] summed gradients of ids left to the displacement pass
  uint32_t* pending;       // [n_max + 1]
  float* part[2];          // per slot: partial rows of multi-item lists
  uint32_t* arrive[2];     // per slot: arrival counters, kept zeroed
  uint32_t* urow[2];       // per slot [n_max]: row handle of unique index u as the forward launch
                           // found it (kNoRow: not in the table), and where its slot is
  unsigned long long* uloc[2];  //          (bucket * 4 + slot): the backward launch need not probe
  uint32_t* uts[2];        // per slot [n_max]: the timestamp the forward launch saw in that slot — the
                           // backward skips its timestamp store when the step's is the same value
  int64_t n_max;           // capacity of the dense arrays (largest batch of the table)
  uint32_t g;              // lanes per id: 8 / 16 / 32 / 64 >= dim / 4
  uint32_t oneseg;         // the table has one segment (scalar descriptor loads, seg_of)
  uint32_t nblk_build;     // workgroups of the build role (a function of the scratch capacity)
  uint32_t count_hits;
};
typedef const MHTE_CONST MStepStatic* ConstStatics;

struct MFwdTab {
  uint32_t n;                  // ids of this batch (its dedup is in slot cur); 0: no lookup
  uint32_t nblk_s;             // workgroups of the lookup role
  uint32_t emb_off;            // floats
  uint32_t gv;                 // lane-group shape of the table in this launch (MHTE_SWITCH_G)
};
struct MFwdArgs {
  ConstViews views;
  ConstStatics st;
  float* out;
  uint32_t cur;                // slot that holds the (numbered) batch being looked up
  uint32_t item_split;         // wavefronts per heavy work item (>= 1)
  unsigned long long* trace;   // per-wavefront timeline (mhte_trace_begin) or nullptr
  MFwdTab tab[kMaxStepTables];
};

struct MBwdTab {
  uint32_t grad_off;           // floats
  uint32_t apply;              // 0: nothing to apply for this table (empty batch / build only); bit 0: apply,
                               // bit 1 (MHTE_EXACT_ORDER): the heavy lists' strictly sequential sums are in
                               // part[] (mstep_exact_sum_kernel, launched in front): the item workgroups only apply
  uint32_t nblk_items, nblk_ids;
  uint32_t build_next;         // 1: slot cur ^ 1 holds a deduplicated batch to number
  uint32_t light_max;
  uint32_t hints;              // 1: urow / uloc of slot cur are what the forward launch left and
                               // nothing has touched the table since
  uint32_t n;                  // ids of the batch in slot cur
  uint32_t n_next;             // ids of the batch in slot cur ^ 1
  uint32_t full;               // 1: the table uses optimizers beyond SGD / Adagrad / FTRL (the
                               // mstep_bwd_kernel<true> launch serves it, <false> the others)
  uint32_t gv;                 // lane-group shape of the table in this launch (MHTE_SWITCH_G)
  ApplyArgs a;
};
struct MBwdArgs {
  ConstViews views;
  ConstStatics st;
  const float* grads;
  uint32_t cur;
  uint32_t pad;
  unsigned long long* trace;
  MBwdTab tab[kMaxStepTables];
};
static_assert(sizeof(MFwdArgs) <= 4096 && sizeof(MBwdArgs) <= 4096, "kernel arguments exceed 4 KB");

template <typename T>
__device__ __forceinline__ const T& deref_const(const MHTE_CONST T* p) {
  return *(const T*)(p);
}

// ---------------------------------------------------------------------------------------------
// scratch reset of the tables' run-dedup slots (creation; a slot whose dedup was never numbered)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mstep_clear_kernel(ConstStatics st, uint32_t slot_mask) {
  const MStepStatic& s = deref_const(st + blockIdx.y);
#pragma unroll
  for (uint32_t sl = 0; sl < 2; ++sl) {
    if (!((slot_mask >> sl) & 1u)) continue;
    const RunView d = s.rv[sl];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= d.cap_mask + 1u) {
      d.hs[i] = RdSlot{kEmptyKey, 0ull};
    }
    if (i < 4) d.ctr[i] = 0;
  }
}

// ---------------------------------------------------------------------------------------------
// run dedup of the NEXT batch of every table, on its own (side) stream beside the step's launches.
// What paces this role is not bandwidth but device-scope atomics (two per distinct id and dedup
// workgroup: the slot claim and the count bump; ~36 G/s on MI355X, scripts/mstep_probe.py), so it
// gets a fixed, small number of persistent workgroups — the memory-bound lookup / apply launches
// keep the rest of the chip — and walks the (table, 1024 positions) items in grid-stride order.
// ---------------------------------------------------------------------------------------------
struct MDedupArgs {
  ConstStatics st;
  const int64_t* ids;
  uint32_t slot;
  uint32_t T;
  unsigned long long* trace;
  uint32_t id_off[kMaxStepTables + 1];      // table t: ids[id_off[t], id_off[t + 1])
  uint32_t blk_start[kMaxStepTables + 1];   // first item of table t
};

__global__ __launch_bounds__(kRdBlock, 8) __attribute__((amdgpu_num_sgpr(80))) void mstep_dedup_kernel(
    MDedupArgs A) {
  __shared__ __attribute__((aligned(16))) RdLds L;
  WaveTrace wt(A.trace);
  const uint32_t total = A.blk_start[A.T];
#pragma unroll 1
  for (uint32_t w = blockIdx.x; w < total; w += gridDim.x) {
    uint32_t t = 0;
    while (t + 1 < A.T && A.blk_start[t + 1] <= w) ++t;  // (uniform; T <= 32)
    const MStepStatic& s = deref_const(A.st + t);
    RunView d = s.rv[A.slot & 1u];
    d.ids = A.ids + A.id_off[t];
    d.n = A.id_off[t + 1] - A.id_off[t];
    d.nblk = A.blk_start[t + 1] - A.blk_start[t];
    rd_dedup_role(d, w - A.blk_start[t], L, wt);
    __syncthreads();  // (the LDS is reused by the next item)
  }
  wt.end(3u);
}

// ---------------------------------------------------------------------------------------------
// forward: per table, lookup of this batch (the run dedup of the NEXT batch is its own launch,
// mstep_dedup_kernel: 1024-thread workgroups do not share a launch with these 256-thread ones)
//
// The batch being looked up was deduplicated and numbered a step ahead, so the lookup probes each
// DISTINCT id once (U ~ 12 k of B = 65 536 under Zipf(1.2)) and scatters its row to the id's
// occurrences through the run format — the reference's lookup(unique) + MonolithFillWithOffsetMap
// (RT/ops/unique_mapping_ops.cc:204-268) as one role: stores are fire and forget, so a wavefront
// has one dependent chain (dense arrays -> buckets -> row) per DISTINCT id instead of per
// occurrence.  What the probe found (row handle, bucket slot) is left for the backward launch,
// which then reaches the row of a resident id without reading a bucket.
// (The displacement pass of the previous update has its own launch, mstep_slow_kernel: with T
// tables in a launch a 3 us launch is noise, and the lookups need no gate.)
// ---------------------------------------------------------------------------------------------
#ifndef MHTE_SCATTER_UNR
#define MHTE_SCATTER_UNR 2
#endif
#ifdef MHTE_SCATTER_PLAIN_STORES
#define MHTE_SCATTER_STORE(P, V) (V).store(P)
#else
#define MHTE_SCATTER_STORE(P, V) store_stream<VEC>(P, V)
#endif
template <int G, int BLOCK, int UNR, int VEC = 4>
__device__ __forceinline__ void mstep_scatter_role(const TableView& tv, const RunView& d,
                                                   float* __restrict__ out,
                                                   uint32_t* __restrict__ urow,
                                                   unsigned long long* __restrict__ uloc,
                                                   uint32_t* __restrict__ uts,
                                                   int64_t n_max, int count_hits, uint32_t bid,
                                                   uint32_t nblk, uint32_t item_split,
                                                   WaveTrace& wt) {
  constexpr int NG = BLOCK / G;
  constexpr int GPW = 64 / G;  // groups per wavefront
  const int lane = threadIdx.x & 63;
  const int j = lane & (G - 1);
  const int gbase = lane & ~(G - 1);
  const int grp = threadIdx.x / G;
  const uint32_t dim = tv.dim;
  const uint32_t e = uint32_t(j) * VEC;
  const bool ev = e < dim;
  const uint32_t n_unique = d.ctr[0];
  const uint32_t n_items = d.ctr[2];
  const int64_t nu = min(n_max, int64_t(n_unique));
  uint64_t hits = 0;
  // ---------------------------------------------------------------- distinct ids, UNR per group
  // (a wavefront's throughput is ids per trip / the trip's dependent round trips, so UNR ids per
  // lane group are in flight together: their probes, then their rows)
  constexpr int PER = (kStepLightMax + G - 1) / G;
  const int64_t stride = int64_t(nblk) * NG * UNR;
#pragma unroll 1
  for (int64_t g0 = int64_t(bid) * NG * UNR; g0 < nu; g0 += stride) {  // workgroup-uniform
    int64_t id[UNR];
    uint32_t cnt[UNR], hp[UNR], gs[UNR];
    bool valid[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t g = g0 + int64_t(grp) * UNR + u;
      valid[u] = g < nu;
      const int64_t gi = valid[u] ? g : 0;  // (loads from a safe index, masked afterwards)
      id[u] = d.uids[gi];
      cnt[u] = d.ucnt[gi];
      hp[u] = d.upos[gi];
      gs[u] = d.uslot[gi];
    }
    if (wt.rec && g0 == int64_t(bid) * NG * UNR) {  // (traced runs: when the first trip's ids arrive)
      asm volatile("" ::"v"(cnt[0]));
      wt.mark(1);
    }
    int64_t kk[UNR];
    uint32_t row[UNR], tsv[UNR], x[UNR][PER];
    uint64_t i1[UNR], i2[UNR];
    bool flat[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (!valid[u]) cnt[u] = 0;
      const uint64_t hv = hash_key(id[u]);
      i1[u] = index_hash(tv.hp, hv);
      i2[u] = alt_index(tv.hp, partial_key(hv), i1[u]);
      const GBucket* b = global_bucket(tv.buckets + ((j & 4) ? i2[u] : i1[u]));
      kk[u] = b->key[j & 3];
      row[u] = b->row[j & 3];
      tsv[u] = b->ts[j & 3];   // (same 64-byte line)
      flat[u] = cnt[u] > 1 && cnt[u] <= uint32_t(kStepLightMax);
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const uint32_t idx = uint32_t(j) + uint32_t(q) * G;
        x[u][q] = (flat[u] && idx < cnt[u]) ? d.hlist[size_t(gs[u]) * kLightMax + idx] : 0xffffffffu;
      }
    }
    if (wt.rec && g0 == int64_t(bid) * NG * UNR) {  // (... its bucket lines and position lists)
      asm volatile("" ::"v"(row[0]), "v"(x[0][0]));
      wt.mark(2);
    }
    Vec<VEC> v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const bool use = valid[u] && j < 8 && id[u] != kEmptyKey;
      const uint64_t m = group_mask_of<G>(__ballot(use && kk[u] == id[u]), gbase);
      bool found = m != 0;
      const int src = found ? (__ffsll(static_cast<long long>(m)) - 1) : 0;
      uint32_t r = __shfl(use ? row[u] : kNoRow, gbase + src);
      const uint32_t ots = __shfl(tsv[u], gbase + src);
      const bool special = valid[u] && id[u] == kEmptyKey;
      if (special) {
        found = tv.ctr->special_state == 1;
        r = tv.ctr->special_row;
      }
      found = found && valid[u];
      if (valid[u] && j == 0) {  // (the side slot's key is left to the update's own path)
        const int64_t g = g0 + int64_t(grp) * UNR + u;
        urow[g] = (found && !special) ? r : kNoRow;
        uloc[g] = (((src & 4) ? i2[u] : i1[u]) << 2) | uint64_t(src & 3);
        uts[g] = ots;
      }
      if (count_hits && found && j == 0) hits += cnt[u];  // (per occurrence, as the direct lookup counts)
      const bool light = valid[u] && cnt[u] <= uint32_t(kStepLightMax);
      vec_zero(v[u]);
      if (found && light && ev) v[u].load(row_ptr(tv, r) + e);
    }
    if (wt.rec && g0 == int64_t(bid) * NG * UNR) {  // (... its rows)
      asm volatile("" ::"v"(v[0].v[0]));
      wt.mark(3);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (cnt[u] == 1) {
        if (ev) MHTE_SCATTER_STORE(out + int64_t(hp[u]) * dim + e, v[u]);
      } else if (flat[u]) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
#pragma unroll 4
          for (int t2 = 0; t2 < G; ++t2) {
            const uint32_t p = __shfl(x[u][q], gbase + t2);
            if (p != 0xffffffffu && ev) MHTE_SCATTER_STORE(out + int64_t(p) * dim + e, v[u]);
          }
        }
      }
    }
  }
  wt.mark(0);
  // ---------------------------------------------------------------- heavy lists: one work item
  // (~256 occurrences of one id, a power-of-two range of dedup workgroups) per WAVEFRONT; run
  // starts by wave scan, no LDS, no barrier
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll 1
  // (item_split wavefronts share an item, each taking every item_split-th pass of 64 occurrences:
  // items are cut for the update launch, where a bigger item amortises its fixed round trips)
  for (uint32_t unit = bid * (BLOCK / 64) + wave; unit < n_items * item_split;
       unit += nblk * (BLOCK / 64)) {
    const uint32_t w = unit / item_split, sub = unit % item_split;
    const ItemHdr hd = d.item_hdr[w];
    const uint32_t rval = d.item_runs[size_t(w) * 64 + lane];
    const uint32_t b0 = hd.meta & 0xffu, nbk = (hd.meta >> 8) & 0xffu;
    const uint32_t val = (uint32_t(lane) < nbk) ? rval : 0u;
    uint32_t incl = run_cnt(val);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    const uint32_t excl = incl - run_cnt(val);
    const uint32_t E = __shfl(incl, 63);
    // the id's row (every group fetches the same lines: one request)
    const uint64_t hv = hash_key(hd.id);
    const uint64_t i1 = index_hash(tv.hp, hv);
    const uint64_t i2 = alt_index(tv.hp, partial_key(hv), i1);
    const GBucket* b = global_bucket(tv.buckets + ((j & 4) ? i2 : i1));
    const int64_t kk = b->key[j & 3];
    const uint32_t row = b->row[j & 3];
    const bool use = j < 8 && hd.id != kEmptyKey;
    const uint64_t m = group_mask_of<G>(__ballot(use && kk == hd.id), gbase);
    bool found = m != 0;
    const int src = found ? (__ffsll(static_cast<long long>(m)) - 1) : 0;
    uint32_t r = __shfl(use ? row : kNoRow, gbase + src);
    if (hd.id == kEmptyKey) {
      found = tv.ctr->special_state == 1;
      r = tv.ctr->special_row;
    }
    Vec<VEC> v;
    vec_zero(v);
    if (found && ev) v.load(row_ptr(tv, r) + e);
#pragma unroll 1
    for (uint32_t qb = sub * 64; qb < E; qb += 64 * item_split) {
      const uint32_t q = qb + uint32_t(lane);
      const bool has = q < E;
      const uint32_t qq = has ? q : 0u;
      uint32_t lo = 0, hi = 63;  // run lo with excl[lo] <= q < incl[lo]
#pragma unroll
      for (int it2 = 0; it2 < 6; ++it2) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        const bool le = uint32_t(__shfl(excl, int(mid))) <= qq;
        lo = le ? mid : lo;
        hi = le ? hi : mid - 1;
      }
      const uint32_t vr = __shfl(val, int(lo));
      const uint32_t er = __shfl(excl, int(lo));
      const uint32_t base = (b0 + lo) * kRdBlock;
      uint32_t p = base + run_first(vr);
      if (has && run_cnt(vr) != 1) p = base + uint32_t(d.seg[base + run_off(vr) + (qq - er)]);
#pragma unroll 4
      for (int t = 0; t < G; ++t) {
        const int idx = t * GPW + (lane / G);
        const uint32_t pt = __shfl(p, idx);
        if (qb + uint32_t(idx) < E && ev) MHTE_SCATTER_STORE(out + int64_t(pt) * dim + e, v);
      }
    }
  }
  wt.mark(4);
  if (count_hits) {
    unsigned long long tot = hits;  // (leader lanes hold partial counts)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) tot += __shfl_xor(tot, o);
    if (lane == 0 && tot) atomicAdd(&tv.ctr->hits, tot);
  }
}

template <int BLOCK, int VW = 4>
__global__ __launch_bounds__(BLOCK) void mstep_fwd_kernel(MFwdArgs A) {
  const uint32_t t = blockIdx.y;
  const MFwdTab ft = A.tab[t];
  const uint32_t bid = blockIdx.x;
  if (ft.n == 0 || bid >= ft.nblk_s || !MHTE_SHAPE_IS(VW, ft.gv)) return;
  const MStepStatic& s = deref_const(A.st + t);
  WaveTrace wt(A.trace);
  const TableView& tv = deref_const(A.views + t);
  const uint32_t cur = A.cur & 1u;
  const RunView d = s.rv[cur];
  float* out = A.out + size_t(ft.emb_off);
  const int ch = int(s.count_hits);
#define MHTE_FWD_CALL(G_, V_) \
  mstep_scatter_role<G_, BLOCK, MHTE_SCATTER_UNR, V_>(tv, d, out, s.urow[cur], s.uloc[cur], s.uts[cur], s.n_max, ch, bid, ft.nblk_s, A.item_split, wt)
  MHTE_SWITCH_G(VW, ft.gv, MHTE_FWD_CALL)
#undef MHTE_FWD_CALL
  wt.end(5u);
}

// ---------------------------------------------------------------------------------------------
// forward + the run dedup of the NEXT batch in ONE launch (as the single-table step_fwd does): the
// lookup / scatter is bound by HBM bandwidth, the dedup by device-scope atomics — side by side they
// take about as long as the longer one (measured: 97 + 103 us as two launches at 26 x 65 536 ids).
// 1024-thread workgroups: the first `nd` are the dedup's persistent workgroups, the rest belong to
// the tables' lookups (fwd_start[t] .. fwd_start[t + 1]).
// ---------------------------------------------------------------------------------------------
struct MFwdFuse {
  uint32_t nd;                                // dedup workgroups (persistent: item w, w + nd, ...)
  uint32_t period;                            // workgroup b is dedup workgroup b / period when
                                              // b % period == 0 and b / period < nd: the two roles'
                                              // workgroups start interleaved, not one role first
  uint32_t fwd_start[kMaxStepTables + 1];     // first lookup workgroup of table t
};
#ifndef MHTE_FUSED_SCATTER_UNR
#define MHTE_FUSED_SCATTER_UNR 1
#endif
// (8 waves per SIMD = two 1024-thread workgroups per CU: at 76 VGPRs only one fits and the lookups
// wait for the persistent dedup workgroups to leave — measured 12.7 ms with 256 of them)
__global__ __launch_bounds__(kRdBlock, 8) __attribute__((amdgpu_num_sgpr(80))) void mstep_fwd_dedup_kernel(
    MFwdArgs A, MDedupArgs D, MFwdFuse F) {
  __shared__ __attribute__((aligned(16))) RdLds L;
  WaveTrace wt(A.trace);
  const uint32_t bq = blockIdx.x / F.period;
  if (blockIdx.x % F.period == 0 && bq < F.nd) {
    const uint32_t total = D.blk_start[D.T];
#pragma unroll 1
    for (uint32_t w = bq; w < total; w += F.nd) {
      uint32_t t = 0;
      while (t + 1 < D.T && D.blk_start[t + 1] <= w) ++t;
      const MStepStatic& s = deref_const(D.st + t);
      RunView d = s.rv[D.slot & 1u];
      d.ids = D.ids + D.id_off[t];
      d.n = D.id_off[t + 1] - D.id_off[t];
      d.nblk = D.blk_start[t + 1] - D.blk_start[t];
      rd_dedup_role(d, w - D.blk_start[t], L, wt);
      __syncthreads();
    }
    wt.end(3u);
    return;
  }
  const uint32_t lin = blockIdx.x - min(F.nd, (blockIdx.x + F.period - 1) / F.period);   // dedup workgroups before it
  uint32_t t = 0;
  while (t + 1 < D.T && F.fwd_start[t + 1] <= lin) ++t;
  const MFwdTab ft = A.tab[t];
  const uint32_t bid = lin - F.fwd_start[t];
  // (tables that move one float per lane are looked up by a launch of their own,
  // mstep_fwd_kernel<.., 1>: this kernel must stay inside 64 VGPRs)
  if (ft.n == 0 || bid >= ft.nblk_s || !MHTE_SHAPE_IS(4, ft.gv)) return;
  const MStepStatic& s = deref_const(A.st + t);
  const TableView& tv = deref_const(A.views + t);
  const uint32_t cur = A.cur & 1u;
  const RunView d = s.rv[cur];
  float* out = A.out + size_t(ft.emb_off);
  const int ch = int(s.count_hits);
  constexpr int U = MHTE_FUSED_SCATTER_UNR;
#define MHTE_FWD_CALL(G_, V_) \
  mstep_scatter_role<G_, kRdBlock, U, V_>(tv, d, out, s.urow[cur], s.uloc[cur], s.uts[cur], s.n_max, ch, bid, ft.nblk_s, A.item_split, wt)
  MHTE_SWITCH_G(4, ft.gv, MHTE_FWD_CALL)
#undef MHTE_FWD_CALL
  wt.end(5u);
}

// ---------------------------------------------------------------------------------------------
// backward: per table   numbering + heavy work list of the NEXT batch | apply of this batch
// ---------------------------------------------------------------------------------------------
// FILT: a table of the launch consults an admission filter (rd_apply_role's run-time form, -1); without
// one the instance carries no filter code (0): the float4 one-segment BASIC instance went from 38 spilled
// VGPRs and 472 spilled SGPRs to 4 and 181 when the code left (round 5)
template <bool ONESEG, bool FULL, int VW, bool FILT>
__device__ __forceinline__ void mstep_apply_switch(uint32_t gv, const TableView& tv, const RunView& d,
                                                   const ApplyCtl& c, const ApplyArgs& a,
                                                   uint32_t bid, WaveTrace& wt, ApplyLds& L) {
#define MHTE_BWD_CALL(G_, V_) \
  rd_apply_role<G_, V_, ONESEG, true, FULL, -1, FILT ? -1 : 0>(tv, d, c, a, bid, wt, L)
  MHTE_SWITCH_G(VW, gv, MHTE_BWD_CALL)
#undef MHTE_BWD_CALL
}

#ifndef MHTE_MBWD_OCC
#define MHTE_MBWD_OCC kBwdBlocksPerCu
#endif
// One instance per (optimizer family, lane width, one segment / several): a table is served by its
// instance, the host launches the instances the model has tables for (one, for configs[4]).
template <bool FULL, int VW, bool ONESEG, bool FILT = false>
__global__ __launch_bounds__(256, MHTE_MBWD_OCC) void mstep_bwd_kernel(MBwdArgs A) {
  __shared__ ApplyLds L;
  const uint32_t t = blockIdx.y;
  const MBwdTab& bt = A.tab[t];
  if ((bt.full != 0u) != FULL || !MHTE_SHAPE_IS(VW, bt.gv)) return;   // another instance serves this table
  const MStepStatic& s = deref_const(A.st + t);
  if ((s.oneseg != 0u) != ONESEG) return;
  WaveTrace wt(A.trace);
  uint32_t bid = blockIdx.x;
  const uint32_t cur = A.cur & 1u;
  const uint32_t nblk_build = bt.build_next ? s.nblk_build : 0u;
  if (bid < nblk_build) {
    RunView nxt = s.rv[cur ^ 1u];
    nxt.nblk = (bt.n_next + kRdBlock - 1) / kRdBlock;
    rd_build_role(nxt, uint32_t(kStepLightMax), bid, nblk_build);
    wt.end(6u);
    return;
  }
  bid -= nblk_build;
  if (!bt.apply || bid >= bt.nblk_items + bt.nblk_ids) return;
  const TableView& tv = deref_const(A.views + t);
  RunView d = s.rv[cur];
  d.nblk = (bt.n + kRdBlock - 1) / kRdBlock;
  ApplyCtl c;
  c.grads = A.grads + size_t(bt.grad_off);
  c.grad_u = s.grad_u;
  c.pending = s.pending;
  c.part = s.part[cur];
  c.arrive = s.arrive[cur];
  c.n_max = s.n_max;
  c.light_max = bt.light_max;
  c.nblk_items = bt.nblk_items;
  c.nblk_ids = bt.nblk_ids;
  c.urow = bt.hints ? s.urow[cur] : nullptr;
  c.uloc = bt.hints ? s.uloc[cur] : nullptr;
  c.uts = bt.hints ? s.uts[cur] : nullptr;
  c.trusted = 1;   // (the host hands the hints over only while Table::mut_epoch is unchanged)
  c.urec = nullptr;
  c.pre_summed = (bt.apply >> 1) & 1u;
  mstep_apply_switch<ONESEG, FULL, VW, FILT>(bt.gv, tv, d, c, bt.a, bid, wt, L);
  wt.end(bid < bt.nblk_items ? 7u : 8u);
}

// MHTE_EXACT_ORDER for the multi-table step (round 6): every table's heavy lists summed strictly in occurrence
// order in front of mstep_bwd — rd_exact_sum_role, the single-table step's kernel, per table (blockIdx.y)
__global__ __launch_bounds__(kExactThreads) void mstep_exact_sum_kernel(MBwdArgs A) {
  __shared__ ExactLds L;
  const uint32_t t = blockIdx.y;
  const MBwdTab& bt = A.tab[t];
  if (!(bt.apply & 2u)) return;
  const MStepStatic& s = deref_const(A.st + t);
  const TableView& tv = deref_const(A.views + t);
  const uint32_t cur = A.cur & 1u;
  RunView d = s.rv[cur];
  d.nblk = (bt.n + kRdBlock - 1) / kRdBlock;
  const float* grads = A.grads + size_t(bt.grad_off);
  const ExactToPart dst{s.part[cur], tv.dim};
  if (bt.gv & 1u) rd_exact_sum_role<1>(d, grads, tv.dim, dst, blockIdx.x, gridDim.x, L);
  else rd_exact_sum_role<4>(d, grads, tv.dim, dst, blockIdx.x, gridDim.x, L);
}

// displacement pass of every table's update: one workgroup per table, kSlowWaves deferred ids at a
// time (slowpath_par_role; a table defers 1-5 ids per step at load 0.3)
constexpr int kSlowWaves = 4;
__global__ __launch_bounds__(64 * kSlowWaves) void mstep_slow_kernel(MBwdArgs A) {
  __shared__ SlowParLds<kSlowWaves> L;
  const uint32_t t = blockIdx.x;
  const MBwdTab& bt = A.tab[t];
  if (!bt.apply) return;
  const MStepStatic& s = deref_const(A.st + t);
  const TableView& tv = deref_const(A.views + t);
  if (bt.gv & 1u)
    slowpath_par_role<1, kOpOptimize, kSlowWaves, false>(tv, s.rv[A.cur & 1u].uids, s.grad_u, bt.a,
                                                         s.pending, L);
  else
    slowpath_par_role<4, kOpOptimize, kSlowWaves, false>(tv, s.rv[A.cur & 1u].uids, s.grad_u, bt.a,
                                                         s.pending, L);
}

// ---------------------------------------------------------------------------------------------
// Fused ops of the sync-training path: the ids of all (shard, table) segments in ONE launch
// (the reference loops over the tables inside Shard() over the shards,
// multi_hash_table_lookup_op.cc:150-196, multi_hash_table_update_op.cc:270-306).
// Segment y = shard * T + table: ids[id_off[y], id_off[y+1]), rows at emb_off[y].
// ---------------------------------------------------------------------------------------------
struct SegLookupArgs {
  ConstViews views;
  const int64_t* ids;
  float* out;
  uint32_t T;
  uint32_t seg0;                      // first segment of this launch (its table = (seg0 + y) % T)
  uint32_t id_off[kMaxSegs + 1];
  uint32_t emb_off[kMaxSegs + 1];
  uint8_t g[kMaxStepTables * 4];      // per table: lane-group shape (MHTE_SWITCH_G)
  uint8_t count_hits[kMaxStepTables * 4];
};
static_assert(sizeof(SegLookupArgs) <= 4096, "kernel arguments exceed 4 KB");

template <int G, int VEC = 4>
__device__ __forceinline__ void seg_lookup_loop(const TableView& tv, const int64_t* ids, int64_t n,
                                                float* out, int count_hits) {
  const int64_t ngroups = (n + 1) / 2;
#pragma unroll 1
  for (int64_t g = (int64_t(blockIdx.x) * 512 + threadIdx.x) / G; g < ngroups;
       g += int64_t(gridDim.x) * 512 / G)
    lookup_role_u<G, VEC, 2, 1>(tv, ids, n, nullptr, out, count_hits, g);
}

template <int VW>
__global__ __launch_bounds__(512) void seg_lookup_kernel(SegLookupArgs A) {
  const uint32_t y = blockIdx.y;
  const uint32_t n = A.id_off[y + 1] - A.id_off[y];
  if (n == 0) return;
  const uint32_t t = (A.seg0 + y) % A.T;
  if (!MHTE_SHAPE_IS(VW, A.g[t])) return;
  const TableView& tv = deref_const(A.views + t);
  const int64_t* ids = A.ids + A.id_off[y];
  float* out = A.out + size_t(A.emb_off[y]);
  const int ch = A.count_hits[t];
#define MHTE_SEGL_CALL(G_, V_) seg_lookup_loop<G_, V_>(tv, ids, n, out, ch)
  MHTE_SWITCH_G(VW, A.g[t], MHTE_SEGL_CALL)
#undef MHTE_SEGL_CALL
}

// FusedOptimize on ids that are distinct within every segment (they come out of
// FusedReorderByIndices' per-table dedup, fused_reorder_by_indices.cc:52-60; the shards of a
// table hold different ids by construction): probe + insert + optimizer, one launch.  An id whose
// two buckets are full goes to its table's pending list as (position in the flat id array,
// segment); seg_slow_kernel finishes those.
struct SegUpsertArgs {
  ConstViews views;
  const int64_t* ids;
  const float* grads;
  uint32_t* pending[kMaxStepTables];  // per table, 2 words per entry
  uint32_t T;
  uint32_t seg0;
  uint32_t nseg;
  uint32_t pad;
  uint32_t id_off[kMaxSegs + 1];
  uint32_t grad_off[kMaxSegs];
  uint8_t g[kMaxStepTables * 4];
  ApplyArgs a[kMaxStepTables];
};
static_assert(sizeof(SegUpsertArgs) <= 4096, "kernel arguments exceed 4 KB");

template <int G, int VEC = 4, bool GROUP = false>
__device__ __forceinline__ void seg_upsert_loop(const TableView& tv, const int64_t* ids, uint32_t n,
                                                const float* values, const ApplyArgs& a,
                                                uint32_t* pending, uint32_t id_base, uint32_t seg) {
  const int lane = threadIdx.x & 63;
  const int j = lane & (G - 1);
  const uint32_t ngroups_wg = 256 / G;
#pragma unroll 1
  for (uint32_t g0 = blockIdx.x * ngroups_wg; g0 < n; g0 += gridDim.x * ngroups_wg) {  // wave-uniform
    const uint32_t g = g0 + threadIdx.x / G;
    bool valid = g < n;
    const int64_t id = valid ? ids[g] : 0;
    Probe<G> pr = probe_issue<G>(tv, id, valid, j);
    // occurrence filter: the ids of a segment are distinct (one sender's, or one shard's), so an id
    // the table does not hold asks with count 1 — what the reference's fused optimize does per
    // (sender, id) (tf_bridge.cc:300-321 behind multi_hash_table_update_op.cc:270-306)
    if (tv.flt_slots && a.filter_mode) {
      const int gbase = lane & ~(G - 1);
      bool contained = group_mask_of<G>(__ballot(valid && id != kEmptyKey && j < 8 && pr.k == id), gbase) != 0;
      if (valid && id == kEmptyKey) contained = tv.ctr->special_state == 1;
      uint32_t first = 0;
      if (valid && j == 0) first = filter_consult(tv, id, 1u, 1, contained);
      if (__shfl(first, gbase) != 0u) valid = false;
    }
    const SlotResult sr = upsert_resolve<G>(tv, (Bucket*)pr.b, id, valid, pr.k, pr.row, lane, a.ts);
    if (sr.deferred && j == 0) {
      const uint32_t slot = atomicAdd(&tv.ctr->n_pending, 1u);
      pending[2 * slot] = id_base + g;
      pending[2 * slot + 1] = seg;
    }
    if (valid && !sr.deferred)
      apply_row<G, VEC, kOpOptimize, false, GROUP>(tv, row_ptr(tv, sr.r), sr.is_new, j, values, nullptr,
                                                   0u, 1u, int64_t(g), a);
  }
}

// (four workgroups per CU = 128 VGPRs, which the loop fits without a spill; left to itself the
// compiler takes 132 and the owner's upsert of the sharded step runs 17 us instead of 13.8)
#ifndef MHTE_SEGU_OCC
#define MHTE_SEGU_OCC 4
#endif
template <int VW, bool GROUP = false>
__global__ __launch_bounds__(256, GROUP ? 1 : MHTE_SEGU_OCC) void seg_upsert_kernel(SegUpsertArgs A) {
  const uint32_t y = blockIdx.y;
  const uint32_t n = A.id_off[y + 1] - A.id_off[y];
  if (n == 0) return;
  const uint32_t t = (A.seg0 + y) % A.T;
  if (!MHTE_SHAPE_IS(VW, A.g[t]) || ((A.g[t] & kShapeGroupBit) != 0u) != GROUP) return;
  const TableView& tv = deref_const(A.views + t);
  const int64_t* ids = A.ids + A.id_off[y];
  const float* values = A.grads + size_t(A.grad_off[y]);
  uint32_t* pend = A.pending[t];
#define MHTE_SEGU_CALL(G_, V_) seg_upsert_loop<G_, V_, GROUP>(tv, ids, n, values, A.a[t], pend, A.id_off[y], y)
  MHTE_SWITCH_G(VW, A.g[t] & ~kShapeGroupBit, MHTE_SEGU_CALL)
#undef MHTE_SEGU_CALL
}

// displacement pass of a fused optimize: one wavefront per table
__global__ __launch_bounds__(64) void seg_slow_kernel(SegUpsertArgs A) {
  __shared__ BfsSlot q[kMaxCuckooCount];
  __shared__ CuckooRecord path[kMaxBfsPathLen];
  const uint32_t t = blockIdx.x;
  const TableView& tv = deref_const(A.views + t);
  const int lane = threadIdx.x;
  const uint32_t np = tv.ctr->n_pending;
  if (np == 0) return;
  const uint32_t* pending = A.pending[t];
  const ApplyArgs& a = A.a[t];
  for (uint32_t i = 0; i < np; ++i) {
    const uint32_t gp = pending[2 * i], seg = pending[2 * i + 1];
    const int64_t id = A.ids[gp];
    uint32_t r;  // (only lane 0's value is read, after the search: not merged with a constant on purpose,
                 // slowpath_role)
    if (lane == 0) r = static_cast<uint32_t>(atomicAdd(&tv.ctr->alloc, (1ull << 32) | 1ull));
    const long long pos = wave_insert_slot(tv.buckets, tv.hp, id, q, path, lane);
    if (lane == 0) {
      if (pos >= 0) {
        Bucket* b = tv.buckets + (pos >> 2);
        b->row[pos & 3] = r;
        b->ts[pos & 3] = a.ts;
      } else {
        atomicAdd(&tv.ctr->alloc, ~((1ull << 32) - 1ull));
        atomicOr(&tv.ctr->error, 1u);
        atomicAdd(&tv.ctr->n_dropped, 1u);
      }
    }
    r = __shfl(r, 0);
    if (pos >= 0) {
      const float* values = A.grads + size_t(A.grad_off[seg]);
      const bool grp = (A.g[t] & kShapeGroupBit) != 0u;   // (rare path: both forms in one kernel)
      if (A.g[t] & 1u) {
        if (grp) apply_row<64, 1, kOpOptimize, false, true>(tv, row_ptr(tv, r), true, lane, values, nullptr, 0u,
                                                            1u, int64_t(gp - A.id_off[seg]), a);
        else apply_row<64, 1, kOpOptimize, false, false>(tv, row_ptr(tv, r), true, lane, values, nullptr, 0u,
                                                         1u, int64_t(gp - A.id_off[seg]), a);
      } else {
        if (grp) apply_row<64, 4, kOpOptimize, false, true>(tv, row_ptr(tv, r), true, lane, values, nullptr, 0u,
                                                            1u, int64_t(gp - A.id_off[seg]), a);
        else apply_row<64, 4, kOpOptimize, false, false>(tv, row_ptr(tv, r), true, lane, values, nullptr, 0u,
                                                         1u, int64_t(gp - A.id_off[seg]), a);
      }
    }
    __syncthreads();
  }
  if (lane == 0) tv.ctr->n_pending = 0;
}

}  // namespace mhte
#endif  // MHTE_MSTEP_KERNELS_H_
