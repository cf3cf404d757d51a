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

struct MStepStatic {
  RunView rv[2];           // run-dedup workspaces: slot s holds the batch deduplicated into it
                           // (ids / n / nblk are per launch: kernel arguments)
  float* grad_u;           // [n_max, dim] summed gradients of ids left to the displacement pass
  uint32_t* pending;       // [n_max + 1]
  float* part[2];          // per slot: partial rows of multi-item lists
  uint32_t* arrive[2];     // per slot: arrival counters, kept zeroed
  int64_t n_max;           // capacity of the dense arrays (largest batch of the table)
  uint32_t g;              // lanes per id: 8 / 16 / 32 / 64 >= dim / 4
  uint32_t oneseg;         // the table has one segment (scalar descriptor loads, seg_of)
  uint32_t nblk_build;     // workgroups of the build role (a function of the scratch capacity)
  uint32_t count_hits;
};
typedef const MHTE_CONST MStepStatic* ConstStatics;

struct MFwdTab {
  uint32_t id_off, n;          // this batch: ids[id_off, id_off + n)
  uint32_t next_off, n_next;   // next batch (n_next = 0: none)
  uint32_t emb_off;            // floats
};
struct MFwdArgs {
  ConstViews views;
  ConstStatics st;
  const int64_t* ids;
  const int64_t* ids_next;
  float* out;
  uint32_t cur;                // slot of the batch being trained; the next one dedups into cur ^ 1
  uint32_t pad;
  MFwdTab tab[kMaxStepTables];
};

struct MBwdTab {
  uint32_t grad_off;           // floats
  uint32_t apply;              // 0: nothing to apply for this table (empty batch / build only)
  uint32_t nblk_items, nblk_ids;
  uint32_t build_next;         // 1: slot cur ^ 1 holds a deduplicated batch to number
  uint32_t light_max;
  ApplyArgs a;
};
struct MBwdArgs {
  ConstViews views;
  ConstStatics st;
  const float* grads;
  uint32_t cur;
  uint32_t pad;
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
      d.hkey[i] = kEmptyKey;
      d.hcnt[i] = 0;
      d.hblk[i] = 0ull;
      d.hpos[i] = 0;
    }
    if (i < 4) d.ctr[i] = 0;
  }
}

// ---------------------------------------------------------------------------------------------
// forward: per table   run dedup of the NEXT batch | lookup of this batch
// (the displacement pass of the previous update has its own launch here, mstep_slow_kernel: with
// T tables in a launch a 3 us launch is noise, and the lookups need no gate)
// ---------------------------------------------------------------------------------------------
template <int G, int UNR>
__device__ __forceinline__ void mstep_lookup_loop(const TableView& tv, const int64_t* ids, int64_t n,
                                                  float* out, int count_hits, uint32_t bid,
                                                  uint32_t nblk_l) {
  const int64_t ngroups = (n + UNR - 1) / UNR;
#pragma unroll 1
  for (int64_t g = (int64_t(bid) * kRdBlock + threadIdx.x) / G; g < ngroups;
       g += int64_t(nblk_l) * kRdBlock / G)
    lookup_role_u<G, 4, UNR, true>(tv, ids, n, nullptr, out, count_hits, g);
}

template <int UNR>
__global__ __launch_bounds__(kRdBlock, 8) __attribute__((amdgpu_num_sgpr(80))) void mstep_fwd_kernel(
    MFwdArgs A) {
  __shared__ __attribute__((aligned(16))) RdLds L;
  const uint32_t t = blockIdx.y;
  const MFwdTab ft = A.tab[t];
  const MStepStatic& s = deref_const(A.st + t);
  WaveTrace wt(nullptr);
  uint32_t bid = blockIdx.x;
  const uint32_t nblk_d = (ft.n_next + kRdBlock - 1) / kRdBlock;
  if (bid < nblk_d) {
    RunView d = s.rv[(A.cur ^ 1u) & 1u];
    d.ids = A.ids_next + ft.next_off;
    d.n = ft.n_next;
    d.nblk = nblk_d;
    rd_dedup_role(d, bid, L, wt);
    return;
  }
  bid -= nblk_d;
  if (ft.n == 0 || gridDim.x <= nblk_d) return;
  const uint32_t nblk_l = gridDim.x - nblk_d;
  const TableView& tv = deref_const(A.views + t);
  const int64_t* ids = A.ids + ft.id_off;
  float* out = A.out + size_t(ft.emb_off);
  const int ch = int(s.count_hits);
  switch (s.g) {
    case 8: mstep_lookup_loop<8, UNR>(tv, ids, ft.n, out, ch, bid, nblk_l); break;
    case 16: mstep_lookup_loop<16, UNR>(tv, ids, ft.n, out, ch, bid, nblk_l); break;
    case 32: mstep_lookup_loop<32, UNR>(tv, ids, ft.n, out, ch, bid, nblk_l); break;
    default: mstep_lookup_loop<64, UNR>(tv, ids, ft.n, out, ch, bid, nblk_l); break;
  }
}

// ---------------------------------------------------------------------------------------------
// backward: per table   numbering + heavy work list of the NEXT batch | apply of this batch
// ---------------------------------------------------------------------------------------------
template <bool ONESEG>
__device__ __forceinline__ void mstep_apply_switch(uint32_t g, const TableView& tv, const RunView& d,
                                                   const ApplyCtl& c, const ApplyArgs& a,
                                                   uint32_t bid, WaveTrace& wt, ApplyLds& L) {
  switch (g) {
    case 8: rd_apply_role<8, 4, ONESEG>(tv, d, c, a, bid, wt, L); break;
    case 16: rd_apply_role<16, 4, ONESEG>(tv, d, c, a, bid, wt, L); break;
    case 32: rd_apply_role<32, 4, ONESEG>(tv, d, c, a, bid, wt, L); break;
    default: rd_apply_role<64, 4, ONESEG>(tv, d, c, a, bid, wt, L); break;
  }
}

__global__ __launch_bounds__(256, kBwdBlocksPerCu) void mstep_bwd_kernel(MBwdArgs A) {
  __shared__ ApplyLds L;
  const uint32_t t = blockIdx.y;
  const MBwdTab& bt = A.tab[t];
  const MStepStatic& s = deref_const(A.st + t);
  WaveTrace wt(nullptr);
  uint32_t bid = blockIdx.x;
  const uint32_t cur = A.cur & 1u;
  const uint32_t nblk_build = bt.build_next ? s.nblk_build : 0u;
  if (bid < nblk_build) {
    const RunView nxt = s.rv[cur ^ 1u];
    rd_build_role(nxt, uint32_t(kStepLightMax), bid, nblk_build);
    return;
  }
  bid -= nblk_build;
  if (!bt.apply || bid >= bt.nblk_items + bt.nblk_ids) return;
  const TableView& tv = deref_const(A.views + t);
  const RunView d = s.rv[cur];
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
  c.spec_row = nullptr;
  if (s.oneseg) mstep_apply_switch<true>(s.g, tv, d, c, bt.a, bid, wt, L);
  else mstep_apply_switch<false>(s.g, tv, d, c, bt.a, bid, wt, L);
}

// displacement pass of every table's update, one wavefront per table (usually nothing to do)
__global__ __launch_bounds__(64) void mstep_slow_kernel(MBwdArgs A) {
  __shared__ BfsSlot q[kMaxCuckooCount];
  __shared__ CuckooRecord path[kMaxBfsPathLen];
  const uint32_t t = blockIdx.x;
  const MBwdTab& bt = A.tab[t];
  if (!bt.apply) return;
  const MStepStatic& s = deref_const(A.st + t);
  const TableView& tv = deref_const(A.views + t);
  slowpath_role<4, kOpOptimize, true>(tv, s.rv[A.cur & 1u].uids, s.grad_u, nullptr, nullptr, bt.a,
                                      nullptr, s.pending, q, path);
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
  uint8_t g[kMaxStepTables * 4];      // per table: lanes per id
  uint8_t count_hits[kMaxStepTables * 4];
};
static_assert(sizeof(SegLookupArgs) <= 4096, "kernel arguments exceed 4 KB");

template <int G>
__device__ __forceinline__ void seg_lookup_loop(const TableView& tv, const int64_t* ids, int64_t n,
                                                float* out, int count_hits) {
  const int64_t ngroups = (n + 1) / 2;
#pragma unroll 1
  for (int64_t g = (int64_t(blockIdx.x) * 512 + threadIdx.x) / G; g < ngroups;
       g += int64_t(gridDim.x) * 512 / G)
    lookup_role_u<G, 4, 2, true>(tv, ids, n, nullptr, out, count_hits, g);
}

__global__ __launch_bounds__(512) void seg_lookup_kernel(SegLookupArgs A) {
  const uint32_t y = blockIdx.y;
  const uint32_t n = A.id_off[y + 1] - A.id_off[y];
  if (n == 0) return;
  const uint32_t t = (A.seg0 + y) % A.T;
  const TableView& tv = deref_const(A.views + t);
  const int64_t* ids = A.ids + A.id_off[y];
  float* out = A.out + size_t(A.emb_off[y]);
  const int ch = A.count_hits[t];
  switch (A.g[t]) {
    case 8: seg_lookup_loop<8>(tv, ids, n, out, ch); break;
    case 16: seg_lookup_loop<16>(tv, ids, n, out, ch); break;
    case 32: seg_lookup_loop<32>(tv, ids, n, out, ch); break;
    default: seg_lookup_loop<64>(tv, ids, n, out, ch); break;
  }
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

template <int G>
__device__ __forceinline__ void seg_upsert_loop(const TableView& tv, const int64_t* ids, uint32_t n,
                                                const float* values, const ApplyArgs& a,
                                                uint32_t* pending, uint32_t id_base, uint32_t seg) {
  const int lane = threadIdx.x & 63;
  const int j = lane & (G - 1);
  const uint32_t ngroups_wg = 256 / G;
#pragma unroll 1
  for (uint32_t g0 = blockIdx.x * ngroups_wg; g0 < n; g0 += gridDim.x * ngroups_wg) {  // wave-uniform
    const uint32_t g = g0 + threadIdx.x / G;
    const bool valid = g < n;
    const int64_t id = valid ? ids[g] : 0;
    Probe<G> pr = probe_issue<G>(tv, id, valid, j);
    const SlotResult sr = upsert_resolve<G>(tv, (Bucket*)pr.b, id, valid, pr.k, pr.row, lane, a.ts);
    if (sr.deferred && j == 0) {
      const uint32_t slot = atomicAdd(&tv.ctr->n_pending, 1u);
      pending[2 * slot] = id_base + g;
      pending[2 * slot + 1] = seg;
    }
    if (valid && !sr.deferred)
      apply_row<G, 4, kOpOptimize, false, false>(tv, row_ptr(tv, sr.r), sr.is_new, j, values, nullptr,
                                                 0u, 1u, int64_t(g), a);
  }
}

__global__ __launch_bounds__(256) void seg_upsert_kernel(SegUpsertArgs A) {
  const uint32_t y = blockIdx.y;
  const uint32_t n = A.id_off[y + 1] - A.id_off[y];
  if (n == 0) return;
  const uint32_t t = (A.seg0 + y) % A.T;
  const TableView& tv = deref_const(A.views + t);
  const int64_t* ids = A.ids + A.id_off[y];
  const float* values = A.grads + size_t(A.grad_off[y]);
  uint32_t* pend = A.pending[t];
  switch (A.g[t]) {
    case 8: seg_upsert_loop<8>(tv, ids, n, values, A.a[t], pend, A.id_off[y], y); break;
    case 16: seg_upsert_loop<16>(tv, ids, n, values, A.a[t], pend, A.id_off[y], y); break;
    case 32: seg_upsert_loop<32>(tv, ids, n, values, A.a[t], pend, A.id_off[y], y); break;
    default: seg_upsert_loop<64>(tv, ids, n, values, A.a[t], pend, A.id_off[y], y); break;
  }
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
    uint32_t r;
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
      apply_row<64, 4, kOpOptimize, false, false>(tv, row_ptr(tv, r), true, lane, values, nullptr, 0u,
                                                  1u, int64_t(gp - A.id_off[seg]), a);
    }
    __syncthreads();
  }
  if (lane == 0) tv.ctr->n_pending = 0;
}

}  // namespace mhte
#endif  // MHTE_MSTEP_KERNELS_H_
