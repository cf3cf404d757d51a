// Kernels of the id-sharded multi-table step (host: mhte_shard_host.h): every table of the model in
// ONE exchange per direction, as the reference's sync-training path moves them
// (NT/distributed_ps_sync.py:95-287 lookup, :289-490 apply_gradients; shard = floormod(id, N),
// NT/distributed_ps.py:289; packing RT/ops/fused_reorder_by_indices.cc:75-123).
//
// Wire format.  Rank r keeps, for every peer p, one fixed-capacity block in each direction:
//   id block    int64[hdr + sum_t cap_t]   word t (< T) = number of ids of table t in the block (the
//                                          exchange carries its own counts: no size all-to-all, no
//                                          D2H), then per table cap_t id slots at id_off[t]
//   row block   float[sum_t cap_t * dim_t] per table cap_t rows at row_off[t]; slot s of the id
//                                          block's table t <-> row s — lookups come back and
//                                          gradients leave in the same slots
// Buffers hold N blocks back to back, peer-major, so an exchange is N fixed-size send / recv pairs.
//
//   owner   shard_lookup   the received id blocks' rows (no insert) into the row blocks
//   sender  shard_scatter  rows back -> every occurrence (rd_gather_role)
//   sender  shard_build    numbering of the NEXT batch's distinct ids, each also packed into its
//                          owner's id block (PackCtl in rd_build_role: slot_off[u] = float offset of
//                          u's row slot in the row buffers)  |  per-id gradient sums of this batch
//                          (same order as the single-GPU step) -> row slots
//   owner   shard_upsert   one peer's block: probe / insert / optimizer per id (ids of a block are
//                          distinct), then shard_slow: the displacement pass; the peers are
//                          applied one after the other in rank order — the reference's N separate
//                          optimizer applications
#ifndef MHTE_SHARD_KERNELS_H_
#define MHTE_SHARD_KERNELS_H_

namespace mhte {

struct ShardTab {
  uint32_t cap;       // id / row slots per peer block
  uint32_t id_off;    // int64 words from the start of a block (header included)
  uint32_t row_off;   // floats from the start of a row block
  uint32_t dim;
};

struct ShardGeom {
  uint32_t world;
  uint32_t T;
  uint32_t ids_block;    // int64 words per peer block
  uint32_t rows_block;   // floats per peer block
};

enum : uint32_t { kShardOverflow = 1u };

// ---- owner: what it keeps about a received id between its lookup (forward) and its update (backward) --
// The update of a step works on the SAME id blocks its lookup read, so the lookup leaves, per entry
// (peer, table, slot) — indexed like the id inside the received id buffer, p * ids_block + id_off + s:
//   * OwnRec: where the probe found the id (row handle, bucket slot, the slot's timestamp).  While
//     Table::mut_epoch stands still the update reaches the row without reading a bucket line, and
//     skips the timestamp store when the slot already carries the step's second (the multi-table
//     step's forward -> backward hints, mhte_mstep_kernels.h).
//   * world > 1: the id's slot in a per-table scratch hash that groups the entries of one id across
//     the peers' blocks (an id occurs at most once per block).  Slot = key (2 words) | peer mask (2)
//     | ent[world] (slot + 1 of the id in that peer's block).  ONE lane group — the lowest peer's —
//     then applies every sender's gradient in rank order (the reference's one optimizer application
//     per sender, NT/distributed_ps_sync.py:357-479) and hands the slot back empty, so the owner's
//     update is ONE launch for all peers (the shape of MonolithMultiHashTableFusedOptimize,
//     RT/ops/multi_hash_table_update_op.cc:247-308).  The peer mask is read with one 8-byte load: a
//     group sees the id's entries either all registered or already consumed, never half.
struct __attribute__((aligned(16))) OwnRec {
  unsigned long long loc;   // bucket * 4 + slot
  uint32_t row;             // kNoRow: the table did not hold the id (or it is the side-slot key)
  uint32_t ts;
};
struct ShardX {
  OwnRec* orec;             // [world][ids_block]
  uint32_t* oslot;          // [world][ids_block] (world > 1)
  uint32_t* xs;             // [T][xmask + 2] slots of xstride words (world > 1; the last slot of a table:
                            // the key that marks an empty slot itself)
  uint32_t xmask;
  uint32_t xstride;         // words per slot: 4 + world rounded up to a multiple of 4
  uint32_t hints;           // bit i: OwnRec of the launch's i-th table may be trusted (nothing touched the
                            // table since the lookup wrote them)
  uint32_t pad;
};
constexpr uint32_t kXKeyWords = 4;   // key + peer mask in front of a slot's entries

__global__ __launch_bounds__(256) void shard_x_clear_kernel(uint32_t* xs, uint64_t nslots, uint32_t xstride) {
  for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nslots;
       i += uint64_t(gridDim.x) * blockDim.x) {
    uint32_t* sl = xs + i * xstride;
    *reinterpret_cast<int64_t*>(sl) = kEmptyKey;
    for (uint32_t w = 2; w < xstride; ++w) sl[w] = 0u;
  }
}

// ---- owner: lookup of the received blocks ---------------------------------------------------------
struct ShardOwnerArgs {
  ConstViews views;
  ShardGeom geo;
  ShardX x;
  const int64_t* recv_ids;    // [world][ids_block]
  float* rows;                // lookup: out [world][rows_block]; upsert: gradients in
  uint32_t* flags;
  uint32_t peer;              // upsert / slow: the block being applied
  uint32_t zero_headers;      // slow: 1 = clear the headers of `clear_ids` when done
  int64_t* clear_ids;
  // the lookup launch that also runs the displacement pass of the PREVIOUS owner update (slow_on): that
  // update's id blocks and gradient blocks; pending[] / a[] / g[] below are that update's too
  const int64_t* slow_ids;
  const float* slow_rows;
  uint32_t slow_on;
  uint32_t pad0;
  // lookup with direct peer stores: the rows of peer p's ids go straight into p's window — peer_win[p] +
  // peer_rows_off (this rank's block of the row buffer there) — instead of rows + p * rows_block
  const unsigned long long* peer_win;   // [world] device array; nullptr: `rows`
  unsigned long long peer_rows_off;
  // a launch serves tables [t0, t0 + tc) of the model (the host chunks: kernel-argument budget); the
  // arrays below are indexed by the table's position in the launch, `views`, the header words and
  // the per-table device arrays by its index in the model
  uint32_t t0, tc;
  uint32_t* pending[kMaxStepTables];
  ShardTab tab[kMaxStepTables];
  uint8_t g[kMaxStepTables];          // lane-group shape per table (MHTE_SWITCH_G; | kShapeGroupBit)
  uint8_t count_hits[kMaxStepTables];
  uint8_t fast[kMaxStepTables];       // 1: SGD / Adagrad / FTRL rows — shard_apply_kernel's FAST
                                      // instance (the step kernels' register-resident update) serves the table
  ApplyArgs a[kMaxStepTables];
};
static_assert(sizeof(ShardOwnerArgs) <= 4096, "kernel arguments exceed 4 KB");

// (t: position of the table in the launch)
__device__ __forceinline__ uint32_t shard_block_count(const ShardOwnerArgs& A, uint32_t p, uint32_t t) {
  const uint64_t c = uint64_t(A.recv_ids[size_t(p) * A.geo.ids_block + A.t0 + t]);
  if (c > A.tab[t].cap) {
    // the sender saw the same count and dropped the surplus ids' rows; flagged on both sides
    if (threadIdx.x == 0) atomicOr(A.flags, uint32_t(kShardOverflow));
    return A.tab[t].cap;
  }
  return uint32_t(c);
}

template <bool GATED>
__device__ __forceinline__ void shard_slow_all_role(const ShardOwnerArgs& A, uint32_t t, const int64_t* recv_ids,
                                                    const float* rows, BfsSlot* q, CuckooRecord* path, int lane);

// Lookup of one (peer, table) segment: two ids per lane group, all probes, then all row loads in flight
// (lookup_role_u), + what the update will want to know (OwnRec) + (xs != nullptr: world > 1) the
// registration of the id in the cross-peer scratch.  The registration's claim is issued behind the row
// loads, so the rows leave while it is in flight.
template <int G, int VEC, int UNR>
__device__ __forceinline__ void shard_lookup_loop(const ShardOwnerArgs& A, uint32_t p, uint32_t t,
                                                  const TableView& tv, const int64_t* __restrict__ ids, uint32_t cap,
                                                  float* __restrict__ out, int count_hits,
                                                  OwnRec* __restrict__ orec, uint32_t* __restrict__ oslot,
                                                  uint32_t* __restrict__ xs, uint32_t xmask, uint32_t xstride,
                                                  uint32_t /*p again*/) {
  const int lane = threadIdx.x & 63;
  const int j = lane & (G - 1);
  const int gbase = lane & ~(G - 1);
  // (the block's count is fetched WITH the first trip's ids — every index below the block's capacity
  // is memory that exists — instead of in front of them: one dependent round trip less on a chain of four)
  uint32_t n = cap;
  bool have_n = false;
  uint64_t hits = 0;
#pragma unroll 1
  for (uint32_t grp = (blockIdx.x * 512u + threadIdx.x) / G; grp < (n + UNR - 1u) / UNR; grp += gridDim.x * 512u / G) {
    const uint32_t g0 = grp * UNR;
    int64_t myid = (j < UNR && g0 + uint32_t(j) < cap) ? ids[g0 + j] : 0;
    if (!have_n) {
      n = shard_block_count(A, p, t);
      have_n = true;
      if (A.slow_on) {
        // The displacement pass of the previous owner update runs in this launch (the first workgroups):
        // no word of the table is read before it has finished — n_pending drops to 0 only after its
        // stores were written back (release), and every load below is control-dependent on having seen
        // the 0 (lookup_role's gate).  Usually the list is empty and this is one load beside the ids'.
        // (the first look is a plain load: the L2s are clean at a launch's start, the count only falls
        // inside it, so a 0 read now is the truth — and a plain load is a round trip shorter than an atomic)
        if (*reinterpret_cast<const volatile unsigned int*>(&tv.ctr->n_pending) != 0u) {
          while (__hip_atomic_load(&tv.ctr->n_pending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
            __builtin_amdgcn_s_sleep(8);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
      }
    }
    const bool mine_valid = j < UNR && g0 + uint32_t(j) < n;
    if (!mine_valid) myid = 0;
    int64_t id[UNR], kk[UNR];
    bool valid[UNR];
    uint32_t row[UNR], tsv[UNR];
    uint64_t i1[UNR], i2[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      id[u] = __shfl(myid, gbase + u);
      valid[u] = g0 + u < n;
      const uint64_t hv = hash_key(id[u]);
      i1[u] = index_hash(tv.hp, hv);
      i2[u] = alt_index(tv.hp, partial_key(hv), i1[u]);
      const GBucket* b = global_bucket(tv.buckets + ((j & 4) ? i2[u] : i1[u]));
      kk[u] = b->key[j & 3];
      row[u] = b->row[j & 3];
      tsv[u] = b->ts[j & 3];   // (same 64-byte line)
    }
    bool found[UNR];
    const float* rp[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const bool use = valid[u] && j < 8 && id[u] != kEmptyKey;
      const uint64_t m = group_mask_of<G>(__ballot(use && kk[u] == id[u]), gbase);
      found[u] = m != 0;
      const int src = found[u] ? (__ffsll(static_cast<long long>(m)) - 1) : 0;
      uint32_t r = __shfl(use ? row[u] : kNoRow, gbase + src);
      const uint32_t ots = __shfl(tsv[u], gbase + src);
      const bool special = valid[u] && id[u] == kEmptyKey;
      if (special) {
        found[u] = tv.ctr->special_state == 1;
        r = tv.ctr->special_row;
      }
      found[u] = found[u] && valid[u];
      if (valid[u] && j == u) {   // (the side slot's key is left to the update's own path)
        OwnRec rec;
        rec.loc = (((src & 4) ? i2[u] : i1[u]) << 2) | uint64_t(src & 3);
        rec.row = (found[u] && !special) ? r : kNoRow;
        rec.ts = ots;
        orec[g0 + u] = rec;
      }
      rp[u] = found[u] ? row_ptr(tv, r) : nullptr;
      if (count_hits) hits += __popcll(__ballot(found[u] && j == 0));
    }
    for (uint32_t e = j * VEC; e < tv.dim; e += G * VEC) {
      Vec<VEC> v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        vec_zero(v[u]);
        if (found[u]) v[u].load(rp[u] + e);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u)
        if (valid[u]) store_stream<VEC>(out + size_t(g0 + u) * tv.dim + e, v[u]);
    }
    if (xs && mine_valid) {   // lane u of the group registers id u
      uint32_t h = xmask + 1u;
      if (myid != kEmptyKey) {
        h = uint32_t(hash_key(myid) >> 20) & xmask;
        for (;;) {
          const unsigned long long old =
              atomicCAS(reinterpret_cast<unsigned long long*>(xs + size_t(h) * xstride),
                        static_cast<unsigned long long>(kEmptyKey), static_cast<unsigned long long>(myid));
          if (old == static_cast<unsigned long long>(kEmptyKey) || old == static_cast<unsigned long long>(myid)) break;
          h = (h + 1u) & xmask;
        }
      }
      uint32_t* sl = xs + size_t(h) * xstride;
      atomicOr(reinterpret_cast<unsigned long long*>(sl + 2), 1ull << p);
      sl[kXKeyWords + p] = g0 + uint32_t(j) + 1u;
      oslot[g0 + j] = h;
    }
  }
  if (count_hits && hits && lane == __ffsll(static_cast<long long>(__ballot(1))) - 1)
    atomicAdd(&tv.ctr->hits, (unsigned long long)hits);
}

// grid (x, world * tc + slow_on): y = peer * tc + table of the launch (+ 1 with slow_on: row 0 — dispatched
// first — is the displacement pass of the previous owner update, one wavefront per table; A.g / A.a /
// A.pending describe that update's tables, which are this launch's)
// SLOW: the instance that can carry the owed pass (its update code costs the lookups registers: 74 VGPRs
// against 52 — the host folds the pass in only where a launch of its own would show, a model of a few tables)
// UNR: ids per lane group in flight (2: one table's launch, where the chain's length is what counts; 4: a
// model's — half the workgroups to dispatch for the same rows)
template <int VW, bool SLOW = false, int UNR = 2>
__global__ __launch_bounds__(512) void shard_lookup_kernel(ShardOwnerArgs A) {
  __shared__ BfsSlot sq[SLOW ? kMaxCuckooCount : 1];
  __shared__ CuckooRecord spath[SLOW ? kMaxBfsPathLen : 1];
  if (SLOW && A.slow_on && blockIdx.y == 0) {
    // (the float4 instance runs the pass for every table: the role picks the lane width per table)
    if (VW == 4 && blockIdx.x < A.tc && threadIdx.x < 64)
      shard_slow_all_role<true>(A, blockIdx.x, A.slow_ids, A.slow_rows, sq, spath, int(threadIdx.x));
    return;
  }
  const uint32_t y = blockIdx.y - ((SLOW && A.slow_on) ? 1u : 0u);
  const uint32_t p = y / A.tc, t = y % A.tc;
  if (!MHTE_SHAPE_IS(VW, A.g[t])) return;
  const ShardTab tb = A.tab[t];
  const TableView& tv = deref_const(A.views + (A.t0 + t));
  const size_t eb = size_t(p) * A.geo.ids_block + tb.id_off;
  const int64_t* ids = A.recv_ids + eb;
  float* out = (A.peer_win ? reinterpret_cast<float*>(A.peer_win[p] + A.peer_rows_off)
                           : A.rows + size_t(p) * A.geo.rows_block) + tb.row_off;
  const int ch = A.count_hits[t];
  uint32_t* xs = A.x.xs ? A.x.xs + size_t(A.t0 + t) * (size_t(A.x.xmask) + 2u) * A.x.xstride : nullptr;
  uint32_t* oslot = A.x.oslot ? A.x.oslot + eb : nullptr;
#define MHTE_SEGL_CALL(G_, V_) \
  shard_lookup_loop<G_, V_, UNR>(A, p, t, tv, ids, tb.cap, out, ch, A.x.orec + eb, oslot, xs, A.x.xmask, A.x.xstride, p)
  MHTE_SWITCH_G(VW, A.g[t] & ~kShapeGroupBit, MHTE_SEGL_CALL)
#undef MHTE_SEGL_CALL
}

// displacement pass of one peer's block for table t, by one wavefront (lane 0 alone touches q, path
// and the buckets; the waits order its stores before the wavefront's next loads)
__device__ __forceinline__ void shard_slow_role(const ShardOwnerArgs& A, uint32_t t, BfsSlot* q,
                                                CuckooRecord* path, int lane) {
  const TableView& tv = deref_const(A.views + (A.t0 + t));
  const uint32_t np = tv.ctr->n_pending;
  if (!np) return;
  const ShardTab tb = A.tab[t];
  const int64_t* ids = A.recv_ids + size_t(A.peer) * A.geo.ids_block + tb.id_off;
  const float* values = A.rows + size_t(A.peer) * A.geo.rows_block + tb.row_off;
  const uint32_t* pending = A.pending[t];
  const ApplyArgs& a = A.a[t];
  for (uint32_t i = 0; i < np; ++i) {
    const uint32_t g = pending[2 * i];
    const int64_t id = ids[g];
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
      const bool grp = (A.g[t] & kShapeGroupBit) != 0u;   // (rare path: both forms in one kernel)
      if (A.g[t] & 1u) {
        if (grp) apply_row<64, 1, kOpOptimize, false, true>(tv, row_ptr(tv, r), true, lane, values, nullptr, 0u,
                                                            1u, int64_t(g), a);
        else apply_row<64, 1, kOpOptimize, false, false>(tv, row_ptr(tv, r), true, lane, values, nullptr, 0u,
                                                         1u, int64_t(g), a);
      } else {
        if (grp) apply_row<64, 4, kOpOptimize, false, true>(tv, row_ptr(tv, r), true, lane, values, nullptr, 0u,
                                                            1u, int64_t(g), a);
        else apply_row<64, 4, kOpOptimize, false, false>(tv, row_ptr(tv, r), true, lane, values, nullptr, 0u,
                                                         1u, int64_t(g), a);
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  if (lane == 0) tv.ctr->n_pending = 0;
}

// grid (x, tc): the block of peer A.peer.  Ids whose two buckets are full go to the table's pending
// list, finished by shard_slow_kernel.  (A launch of its own, not the last workgroup of this one: on
// the 8-XCD part the other workgroups' bucket and pending-list stores sit in their XCDs' L2s until
// the kernel ends — making them visible earlier is an L2 write-back per workgroup, measured 295 us
// against 5 us for the extra launch.)
template <int VW, bool GROUP = false>
__global__ __launch_bounds__(256, GROUP ? 1 : MHTE_SEGU_OCC) void shard_upsert_kernel(ShardOwnerArgs A) {
  const uint32_t p = A.peer, t = blockIdx.y;
  if (!MHTE_SHAPE_IS(VW, A.g[t]) || ((A.g[t] & kShapeGroupBit) != 0u) != GROUP) return;
  const uint32_t n = shard_block_count(A, p, t);
  if (n == 0) return;
  const ShardTab tb = A.tab[t];
  const TableView& tv = deref_const(A.views + (A.t0 + t));
  const int64_t* ids = A.recv_ids + size_t(p) * A.geo.ids_block + tb.id_off;
  const float* values = A.rows + size_t(p) * A.geo.rows_block + tb.row_off;
  uint32_t* pend = A.pending[t];
  // (pending entry = (position in the block's table segment, unused))
#define MHTE_SEGU_CALL(G_, V_) seg_upsert_loop<G_, V_, GROUP>(tv, ids, n, values, A.a[t], pend, 0u, 0u)
  MHTE_SWITCH_G(VW, A.g[t] & ~kShapeGroupBit, MHTE_SEGU_CALL)
#undef MHTE_SEGU_CALL
}

// displacement pass of one peer's block, one wavefront per table; the last one of a step also
// clears the headers of the step's send blocks (the next numbering into them counts from zero)
__global__ __launch_bounds__(64) void shard_slow_kernel(ShardOwnerArgs A) {
  __shared__ BfsSlot q[kMaxCuckooCount];
  __shared__ CuckooRecord path[kMaxBfsPathLen];
  const uint32_t t = blockIdx.x;
  const int lane = threadIdx.x;
  shard_slow_role(A, t, q, path, lane);
  if (A.zero_headers && uint32_t(lane) < A.geo.world)
    A.clear_ids[size_t(lane) * A.geo.ids_block + A.t0 + t] = 0;
}

// ---- owner: ONE launch applies every peer's gradient block -----------------------------------------
// grid (x, world * tc): y = peer * tc + table of the launch, a lane group per (peer, slot).  The group of
// the LOWEST peer that sent an id applies all of the id's entries, one optimizer step per sender in rank
// order, and hands the cross-peer slot back empty; the other peers' groups of that id leave.  The row
// is reached through the lookup's record when it can be trusted (no bucket line is read), otherwise
// by probe / insert as seg_upsert_loop does.  Ids whose two buckets are full go to the table's pending
// list as (slot, peer); shard_slow_all_kernel finishes them.
// (the cross-peer form at three wavefronts per SIMD: at four it spills 38 registers)
#ifndef MHTE_SEGX_OCC
#define MHTE_SEGX_OCC 3
#endif
template <int G, int VEC, bool GROUP, bool MULTI>
__device__ __forceinline__ void shard_apply_loop(const ShardOwnerArgs& A, const TableView& tv, uint32_t p,
                                                 uint32_t t) {
  const ShardTab tb = A.tab[t];
  const int lane = threadIdx.x & 63;
  const int j = lane & (G - 1);
  const int gbase = lane & ~(G - 1);
  const uint32_t world = A.geo.world;
  constexpr bool multi = MULTI;   // (an instance of its own: the one-rank / no-duplicate form keeps its registers)
  const size_t eb = size_t(p) * A.geo.ids_block + tb.id_off;
  const int64_t* ids = A.recv_ids + eb;
  const OwnRec* orec = A.x.orec + eb;
  uint32_t* const xs = multi ? A.x.xs + size_t(A.t0 + t) * (size_t(A.x.xmask) + 2u) * A.x.xstride : nullptr;
  const bool trust = ((A.x.hints >> t) & 1u) != 0u;
  const ApplyArgs& a = A.a[t];
  const uint32_t ngroups_wg = 256 / G;
  // (the block's count arrives WITH the first trip's records, not in front of them: shard_lookup_loop)
  uint32_t n = tb.cap;
  bool have_n = false;
#pragma unroll 1
  for (uint32_t g0 = blockIdx.x * ngroups_wg; g0 < n; g0 += gridDim.x * ngroups_wg) {  // wave-uniform
    const uint32_t g = g0 + threadIdx.x / G;
    const uint32_t gs = g < tb.cap ? g : 0u;   // (loads from an index that exists, masked afterwards)
    const int64_t id = ids[gs];
    const OwnRec rec = orec[gs];
    uint32_t hs = 0;
    if (multi) hs = A.x.oslot[eb + gs];
    if (!have_n) {
      n = shard_block_count(A, p, t);
      have_n = true;
    }
    const bool valid = g < n;
    // ---- the id's entries (one per peer that sent it); who applies them
    unsigned long long pm = 1ull << p;
    bool mine = valid;
    if (multi) {
      const uint32_t* sl = xs + size_t(hs) * A.x.xstride;
      pm = *reinterpret_cast<const unsigned long long*>(sl + 2);   // one 8-byte load: all or nothing
      if (!valid) pm = 0ull;
      mine = ((pm >> p) & 1ull) != 0ull && (pm & ((1ull << p) - 1ull)) == 0ull;
    }
    // ---- the row: through the lookup's record, else probe / insert
    const bool hinted = mine && trust && rec.row != kNoRow;
    const bool need = mine && !hinted;
    uint32_t r = rec.row;
    bool is_new = false, deferred = false;
    {   // (every lane: wave ballots inside; a group that needs no probe issues no load — under `if (__any(need))`
        // the float4 instance spills six registers)
      const Probe<G> pr = probe_issue<G>(tv, id, need, j);
      const SlotResult sr = upsert_resolve<G>(tv, (Bucket*)pr.b, id, need, pr.k, pr.row, lane, a.ts);
      if (need) {
        r = sr.r;
        is_new = sr.is_new;
        deferred = sr.deferred;
      }
    }
    if (hinted && j == 0 && rec.ts != a.ts)   // SetTimestamp(update_time): the slot is known, its line is not read
      global_bucket(tv.buckets + (rec.loc >> 2))->ts[rec.loc & 3ull] = a.ts;
    if (deferred && j == 0) {
      const uint32_t slot = atomicAdd(&tv.ctr->n_pending, 1u);
      A.pending[t][2 * slot] = g;
      A.pending[t][2 * slot + 1] = p;
    }
    if (mine && !deferred) {
      float* rp = row_ptr(tv, r);
      // the senders, ascending: one optimizer step each.  The first is this group's own entry; the
      // others' slots are fetched when their turn comes (the slot's lines are in the L2 by then, and
      // nothing of them has to stay in registers across the row update)
      uint32_t q = p, sq = g;
#pragma unroll 1
      for (;;) {   // (group-uniform)
        const float* values = A.rows + size_t(q) * A.geo.rows_block + tb.row_off;
        apply_row<G, VEC, kOpOptimize, false, GROUP>(tv, rp, is_new, j, values, nullptr, 0u, 1u, int64_t(sq), a);
        is_new = false;
        if (!multi || q >= 63u) break;
        const uint32_t* sl = xs + size_t(hs) * A.x.xstride;
        const unsigned long long rest = *reinterpret_cast<const unsigned long long*>(sl + 2) & ~((2ull << q) - 1ull);
        if (!rest) break;
        q = uint32_t(__ffsll(static_cast<long long>(rest)) - 1);
        sq = sl[kXKeyWords + q] - 1u;
      }
      if (multi) {   // the slot goes back empty (its entries are consumed)
        uint32_t* sl = xs + size_t(hs) * A.x.xstride;
        for (uint32_t q = uint32_t(j); q < world; q += G) sl[kXKeyWords + q] = 0u;
        if (j == 0) {
          *reinterpret_cast<unsigned long long*>(sl + 2) = 0ull;
          *reinterpret_cast<int64_t*>(sl) = kEmptyKey;
        }
      }
    }
  }
}

// The same for a table of SGD / Adagrad / FTRL rows (the shapes of BASELINE's configs), with
// the update the step kernels use (optimize_row_pre: segment descriptor in scalar registers, row and
// gradient in vector registers).  The sender's gradient row is fetched WITH the id and the lookup's record
// (its address is the slot's), the row as soon as the record is there: two dependent round trips
// (id | record | gradient -> row -> store) where apply_row makes four (.. -> descriptor -> row | gradient).
template <int G, int VEC, bool MULTI, bool ONESEG>
__device__ __forceinline__ void shard_apply_fast_loop(const ShardOwnerArgs& A, const TableView& tv, uint32_t p,
                                                      uint32_t t) {
  const ShardTab tb = A.tab[t];
  const int lane = threadIdx.x & 63;
  const int j = lane & (G - 1);
  const uint32_t world = A.geo.world;
  constexpr bool multi = MULTI;
  const size_t eb = size_t(p) * A.geo.ids_block + tb.id_off;
  const int64_t* ids = A.recv_ids + eb;
  const OwnRec* orec = A.x.orec + eb;
  uint32_t* const xs = multi ? A.x.xs + size_t(A.t0 + t) * (size_t(A.x.xmask) + 2u) * A.x.xstride : nullptr;
  const bool trust = ((A.x.hints >> t) & 1u) != 0u;
  const ApplyArgs& a = A.a[t];
  const float* my_rows = A.rows + size_t(p) * A.geo.rows_block + tb.row_off;
  const uint32_t dim = tv.dim;
  const uint32_t e = uint32_t(j) * VEC;
  const bool ev = e < dim;
  const uint32_t ngroups_wg = 256 / G;
  uint32_t n = tb.cap;
  bool have_n = false;
#pragma unroll 1
  for (uint32_t g0 = blockIdx.x * ngroups_wg; g0 < n; g0 += gridDim.x * ngroups_wg) {  // wave-uniform
    const uint32_t g = g0 + threadIdx.x / G;
    const uint32_t gs = g < tb.cap ? g : 0u;   // (loads from an index that exists, masked afterwards)
    const int64_t id = ids[gs];
    const OwnRec rec = orec[gs];
    uint32_t hs = 0;
    if (multi) hs = A.x.oslot[eb + gs];
    Vec<VEC> gv;
    vec_zero(gv);
    if (ev) gv.load(my_rows + size_t(gs) * dim + e);
    if (!have_n) {
      n = shard_block_count(A, p, t);
      have_n = true;
    }
    const bool valid = g < n;
    unsigned long long pm = 1ull << p;
    bool mine = valid;
    if (multi) {
      pm = *reinterpret_cast<const unsigned long long*>(xs + size_t(hs) * A.x.xstride + 2);
      if (!valid) pm = 0ull;
      mine = ((pm >> p) & 1ull) != 0ull && (pm & ((1ull << p) - 1ull)) == 0ull;
    }
    const bool hinted = mine && trust && rec.row != kNoRow;
    const bool need = mine && !hinted;
    RowRegs<VEC> rr;
    vec_zero(rr.w);
    vec_zero(rr.s1);
    // (rows of the first slab are addressed without the slab table: rd_apply_role)
    const bool slab0 = (rec.row >> tv.chunk_shift) == 0u;
    if (hinted && slab0) row_prefetch<VEC, ONESEG>(tv, assume_global(tv.chunk0 + size_t(rec.row) * tv.row_floats), e, rr);
    if (__any(hinted && !slab0)) {
      if (hinted && !slab0) row_prefetch<VEC, ONESEG>(tv, row_ptr(tv, rec.row), e, rr);
    }
    uint32_t r = rec.row;
    bool is_new = false, deferred = false, found_now = false;
    {
      const Probe<G> pr = probe_issue<G>(tv, id, need, j);
      const SlotResult sr = upsert_resolve<G>(tv, (Bucket*)pr.b, id, need, pr.k, pr.row, lane, a.ts);
      if (need) {
        r = sr.r;
        is_new = sr.is_new;
        deferred = sr.deferred;
        found_now = !sr.is_new && !sr.deferred;
      }
    }
    if (hinted && j == 0 && rec.ts != a.ts)
      global_bucket(tv.buckets + (rec.loc >> 2))->ts[rec.loc & 3ull] = a.ts;
    if (deferred && j == 0) {
      const uint32_t slot = atomicAdd(&tv.ctr->n_pending, 1u);
      A.pending[t][2 * slot] = g;
      A.pending[t][2 * slot + 1] = p;
    }
    if (mine && !deferred) {
      float* rp = row_ptr(tv, r);
      if (found_now) row_prefetch<VEC, ONESEG>(tv, rp, e, rr);   // (a resident id the record did not cover)
      optimize_row_pre<VEC, ONESEG>(tv, rp, is_new, e, gv, a, rr);
      if (multi) {
        // the other senders of the id, ascending, one optimizer step each: their slots and gradients are
        // fetched when their turn comes, the row is read back through the L2
        uint32_t* sl = xs + size_t(hs) * A.x.xstride;
        unsigned long long rest = pm & ~((2ull << p) - 1ull);
        if (p >= 63u) rest = 0ull;
#pragma unroll 1
        while (rest) {   // (group-uniform)
          const uint32_t q = uint32_t(__ffsll(static_cast<long long>(rest)) - 1);
          rest &= rest - 1ull;
          const uint32_t sq = sl[kXKeyWords + q] - 1u;
          Vec<VEC> g2;
          vec_zero(g2);
          if (ev) g2.load(A.rows + size_t(q) * A.geo.rows_block + tb.row_off + size_t(sq) * dim + e);
          optimize_row_reg<VEC, ONESEG>(tv, rp, false, e, g2, a);
        }
        for (uint32_t q = uint32_t(j); q < world; q += G) sl[kXKeyWords + q] = 0u;
        if (j == 0) {
          *reinterpret_cast<unsigned long long*>(sl + 2) = 0ull;
          *reinterpret_cast<int64_t*>(sl) = kEmptyKey;
        }
      }
    }
  }
}

// MULTI: the world has more than one rank (the cross-peer scratch is consulted); FAST: the tables with
// A.fast set (shard_apply_fast_loop), else the others
template <int VW, bool GROUP = false, bool MULTI = false, bool FAST = false>
__global__ __launch_bounds__(256, GROUP ? 1 : ((MULTI && !FAST) ? MHTE_SEGX_OCC : MHTE_SEGU_OCC)) void shard_apply_kernel(
    ShardOwnerArgs A) {
  const uint32_t p = blockIdx.y / A.tc, t = blockIdx.y % A.tc;
  if (!MHTE_SHAPE_IS(VW, A.g[t]) || ((A.g[t] & kShapeGroupBit) != 0u) != GROUP || (A.fast[t] != 0) != FAST) return;
  const TableView& tv = deref_const(A.views + (A.t0 + t));
  if constexpr (FAST) {
    // (one segment — the usual row — reads its descriptor with scalar loads: seg_of)
#define MHTE_SEGF_CALL(G_, V_)                                                      \
  do {                                                                              \
    if (tv.nseg == 1u) shard_apply_fast_loop<G_, V_, MULTI, true>(A, tv, p, t);     \
    else shard_apply_fast_loop<G_, V_, MULTI, false>(A, tv, p, t);                  \
  } while (0)
    MHTE_SWITCH_G(VW, A.g[t] & ~kShapeGroupBit, MHTE_SEGF_CALL)
#undef MHTE_SEGF_CALL
  } else {
#define MHTE_SEGU_CALL(G_, V_) shard_apply_loop<G_, V_, GROUP, MULTI>(A, tv, p, t)
    MHTE_SWITCH_G(VW, A.g[t] & ~kShapeGroupBit, MHTE_SEGU_CALL)
#undef MHTE_SEGU_CALL
  }
}

// displacement pass behind shard_apply_kernel, one wavefront per table: every deferred id's entries in
// rank order, as the fast path would have applied them; then the headers of the step's send blocks are
// cleared (the next numbering into them counts from zero).  Runs as the last launch of a step
// (shard_slow_all_kernel) or — GATED — inside the NEXT owner lookup's launch, whose lookups of the table
// wait for n_pending == 0 (slowpath_role's protocol: agent-scope release of the pass's stores, then the 0).
template <bool GATED>
__device__ __forceinline__ void shard_slow_all_role(const ShardOwnerArgs& A, uint32_t t, const int64_t* recv_ids,
                                                    const float* rows, BfsSlot* q, CuckooRecord* path, int lane) {
  const TableView& tv = deref_const(A.views + (A.t0 + t));
  const uint32_t np = tv.ctr->n_pending;
  const ShardTab tb = A.tab[t];
  const ApplyArgs& a = A.a[t];
  const uint32_t world = A.geo.world;
  const bool multi = world > 1u;
  for (uint32_t i = 0; i < np; ++i) {
    const uint32_t g = A.pending[t][2 * i], p = A.pending[t][2 * i + 1];
    const size_t eb = size_t(p) * A.geo.ids_block + tb.id_off;
    const int64_t id = recv_ids[eb + g];
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
    unsigned long long pm = 1ull << p;
    uint32_t* sl = nullptr;
    if (multi) {
      sl = A.x.xs + (size_t(A.t0 + t) * (size_t(A.x.xmask) + 2u) + A.x.oslot[eb + g]) * A.x.xstride;
      pm = *reinterpret_cast<const unsigned long long*>(sl + 2);
    }
    bool fresh = true;
    while (pm) {
      const int k = __ffsll(static_cast<long long>(pm)) - 1;
      pm &= pm - 1ull;
      const uint32_t sq = multi ? sl[kXKeyWords + k] - 1u : g;
      const float* values = rows + size_t(k) * A.geo.rows_block + tb.row_off;
      if (pos >= 0) {
        if (GATED) {
          // (inside the lookup launch only tables of SGD / Adagrad / FTRL rows: the BASIC update code keeps
          // that launch's registers — with all twelve optimizers it went from 54 to 143 VGPRs)
          if (A.g[t] & 1u) apply_row<64, 1, kOpOptimize, true, false>(tv, row_ptr(tv, r), fresh, lane, values, nullptr,
                                                                      0u, 1u, int64_t(sq), a);
          else apply_row<64, 4, kOpOptimize, true, false>(tv, row_ptr(tv, r), fresh, lane, values, nullptr, 0u, 1u,
                                                          int64_t(sq), a);
        } else {
          const bool grp = (A.g[t] & kShapeGroupBit) != 0u;   // (rare path: both forms in one kernel)
          if (A.g[t] & 1u) {
            if (grp) apply_row<64, 1, kOpOptimize, false, true>(tv, row_ptr(tv, r), fresh, lane, values, nullptr, 0u,
                                                                1u, int64_t(sq), a);
            else apply_row<64, 1, kOpOptimize, false, false>(tv, row_ptr(tv, r), fresh, lane, values, nullptr, 0u,
                                                             1u, int64_t(sq), a);
          } else {
            if (grp) apply_row<64, 4, kOpOptimize, false, true>(tv, row_ptr(tv, r), fresh, lane, values, nullptr, 0u,
                                                                1u, int64_t(sq), a);
            else apply_row<64, 4, kOpOptimize, false, false>(tv, row_ptr(tv, r), fresh, lane, values, nullptr, 0u,
                                                             1u, int64_t(sq), a);
          }
        }
        fresh = false;
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    if (multi) {
      for (uint32_t w = kXKeyWords + uint32_t(lane); w < kXKeyWords + world; w += 64u) sl[w] = 0u;
      if (lane == 0) {
        *reinterpret_cast<unsigned long long*>(sl + 2) = 0ull;
        *reinterpret_cast<int64_t*>(sl) = kEmptyKey;
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  if (np) {
    if (GATED) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (lane == 0) __hip_atomic_store(&tv.ctr->n_pending, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (lane == 0) {
      tv.ctr->n_pending = 0;
    }
  }
  if (A.zero_headers && uint32_t(lane) < world)
    A.clear_ids[size_t(lane) * A.geo.ids_block + A.t0 + t] = 0;
}

__global__ __launch_bounds__(64) void shard_slow_all_kernel(ShardOwnerArgs A) {
  __shared__ BfsSlot q[kMaxCuckooCount];
  __shared__ CuckooRecord path[kMaxBfsPathLen];
  shard_slow_all_role<false>(A, blockIdx.x, A.recv_ids, A.rows, q, path, threadIdx.x);
}

// ---- sender: rows -> occurrences, occurrence gradients -> row slots ---------------------------------
struct ShardGatherTab {
  uint32_t nblk_items;
  uint32_t nblk_ids;
  uint32_t io_off;      // floats: SCATTER the table's embeddings in `flat`, SUM its gradients
  uint32_t n;           // occurrences of the batch (0: nothing to do)
  uint32_t gv;          // lane-group shape of the table in this launch (MHTE_SWITCH_G)
};
struct ShardGatherArgs {
  ConstStatics st;
  const float* in;      // SCATTER: row buffer; SUM: flat gradients
  float* out;           // SCATTER: flat embeddings; SUM: row buffer
  const uint32_t* slot_off;   // [T][n_max]
  uint32_t slot;
  uint32_t n_max;
  uint32_t t0, tc;      // tables [t0, t0 + tc) of the model (ShardOwnerArgs); tab / gt by position in the launch
  ShardTab tab[kMaxStepTables];
  ShardGatherTab gt[kMaxStepTables];
};
static_assert(sizeof(ShardGatherArgs) <= 4096, "kernel arguments exceed 4 KB");

__device__ __forceinline__ void shard_gather_ctl(GatherCtl& c, const MStepStatic& s, uint32_t cur,
                                                 const uint32_t* slot_off, uint32_t n_max, uint32_t t,
                                                 uint32_t dim, const ShardGatherTab& gt) {
  c.index = slot_off + size_t(t) * n_max;
  c.part = s.part[cur];
  c.arrive = s.arrive[cur];
  c.n_max = s.n_max;
  c.dim = dim;
  c.nblk_items = gt.nblk_items;
  c.nblk_ids = gt.nblk_ids;
  c.index_is_offset = 1;
  c.peer_win = nullptr;
  c.peer_out_off = 0;
  c.rows_block = 1;
  c.pre_summed = 0;
}

template <bool SCATTER, int VW>
__device__ __forceinline__ void shard_gather_switch(uint32_t gv, const RunView& d, const GatherCtl& c,
                                                    uint32_t bid, char* raw) {
#define MHTE_GATHER_CALL(G_, V_) \
  rd_gather_role<G_, V_, SCATTER>(d, c, bid, *reinterpret_cast<GatherLds<G_, V_>*>(raw))
  MHTE_SWITCH_G(VW, gv, MHTE_GATHER_CALL)
#undef MHTE_GATHER_CALL
}
static_assert(sizeof(GatherLds<8, 4>) >= sizeof(GatherLds<16, 4>) &&
                  sizeof(GatherLds<8, 4>) >= sizeof(GatherLds<32, 4>) &&
                  sizeof(GatherLds<8, 4>) >= sizeof(GatherLds<64, 4>) &&
                  sizeof(GatherLds<8, 4>) >= sizeof(GatherLds<8, 1>) &&
                  sizeof(GatherLds<8, 4>) >= sizeof(GatherLds<64, 1>),
              "LDS of the gather role");

// rows back -> every occurrence of the batch in `slot`
template <int VW>
__global__ __launch_bounds__(256) void shard_scatter_kernel(ShardGatherArgs A) {
  __shared__ __attribute__((aligned(16))) char raw[sizeof(GatherLds<8, 4>)];
  const uint32_t tl = blockIdx.y, t = A.t0 + tl;
  const ShardGatherTab gt = A.gt[tl];
  if (gt.n == 0 || blockIdx.x >= gt.nblk_items + gt.nblk_ids || !MHTE_SHAPE_IS(VW, gt.gv)) return;
  const MStepStatic& s = deref_const(A.st + t);
  const uint32_t cur = A.slot & 1u;
  RunView d = s.rv[cur];
  d.nblk = (gt.n + kRdBlock - 1) / kRdBlock;
  GatherCtl c;
  c.in = A.in;
  c.out = A.out + size_t(gt.io_off);
  shard_gather_ctl(c, s, cur, A.slot_off, A.n_max, t, A.tab[tl].dim, gt);
  shard_gather_switch<true, VW>(gt.gv, d, c, blockIdx.x, raw);
}

// ---- sender: rows back -> occurrences AND the run dedup of the next batch, ONE launch ----------------
// (what mstep_fwd_dedup_kernel is to the unsharded step: the scatter is bound by HBM bandwidth, the
// dedup by device-scope atomics; side by side they take about as long as the longer one.)
// 1024-thread workgroups: the first F.nd are the dedup's persistent workgroups, the rest scatter —
// the multi-table forward's role (mstep_scatter_role) with the table probe replaced by the row the
// owner sent back: distinct ids UNR per lane group, heavy lists one work item per wavefront (run starts
// by wave scan, no LDS, no barrier).
template <int G, int BLOCK, int UNR, int VEC>
__device__ __forceinline__ void shard_scatter_role(const RunView& d, const float* __restrict__ rows,
                                                   const uint32_t* __restrict__ slot_off,
                                                   float* __restrict__ out, uint32_t dim, int64_t n_max,
                                                   uint32_t bid, uint32_t nblk, uint32_t item_split) {
  constexpr int NG = BLOCK / G;
  constexpr int GPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int j = lane & (G - 1);
  const int gbase = lane & ~(G - 1);
  const int grp = threadIdx.x / G;
  const uint32_t e = uint32_t(j) * VEC;
  const bool ev = e < dim;
  const uint32_t n_unique = d.ctr[0];
  const uint32_t n_items = d.ctr[2];
  const int64_t nu = min(n_max, int64_t(n_unique));
  constexpr int PER = (kStepLightMax + G - 1) / G;
  // Unique index g = w * nblk + bid for the workgroup's w-th group-id: consecutive indices land in
  // DIFFERENT workgroups, so the U distinct ids of a batch (a fifth of its positions under Zipf) keep
  // every workgroup — every CU — busy with a few groups instead of filling the first U / (NG * UNR)
  // workgroups and leaving the others empty (measured: the fused launch took the sum of its two roles).
#pragma unroll 1
  for (int64_t w0 = 0; w0 * int64_t(nblk) < nu; w0 += int64_t(NG) * UNR) {  // workgroup-uniform
    uint32_t cnt[UNR], hp[UNR], gs[UNR], ix[UNR];
    bool valid[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t g = (w0 + int64_t(grp) * UNR + u) * int64_t(nblk) + int64_t(bid);
      valid[u] = g < nu;
      const int64_t gi = valid[u] ? g : 0;  // (loads from a safe index, masked afterwards)
      cnt[u] = d.ucnt[gi];
      hp[u] = d.upos[gi];
      gs[u] = d.uslot[gi];
      ix[u] = slot_off[gi];
    }
    Vec<VEC> v[UNR];
    uint32_t x[UNR][PER];
    bool flat[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (!valid[u]) cnt[u] = 0;
      flat[u] = cnt[u] > 1 && cnt[u] <= uint32_t(kStepLightMax);
      vec_zero(v[u]);
      if (valid[u] && cnt[u] <= uint32_t(kStepLightMax) && ix[u] != 0xffffffffu && ev) v[u].load(rows + size_t(ix[u]) + e);
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const uint32_t idx = uint32_t(j) + uint32_t(q) * G;
        x[u][q] = (flat[u] && idx < cnt[u]) ? d.hlist[size_t(gs[u]) * kLightMax + idx] : 0xffffffffu;
      }
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
  // ---- heavy lists: one work item per WAVEFRONT (mstep_scatter_role)
  // (unit k * nblk + bid: the items too are dealt out one per workgroup first)
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll 1
  for (uint32_t unit = wave * nblk + bid; unit < n_items * item_split; unit += nblk * (BLOCK / 64)) {
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
    const uint32_t ixh = slot_off[hd.u];
    Vec<VEC> v;
    vec_zero(v);
    if (ixh != 0xffffffffu && ev) v.load(rows + size_t(ixh) + e);
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
}

#ifndef MHTE_SHARD_SCATTER_UNR
#define MHTE_SHARD_SCATTER_UNR 1
#endif
__global__ __launch_bounds__(kRdBlock, 8) __attribute__((amdgpu_num_sgpr(80))) void shard_scatter_dedup_kernel(
    ShardGatherArgs A, MDedupArgs D, MFwdFuse F, uint32_t item_split) {
  __shared__ __attribute__((aligned(16))) RdLds L;
  WaveTrace wt(nullptr);
  if (blockIdx.x < F.nd) {
    const uint32_t total = D.blk_start[D.T];
#pragma unroll 1
    for (uint32_t w = blockIdx.x; w < total; w += F.nd) {
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
    return;
  }
  const uint32_t lin = blockIdx.x - F.nd;
  uint32_t tl = 0;
  while (tl + 1 < A.tc && F.fwd_start[tl + 1] <= lin) ++tl;
  const ShardGatherTab gt = A.gt[tl];
  const uint32_t bid = lin - F.fwd_start[tl];
  const uint32_t nblk = F.fwd_start[tl + 1] - F.fwd_start[tl];
  // (tables that move one float per lane are scattered by a launch of their own, shard_scatter_kernel<1>)
  if (gt.n == 0 || bid >= nblk || !MHTE_SHAPE_IS(4, gt.gv)) return;
  const uint32_t t = A.t0 + tl;
  const MStepStatic& s = deref_const(A.st + t);
  const uint32_t cur = A.slot & 1u;
  const RunView d = s.rv[cur];
  float* out = A.out + size_t(gt.io_off);
  const uint32_t* so = A.slot_off + size_t(t) * A.n_max;
  const uint32_t dim = A.tab[tl].dim;
#define MHTE_SSC_CALL(G_, V_) \
  shard_scatter_role<G_, kRdBlock, MHTE_SHARD_SCATTER_UNR, V_>(d, A.in, so, out, dim, s.n_max, bid, nblk, item_split)
  MHTE_SWITCH_G(4, gt.gv, MHTE_SSC_CALL)
#undef MHTE_SSC_CALL
}

// backward launch of the sender side, per table:
//   numbering + owner packing of the batch deduplicated into build_slot | gradient sums of the
//   batch in slot -> row slots
// (either half may be absent: n_build[t] == 0, gt[t].n == 0)
struct ShardBuildArgs {
  ConstStatics st;
  ShardGeom geo;
  int64_t* send_ids;              // blocks of build_slot; headers zero on entry
  uint32_t* slot_off_build;       // [T][n_max] of build_slot
  uint32_t* flags;
  const float* grads;             // flat gradients of the batch in `slot`
  float* rows_out;                // sender-side row buffer
  const uint32_t* slot_off;       // [T][n_max] of `slot`
  uint32_t build_slot;
  uint32_t slot;
  uint32_t n_max;
  uint32_t t0;                    // tables [t0, t0 + gridDim.y) of the model; tab / n_build / gt by position
  uint32_t exact;                 // 1 (mhte_shard_step_set_exact_order): heavy lists summed strictly in occurrence
                                  // order by shard_exact_sum_kernel, launched in front of this one
  // direct peer stores (nullptr: send_ids / rows_out): ids into the owners' id buffers, sums into their
  // gradient buffers
  const unsigned long long* peer_win;   // [world] device array
  unsigned long long peer_ids_off, peer_grads_off;
  ShardTab tab[kMaxStepTables];
  uint32_t n_build[kMaxStepTables];
  ShardGatherTab gt[kMaxStepTables];
};
static_assert(sizeof(ShardBuildArgs) <= 4096, "kernel arguments exceed 4 KB");

// (a table is served — its numbering included — by the instance of its gradient rows' lane width)
template <int VW>
__global__ __launch_bounds__(256) void shard_build_kernel(ShardBuildArgs A) {
  __shared__ __attribute__((aligned(16))) char raw[sizeof(GatherLds<8, 4>)];
  const uint32_t tl = blockIdx.y, t = A.t0 + tl;
  if (!MHTE_SHAPE_IS(VW, A.gt[tl].gv)) return;
  const MStepStatic& s = deref_const(A.st + t);
  uint32_t bid = blockIdx.x;
  const uint32_t nb = A.n_build[tl] ? s.nblk_build : 0u;
  if (bid < nb) {
    RunView nxt = s.rv[A.build_slot & 1u];
    nxt.nblk = (A.n_build[tl] + kRdBlock - 1) / kRdBlock;
    const ShardTab tb = A.tab[tl];
    PackCtl pc;
    pc.send_ids = A.send_ids;
    pc.peer_win = A.peer_win;
    pc.peer_ids_off = A.peer_ids_off;
    pc.slot_off = A.slot_off_build + size_t(t) * A.n_max;
    pc.flags = A.flags;
    pc.world = A.geo.world;
    pc.ids_block = A.geo.ids_block;
    pc.rows_block = A.geo.rows_block;
    pc.hdr_word = t;
    pc.cap = tb.cap;
    pc.id_off = tb.id_off;
    pc.row_off = tb.row_off;
    pc.dim = tb.dim;
    rd_build_role<true>(nxt, uint32_t(kStepLightMax), bid, nb, &pc);
    return;
  }
  bid -= nb;
  const ShardGatherTab gt = A.gt[tl];
  if (gt.n == 0 || bid >= gt.nblk_items + gt.nblk_ids) return;
  const uint32_t cur = A.slot & 1u;
  RunView d = s.rv[cur];
  d.nblk = (gt.n + kRdBlock - 1) / kRdBlock;
  GatherCtl c;
  c.in = A.grads + size_t(gt.io_off);
  c.out = A.rows_out;
  shard_gather_ctl(c, s, cur, A.slot_off, A.n_max, t, A.tab[tl].dim, gt);
  c.peer_win = A.peer_win;
  c.peer_out_off = A.peer_grads_off;
  c.rows_block = A.geo.rows_block;
  c.pre_summed = A.exact;
  shard_gather_switch<false, VW>(gt.gv, d, c, bid, raw);
}

// Exact order on the sender side of the sharded step (round 6): every table's heavy lists summed strictly in
// occurrence order — rd_exact_sum_role, the single-table step's kernel — straight into the row slot the sum
// travels in (the owner's gradient block with peer stores), in front of shard_build_kernel, whose item
// workgroups then leave.  With it every sender's per-id sum is the reference's sequential sum
// (RT/ops/unique_mapping_ops.cc:284-329) bit for bit, and so is every owner's row.
struct ExactToSlot {
  GatherCtl c;
  __device__ __forceinline__ float* operator()(uint32_t, const ItemHdr& hd) const {
    const uint32_t ix = c.index[hd.u];
    return ix == 0xffffffffu ? nullptr : gather_out_ptr(c, ix);
  }
};
__global__ __launch_bounds__(kExactThreads) void shard_exact_sum_kernel(ShardBuildArgs A) {
  __shared__ ExactLds L;
  const uint32_t tl = blockIdx.y, t = A.t0 + tl;
  const ShardGatherTab gt = A.gt[tl];
  if (gt.n == 0) return;
  const MStepStatic& s = deref_const(A.st + t);
  const uint32_t cur = A.slot & 1u;
  RunView d = s.rv[cur];
  d.nblk = (gt.n + kRdBlock - 1) / kRdBlock;
  ExactToSlot dst;
  dst.c.in = A.grads + size_t(gt.io_off);
  dst.c.out = A.rows_out;
  shard_gather_ctl(dst.c, s, cur, A.slot_off, A.n_max, t, A.tab[tl].dim, gt);
  dst.c.peer_win = A.peer_win;
  dst.c.peer_out_off = A.peer_grads_off;
  dst.c.rows_block = A.geo.rows_block;
  const float* grads = A.grads + size_t(gt.io_off);
  if (gt.gv & 1u) rd_exact_sum_role<1>(d, grads, A.tab[tl].dim, dst, blockIdx.x, gridDim.x, L);
  else rd_exact_sum_role<4>(d, grads, A.tab[tl].dim, dst, blockIdx.x, gridDim.x, L);
}

// ---- peer-store transport: one process per GPU, every rank maps every other rank's WINDOW ----------
// (hipIpcGetMemHandle / hipIpcOpenMemHandle; works over xGMI between devices and between two
// processes on one device alike.)  A window holds what peers write into a rank:
//   flags      credit[kIpcChannels][kMaxShards]   written by peer p: "p is ready to RECEIVE exchange n
//                                                  of this channel" (its buffer of the channel is free)
//              arrived[kIpcChannels][kMaxShards]  written by peer p: "p's block of exchange n landed"
//   ids_recv[2], snd_rows (rows coming back), own_grads (gradient sums coming in)
// Channels: the two id slots, rows, gradients, and a data-less one for the creation self test.  Every
// exchange of a channel has a sequence number the ranks count in lockstep (the calls are collective).
//
// shard_push_kernel (grid (x, world), y = peer): post my credit to the peer, wait for the peer's
// credit, copy the OCCUPIED part of every (peer, table) segment straight into the peer's window — the
// counts are the id blocks' headers, read on the device: exact-size exchanges and still nothing
// reaches the host.  The stores are made visible by the END of the launch (measured: a system-scope
// fence per workgroup is an L2 write-back each, 45 us for a 3 MB push; none at all: 4 us), so
// `arrived` is published by the next launch on the stream: shard_sync_kernel (one wavefront) first
// posts the arrival of every push since the last sync, then holds the stream until the blocks of
// the wanted peers have landed; consumers are separate launches behind it.  Every rank posts what
// it owes before it waits, so the waits cannot cross.  Waits are bounded (wall clock): a peer that
// never shows up sets kShardPeerTimeout in the host-mapped flag word instead of hanging the queue.
enum : uint32_t { kShardPeerTimeout = 2u };
constexpr int kIpcChannels = 5;
enum IpcChannel : uint32_t { kChIds0 = 0, kChIds1 = 1, kChRows = 2, kChGrads = 3, kChTest = 4 };
constexpr size_t kIpcFlagBytes = size_t(2) * kIpcChannels * kMaxShards * sizeof(uint32_t);

__device__ __forceinline__ uint32_t* ipc_credit(char* win, uint32_t chan, uint32_t from) {
  return reinterpret_cast<uint32_t*>(win) + size_t(chan) * kMaxShards + from;
}
__device__ __forceinline__ uint32_t* ipc_arrived(char* win, uint32_t chan, uint32_t from) {
  return reinterpret_cast<uint32_t*>(win) + size_t(kIpcChannels + chan) * kMaxShards + from;
}
__device__ __forceinline__ uint32_t ld_sys(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// sequence numbers wrap after 2^32 exchanges: compared as signed differences
__device__ __forceinline__ bool ipc_spin(const uint32_t* flag, uint32_t seq, uint64_t timeout_ticks) {
  if (int32_t(ld_sys(flag) - seq) >= 0) return true;
  const uint64_t t0 = wall_clock64();
  for (;;) {
    for (int i = 0; i < 64; ++i) {
      if (int32_t(ld_sys(flag) - seq) >= 0) return true;
      __builtin_amdgcn_s_sleep(2);
    }
    if (wall_clock64() - t0 > timeout_ticks) return false;
  }
}

struct ShardPushArgs {
  char* win[kMaxShards];      // every rank's window as mapped here (win[rank] = my own)
  const char* src;            // [world][block] of this rank's send buffer; nullptr: no data (test)
  const int64_t* counts;      // [world][ids_block]: the headers that size the segments
  uint64_t dst_off;           // byte offset of the destination buffer inside a window
  uint32_t* flags;            // host-mapped error word
  uint64_t timeout_ticks;
  ShardGeom geo;
  uint32_t rank;
  uint32_t chan;
  uint32_t seq;
  uint32_t ids;               // 1: id blocks (header + int64 slots), 0: row blocks (floats)
  uint32_t half;              // row blocks of 16-bit elements (the fp16 gradient wire): dense
                              // [world][rows_block] arrays of 16-bit words on both sides
  const ShardTab* tab;        // [T], device memory (the step's geometry: fixed at creation)
};
static_assert(sizeof(ShardPushArgs) <= 4096, "kernel arguments exceed 4 KB");

__device__ __forceinline__ void ipc_copy16(char* dst, const char* src, uint32_t n16, uint32_t first,
                                           uint32_t stride) {
  const uint4* s = reinterpret_cast<const uint4*>(src);
  uint4* d = reinterpret_cast<uint4*>(dst);
  for (uint32_t i = first; i < n16; i += stride) {
    const uint4 v = s[i];
    __builtin_nontemporal_store(v.x, &d[i].x);
    __builtin_nontemporal_store(v.y, &d[i].y);
    __builtin_nontemporal_store(v.z, &d[i].z);
    __builtin_nontemporal_store(v.w, &d[i].w);
  }
}

__global__ __launch_bounds__(256) void shard_push_kernel(ShardPushArgs A) {
  __shared__ uint32_t ok;
  const uint32_t p = blockIdx.y;
  char* mine = A.win[A.rank];
  char* peer = A.win[p];
  if (threadIdx.x == 0) {
    if (blockIdx.x == 0) st_sys(ipc_credit(peer, A.chan, A.rank), A.seq);
    ok = ipc_spin(ipc_credit(mine, A.chan, p), A.seq, A.timeout_ticks) ? 1u : 0u;
    if (!ok) atomicOr(A.flags, uint32_t(kShardPeerTimeout));
  }
  __syncthreads();
  if (!ok || !A.src) return;
  const size_t blk = A.ids ? size_t(A.geo.ids_block) * 8 : size_t(A.geo.rows_block) * (A.half ? 2 : 4);
  const char* src = A.src + size_t(p) * blk;
  char* dst = peer + A.dst_off + size_t(A.rank) * blk;
  const int64_t* hdr = A.counts + size_t(p) * A.geo.ids_block;
  const uint32_t first = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  if (A.ids) {
    const uint32_t hdr16 = (((A.geo.T + 7u) & ~7u) * 8u) / 16u;
    ipc_copy16(dst, src, hdr16, first, stride);
  }
  for (uint32_t t = 0; t < A.geo.T; ++t) {
    const ShardTab tb = A.tab[t];
    const uint64_t c = uint64_t(hdr[t]);
    const uint32_t n = c > tb.cap ? tb.cap : uint32_t(c);
    if (!n) continue;
    if (A.ids)
      ipc_copy16(dst + size_t(tb.id_off) * 8, src + size_t(tb.id_off) * 8, (n + 1u) / 2u, first, stride);
    else if (A.half)
      ipc_copy16(dst + size_t(tb.row_off) * 2, src + size_t(tb.row_off) * 2, (n * tb.dim * 2u + 15u) / 16u, first,
                 stride);
    else
      ipc_copy16(dst + size_t(tb.row_off) * 4, src + size_t(tb.row_off) * 4, (n * tb.dim + 3u) / 4u, first, stride);
  }
}

// ---- RCCL transport, exact-size form: ONE send / recv pair per peer and exchange (round 6) -----------------
// A (peer, table) segment of a wire block sits at a fixed offset and is occupied from its start; what crosses
// is the occupied part.  Round 5 issued one ncclSend / ncclRecv pair per (peer, table) — 208 pairs per
// exchange at 8 ranks x 26 tables.  Now the occupied parts of a peer's segments are packed back to back into
// a staging block by this kernel (PACK; every segment rounded up to 16 bytes), ONE pair per peer moves the
// packed stream, and the receiver spreads it over its block again (UNPACK) — the counts are the id blocks'
// headers, read on the device, the same ones the host sizes the pair with.  grid (x, world), y = peer.
struct ShardPackArgs {
  const char* src;            // [world][block]
  char* dst;                  // [world][block]
  const int64_t* counts;      // [world][ids_block]: the headers that size the segments
  ShardGeom geo;
  uint32_t ids;               // 1: id blocks (the int64 slots; the header is not part of the stream)
  uint32_t half;              // row blocks of 16-bit elements (the fp16 gradient wire)
  const ShardTab* tab;        // [T], device memory
};
// bytes of the occupied part of a segment on the wire, rounded up to 16 (the host computes the same)
__host__ __device__ __forceinline__ uint64_t shard_packed_bytes(uint32_t n, uint32_t dim, bool ids, bool half) {
  const uint64_t b = ids ? uint64_t(n) * 8u : uint64_t(n) * dim * (half ? 2u : 4u);
  return (b + 15u) & ~uint64_t(15);
}
// exactly `bytes` bytes (all offsets on the wire are multiples of 4): 16-byte pieces where both sides are
// aligned for them, 4-byte pieces otherwise (fp16 blocks of odd-dim tables), then the last bytes
__device__ __forceinline__ void pack_copy(char* dst, const char* src, uint64_t bytes, uint32_t first, uint32_t stride) {
  if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15u) == 0u) {
    const uint4* s = reinterpret_cast<const uint4*>(src);
    uint4* d = reinterpret_cast<uint4*>(dst);
    const uint64_t n16 = bytes / 16u;
    for (uint64_t i = first; i < n16; i += stride) d[i] = s[i];
    for (uint64_t i = n16 * 16u + first; i < bytes; i += stride) dst[i] = src[i];
  } else {
    const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    const uint64_t n4 = bytes / 4u;
    for (uint64_t i = first; i < n4; i += stride) d[i] = s[i];
    for (uint64_t i = n4 * 4u + first; i < bytes; i += stride) dst[i] = src[i];
  }
}
template <bool UNPACK>
__global__ __launch_bounds__(256) void shard_pack_kernel(ShardPackArgs A) {
  const uint32_t p = blockIdx.y;
  const size_t es = A.half ? 2 : 4;
  const size_t blk = A.ids ? size_t(A.geo.ids_block) * 8 : size_t(A.geo.rows_block) * es;
  const char* src = A.src + size_t(p) * blk;
  char* dst = A.dst + size_t(p) * blk;
  const int64_t* hdr = A.counts + size_t(p) * A.geo.ids_block;
  const uint32_t first = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  uint64_t poff = 0;
  for (uint32_t t = 0; t < A.geo.T; ++t) {
    const ShardTab tb = A.tab[t];
    const uint64_t c = uint64_t(hdr[t]);
    const uint32_t n = c > tb.cap ? tb.cap : uint32_t(c);
    if (!n) continue;
    const uint64_t bytes = A.ids ? uint64_t(n) * 8u : uint64_t(n) * tb.dim * es;   // (the occupied part, exactly:
    const size_t uoff = A.ids ? size_t(tb.id_off) * 8 : size_t(tb.row_off) * es;   //  the next segment is not touched)
    if (UNPACK) pack_copy(dst + uoff, src + poff, bytes, first, stride);
    else pack_copy(dst + poff, src + uoff, bytes, first, stride);
    poff += shard_packed_bytes(n, tb.dim, A.ids != 0, A.half != 0);
  }
}

// The fp16 gradient wire (the reference's optional cast of the gradient all-to-all,
// NT/distributed_ps_sync.py:47,334-337: `grad_flat` goes out as tf.float16 and is cast back by the
// owner): NARROW = sender, fp32 sums -> fp16 (round to nearest even) before the exchange; !NARROW =
// owner, back to fp32 before the update.  The occupied part of every (peer, table) segment, sized by
// the id blocks' headers.  grid (x, world * T).
struct ShardCvtArgs {
  const int64_t* counts;      // [world][ids_block]
  const void* src;            // [world][rows_block] fp32 (NARROW) / 16-bit words at the same element offsets
  void* dst;
  ShardGeom geo;
  uint32_t peer_lo, peer_n;   // peers [peer_lo, peer_lo + peer_n)
  const ShardTab* tab;        // [T], device memory
};
template <bool NARROW>
__global__ __launch_bounds__(256) void shard_cvt_kernel(ShardCvtArgs A) {
  const uint32_t p = A.peer_lo + blockIdx.y / A.geo.T, t = blockIdx.y % A.geo.T;
  const ShardTab tb = A.tab[t];
  const uint64_t c = uint64_t(A.counts[size_t(p) * A.geo.ids_block + t]);
  const uint32_t n = c > tb.cap ? tb.cap : uint32_t(c);
  const size_t base = size_t(p) * A.geo.rows_block + tb.row_off;   // element offset
  const uint32_t total = (n * tb.dim + 3u) / 4u;                   // 4 elements per thread and trip (a table's
                                                                   // region is whole float4s: cap is a multiple of 4)
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    if (NARROW) {
      const float4 v = reinterpret_cast<const float4*>(static_cast<const float*>(A.src) + base)[i];
      ushort4 h;
      h.x = __half_as_ushort(__float2half_rn(v.x));
      h.y = __half_as_ushort(__float2half_rn(v.y));
      h.z = __half_as_ushort(__float2half_rn(v.z));
      h.w = __half_as_ushort(__float2half_rn(v.w));
      reinterpret_cast<ushort4*>(static_cast<unsigned short*>(A.dst) + base)[i] = h;
    } else {
      const ushort4 h = reinterpret_cast<const ushort4*>(static_cast<const unsigned short*>(A.src) + base)[i];
      float4 v;
      v.x = __half2float(__ushort_as_half(h.x));
      v.y = __half2float(__ushort_as_half(h.y));
      v.z = __half2float(__ushort_as_half(h.z));
      v.w = __half2float(__ushort_as_half(h.w));
      reinterpret_cast<float4*>(static_cast<float*>(A.dst) + base)[i] = v;
    }
  }
}

// Creation self test with DATA (ipc_selftest): round r, rank a -> peer b: a pattern of (r, a, b, word)
// into the head of b's gradient block for sender a, with the push kernel's store form; b checks it
// from its own window with plain loads in a launch of its own behind the sync — the path every real
// exchange takes.  Several rounds over the SAME addresses with different patterns: a stale cached
// copy of an earlier round, or stores that have not landed when `arrived` is seen, fail the check.
enum : uint32_t { kShardSelftestBad = 4u };
struct ShardSelftestArgs {
  char* win[kMaxShards];
  uint64_t off;               // byte offset of the gradient blocks inside a window
  uint64_t blk;               // bytes between two senders' blocks
  uint32_t* flags;
  uint32_t n16;               // 16-byte words of the pattern
  uint32_t rank, round, check;
  uint32_t corrupt;           // test hook (MHTE_SHARD_SELFTEST_CORRUPT): the checker expects another round
};
__device__ __forceinline__ uint32_t selftest_word(uint32_t round, uint32_t from, uint32_t to, uint32_t i) {
  uint32_t x = (round + 1u) * 0x9e3779b9u ^ (from * 0x85ebca6bu + to * 0xc2b2ae35u + i * 0x27d4eb2fu);
  x ^= x >> 15;
  x *= 0x2c1b3c6du;
  x ^= x >> 12;
  return x;
}
__global__ __launch_bounds__(256) void shard_selftest_kernel(ShardSelftestArgs A) {
  const uint32_t p = blockIdx.x;   // the peer
  if (!A.check) {
    uint4* d = reinterpret_cast<uint4*>(A.win[p] + A.off + size_t(A.rank) * A.blk);
    for (uint32_t i = threadIdx.x; i < A.n16; i += blockDim.x) {
      __builtin_nontemporal_store(selftest_word(A.round, A.rank, p, 4 * i), &d[i].x);
      __builtin_nontemporal_store(selftest_word(A.round, A.rank, p, 4 * i + 1), &d[i].y);
      __builtin_nontemporal_store(selftest_word(A.round, A.rank, p, 4 * i + 2), &d[i].z);
      __builtin_nontemporal_store(selftest_word(A.round, A.rank, p, 4 * i + 3), &d[i].w);
    }
    return;
  }
  const uint4* s = reinterpret_cast<const uint4*>(A.win[A.rank] + A.off + size_t(p) * A.blk);
  const uint32_t r = A.round + A.corrupt;
  bool bad = false;
  for (uint32_t i = threadIdx.x; i < A.n16; i += blockDim.x) {
    const uint4 v = s[i];
    bad |= v.x != selftest_word(r, p, A.rank, 4 * i) || v.y != selftest_word(r, p, A.rank, 4 * i + 1) ||
           v.z != selftest_word(r, p, A.rank, 4 * i + 2) || v.w != selftest_word(r, p, A.rank, 4 * i + 3);
  }
  if (__any(bad) && (threadIdx.x & 63u) == 0u) atomicOr(A.flags, uint32_t(kShardSelftestBad));
}

// one wavefront: (1) the pushes completed before this launch are published to every peer, (2) the
// stream is held until peers [lo, hi) have published exchange `wait_seq` of `wait_chan`
struct ShardSyncArgs {
  char* win[kMaxShards];
  uint32_t* flags;
  uint64_t timeout_ticks;
  uint32_t rank, world;
  uint32_t n_sig;
  uint32_t sig_chan[kIpcChannels], sig_seq[kIpcChannels];
  uint32_t wait_chan;         // kIpcChannels: nothing to wait for
  uint32_t wait_seq, lo, hi;
};

// ---- direct peer stores: the data kernels write straight into the peers' windows (no push kernels);
// every wait of the protocol sits in ONE one-wavefront launch per step phase.  At a sync point a rank
//   * copies the headers (per-table counts) of an id exchange it has just written to the peers' id buffers,
//   * publishes `arrived` for every exchange its earlier launches wrote (their stores are complete: the
//     kernels have ended), and `credit` for every buffer of its own whose consumers are enqueued,
//   * waits for the same set from every peer.
// Every rank runs the same program, so what a rank waits for at a point is what its peers publish at
// that same point — they publish before they wait: no cycle.  Spins are bounded (ipc_spin).
struct ShardSync2Args {
  char* win[kMaxShards];
  uint32_t* flags;
  uint64_t timeout_ticks;
  uint32_t rank, world;
  uint32_t n_arr, n_cred;
  uint32_t arr_chan[kIpcChannels], arr_seq[kIpcChannels];
  uint32_t cred_chan[kIpcChannels], cred_seq[kIpcChannels];
  const int64_t* hdr_src;     // send blocks whose headers go out (nullptr: none): [world][ids_block]
  uint64_t hdr_dst_off;       // byte offset of this rank's block of that id buffer inside a window
  uint32_t hdr_words;         // words of a header (T rounded up)
  uint32_t ids_block;
};
__global__ __launch_bounds__(64) void shard_sync2_kernel(ShardSync2Args A) {
  const uint32_t lane = threadIdx.x;
  if (A.hdr_src) {
    for (uint32_t i = lane; i < A.world * A.hdr_words; i += 64u) {
      const uint32_t p = i / A.hdr_words, w = i % A.hdr_words;
      const int64_t v = A.hdr_src[size_t(p) * A.ids_block + w];
      __builtin_nontemporal_store(v, reinterpret_cast<int64_t*>(A.win[p] + A.hdr_dst_off) + w);
    }
    __threadfence_system();
  }
  for (uint32_t k = 0; k < A.n_arr; ++k)
    for (uint32_t p = lane; p < A.world; p += 64u) st_sys(ipc_arrived(A.win[p], A.arr_chan[k], A.rank), A.arr_seq[k]);
  for (uint32_t k = 0; k < A.n_cred; ++k)
    for (uint32_t p = lane; p < A.world; p += 64u) st_sys(ipc_credit(A.win[p], A.cred_chan[k], A.rank), A.cred_seq[k]);
  char* mine = A.win[A.rank];
  bool ok = true;
  for (uint32_t k = 0; k < A.n_arr; ++k)
    for (uint32_t p = lane; p < A.world; p += 64u)
      ok = ipc_spin(ipc_arrived(mine, A.arr_chan[k], p), A.arr_seq[k], A.timeout_ticks) && ok;
  for (uint32_t k = 0; k < A.n_cred; ++k)
    for (uint32_t p = lane; p < A.world; p += 64u)
      ok = ipc_spin(ipc_credit(mine, A.cred_chan[k], p), A.cred_seq[k], A.timeout_ticks) && ok;
  if (!ok) atomicOr(A.flags, uint32_t(kShardPeerTimeout));
  __threadfence_system();
}

__global__ __launch_bounds__(64) void shard_sync_kernel(ShardSyncArgs A) {
  for (uint32_t k = 0; k < A.n_sig; ++k)
    for (uint32_t p = threadIdx.x; p < A.world; p += 64)
      st_sys(ipc_arrived(A.win[p], A.sig_chan[k], A.rank), A.sig_seq[k]);
  if (A.wait_chan >= uint32_t(kIpcChannels)) return;
  char* mine = A.win[A.rank];
  for (uint32_t p = A.lo + threadIdx.x; p < A.hi; p += 64)
    if (!ipc_spin(ipc_arrived(mine, A.wait_chan, p), A.wait_seq, A.timeout_ticks))
      atomicOr(A.flags, uint32_t(kShardPeerTimeout));
  __threadfence_system();
}

}  // namespace mhte
#endif  // MHTE_SHARD_KERNELS_H_
