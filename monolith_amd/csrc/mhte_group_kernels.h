// Grouping rows by a bounded key (AuxWs::group_sorted): a stable least-significant-digit radix sort of
// (32-bit key, position) pairs written for 64-wide wavefronts, then the runs of the sorted keys compacted
// into the (key, list offset) arrays the pooling-gradient kernels read.  Round 6: replaces rocPRIM's
// Onesweep sort + run-length encode + scan (≈ 20 launches per grouping, profiles/r05/kernel_stats_pooling.md)
// — no library primitive is left in the product.
//
// What it serves: the deterministic gradient of FusedGatherEmbeddingsByInput (the reference adds with
// GpuAtomicAdd in arrival order, runtime/ops/map_id_to_embedding.cu.cc:76-118) and the unsorted ragged
// reductions (runtime/ops/reduce_op.cc:46-49,77-81,110-116): rows that share a destination must be added in
// row order, so equal keys have to keep their positions ascending — a STABLE sort.
//
// One pass (digit of <= 10 bits, three launches):
//   gs_hist      a workgroup (4 wavefronts) counts the digits of its tile of 4 096 keys in LDS and stores the
//                column H[bin][tile]
//   gs_rowscan   one wavefront per bin: exclusive prefix of its row over the tiles, row total -> tot[bin]
//   gs_scatter   the tile again: every WAVEFRONT owns a contiguous quarter of the tile, keys held in
//                registers; per-wavefront digit counts in LDS give each wavefront its first output slot per
//                bin (bin start from tot[], tile prefix from H, the wavefronts before it), then the
//                wavefront walks its keys 64 at a time IN ORDER: lanes with equal digits find each other
//                with one ballot per digit bit (rank = equal-digit lanes below), the lowest of them moves the
//                bin's cursor in LDS.  Stability needs no sorting inside the tile and no atomics at all.
// Between passes a key and its position travel as ONE 8-byte word (half the scattered stores of a pass); the
// last pass writes the sorted keys and the positions as the two arrays the consumers read.
// Keys beyond the register-held tile size (n > 2 M) are walked in rounds of the same shape.
// Then gs_heads_count / gs_heads_emit compact the heads of the runs (two launches).
// 1 M keys of 18 bits: 2 passes = 8 launches; 27 bits: 3 passes = 11 launches.
#ifndef MHTE_GROUP_KERNELS_H_
#define MHTE_GROUP_KERNELS_H_

#include "mhte_kernels.h"

namespace mhte {

constexpr int kGsWaves = 4;                   // wavefronts per workgroup
constexpr int kGsThreads = kGsWaves * 64;
constexpr int kGsIters = 16;                  // 64-key steps a wavefront holds in registers per round
constexpr int kGsRound = kGsIters * 64;       // keys per wavefront and round (1 024: a tile of 4 096 keys — 1 M keys
                                              // are 256 workgroups, one per CU; with 8 192 per tile half the chip
                                              // took the scattered stores of a pass: 24.6 us per gs_scatter launch)
constexpr int kGsMaxBits = 10;                // digit bits per pass
constexpr int kGsMaxBins = 1 << kGsMaxBits;
constexpr int kGsMaxTiles = 512;              // (gs_rowscan: 8 row entries per lane)
constexpr int kGsSelTile = 4096;              // sorted keys per workgroup of the run compaction

struct GsPass {
  const int64_t* k64;      // first pass: the caller's keys, position = index (keys outside [0, limit) -> limit)
  const uint2* kvin;       // later passes: (key, position) words as the previous pass left them
  uint2* kvout;            // every pass but the last: (key, position) words
  uint32_t* kout;          // the last pass: sorted keys | their positions
  uint32_t* pout;
  uint32_t* hist;          // [bins][tstride]
  uint32_t* tot;           // [bins]
  uint32_t n, limit;
  uint32_t shift, bits;    // this pass's digit
  uint32_t rounds;         // rounds of kGsRound keys per wavefront; a tile = kGsWaves * rounds * kGsRound keys
  uint32_t ntiles, tstride;
};

// (key, position) of entry i of the pass's input
__device__ __forceinline__ uint2 gs_load(const GsPass& P, uint32_t i) {
  if (P.k64) {
    const int64_t v = P.k64[i];
    return make_uint2((v >= 0 && v < int64_t(P.limit)) ? uint32_t(v) : P.limit, i);
  }
  return P.kvin[i];
}

__global__ __launch_bounds__(kGsThreads) void gs_hist_kernel(GsPass P) {
  __shared__ uint32_t h[kGsMaxBins];
  const uint32_t bins = 1u << P.bits, mask = bins - 1u;
  for (uint32_t b = threadIdx.x; b < bins; b += kGsThreads) h[b] = 0;
  __syncthreads();
  const uint32_t tile_keys = kGsWaves * P.rounds * kGsRound;
  const uint32_t lo = blockIdx.x * tile_keys;
  const uint32_t hi = min(P.n, lo + tile_keys);   // (tile_keys * ntiles < 2^32: the host keeps n < 2^31)
  for (uint32_t i0 = lo + threadIdx.x; i0 < hi; i0 += kGsThreads * 8) {
    uint32_t k[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const uint32_t i = i0 + uint32_t(q) * kGsThreads;
      k[q] = gs_load(P, i < hi ? i : lo).x;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (i0 + uint32_t(q) * kGsThreads < hi) atomicAdd(&h[(k[q] >> P.shift) & mask], 1u);
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < bins; b += kGsThreads) P.hist[size_t(b) * P.tstride + blockIdx.x] = h[b];
}

// one wavefront per bin: H[bin][0 .. ntiles) -> its exclusive prefix, the row's sum -> tot[bin]
__global__ __launch_bounds__(64) void gs_rowscan_kernel(GsPass P) {
  constexpr int E = kGsMaxTiles / 64;
  uint32_t* row = P.hist + size_t(blockIdx.x) * P.tstride;
  const uint32_t lane = threadIdx.x;
  uint32_t v[E], sum = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const uint32_t t = lane * E + uint32_t(e);
    v[e] = t < P.ntiles ? row[t] : 0u;
    sum += v[e];
  }
  uint32_t incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t x = __shfl_up(incl, o);
    if (int(lane) >= o) incl += x;
  }
  uint32_t run = incl - sum;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const uint32_t t = lane * E + uint32_t(e);
    if (t < P.ntiles) row[t] = run;
    run += v[e];
  }
  if (lane == 63) P.tot[blockIdx.x] = incl;
}

__global__ __launch_bounds__(kGsThreads) void gs_scatter_kernel(GsPass P) {
  __shared__ uint32_t cur[kGsWaves][kGsMaxBins];   // digit counts of a wavefront, then its output cursors
  __shared__ uint32_t binstart[kGsMaxBins];
  __shared__ uint32_t wsum[kGsWaves];
  const uint32_t bins = 1u << P.bits, mask = bins - 1u;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // ---- where every bin starts in the output: exclusive scan of tot[] (<= 4 bins per thread)
  {
    uint32_t t4[kGsMaxBins / kGsThreads], s = 0;
#pragma unroll
    for (int q = 0; q < kGsMaxBins / kGsThreads; ++q) {
      const uint32_t b = threadIdx.x * (kGsMaxBins / kGsThreads) + uint32_t(q);
      t4[q] = b < bins ? P.tot[b] : 0u;
      s += t4[q];
    }
    uint32_t incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t x = __shfl_up(incl, o);
      if (int(lane) >= o) incl += x;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t run = incl - s;
    for (uint32_t i = 0; i < w; ++i) run += wsum[i];
#pragma unroll
    for (int q = 0; q < kGsMaxBins / kGsThreads; ++q) {
      const uint32_t b = threadIdx.x * (kGsMaxBins / kGsThreads) + uint32_t(q);
      if (b < bins) binstart[b] = run;
      run += t4[q];
    }
  }
  for (uint32_t b = threadIdx.x; b < bins; b += kGsThreads) {
#pragma unroll
    for (int x = 0; x < kGsWaves; ++x) cur[x][b] = 0;
  }
  __syncthreads();
  const uint32_t tile_keys = kGsWaves * P.rounds * kGsRound;
  const uint32_t wave_keys = P.rounds * kGsRound;
  const uint32_t wlo = blockIdx.x * tile_keys + w * wave_keys;   // this wavefront's keys: [wlo, whi)
  const uint32_t whi = min(P.n, wlo + wave_keys);
  uint32_t k[kGsIters], p[kGsIters];
  // ---- digit counts of this wavefront (LDS atomics on its own row: no other wavefront touches it)
  for (uint32_t r = 0; r < P.rounds; ++r) {
    const uint32_t base = wlo + r * kGsRound + lane;
#pragma unroll
    for (int it = 0; it < kGsIters; ++it) {
      const uint32_t i = base + uint32_t(it) * 64u;
      const uint2 kv = gs_load(P, i < whi ? i : 0u);
      k[it] = kv.x;
      p[it] = kv.y;
    }
#pragma unroll
    for (int it = 0; it < kGsIters; ++it)
      if (base + uint32_t(it) * 64u < whi) atomicAdd(&cur[w][(k[it] >> P.shift) & mask], 1u);
  }
  __syncthreads();
  // ---- counts -> cursors: bin start + what the tiles before this one hold + the wavefronts before this one
  for (uint32_t b = threadIdx.x; b < bins; b += kGsThreads) {
    uint32_t run = binstart[b] + P.hist[size_t(b) * P.tstride + blockIdx.x];
#pragma unroll
    for (int x = 0; x < kGsWaves; ++x) {
      const uint32_t c = cur[x][b];
      cur[x][b] = run;
      run += c;
    }
  }
  __syncthreads();
  // ---- the walk, in order.  (with one round the keys are still in registers)
  for (uint32_t r = 0; r < P.rounds; ++r) {
    const uint32_t base = wlo + r * kGsRound + lane;
    if (P.rounds > 1) {
#pragma unroll
      for (int it = 0; it < kGsIters; ++it) {
        const uint32_t i = base + uint32_t(it) * 64u;
        const uint2 kv = gs_load(P, i < whi ? i : 0u);
        k[it] = kv.x;
        p[it] = kv.y;
      }
    }
#pragma unroll
    for (int it = 0; it < kGsIters; ++it) {
      const bool valid = base + uint32_t(it) * 64u < whi;
      const uint32_t d = (k[it] >> P.shift) & mask;
      uint64_t peers = __ballot(valid);
      for (uint32_t b = 0; b < P.bits; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t m = __ballot(bit);
        peers &= bit ? m : ~m;
      }
      const uint64_t below = peers & ((uint64_t(1) << lane) - 1ull);
      if (valid) {
        const uint32_t c = cur[w][d];
        if (below == 0) cur[w][d] = c + uint32_t(__popcll(peers));
        const uint32_t o = c + uint32_t(__popcll(below));
        if (P.kvout) {
          P.kvout[o] = make_uint2(k[it], p[it]);
        } else {
          P.kout[o] = k[it];
          P.pout[o] = p[it];
        }
      }
      // (LDS operations of one wavefront execute in order: the next step's read sees this step's write)
    }
  }
}

// ---- runs of the sorted keys -> uids / seg_off / nu (the run of dropped keys, == limit, is cut off) -----
__device__ __forceinline__ uint32_t gs_head_flags(const uint32_t* __restrict__ skey, uint32_t n, uint32_t i0,
                                                  uint32_t (&kk)[4]) {
  uint32_t prev = 0;
  if (i0 > 0 && i0 < n) prev = skey[i0 - 1];
  uint32_t f = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t i = i0 + uint32_t(q);
    kk[q] = i < n ? skey[i] : 0u;
    if (i < n && (i == 0 || kk[q] != prev)) f |= 1u << q;
    prev = kk[q];
  }
  return f;
}

__global__ __launch_bounds__(1024) void gs_heads_count_kernel(const uint32_t* __restrict__ skey, uint32_t n,
                                                              uint32_t* __restrict__ hc) {
  __shared__ uint32_t ws[16];
  uint32_t kk[4];
  const uint32_t f = gs_head_flags(skey, n, blockIdx.x * kGsSelTile + threadIdx.x * 4u, kk);
  uint32_t c = __popc(f);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63u) == 0) ws[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += ws[i];
    hc[blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(1024) void gs_heads_emit_kernel(const uint32_t* __restrict__ skey, uint32_t n,
                                                             uint32_t limit, const uint32_t* __restrict__ hc,
                                                             int64_t* __restrict__ uids,
                                                             uint32_t* __restrict__ seg_off,
                                                             uint32_t* __restrict__ nu) {
  __shared__ uint32_t ws[16];
  __shared__ uint32_t s_base;
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  // heads in the tiles before this one
  uint32_t b = 0;
  for (uint32_t t = threadIdx.x; t < blockIdx.x; t += 1024) b += hc[t];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) b += __shfl_xor(b, o);
  if (lane == 0) ws[w] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += ws[i];
    s_base = s;
  }
  __syncthreads();
  const uint32_t i0 = blockIdx.x * kGsSelTile + threadIdx.x * 4u;
  uint32_t kk[4];
  const uint32_t f = gs_head_flags(skey, n, i0, kk);
  const uint32_t c = __popc(f);
  uint32_t incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t x = __shfl_up(incl, o);
    if (int(lane) >= o) incl += x;
  }
  __syncthreads();   // (ws is reused)
  if (lane == 63) ws[w] = incl;
  __syncthreads();
  uint32_t r = s_base + incl - c;
  for (uint32_t i = 0; i < w; ++i) r += ws[i];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (f & (1u << q)) {
      uids[r] = int64_t(kk[q]);
      seg_off[r] = i0 + uint32_t(q);
      ++r;
    }
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 1023) {   // r = number of runs
    seg_off[r] = n;
    *nu = (n && skey[n - 1] == limit) ? r - 1u : r;
  }
}

}  // namespace mhte
#endif  // MHTE_GROUP_KERNELS_H_
