// Dense MLP leg downstream of the embedding path (north_star: "MFMA only for the downstream
// pooled-embedding x dense MLP GEMM"; the reference side is native_training/layers/mlp.py behind
// fused_embedding_to_layout): bf16 GEMMs on the gfx950 matrix cores, written for 64-wide wavefronts.
//
//   C[m][n] = sum_k A[m][k] * B[n][k]          ("NT": both operands K-contiguous, bf16, fp32 accumulate)
//
// is the ONLY matrix kernel: the forward (x W^T), the input gradient (dz W, with W^T kept as a second
// bf16 copy) and the weight gradient (dz^T h: both operands stored TRANSPOSED by the kernel that
// produced them, so the long batch reduction is K-contiguous too) are all this form — the producing
// kernel's epilogue writes the transposed copy as well.
//
// Two tile shapes.  256 x 256 x 64 per workgroup of 8 wavefronts (4 x 2; a wavefront owns 64 x 128 =
// 2 x 4 v_mfma_f32_32x32x16_bf16 tiles, 128 accumulator registers) when M and N are multiples of 256:
// the 128 x 128 tile (4 wavefronts, 2 x 2 tiles each) pulls 2.1 GB of operands through the L2s for a
// 65 536 x 1024 x 1024 product — 8.8 TB/s at the 575 TFLOP/s it measured, i.e. L2-bound — and the
// larger tile halves that.  Operand tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4; the
// bank spread of the unpadded image comes from a swizzle on the SOURCE address), two LDS buffers, the
// next tile's DMA in flight behind the current tile's MFMAs.  Workgroups are numbered so that one XCD
// works on a contiguous range of tiles and K slices (its L2 serves the re-reads).
// Fragment layout (cdna_hip_programming.md, MFMA): A / B lane l holds row (l & 31), k = 8 * (l >> 5) +
// [0, 8); C / D lane l holds column (l & 31), rows (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5).
#ifndef MHTE_GEMM_KERNELS_H_
#define MHTE_GEMM_KERNELS_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mhte {

constexpr int kGemmBK = 64;
constexpr int kGemmLdsRow = kGemmBK + 8;   // bf16 elements per LDS row (144 bytes)

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint16_t f32_to_bf16(float f) {   // round to nearest even
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return uint16_t(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(uint32_t(h) << 16); }
// LDS traffic of ONE wavefront: its accesses complete in order, the wait makes them visible to itself
__device__ __forceinline__ void gemm_wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

enum GemmEpilogue : int {
  kEpiFwd = 0,      // + bias[n], ReLU (relu != 0), bf16 C [M][ldc] and C^T [N][ldct]
  kEpiDgrad = 1,    // * (mask[m][n] > 0), bf16 C and C^T            (dz of the layer below)
  kEpiDgradF32 = 2, // fp32 C [M][ldcf]                              (gradient of the MLP's input)
  kEpiWgrad = 3     // fp32 slab z: Cf + z * M * ldcf                (split-K partial of dW)
};

struct GemmArgs {
  const uint16_t* A;     // [M][lda] bf16 bits
  const uint16_t* B;     // [N][ldb]
  int64_t lda, ldb;
  uint32_t M, N;         // multiples of the tile
  uint32_t klen;         // reduction length of ONE z-slice (multiple of 64); slice z starts at z * klen
  const float* bias;     // kEpiFwd
  uint32_t relu;
  const uint16_t* mask;  // kEpiDgrad: [M][ldc] bf16 (the layer's forward output)
  uint16_t* C;           // bf16 out
  uint16_t* Ct;          // bf16 out, transposed (nullptr: none)
  float* Cf;             // fp32 out
  int64_t ldc, ldct, ldcf;
};

// WAVES_M x WAVES_N wavefronts, each TM x TN MFMA tiles of 32 x 32: (2, 2, 2, 2) = 128 x 128 with 256
// threads, (4, 2, 2, 4) = 256 x 256 with 512
template <int EPI, int WAVES_M, int WAVES_N, int TM, int TN>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, WAVES_M* WAVES_N == 4 ? 2 : 1) void gemm_nt_bf16_kernel(
    GemmArgs g) {
  constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32, THREADS = 64 * WAVES_M * WAVES_N;
  constexpr int RPP = THREADS / 8;   // rows per staging pass (8 pieces of 16 bytes per row)
  static_assert(BM / RPP == 4 && BN / RPP == 4, "four 16-byte pieces per thread, operand and tile");
  static_assert(TM == 2 && TN % 2 == 0, "the epilogue stages 64 x 64 parts");
  __shared__ __attribute__((aligned(16))) uint16_t smem[2][(BM + BN) * kGemmLdsRow];   // A rows, then B rows
  const uint32_t t = threadIdx.x;
  const uint32_t lane = t & 63u, wave = t >> 6;
  const uint32_t wm = wave / uint32_t(WAVES_N), wn = wave % uint32_t(WAVES_N);
  // tile numbering: hardware workgroup i runs on XCD i % 8; give every XCD a contiguous run of the
  // (z, y, x) order — x fastest: the tiles of one A row block follow each other, and a K slice of a
  // split product (z) stays on ONE XCD, whose L2 then serves every re-read of that slice's operands
  // (spread over the XCDs, each of them pulled the whole slice from memory)
  const uint32_t plane = gridDim.x * gridDim.y;
  uint32_t lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  {
    const uint32_t nwg = plane * gridDim.z;
    if ((nwg & 7u) == 0u) lin = (lin & 7u) * (nwg >> 3) + (lin >> 3);
  }
  const uint32_t z = lin / plane, tile = lin % plane;
  const uint32_t m0 = (tile / gridDim.x) * BM, n0 = (tile % gridDim.x) * BN;
  const uint64_t kbeg = uint64_t(z) * g.klen;
  const uint32_t nk = g.klen / kGemmBK;

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const uint16_t* gA = g.A + int64_t(m0) * g.lda + kbeg;
  const uint16_t* gB = g.B + int64_t(n0) * g.ldb + kbeg;
  const uint32_t frow = lane & 31u, fk = (lane >> 5) * 8u;

#ifndef MHTE_GEMM_REGSTAGE
  // ---- operand tiles by LDS-DMA (global_load_lds_dwordx4): no staging registers and, above all, no
  // ds_write pass — timing-only builds of the register-staged loop put 37 % of a launch in its
  // ds_write_b128 (13 LDS cycles each, and the store path of a SIMD pair serves one wavefront at a
  // time), 22 % in the barrier, 10 % in the loads.  A wavefront's DMA image is LINEAR (base + lane *
  // 16 bytes = 8 rows of 128 bytes), so rows are unpadded and the bank spread comes from the SOURCE
  // side: 16-byte slot s of row r holds k-chunk s ^ ((r >> 1) & 7) — with the row's parity that gives
  // the 16 rows of every ds_read_b128 lane group 16 different bank quads (MI355X_MICROARCH.md, LDS:
  // groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...).  The DMA of tile k + 1 is issued in front of
  // tile k's MFMAs and drained (vmcnt(0), by the compiler) in front of the barrier that ends them.
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  constexpr int kRows = (BM + BN) / 8;                 // 8-row DMA instructions per tile (A then B)
  constexpr int kPerWave = kRows / (THREADS / 64);     // ... per wavefront: 8
  static_assert(kRows % (THREADS / 64) == 0, "whole instructions per wavefront");
  const uint32_t lrow = lane >> 3, lslot = lane & 7u;
#define MHTE_GEMM_DMA(KT_, BUF_)                                                                            \
  {                                                                                                         \
    _Pragma("unroll") for (int i_ = 0; i_ < kPerWave; ++i_) {                                               \
      const uint32_t rb_ = (uint32_t(i_) * uint32_t(THREADS / 64) + wave) * 8u; /* first of 8 rows */       \
      const uint32_t r_ = rb_ + lrow;                                                                       \
      const uint32_t kc_ = (lslot ^ ((r_ >> 1) & 7u)) * 8u;                                                 \
      const uint16_t* src_ = r_ < uint32_t(BM) ? gA + int64_t(r_) * g.lda                                   \
                                               : gB + int64_t(r_ - uint32_t(BM)) * g.ldb;                   \
      __builtin_amdgcn_global_load_lds((gptr_t)(src_ + (KT_) * kGemmBK + kc_),                             \
                                       (lptr_t)(&smem[BUF_][rb_ * 64u]), 16, 0, 0);                         \
    }                                                                                                       \
  }
#define MHTE_GEMM_KSTEP(KS_)                                                                                \
  {                                                                                                         \
    bf16x8_t fa[TM], fb[TN];                                                                                \
    const uint32_t c_ = uint32_t(KS_) * 2u + (lane >> 5); /* this lane's k-chunk */                         \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                        \
      const uint32_t r_ = wm * uint32_t(TM * 32) + uint32_t(i) * 32u + frow;                                \
      fa[i] = *reinterpret_cast<const bf16x8_t*>(&smem[buf][r_ * 64u + ((c_ ^ ((r_ >> 1) & 7u)) * 8u)]);    \
    }                                                                                                       \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                        \
      const uint32_t r_ = uint32_t(BM) + wn * uint32_t(TN * 32) + uint32_t(j) * 32u + frow;                 \
      fb[j] = *reinterpret_cast<const bf16x8_t*>(&smem[buf][r_ * 64u + ((c_ ^ ((r_ >> 1) & 7u)) * 8u)]);    \
    }                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                          \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                        \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);              \
  }
  MHTE_GEMM_DMA(0u, 0)
  __syncthreads();
#pragma unroll 1
  for (uint32_t kt = 0; kt < nk; ++kt) {
    const uint32_t buf = kt & 1u;
    if (kt + 1 < nk) MHTE_GEMM_DMA(kt + 1, buf ^ 1u)
    MHTE_GEMM_KSTEP(0)
    MHTE_GEMM_KSTEP(1)
    MHTE_GEMM_KSTEP(2)
    MHTE_GEMM_KSTEP(3)
    __syncthreads();
  }
#undef MHTE_GEMM_DMA
#undef MHTE_GEMM_KSTEP
#else
  // ---- (A/B builds, -DMHTE_GEMM_REGSTAGE: 4 % slower than the DMA form on the tower)
  // global -> registers -> LDS rows padded to 72 elements, one tile of look-ahead (a second one, 32
  // more registers, bought nothing).  4 + 4 16-byte pieces per thread and tile (piece i: row prow +
  // RPP i, k offset pk: eight consecutive threads fetch one 128-byte row segment).  (Named registers
  // and macros, not arrays captured by a lambda: those ended up in scratch memory.)
  uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  const uint32_t prow = t >> 3, pk = (t & 7u) * 8u;
#define MHTE_GEMM_FETCH1(I_, RA_, RB_)                                                                          \
  RA_ = *reinterpret_cast<const uint4*>(gA + int64_t(prow + uint32_t(RPP) * (I_)) * g.lda + kt_ * kGemmBK + pk); \
  RB_ = *reinterpret_cast<const uint4*>(gB + int64_t(prow + uint32_t(RPP) * (I_)) * g.ldb + kt_ * kGemmBK + pk);
#define MHTE_GEMM_FETCH(KT_)            \
  {                                     \
    const uint32_t kt_ = (KT_);         \
    MHTE_GEMM_FETCH1(0u, ra0, rb0)      \
    MHTE_GEMM_FETCH1(1u, ra1, rb1)      \
    MHTE_GEMM_FETCH1(2u, ra2, rb2)      \
    MHTE_GEMM_FETCH1(3u, ra3, rb3)      \
  }
#define MHTE_GEMM_STAGE1(BUF_, I_, RA_, RB_)                                                                      \
  *reinterpret_cast<uint4*>(&smem[BUF_][(prow + uint32_t(RPP) * (I_)) * kGemmLdsRow + pk]) = RA_;                 \
  *reinterpret_cast<uint4*>(&smem[BUF_][(uint32_t(BM) + prow + uint32_t(RPP) * (I_)) * kGemmLdsRow + pk]) = RB_;
#define MHTE_GEMM_STAGE(BUF_)              \
  {                                        \
    MHTE_GEMM_STAGE1(BUF_, 0u, ra0, rb0)   \
    MHTE_GEMM_STAGE1(BUF_, 1u, ra1, rb1)   \
    MHTE_GEMM_STAGE1(BUF_, 2u, ra2, rb2)   \
    MHTE_GEMM_STAGE1(BUF_, 3u, ra3, rb3)   \
  }
  MHTE_GEMM_FETCH(0u)
  MHTE_GEMM_STAGE(0)
  __syncthreads();
#define MHTE_GEMM_KSTEP(KS_)                                                                                          \
  {                                                                                                                   \
    bf16x8_t fa[TM], fb[TN];                                                                                          \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(                        \
        &smem[buf][(wm * uint32_t(TM * 32) + uint32_t(i) * 32u + frow) * kGemmLdsRow + uint32_t(KS_) * 16u + fk]);    \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(                        \
        &smem[buf][(uint32_t(BM) + wn * uint32_t(TN * 32) + uint32_t(j) * 32u + frow) * kGemmLdsRow +                 \
                   uint32_t(KS_) * 16u + fk]);                                                                        \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                    \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                                  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);                        \
  }
  // (tried: the next tile's LDS writes between this tile's k-steps instead of behind them, pinned
  // with sched_barrier — no change; a second tile of look-ahead in registers — no change.  SQ
  // counters of the 256 x 256 form: matrix pipe busy 30 % of the time, wavefronts parked on waitcnt /
  // barrier 34 %, issue-stalled 52 %, LDS 18 % busy, no bank conflict.)
#pragma unroll 1
  for (uint32_t kt = 0; kt < nk; ++kt) {
    const uint32_t buf = kt & 1u;
    if (kt + 1 < nk) MHTE_GEMM_FETCH(kt + 1)   // in flight behind this tile's MFMAs
    MHTE_GEMM_KSTEP(0)
    MHTE_GEMM_KSTEP(1)
    MHTE_GEMM_KSTEP(2)
    MHTE_GEMM_KSTEP(3)
    if (kt + 1 < nk) MHTE_GEMM_STAGE(buf ^ 1u)
    __syncthreads();
  }
#undef MHTE_GEMM_KSTEP
#undef MHTE_GEMM_FETCH
#undef MHTE_GEMM_FETCH1
#undef MHTE_GEMM_STAGE
#undef MHTE_GEMM_STAGE1

#endif

  // ---- epilogue.  bf16 outputs (forward, dgrad) leave through LDS: the accumulator layout gives a
  // lane one column and 4-row pieces of it, i.e. 2-byte row-major stores and scattered 8-byte
  // transposed ones — measured 113-184 TFLOP/s on the short-K dgrads.  Each wavefront stages 64 x 64
  // parts of its own tile in its own slice of the (now free) operand buffers — row-major T and
  // transposed Tt — and moves both outputs, and the ReLU mask of dgrad on the way in, in 16-byte
  // pieces, 8 consecutive lanes per 128-byte row segment.  No workgroup barrier: a wavefront's LDS
  // accesses complete in order.
  const uint32_t col_l = lane & 31u, rhalf = (lane >> 5) * 4u;
  const uint32_t mw = m0 + wm * uint32_t(TM * 32), nw = n0 + wn * uint32_t(TN * 32);
  if (EPI == kEpiFwd || EPI == kEpiDgrad) {
    constexpr int kTRow = 72;   // 64 + 8
    static_assert(2 * 64 * kTRow * (THREADS / 64) <= 2 * (BM + BN) * kGemmLdsRow, "staging slices fit");
    uint16_t* const T = &smem[0][0] + wave * uint32_t(2 * 64 * kTRow);
    uint16_t* const Tt = T + 64 * kTRow;
#pragma unroll
    for (int jp = 0; jp < TN / 2; ++jp) {
      const uint32_t np = nw + uint32_t(jp) * 64u;   // first column of the part
      if (EPI == kEpiDgrad) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const uint32_t p = lane + uint32_t(i) * 64u, row = p >> 3, cp = (p & 7u) * 8u;
          *reinterpret_cast<uint4*>(&T[row * kTRow + cp]) =
              *reinterpret_cast<const uint4*>(g.mask + int64_t(mw + row) * g.ldc + np + cp);
        }
        gemm_wave_lds_sync();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              const uint32_t ml = uint32_t(i) * 32u + rhalf + uint32_t(q >> 2) * 8u + uint32_t(q & 3);
              const uint16_t hm = T[ml * kTRow + uint32_t(j2) * 32u + col_l];
              if (!(bf16_to_f32(hm) > 0.f)) acc[i][jp * 2 + j2][q] = 0.f;
            }
        gemm_wave_lds_sync();
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) {
          const uint32_t nl = uint32_t(j2) * 32u + col_l;
          float bias = 0.f;
          if (EPI == kEpiFwd) bias = g.bias[np + nl];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t ml = uint32_t(i) * 32u + rhalf + uint32_t(q) * 8u;
            uint16_t h[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float v = acc[i][jp * 2 + j2][q * 4 + r];
              if (EPI == kEpiFwd) {
                v += bias;
                if (g.relu) v = fmaxf(v, 0.f);
              }
              h[r] = f32_to_bf16(v);
              T[(ml + uint32_t(r)) * kTRow + nl] = h[r];
            }
            uint2 pk2;
            pk2.x = uint32_t(h[0]) | (uint32_t(h[1]) << 16);
            pk2.y = uint32_t(h[2]) | (uint32_t(h[3]) << 16);
            *reinterpret_cast<uint2*>(&Tt[nl * kTRow + ml]) = pk2;
          }
        }
      gemm_wave_lds_sync();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t p = lane + uint32_t(i) * 64u, row = p >> 3, cp = (p & 7u) * 8u;
        *reinterpret_cast<uint4*>(g.C + int64_t(mw + row) * g.ldc + np + cp) =
            *reinterpret_cast<const uint4*>(&T[row * kTRow + cp]);
        if (g.Ct)
          *reinterpret_cast<uint4*>(g.Ct + int64_t(np + row) * g.ldct + mw + cp) =
              *reinterpret_cast<const uint4*>(&Tt[row * kTRow + cp]);
      }
      gemm_wave_lds_sync();   // (the next part overwrites T / Tt)
    }
  } else {
    // fp32 outputs straight from the accumulators (a row segment of 32 columns per half-wavefront)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const uint32_t n = nw + uint32_t(j) * 32u + col_l;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const uint32_t m = mw + uint32_t(i) * 32u + rhalf + uint32_t(q >> 2) * 8u + uint32_t(q & 3);
          if (EPI == kEpiDgradF32) g.Cf[int64_t(m) * g.ldcf + n] = acc[i][j][q];
          else g.Cf[(int64_t(z) * g.M + m) * g.ldcf + n] = acc[i][j][q];
        }
      }
  }
}

// ---- the MLP's other kernels (bandwidth work around the GEMMs)

// x fp32 [B][K] -> bf16 [B][K] and its transpose [K][B]; 64 x 64 tiles through LDS, every global
// access a 16-byte (input) or 8- / 16-byte (outputs) piece of a contiguous row segment
__global__ __launch_bounds__(256) void mlp_cast_transpose_kernel(const float* __restrict__ x, uint16_t* __restrict__ xb,
                                                                 uint16_t* __restrict__ xt, uint32_t B, uint32_t K) {
  __shared__ uint16_t tile[64][72];   // [b][k]; 144-byte rows
  const uint32_t b0 = blockIdx.y * 64u, k0 = blockIdx.x * 64u, t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t p = t + uint32_t(i) * 256u, row = p >> 4, c4 = (p & 15u) * 4u;
    const float4 v = *reinterpret_cast<const float4*>(x + int64_t(b0 + row) * K + k0 + c4);
    uint2 pk;
    pk.x = uint32_t(f32_to_bf16(v.x)) | (uint32_t(f32_to_bf16(v.y)) << 16);
    pk.y = uint32_t(f32_to_bf16(v.z)) | (uint32_t(f32_to_bf16(v.w)) << 16);
    *reinterpret_cast<uint2*>(xb + int64_t(b0 + row) * K + k0 + c4) = pk;
    *reinterpret_cast<uint2*>(&tile[row][c4]) = pk;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const uint32_t p = t + uint32_t(i) * 256u, krow = p >> 3, bp = (p & 7u) * 8u;
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      w[e] = uint32_t(tile[bp + 2 * e][krow]) | (uint32_t(tile[bp + 2 * e + 1][krow]) << 16);
    *reinterpret_cast<uint4*>(xt + int64_t(k0 + krow) * B + b0 + bp) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// last layer (one output): y[m] = b + sum_k h[m][k] w[k]; 32 lanes per row, 8 elements per lane and trip
__global__ __launch_bounds__(256) void mlp_rowdot_kernel(const uint16_t* __restrict__ h, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ y,
                                                         uint32_t B, uint32_t K) {
  const uint32_t row = blockIdx.x * 8u + (threadIdx.x >> 5), l = threadIdx.x & 31u;
  if (row >= B) return;
  float s = 0.f;
  for (uint32_t k = l * 8u; k < K; k += 256u) {
    const uint4 v = *reinterpret_cast<const uint4*>(h + int64_t(row) * K + k);
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      s += bf16_to_f32(uint16_t(u[q] & 0xffffu)) * w[k + 2 * q];
      s += bf16_to_f32(uint16_t(u[q] >> 16)) * w[k + 2 * q + 1];
    }
  }
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if (l == 0) y[row] = s + b[0];
}

// backward of the last layer: dz[m][k] = h[m][k] > 0 ? dy[m] w[k] : 0 (bf16, and transposed);
// partial sums of dW[k] = sum_m dy[m] h[m][k] and db = sum_m dy[m] per block of 64 rows:
// part [K + 1][nblk], summed in a fixed order by mlp_sgd_vec_kernel (deterministic).  16-byte pieces.
__global__ __launch_bounds__(256) void mlp_last_bwd_kernel(const float* __restrict__ dy, const uint16_t* __restrict__ h,
                                                           const float* __restrict__ w, uint16_t* __restrict__ dz,
                                                           uint16_t* __restrict__ dzt, float* __restrict__ part,
                                                           uint32_t B, uint32_t K) {
  __shared__ uint16_t tile[64][72];   // dz [b][k]
  __shared__ float sdw[32][65];
  const uint32_t b0 = blockIdx.y * 64u, k0 = blockIdx.x * 64u, t = threadIdx.x;
  const uint32_t c8 = (t & 7u) * 8u;        // this thread's 8 columns, rows (t >> 3) and (t >> 3) + 32
  float wk[8], dw[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    wk[e] = w[k0 + c8 + uint32_t(e)];
    dw[e] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const uint32_t row = (t >> 3) + uint32_t(i) * 32u;
    const float d = dy[b0 + row];
    const uint4 hv = *reinterpret_cast<const uint4*>(h + int64_t(b0 + row) * K + k0 + c8);
    const uint32_t u[4] = {hv.x, hv.y, hv.z, hv.w};
    uint32_t o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float h0 = bf16_to_f32(uint16_t(u[q] & 0xffffu)), h1 = bf16_to_f32(uint16_t(u[q] >> 16));
      dw[2 * q] += d * h0;
      dw[2 * q + 1] += d * h1;
      o[q] = uint32_t(f32_to_bf16(h0 > 0.f ? d * wk[2 * q] : 0.f)) |
             (uint32_t(f32_to_bf16(h1 > 0.f ? d * wk[2 * q + 1] : 0.f)) << 16);
    }
    const uint4 ov = make_uint4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<uint4*>(dz + int64_t(b0 + row) * K + k0 + c8) = ov;
    *reinterpret_cast<uint4*>(&tile[row][c8]) = ov;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) sdw[t >> 3][c8 + uint32_t(e)] = dw[e];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const uint32_t p = t + uint32_t(i) * 256u, krow = p >> 3, bp = (p & 7u) * 8u;
    uint32_t wv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      wv[e] = uint32_t(tile[bp + 2 * e][krow]) | (uint32_t(tile[bp + 2 * e + 1][krow]) << 16);
    *reinterpret_cast<uint4*>(dzt + int64_t(k0 + krow) * B + b0 + bp) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
  }
  if (t < 64) {
    float s = 0.f;
    for (int r = 0; r < 32; ++r) s += sdw[r][t];     // (fixed order)
    part[int64_t(k0 + t) * gridDim.y + blockIdx.y] = s;
    if (blockIdx.x == 0 && t == 0) {
      float sb = 0.f;
      for (uint32_t r = 0; r < 64u; ++r) sb += dy[b0 + r];
      part[int64_t(K) * gridDim.y + blockIdx.y] = sb;
    }
  }
}

// db[n] = sum_m dz[m][n] from the transposed copy dzt [N][B] (one block per n; fixed order)
__global__ __launch_bounds__(256) void mlp_bias_grad_kernel(const uint16_t* __restrict__ dzt, float* __restrict__ db,
                                                            uint32_t B) {
  __shared__ float red[256];
  const uint16_t* row = dzt + int64_t(blockIdx.x) * B;
  float s = 0.f;
  for (uint32_t m = threadIdx.x * 8u; m < B; m += 2048u) {
    const uint4 v = *reinterpret_cast<const uint4*>(row + m);
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) s += bf16_to_f32(uint16_t(u[q] & 0xffffu)) + bf16_to_f32(uint16_t(u[q] >> 16));
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (uint32_t o = 128; o >= 1; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) db[blockIdx.x] = red[0];
}

// SGD on the fp32 master weights of one layer from the split-K slabs of its weight gradient, and the
// two bf16 copies the next step's GEMMs read (W [N][K] and W^T [K][N]); bias from db
__global__ __launch_bounds__(256) void mlp_sgd_kernel(float* __restrict__ w, float* __restrict__ b,
                                                      const float* __restrict__ slabs, uint32_t nsplit,
                                                      const float* __restrict__ db, float lr, uint16_t* __restrict__ wb,
                                                      uint16_t* __restrict__ wt, uint32_t N, uint32_t K) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < uint64_t(N) * K) {
    float gsum = 0.f;
    for (uint32_t s = 0; s < nsplit; ++s) gsum += slabs[uint64_t(s) * N * K + i];
    const float v = w[i] - lr * gsum;
    w[i] = v;
    const uint16_t h = f32_to_bf16(v);
    wb[i] = h;
    const uint32_t n = uint32_t(i / K), k = uint32_t(i % K);
    wt[uint64_t(k) * N + n] = h;
  }
  if (i < N) b[i] -= lr * db[i];
}
// ... of the last layer's weight vector from mlp_last_bwd_kernel's per-block partial sums
// part [K + 1][nblk]: one workgroup per element (K = the bias), a fixed-order tree over the blocks
__global__ __launch_bounds__(256) void mlp_sgd_vec_kernel(float* __restrict__ w, float* __restrict__ b,
                                                          const float* __restrict__ part, uint32_t nblk, float lr,
                                                          uint32_t K) {
  __shared__ float red[256];
  const uint32_t k = blockIdx.x;
  float s = 0.f;
  for (uint32_t i = threadIdx.x; i < nblk; i += 256u) s += part[uint64_t(k) * nblk + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (uint32_t o = 128; o >= 1; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (k < K) w[k] -= lr * red[0];
    else b[0] -= lr * red[0];
  }
}
// bf16 copies of a layer's fp32 weights (after set_params)
__global__ __launch_bounds__(256) void mlp_refresh_kernel(const float* __restrict__ w, uint16_t* __restrict__ wb,
                                                          uint16_t* __restrict__ wt, uint32_t N, uint32_t K) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= uint64_t(N) * K) return;
  const uint16_t h = f32_to_bf16(w[i]);
  wb[i] = h;
  wt[uint64_t(i % K) * N + uint32_t(i / K)] = h;
}

}  // namespace mhte
#endif  // MHTE_GEMM_KERNELS_H_
