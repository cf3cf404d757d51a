// The step right after the embedding exchange: gathering rows of the fused
// embedding buffer per merged slot, its gradient, and the ragged reductions of the combiners.
// Reference: runtime/ops/map_id_to_embedding.cu.cc:31-118 (FusedGatherEmbeddingsByInput and its
// gradient, CUDA kernels), runtime/ops/reduce_op.cc:29-125 (ReduceSum / ReduceMean /
// ReduceSquareNorm), native_training/embedding_combiners.py:41-102.  Included by mhte.hip.
//
// All three are HBM-streaming copy/add work: one group of lanes per output row moving float4s
// (scalar forms for dims / offsets that are not multiples of 4 floats).
#ifndef MHTE_POOL_KERNELS_H_
#define MHTE_POOL_KERNELS_H_

#include "mhte_kernels.h"

namespace mhte {

constexpr int kMaxGatherInputs = 32;  // inputs per launch (more: several launches)

struct GatherInputs {
  const int32_t* offsets[kMaxGatherInputs];  // [n_i] float offset of row j in the fused buffer
  float* rows[kMaxGatherInputs];             // [n_i, dim_i] output (gather) / gradient (scatter)
  int64_t start[kMaxGatherInputs + 1];       // row-index prefix over the inputs
  int32_t dim[kMaxGatherInputs];
  int32_t n_inputs;
  int32_t aligned;  // fused and every rows[i] are 16-byte aligned
};

// MonolithFusedGatherEmbeddingsByInput: outputs[i][j, :] = fused[offsets[i][j] + 0 .. dim_i)
// (map_id_to_embedding.cu.cc:31-72).  One thread per element would re-read the offset dim times;
// here a lane group takes one row.  GATHER = false: the gradient,
// fused_grad[offsets[i][j] + k] += grads[i][j, k] * scale (:74-118) — float atomics, like the
// reference's GpuAtomicAdd (rows that share an offset are added in arrival order).
// A row whose offset and dim are multiples of 4 floats moves as float4s (in.aligned: the host
// found every base pointer 16-byte aligned); any other row takes the scalar loop.
// MonolithHashTableLookupGradient (RT/ops/hash_table_lookup_op.cc:110-147): out_ids[i] = id_values[i],
// out_grads[i, :] = input_grads[id_indices[i, 0], :] — the gradient of the embedding of (batch row,
// id) pairs gathered back by batch row; id_indices is the [n, index_cols] index matrix of a sparse
// tensor, column 0 = the batch row.  A lane group of 8 per output row (float4s when dim and the
// buffers allow, else one float per lane).  A row index outside [0, n_rows) reads zeros and raises a
// flag the host turns into InvalidArgument (the reference indexes without a check).
__global__ __launch_bounds__(256) void lookup_gradient_kernel(const int64_t* __restrict__ id_indices,
                                                              int64_t n, int64_t index_cols,
                                                              const int64_t* __restrict__ id_values,
                                                              const float* __restrict__ input_grads,
                                                              int64_t n_rows, int32_t dim, int32_t vec4,
                                                              int64_t* __restrict__ out_ids,
                                                              float* __restrict__ out_grads,
                                                              uint32_t* __restrict__ bad) {
  constexpr int G = 8;
  const int j = threadIdx.x & (G - 1);
  for (int64_t i = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G; i < n;
       i += int64_t(gridDim.x) * blockDim.x / G) {
    const int64_t row = id_indices[i * index_cols];
    if (j == 0) out_ids[i] = id_values[i];
    const bool ok = row >= 0 && row < n_rows;
    if (!ok && j == 0) atomicOr(bad, 1u);
    const float* src = input_grads + (ok ? row : 0) * int64_t(dim);
    float* dst = out_grads + i * int64_t(dim);
    if (vec4) {
      for (int k = j * 4; k < dim; k += G * 4) {
        Vec<4> v;
        v.v[0] = v.v[1] = v.v[2] = v.v[3] = 0.f;
        if (ok) v.load(src + k);
        v.store(dst + k);
      }
    } else {
      for (int k = j; k < dim; k += G) dst[k] = ok ? src[k] : 0.f;
    }
  }
}

template <bool GATHER>
__global__ __launch_bounds__(256) void fused_gather_kernel(float* __restrict__ fused, GatherInputs in,
                                                           float scale) {
  constexpr int G = 8;
  const int j = threadIdx.x & (G - 1);
  const int64_t r = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  if (r >= in.start[in.n_inputs]) return;
  int i = 0;
  while (in.start[i + 1] <= r) ++i;  // n_inputs < 100: linear search, as the reference does
  const int64_t local = r - in.start[i];
  const int32_t dim = in.dim[i];
  const int64_t off = in.offsets[i][local];
  float* row = in.rows[i] + local * dim;
  // (the gradient keeps the scalar loop: neighbouring lanes on neighbouring floats is the shape
  // the atomic units coalesce; four floats per lane halves its rate — measured)
  if (GATHER && in.aligned && ((off | int64_t(dim)) & 3) == 0) {
    for (int k = j * 4; k < dim; k += G * 4) {
      Vec<4> v;
      v.load(fused + off + k);
      v.store(row + k);
    }
    return;
  }
  for (int k = j; k < dim; k += G) {
    if (GATHER) {
      row[k] = fused[off + k];
    } else {
      atomicAdd(&fused[off + k], row[k] * scale);
    }
  }
}

// ReduceSum / ReduceMean / ReduceSquareNorm over sorted row indices (reduce_op.cc:29-125):
// out[b, :] = reduce over {i : indices[i] == b} of values[i, :], IN ORDER of i — the reference's
// sequential accumulation, bit for bit.  One lane group per output row; its segment is found by
// binary search.  mode 0 sum, 1 mean (sum * (1 / count), count 0 -> the reference's 0 * inf),
// 2 square norm (sqrt of the sum of squares).
template <int VEC>
__global__ __launch_bounds__(256) void reduce_rows_sorted_kernel(const int64_t* __restrict__ indices,
                                                                 const float* __restrict__ values,
                                                                 int64_t n, int32_t dim,
                                                                 int64_t batch, int32_t mode,
                                                                 float* __restrict__ out) {
  constexpr int G = 16;
  const int j = threadIdx.x & (G - 1);
  const int64_t b = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  if (b >= batch) return;
  int64_t lo = 0, hi = n;  // first i with indices[i] >= b
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (indices[mid] < b) lo = mid + 1; else hi = mid;
  }
  const int64_t s0 = lo;
  hi = n;                  // first i with indices[i] > b
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (indices[mid] <= b) lo = mid + 1; else hi = mid;
  }
  const int64_t s1 = lo;
  const float mult = 1.0f / static_cast<float>(s1 - s0);
  for (int k = j * VEC; k < dim; k += G * VEC) {
    Vec<VEC> acc;
    vec_zero(acc);
    // 4 rows in flight, added in order
    for (int64_t i = s0; i < s1; i += 4) {
      Vec<VEC> v[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) vec_zero(v[t]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (i + t < s1) v[t].load(values + (i + t) * dim + k);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (i + t < s1) {
#pragma unroll
          for (int c = 0; c < VEC; ++c) {
            const float x = v[t].v[c];
            acc.v[c] = acc.v[c] + (mode == 2 ? x * x : x);
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      if (mode == 1) acc.v[c] = acc.v[c] * mult;
      if (mode == 2) acc.v[c] = sqrtf(acc.v[c]);
    }
    acc.store(out + b * int64_t(dim) + k);
  }
}

// The same for indices in any order: atomics (sums of equal rows in arrival order), then a
// finishing pass for mean / square norm.
__global__ __launch_bounds__(256) void reduce_rows_atomic_kernel(const int64_t* __restrict__ indices,
                                                                 const float* __restrict__ values,
                                                                 int64_t n, int32_t dim, int32_t mode,
                                                                 float* __restrict__ out,
                                                                 uint32_t* __restrict__ count) {
  constexpr int G = 16;
  const int j = threadIdx.x & (G - 1);
  const int64_t i = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  if (i >= n) return;
  const int64_t b = indices[i];
  if (j == 0 && mode == 1) atomicAdd(&count[b], 1u);
  for (int k = j; k < dim; k += G) {
    const float v = values[i * dim + k];
    atomicAdd(&out[b * int64_t(dim) + k], mode == 2 ? v * v : v);
  }
}
__global__ __launch_bounds__(256) void reduce_rows_finish_kernel(float* __restrict__ out,
                                                                 const uint32_t* __restrict__ count,
                                                                 int64_t batch, int32_t dim,
                                                                 int32_t mode) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= batch * dim) return;
  if (mode == 1) out[t] = out[t] * (1.0f / static_cast<float>(count[t / dim]));
  if (mode == 2) out[t] = sqrtf(out[t]);
}

// ---- deterministic forms (no float atomics): the rows that share a destination are found with the
// list-building dedup (DedupWs::unique: distinct keys in first-occurrence order, each key's positions
// in ascending order) and ONE lane group per destination adds them in position order — for the
// reductions that is the reference's sequential loop (reduce_op.cc:46-49,77-81,110-116), bit for bit,
// for indices in any order.
template <int VEC>
__global__ __launch_bounds__(256) void reduce_rows_lists_kernel(const int64_t* __restrict__ keys,
                                                                const uint32_t* __restrict__ n_keys,
                                                                const uint32_t* __restrict__ seg_off,
                                                                const uint32_t* __restrict__ seg_pos,
                                                                const float* __restrict__ values,
                                                                int32_t dim, int64_t batch, int32_t mode,
                                                                float* __restrict__ out) {
  constexpr int G = 16;
  const int j = threadIdx.x & (G - 1);
  const int64_t u = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  if (u >= int64_t(*n_keys)) return;
  const int64_t b = keys[u];
  if (b < 0 || b >= batch) return;
  const uint32_t s0 = seg_off[u], s1 = seg_off[u + 1];
  const float mult = 1.0f / static_cast<float>(s1 - s0);
  for (int k = j * VEC; k < dim; k += G * VEC) {
    Vec<VEC> acc;
    vec_zero(acc);
    for (uint32_t i = s0; i < s1; i += 4) {   // 4 rows in flight, added in order
      Vec<VEC> v[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) vec_zero(v[t]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (i + t < s1) v[t].load(values + int64_t(seg_pos[i + t]) * dim + k);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (i + t < s1) {
#pragma unroll
          for (int c = 0; c < VEC; ++c) {
            const float x = v[t].v[c];
            acc.v[c] = acc.v[c] + (mode == 2 ? x * x : x);
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      if (mode == 1) acc.v[c] = acc.v[c] * mult;
      if (mode == 2) acc.v[c] = sqrtf(acc.v[c]);
    }
    acc.store(out + b * int64_t(dim) + k);
  }
}

// gradient of FusedGatherEmbeddingsByInput without atomics: key of global row r = its float offset
__global__ __launch_bounds__(256) void gather_keys_kernel(GatherInputs in, int64_t* __restrict__ keys) {
  const int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= in.start[in.n_inputs]) return;
  int i = 0;
  while (in.start[i + 1] <= r) ++i;
  keys[r] = in.offsets[i][r - in.start[i]];
}
// one lane group per distinct offset: its rows (of whichever inputs) added in row order, scaled as the
// reference scales every addend (map_id_to_embedding.cu.cc:98-107), then added to fused_grad (plain
// read-modify-write: one group owns the destination; launches of further input chunks follow in
// stream order)
// ... the same with 16 lanes per key moving float4s (every input's rows and the fused buffer 16-byte
// aligned, every dim a multiple of 4 — the host checks): the scalar form below walks a row 8 floats at a
// time and looks the input of every addend up in the kernel-argument table again for every pass (219 us for
// 1 M keys of one row each); here the input table sits in LDS, a lane keeps one float4 column of the row, and
// the addends of a key are fetched four at a time.  Same arithmetic: acc = acc + x * scale in row order, then
// one read-modify-write of the destination.
constexpr int kGatherGradKeys = 4;   // keys a lane group works on at once (their round trips overlap)
// fresh: the fused buffer is known to be all zeros (the entry point has just filled it and this is its first
// launch): a destination is then stored without being read first — 0 + acc is acc bit for bit, a sum that
// started as 0 + x is never -0 — a third of the launch's bytes (round 6: 126 -> ~100 us for 1 M keys).
__global__ __launch_bounds__(256) void gather_grad_lists_vec_kernel(float* __restrict__ fused, GatherInputs in,
                                                                    float scale,
                                                                    const int64_t* __restrict__ keys,
                                                                    const uint32_t* __restrict__ n_keys,
                                                                    const uint32_t* __restrict__ seg_off,
                                                                    const uint32_t* __restrict__ seg_pos,
                                                                    int fresh) {
  constexpr int G = 16, K = kGatherGradKeys;
  __shared__ long long s_start[kMaxGatherInputs + 1];
  __shared__ const float* s_rows[kMaxGatherInputs];
  __shared__ int s_dim[kMaxGatherInputs];
  if (threadIdx.x <= uint32_t(in.n_inputs)) s_start[threadIdx.x] = in.start[threadIdx.x];
  if (threadIdx.x < uint32_t(in.n_inputs)) {
    s_rows[threadIdx.x] = in.rows[threadIdx.x];
    s_dim[threadIdx.x] = in.dim[threadIdx.x];
  }
  __syncthreads();
  const int j = threadIdx.x & (G - 1);
  const int64_t nk = int64_t(*n_keys);
  const int n_in = in.n_inputs;
  auto input_of = [&](long long r) {
    int i = 0;
    while (i + 1 < n_in && s_start[i + 1] <= r) ++i;
    return i;
  };
  // group g of the grid takes keys g * K .. g * K + K - 1: a wavefront that handled one key per lane group
  // lived for four dependent round trips and moved 4 x 256 bytes (206 us for 1 M keys: 32 rounds of
  // wavefronts); with K keys per group the same round trips carry K times the rows
  const int64_t u0 = ((int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G) * K;
  if (u0 >= nk) return;
  int64_t off[K];
  uint32_t s0[K], s1[K];
#pragma unroll
  for (int t = 0; t < K; ++t) {   // round trip 1: key and list bounds of the K keys
    const int64_t u = min(u0 + t, nk - 1);
    off[t] = keys[u];
    s0[t] = seg_off[u];
    s1[t] = (u0 + t < nk) ? seg_off[u + 1] : s0[t];   // (a key past the end: an empty list)
  }
  long long r0[K];
#pragma unroll
  for (int t = 0; t < K; ++t) r0[t] = seg_pos[s0[t] < s1[t] ? s0[t] : 0u];   // round trip 2: first row of each
  int i0[K], dim[K];
  bool fast[K];   // one addend, aligned destination: the common shape — everything of the K keys in flight
#pragma unroll
  for (int t = 0; t < K; ++t) {
    i0[t] = input_of(r0[t]);
    dim[t] = s_dim[i0[t]];
    fast[t] = s1[t] == s0[t] + 1u && (off[t] & 3) == 0 && dim[t] <= G * 4;
  }
  Vec<4> v[K], f[K];
#pragma unroll
  for (int t = 0; t < K; ++t) {   // round trip 3: the row and the destination of every fast key
    vec_zero(v[t]);
    vec_zero(f[t]);
    if (fast[t] && j * 4 < dim[t]) {
      v[t].load(s_rows[i0[t]] + (r0[t] - s_start[i0[t]]) * dim[t] + j * 4);
      if (!fresh) f[t].load(fused + off[t] + j * 4);
    }
  }
#pragma unroll
  for (int t = 0; t < K; ++t) {
    if (fast[t] && j * 4 < dim[t]) {
#pragma unroll
      for (int c = 0; c < 4; ++c) f[t].v[c] = f[t].v[c] + (0.f + v[t].v[c] * scale);
      f[t].store(fused + off[t] + j * 4);
    }
  }
  // the other keys (several addends, an unaligned destination, a row wider than the group): one at a time
#pragma unroll 1
  for (int t = 0; t < K; ++t) {
    if (fast[t] || s0[t] >= s1[t]) continue;
    const int64_t o = off[t];
    const int d0 = dim[t];
    if (o & 3) {   // a destination that is not 16-byte aligned: one float per lane
      for (int k = j; k < d0; k += G) {
        float acc = 0.f;
        for (uint32_t q = s0[t]; q < s1[t]; ++q) {
          const long long r = seg_pos[q];
          const int i = input_of(r);
          if (k < s_dim[i]) acc += s_rows[i][(r - s_start[i]) * s_dim[i] + k] * scale;
        }
        fused[o + k] += acc;
      }
      continue;
    }
    for (int k = j * 4; k < d0; k += G * 4) {
      Vec<4> acc;
      vec_zero(acc);
      for (uint32_t q = s0[t]; q < s1[t]; q += 4) {
        Vec<4> w[4];
        bool has[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          vec_zero(w[x]);
          has[x] = false;
          if (q + x < s1[t]) {
            const long long r = seg_pos[q + x];
            const int i = input_of(r);
            const int d = s_dim[i];
            if (k < d) {
              w[x].load(s_rows[i] + (r - s_start[i]) * d + k);
              has[x] = true;
            }
          }
        }
#pragma unroll
        for (int x = 0; x < 4; ++x)
          if (has[x]) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc.v[c] = acc.v[c] + w[x].v[c] * scale;
          }
      }
      Vec<4> g2;
      vec_zero(g2);
      if (!fresh) g2.load(fused + o + k);
#pragma unroll
      for (int c = 0; c < 4; ++c) g2.v[c] = g2.v[c] + acc.v[c];
      g2.store(fused + o + k);
    }
  }
}
__global__ __launch_bounds__(256) void gather_grad_lists_kernel(float* __restrict__ fused, GatherInputs in,
                                                                float scale,
                                                                const int64_t* __restrict__ keys,
                                                                const uint32_t* __restrict__ n_keys,
                                                                const uint32_t* __restrict__ seg_off,
                                                                const uint32_t* __restrict__ seg_pos) {
  constexpr int G = 8;
  const int j = threadIdx.x & (G - 1);
  const int64_t u = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  if (u >= int64_t(*n_keys)) return;
  const int64_t off = keys[u];
  const uint32_t s0 = seg_off[u], s1 = seg_off[u + 1];
  int32_t dim = 0;
  {
    const int64_t r = seg_pos[s0];
    int i = 0;
    while (in.start[i + 1] <= r) ++i;
    dim = in.dim[i];
  }
  for (int k = j; k < dim; k += G) {
    float acc = 0.f;
    for (uint32_t q = s0; q < s1; ++q) {
      const int64_t r = seg_pos[q];
      int i = 0;
      while (in.start[i + 1] <= r) ++i;
      if (k < in.dim[i]) acc += in.rows[i][(r - in.start[i]) * in.dim[i] + k] * scale;
    }
    fused[off + k] += acc;
  }
}

}  // namespace mhte
#endif  // MHTE_POOL_KERNELS_H_
