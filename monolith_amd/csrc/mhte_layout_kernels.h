// fused_embedding_to_layout forward / backward: the pooling that turns looked-up embeddings into
// the dense model's input tensors.  Reference:
//   runtime/ops/fused_embedding_to_layout.h:50-76 (fid / nfl offset encodings, GetFeatureInfo),
//   :204-261 GatherEmb, :286-346 ScatterGrad; the CUDA kernels that call them,
//   runtime/ops/fused_embedding_to_layout.cu.cc:96-199 (ForwardBatchKernel), :337-... (backward);
//   op inputs / outputs runtime/ops/fused_embedding_to_layout.cc:1037-1066; configuration messages
//   idl/matrix/proto/example.proto:176-221.  Included by mhte.hip.
//
// One lane group of 16 takes one (unit, batch row); lanes stride over the slice's elements, the
// feature's fids are walked in order — the reference's loop, so SUM / MEAN are its sequential sums
// bit for bit.  A unit is one slice of a layout, or a whole ADDN layout: its slices are added in
// configuration order by the same group (the reference's GPU path adds them with float atomics in
// whatever order the blocks arrive, its CPU path in this order).
// HBM-streaming copy / add work: nothing here has a GEMM shape.
#ifndef MHTE_LAYOUT_KERNELS_H_
#define MHTE_LAYOUT_KERNELS_H_

#include "mhte_kernels.h"

namespace mhte {

constexpr int kMaxLayoutTasks = 48;   // slices per launch (more: several launches)
constexpr int kMaxLayoutEmb = 64;     // embedding matrices (shard x sub-table) carried in the kernel
constexpr int kMaxLayoutOut = 32;     // arguments, output tensors likewise; more (the reference's own
                                      // test has ~1000 and 153): pointer tables in device memory

enum LayoutPooling : int32_t { kPoolSum = 0, kPoolMean = 1, kPoolFirstN = 3 };  // example.proto:176-180
constexpr int32_t kPoolZeroFill = 100;   // (internal) a run of output columns no slice writes: layout_rows_kernel zeroes it

struct LayoutTask {
  int32_t nfl_idx;       // SliceConfig.feature_idx
  int32_t start, dim;    // slice [start, start + dim) of the feature's embedding
  int32_t pooling;
  int32_t max_seq;       // FIRSTN: rows kept
  int32_t out_index;     // which output tensor
  int32_t out_offset;    // float offset of the slice inside an output row
  int32_t out_stride;    // floats per output row
};
struct LayoutUnit {
  uint16_t first, count; // tasks [first, first + count); count > 1 only for an ADDN layout
  uint16_t addn, pad;    // addn: 1 an ADDN layout, 2 its continuation in a following launch
};
struct LayoutArgs {
  const float* emb[kMaxLayoutEmb];      // forward: embeddings; backward: gradient buffers (written)
  uint32_t emb_stride[kMaxLayoutEmb];   // floats per row (PtrWrapper.offset)
  uint32_t emb_count[kMaxLayoutEmb];    // floats in the matrix (PtrWrapper.count)
  float* out[kMaxLayoutOut];            // forward: outputs; backward: output gradients (read)
  const unsigned long long* fid_offset;
  const int32_t* feature_offset;
  const uint32_t* nfl_offset;
  int32_t n_fid, n_feature, n_nfl, batch, n_emb, n_units;
  int32_t zero_missing;   // layout_rows_kernel<true>: a (slice, row) without a fid is written as zeros (the
                          // output was not zero-filled beforehand: the launch writes every column of it)
  // when the model has more matrices / outputs than the inline arrays hold: the same four tables in
  // device memory (x_emb != nullptr), uploaded by the host for the call
  const float* const* x_emb;
  const uint32_t* x_stride;
  const uint32_t* x_count;
  float* const* x_out;
  LayoutTask task[kMaxLayoutTasks];
  LayoutUnit unit[kMaxLayoutTasks];
};
static_assert(sizeof(LayoutArgs) <= 4096, "kernel arguments exceed 4 KB");

__device__ __forceinline__ const float* layout_emb(const LayoutArgs& A, uint32_t i) {
  return A.x_emb ? A.x_emb[i] : A.emb[i];
}
__device__ __forceinline__ uint32_t layout_stride(const LayoutArgs& A, uint32_t i) {
  return A.x_emb ? A.x_stride[i] : A.emb_stride[i];
}
__device__ __forceinline__ uint32_t layout_count(const LayoutArgs& A, uint32_t i) {
  return A.x_emb ? A.x_count[i] : A.emb_count[i];
}
__device__ __forceinline__ float* layout_out(const LayoutArgs& A, int32_t i) {
  return A.x_emb ? A.x_out[i] : A.out[i];
}

// zero fill of a table of buffers (the op's SetZeroFunctor over every output / gradient buffer)
__global__ __launch_bounds__(256) void layout_zero_kernel(float* const* buf, const uint64_t* len) {
  float* p = buf[blockIdx.y];
  const uint64_t n = len[blockIdx.y];
  for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += uint64_t(gridDim.x) * blockDim.x)
    p[i] = 0.f;
}

// the same for up to kLayoutZeroBufs buffers named in the kernel arguments, 16 bytes per lane where the
// buffer allows it (hipMemsetAsync moved the 268 MB of a configs[4] layout output at 1.3 TB/s, and the
// gradient side took one call per matrix)
constexpr int kLayoutZeroBufs = 64;
struct LayoutZeroArgs {
  float* p[kLayoutZeroBufs];
  uint64_t len[kLayoutZeroBufs];   // floats
};
__global__ __launch_bounds__(256) void layout_zero_args_kernel(LayoutZeroArgs Z) {
  float* p = Z.p[blockIdx.y];
  const uint64_t n = Z.len[blockIdx.y];
  const uint64_t tid = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x, nth = uint64_t(gridDim.x) * blockDim.x;
  if ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
    Vec<4> z;
    vec_zero(z);
    const uint64_t n4 = n >> 2;
    for (uint64_t i = tid; i < n4; i += nth) z.store(p + i * 4);
    for (uint64_t i = (n4 << 2) + tid; i < n; i += nth) p[i] = 0.f;
  } else {
    for (uint64_t i = tid; i < n; i += nth) p[i] = 0.f;
  }
}

// GetFeatureInfo + the feature's fid range for batch row b (fused_embedding_to_layout.h:56-76,
// :214-221); false: the named feature list is absent or the row has no fids
__device__ __forceinline__ bool layout_fid_range(const LayoutArgs& A, int32_t nfl_idx, int32_t b,
                                                 int32_t* f0, int32_t* f1) {
  if (nfl_idx < 0 || nfl_idx >= A.n_nfl) return false;
  const uint32_t enc = A.nfl_offset[nfl_idx];
  const bool shared = enc >> 31;
  const int32_t off = int32_t(enc & 0x7fffffffu);
  const int32_t off_next = nfl_idx < A.n_nfl - 1 ? int32_t(A.nfl_offset[nfl_idx + 1] & 0x7fffffffu)
                                                 : A.n_feature;
  if (off_next - off <= 0) return false;          // "nfl exits"
  const int32_t f = off + (shared ? 0 : b);
  if (f >= A.n_feature) return false;
  *f0 = A.feature_offset[f];
  *f1 = f < A.n_feature - 1 ? A.feature_offset[f + 1] : A.n_fid;
  return *f1 > *f0;
}

// Fast form for the common shape of a recommendation batch after a per-occurrence lookup: every
// feature instance has exactly ONE fid, SUM / MEAN pooling (a copy), every slice boundary, row
// stride and output offset a multiple of 4 floats, and (gradient) no embedding row referenced twice
// — the host checks what it can and the caller asserts the rest (MHTE_LAYOUT_UNIQUE_ROWS).  A lane
// moves float4s; the gradient is a plain store instead of a float atomic per element.
template <bool FORWARD>
__global__ __launch_bounds__(256) void layout_copy_kernel(LayoutArgs A) {
  constexpr int G = 16;
  const int j = threadIdx.x & (G - 1);
  const int32_t b = int32_t((int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G);
  if (b >= A.batch) return;
  const LayoutUnit u = A.unit[blockIdx.y];
  const LayoutTask t = A.task[u.first];
  int32_t f0 = 0, f1 = 0;
  if (!layout_fid_range(A, t.nfl_idx, b, &f0, &f1)) return;
  const unsigned long long fo = A.fid_offset[f0];
  const uint32_t i1 = uint32_t(fo >> 32), i2 = uint32_t(fo);
  if (i1 >= uint32_t(A.n_emb)) return;
  float* orow = layout_out(A, t.out_index) + int64_t(b) * t.out_stride + t.out_offset;
  float* erow = const_cast<float*>(layout_emb(A, i1)) + uint64_t(i2) * layout_stride(A, i1) + uint32_t(t.start);
  for (int32_t e = j * 4; e < t.dim; e += G * 4) {
    Vec<4> v;
    if (FORWARD) {
      v.load(erow + e);
      v.store(orow + e);
    } else {
      v.load(orow + e);
      v.store(erow + e);
    }
  }
}

// The same copies, a workgroup per kLayoutRowsR batch rows over ALL slices of the launch (round 5).
// layout_copy_kernel gives a 16-lane group one (slice, batch row): four dependent round trips
// (nfl_offset -> feature_offset -> fid_offset -> the row) in front of 64-256 bytes, 12 of the 16 lanes idle
// on a dim-16 slice, and the pieces of one output row written by 26 workgroups of different XCDs at
// different times: 2.2-2.5 TB/s of the two streams at configs[4]'s shape.  Here the workgroup first resolves
// its (slice, row) pairs — one thread per pair, every chain in flight at once, source and destination
// addresses into LDS — then walks the float4 columns of its output rows flat, consecutive lanes on
// consecutive columns across slice boundaries: whole 128-byte lines of a CONCAT row leave one workgroup,
// every lane moves 16 bytes, kLayoutRowsUnroll independent loads in flight per lane.
constexpr int kLayoutRowsR = 16;
constexpr int kLayoutRowsMaxF = 2048;   // float4 columns of one batch row over the launch's slices
constexpr int kLayoutRowsUnroll = 8;
template <bool FORWARD>
__global__ __launch_bounds__(256) void layout_rows_kernel(LayoutArgs A) {
  constexpr int R = kLayoutRowsR;
  __shared__ unsigned long long s_src[kMaxLayoutTasks * R];   // embedding side (0: no row)
  __shared__ unsigned long long s_dst[kMaxLayoutTasks * R];   // output side
  __shared__ uint16_t s_cstart[kMaxLayoutTasks + 1];
  __shared__ uint8_t s_ctask[kLayoutRowsMaxF];
  const int nu = A.n_units;
  const int32_t b0 = int32_t(blockIdx.x) * R;
  if (threadIdx.x < 64) {   // float4 columns in front of slice q (an inclusive scan over <= 64 slices)
    const int q = threadIdx.x;
    uint32_t w = q < nu ? uint32_t(A.task[A.unit[q].first].dim) >> 2 : 0u;
    uint32_t incl = w;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = __shfl_up(incl, o);
      if (q >= o) incl += v;
    }
    if (q <= nu) s_cstart[q] = uint16_t(incl - w);
  }
  for (int p = threadIdx.x; p < nu * R; p += 256) {
    const int q = p / R, r = p % R;
    const int32_t b = b0 + r;
    const LayoutTask t = A.task[A.unit[q].first];
    // src: the embedding side; 0 = nothing to move, 1 = zeros (forward: a run of columns no slice writes,
    // or — in a launch that writes whole output rows instead of a zero fill beforehand — a row without a fid)
    unsigned long long src = 0ull, dst = 0ull;
    int32_t f0 = 0, f1 = 0;
    if (b < A.batch) {
      dst = reinterpret_cast<unsigned long long>(layout_out(A, t.out_index) + int64_t(b) * t.out_stride +
                                                 t.out_offset);
      if (FORWARD && t.pooling == kPoolZeroFill) {
        src = 1ull;
      } else if (layout_fid_range(A, t.nfl_idx, b, &f0, &f1)) {
        const unsigned long long fo = A.fid_offset[f0];
        const uint32_t i1 = uint32_t(fo >> 32), i2 = uint32_t(fo);
        if (i1 < uint32_t(A.n_emb))
          src = reinterpret_cast<unsigned long long>(layout_emb(A, i1) + uint64_t(i2) * layout_stride(A, i1) +
                                                     uint32_t(t.start));
      }
      if (!src && FORWARD && A.zero_missing) src = 1ull;
    }
    s_src[p] = src;
    s_dst[p] = dst;
  }
  __syncthreads();
  const uint32_t F = s_cstart[nu];
  for (uint32_t c = threadIdx.x; c < F; c += 256) {   // column -> slice
    uint32_t lo = 0, hi = uint32_t(nu) - 1u;
    while (lo < hi) {
      const uint32_t mid = (lo + hi + 1u) >> 1;
      if (s_cstart[mid] <= c) lo = mid;
      else hi = mid - 1u;
    }
    s_ctask[c] = uint8_t(lo);
  }
  __syncthreads();
  // a thread keeps its float4 column (slice and offset inside it: looked up once) and walks the workgroup's
  // rows, kLayoutRowsUnroll of them in flight; for a given row consecutive lanes are consecutive columns
  for (uint32_t c = threadIdx.x; c < F; c += 256u) {
    const uint32_t q = s_ctask[c];
    const uint32_t e = (c - s_cstart[q]) * 4u;
    const unsigned long long* const srcs = s_src + q * R;
    const unsigned long long* const dsts = s_dst + q * R;
#pragma unroll
    for (int r0 = 0; r0 < R; r0 += kLayoutRowsUnroll) {
      Vec<4> v[kLayoutRowsUnroll];
      float* to[kLayoutRowsUnroll];
#pragma unroll
      for (int u = 0; u < kLayoutRowsUnroll; ++u) {
        const unsigned long long src = srcs[r0 + u], dst = dsts[r0 + u];
        to[u] = nullptr;
        if (src > 1ull) {
          float* const ep = reinterpret_cast<float*>(src) + e;
          float* const op = reinterpret_cast<float*>(dst) + e;
          v[u].load(FORWARD ? ep : op);
          to[u] = FORWARD ? op : ep;
        } else if (FORWARD && src == 1ull) {
          vec_zero(v[u]);
          to[u] = reinterpret_cast<float*>(dst) + e;
        }
      }
#pragma unroll
      for (int u = 0; u < kLayoutRowsUnroll; ++u)
        if (to[u]) v[u].store(to[u]);
    }
  }
}

template <bool FORWARD>
__global__ __launch_bounds__(256) void layout_kernel(LayoutArgs A) {
  constexpr int G = 16;
  const int j = threadIdx.x & (G - 1);
  const int32_t b = int32_t((int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G);
  if (b >= A.batch) return;
  const LayoutUnit u = A.unit[blockIdx.y];
  // elements of an ADDN layout's row, summed over its slices in order (forward)
  for (int32_t e0 = 0; e0 < A.task[u.first].dim || e0 == 0; e0 += G) {
    const int32_t e = e0 + j;
    float acc = 0.f;
    bool any = false;
    if (FORWARD && u.addn == 2 && e < A.task[u.first].dim) {   // continues the previous launch's sums
      const LayoutTask t0 = A.task[u.first];
      acc = layout_out(A, t0.out_index)[int64_t(b) * t0.out_stride + t0.out_offset + e];
      any = true;
    }
    for (uint32_t k = u.first; k < uint32_t(u.first) + u.count; ++k) {
      const LayoutTask t = A.task[k];
      int32_t f0 = 0, f1 = 0;
      if (e >= t.dim || !layout_fid_range(A, t.nfl_idx, b, &f0, &f1)) continue;
      const int32_t fid_num = f1 - f0;
      float* orow = layout_out(A, t.out_index) + int64_t(b) * t.out_stride + t.out_offset;
      float pooled = 0.f;
      int32_t seq = 0;
      for (int32_t q = f0; q < f1; ++q) {
        const unsigned long long fo = A.fid_offset[q];
        const uint32_t i1 = uint32_t(fo >> 32), i2 = uint32_t(fo);
        if (i1 >= uint32_t(A.n_emb)) continue;
        const uint64_t at = uint64_t(i2) * layout_stride(A, i1) + uint32_t(t.start) + uint32_t(e);
        if (at >= layout_count(A, i1)) continue;      // (CUSTOM_CHECK in the reference)
        if (FORWARD) {
          const float x = layout_emb(A, i1)[at];
          if (t.pooling == kPoolFirstN) {
            if (seq < t.max_seq) orow[int64_t(seq) * t.dim + e] = x;
            ++seq;
          } else if (t.pooling == kPoolMean) {
            pooled = (q == f0) ? x / float(fid_num) : pooled + x / float(fid_num);
          } else {
            pooled = (q == f0) ? x : pooled + x;
          }
        } else {
          float* dst = const_cast<float*>(layout_emb(A, i1)) + at;
          if (t.pooling == kPoolFirstN) {
            if (seq < t.max_seq) atomicAdd(dst, orow[int64_t(seq) * t.dim + e]);
            ++seq;
          } else if (t.pooling == kPoolMean) {
            atomicAdd(dst, orow[e] / float(fid_num));
          } else {
            atomicAdd(dst, orow[e]);
          }
        }
      }
      if (FORWARD && t.pooling != kPoolFirstN) {
        if (u.addn) {
          acc = any ? acc + pooled : pooled;
          any = true;
        } else {
          orow[e] = pooled;
        }
      }
    }
    if (FORWARD && u.addn && any) {
      const LayoutTask t = A.task[u.first];
      layout_out(A, t.out_index)[int64_t(b) * t.out_stride + t.out_offset + e] = acc;
    }
    // (all slices of a unit have its first slice's width; FIRSTN units are single slices)
  }
}

// ---------------------------------------------------------------------------------------------
// The gradient WITHOUT float atomics (the general form: pooled features, a row referenced by several
// batch rows): one lane group per DISTINCT embedding row adds everything that row receives, in the
// order the reference's CPU kernel adds it — slices in configuration order, for each slice the batch
// rows ascending, inside a batch row the fids in list order (ScatterGrad inside the op's slice /
// batch loops, fused_embedding_to_layout.h:286-346, .cc:600-700) — so the result is that sequential
// fp32 sum bit for bit, and the same bits on every run.  (The reference's GPU kernel uses float
// atomics, .cu.cc:337-420; layout_kernel<false> above is that form, kept behind MHTE_POOL_ATOMICS.)
//   host: distinct (matrix, row) keys of fid_offset with their positions ascending (DedupWs::unique)
//   layout_qmap_kernel: position q -> its feature instance and its rank among the instance's valid
//                       fids (FIRSTN); feature instance -> its named feature list
//   layout_grad_lists_kernel: per distinct key, the slices of the features on its list in ascending
//                       slice order; per slice the list is walked once per element stride
// ---------------------------------------------------------------------------------------------
struct LayoutLists {
  const LayoutTask* tasks;          // every slice of the call, configuration order
  const uint32_t* nfl_task_off;     // [n_nfl + 1] slices of a named feature list ...
  const uint32_t* nfl_tasks;        //   ... ascending
  int32_t* qf;                      // [n_fid] feature instance of position q (-1: none)
  uint32_t* qseq;                   // [n_fid] valid fids before q in its instance
  uint32_t* fnfl;                   // [n_feature] named feature list of an instance (~0u: none)
  const int64_t* ukeys;             // distinct fid_offset values,
  const uint32_t* nu;               //   their number,
  const uint32_t* seg_off;          //   and positions: seg_pos[seg_off[u] .. seg_off[u + 1]) ascending
  const uint32_t* seg_pos;
  const uint32_t* inverse;          // [n_fid] distinct-key index of position q
  uint32_t* kflag;                  // [n_fid] per distinct key: 1 = a SHARED list reaches it
  uint32_t* heavy;                  // [n_fid] distinct-key indices of the heavy keys, n_heavy[0] of them
  uint32_t* n_heavy;
  uint32_t* cpos;                   // [n_fid] scratch of the heavy form (a key's slice of it: its list)
};

// A row that receives more than kLayoutLight contributions for one slice is HEAVY: a Zipf head row
// is pooled into tens of thousands of batch rows, a row of a SHARED list into every batch row, and
// one lane group walking that in order would be the launch's long pole (64 ms against 4 ms with
// atomics, scripts/next_rows_bench.py layout).  A heavy row gets a workgroup: the contribution
// sequence of a slice — still in the op's order — is cut into kLayoutHeavyGroups contiguous ranges,
// each summed in order by one lane group, and the partial sums are added to the row in range order.  The result
// depends on the inputs only (same bits on every run); against the strictly sequential sum it
// differs by fp32 re-association, as the training step's heavy lists do.
constexpr uint32_t kLayoutLight = 1024;
__device__ __forceinline__ bool layout_key_heavy(const LayoutArgs& A, const LayoutLists& X, int64_t u) {
  const uint64_t len = X.seg_off[u + 1] - X.seg_off[u];
  return len > kLayoutLight || (X.kflag[u] && len * uint64_t(A.batch) > kLayoutLight);
}

static_assert(sizeof(LayoutArgs) + sizeof(LayoutLists) <= 4096, "kernel arguments exceed 4 KB");

// one thread per (named feature list, instance): the instances a batch row reaches (GetFeatureInfo:
// off + b, or off for a shared list) and their fids
__global__ __launch_bounds__(256) void layout_qmap_kernel(LayoutArgs A, LayoutLists X) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int32_t nfl = int32_t(i / A.batch), b = int32_t(i % A.batch);
  if (nfl >= A.n_nfl) return;
  const uint32_t enc = A.nfl_offset[nfl];
  const bool shared = enc >> 31;
  const int32_t off = int32_t(enc & 0x7fffffffu);
  const int32_t off_next = nfl < A.n_nfl - 1 ? int32_t(A.nfl_offset[nfl + 1] & 0x7fffffffu) : A.n_feature;
  if (off_next - off <= 0 || (shared && b != 0)) return;
  const int32_t f = off + b;
  if (f >= A.n_feature || f >= off_next) return;   // (an instance of the NEXT list belongs to that list)
  X.fnfl[f] = uint32_t(nfl);
  const int32_t f0 = A.feature_offset[f];
  const int32_t f1 = f < A.n_feature - 1 ? A.feature_offset[f + 1] : A.n_fid;
  uint32_t seq = 0;
  for (int32_t q = f0; q < f1; ++q) {
    X.qf[q] = f;
    X.qseq[q] = seq;
    if (uint32_t(A.fid_offset[q] >> 32) < uint32_t(A.n_emb)) ++seq;
    if (shared) X.kflag[X.inverse[q]] = 1u;   // (racing stores of the same value)
  }
}

__global__ __launch_bounds__(256) void layout_heavy_select_kernel(LayoutArgs A, LayoutLists X) {
  const int64_t u = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u >= int64_t(X.nu[0])) return;
  if (uint32_t(static_cast<unsigned long long>(X.ukeys[u]) >> 32) >= uint32_t(A.n_emb)) return;
  if (layout_key_heavy(A, X, u)) X.heavy[atomicAdd(X.n_heavy, 1u)] = uint32_t(u);
}

// one contribution of slice t to element e of its row from list position q and batch row b, loaded
// unconditionally from a valid address (the callers keep several of these dependent chains —
// position -> feature instance -> gradient row — in flight and add the results in order afterwards);
// *ok = false: FIRSTN beyond max_sequence_length, nothing is added
__device__ __forceinline__ float layout_contribution(const LayoutArgs& A, const LayoutLists& X,
                                                    const LayoutTask& t, const float* og, uint32_t q,
                                                    int32_t f, int32_t b, int32_t e, bool* ok) {
  const float* orow = og + int64_t(b) * t.out_stride + t.out_offset;
  if (t.pooling == kPoolFirstN) {
    const uint32_t seq = X.qseq[q];
    *ok = seq < uint32_t(t.max_seq);
    return orow[int64_t(*ok ? seq : 0u) * t.dim + e];
  }
  *ok = true;
  if (t.pooling == kPoolMean) {
    const int32_t f0 = A.feature_offset[f];
    const int32_t f1 = f < A.n_feature - 1 ? A.feature_offset[f + 1] : A.n_fid;
    return orow[e] / float(f1 - f0);
  }
  return orow[e];
}
constexpr int kLayoutUnroll = 8;

__global__ __launch_bounds__(256) void layout_grad_lists_kernel(LayoutArgs A, LayoutLists X) {
  constexpr int G = 16;
  const int j = threadIdx.x & (G - 1);
  const int64_t u = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  if (u >= int64_t(X.nu[0])) return;
  const unsigned long long key = static_cast<unsigned long long>(X.ukeys[u]);
  const uint32_t i1 = uint32_t(key >> 32), i2 = uint32_t(key);
  if (i1 >= uint32_t(A.n_emb) || layout_key_heavy(A, X, u)) return;
  const uint32_t l0 = X.seg_off[u], l1 = X.seg_off[u + 1];
  float* const mat = const_cast<float*>(layout_emb(A, i1));
  const uint64_t rbase = uint64_t(i2) * layout_stride(A, i1);
  const uint64_t cnt = layout_count(A, i1);
  int32_t last_k = -1;
  for (;;) {
    // the next slice: the smallest index > last_k among the slices of the features on the list
    int32_t best = INT32_MAX;
    for (uint32_t p = l0 + uint32_t(j); p < l1; p += G) {
      const int32_t f = X.qf[X.seg_pos[p]];
      if (f < 0) continue;
      const uint32_t nfl = X.fnfl[f];
      if (nfl >= uint32_t(A.n_nfl)) continue;
      for (uint32_t c = X.nfl_task_off[nfl]; c < X.nfl_task_off[nfl + 1]; ++c) {
        const int32_t k = int32_t(X.nfl_tasks[c]);
        if (k > last_k) {
          best = min(best, k);
          break;
        }
      }
    }
#pragma unroll
    for (int o = G / 2; o; o >>= 1) best = min(best, __shfl_xor(best, o, G));
    if (best == INT32_MAX) break;
    last_k = best;
    const LayoutTask t = X.tasks[best];
    const float* og = layout_out(A, t.out_index);
    const uint32_t enc = A.nfl_offset[t.nfl_idx];
    const bool shared = enc >> 31;
    const int32_t off = int32_t(enc & 0x7fffffffu);
    const int32_t nb = shared ? A.batch : 1;
    for (int32_t e = j; e < t.dim; e += G) {
      const uint64_t at = rbase + uint32_t(t.start) + uint32_t(e);
      if (at >= cnt) continue;                         // (CUSTOM_CHECK in the reference)
      // (L2-served load: the previous slice's store to an overlapping element came from another lane)
      float acc = __hip_atomic_load(mat + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int32_t bb = 0; bb < nb; ++bb) {
        for (uint32_t p0 = l0; p0 < l1; p0 += kLayoutUnroll) {
          float x[kLayoutUnroll];
          bool ok[kLayoutUnroll];
#pragma unroll
          for (int r = 0; r < kLayoutUnroll; ++r) {   // (unconditional loads from clamped positions)
            const uint32_t p = min(p0 + uint32_t(r), l1 - 1u);
            const uint32_t q = X.seg_pos[p];
            const int32_t f = X.qf[q];
            const int32_t fc = max(f, 0);
            const int32_t b = shared ? bb : fc - off;
            const bool in = p0 + uint32_t(r) < l1 && f >= 0 && X.fnfl[fc] == uint32_t(t.nfl_idx) && b >= 0 &&
                            b < A.batch;
            bool o2;
            x[r] = layout_contribution(A, X, t, og, q, fc, in ? b : 0, e, &o2);
            ok[r] = in && o2;
          }
#pragma unroll
          for (int r = 0; r < kLayoutUnroll; ++r)
            if (ok[r]) acc = acc + x[r];
        }
      }
      mat[at] = acc;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (this slice's stores before the next one's loads)
  }
}

// the heavy rows: one workgroup per row, see kLayoutLight
constexpr int kLayoutHeavyGroups = 64;   // lane groups of that workgroup = ranges of a slice's sum
__global__ __launch_bounds__(1024) void layout_grad_heavy_kernel(LayoutArgs A, LayoutLists X) {
  constexpr int G = 16, NG = kLayoutHeavyGroups, NT = G * NG, NW = NT / 64;
  __shared__ int32_t s_best;
  __shared__ uint32_t s_wcnt[NW];
  __shared__ float s_part[NG][G];
  __shared__ uint32_t s_has[NG];
  const int tid = threadIdx.x, j = tid & (G - 1), g = tid >> 4, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t nh = X.n_heavy[0];
  for (uint32_t w = blockIdx.x; w < nh; w += gridDim.x) {
    const uint32_t u = X.heavy[w];
    const unsigned long long key = static_cast<unsigned long long>(X.ukeys[u]);
    const uint32_t i1 = uint32_t(key >> 32), i2 = uint32_t(key);
    const uint32_t l0 = X.seg_off[u], l1 = X.seg_off[u + 1];
    float* const mat = const_cast<float*>(layout_emb(A, i1));
    const uint64_t rbase = uint64_t(i2) * layout_stride(A, i1);
    const uint64_t cnt = layout_count(A, i1);
    int32_t last_k = -1;
    for (;;) {
      // ---- the next slice
      if (tid == 0) s_best = INT32_MAX;
      __syncthreads();
      int32_t best = INT32_MAX;
      for (uint32_t p = l0 + uint32_t(tid); p < l1; p += NT) {
        const int32_t f = X.qf[X.seg_pos[p]];
        if (f < 0) continue;
        const uint32_t nfl = X.fnfl[f];
        if (nfl >= uint32_t(A.n_nfl)) continue;
        for (uint32_t c = X.nfl_task_off[nfl]; c < X.nfl_task_off[nfl + 1]; ++c) {
          const int32_t k = int32_t(X.nfl_tasks[c]);
          if (k > last_k) {
            best = min(best, k);
            break;
          }
        }
      }
      if (best != INT32_MAX) atomicMin(&s_best, best);
      __syncthreads();
      best = s_best;
      if (best == INT32_MAX) break;
      last_k = best;
      const LayoutTask t = X.tasks[best];
      const float* og = layout_out(A, t.out_index);
      const uint32_t enc = A.nfl_offset[t.nfl_idx];
      const bool shared = enc >> 31;
      const int32_t off = int32_t(enc & 0x7fffffffu);
      // ---- the slice's entries of the list, in list order -> cpos[l0 .. l0 + m)
      uint32_t m = 0;
      for (uint32_t tile = l0; tile < l1; tile += NT) {
        const uint32_t p = tile + uint32_t(tid);
        bool match = false;
        uint32_t q = 0;
        if (p < l1) {
          q = X.seg_pos[p];
          const int32_t f = X.qf[q];
          if (f >= 0 && X.fnfl[f] == uint32_t(t.nfl_idx)) {
            const int32_t b = shared ? 0 : f - off;
            match = b >= 0 && b < A.batch;
          }
        }
        const unsigned long long mask = __ballot(match);
        if (lane == 0) s_wcnt[wave] = uint32_t(__popcll(mask));
        __syncthreads();
        uint32_t base = m;
        for (int w2 = 0; w2 < wave; ++w2) base += s_wcnt[w2];
        if (match) X.cpos[l0 + base + uint32_t(__popcll(mask & ((1ull << lane) - 1ull)))] = q;
        for (int w2 = 0; w2 < NW; ++w2) m += s_wcnt[w2];
        __syncthreads();
      }
      // (cpos is read back below by other lanes of this workgroup: L2-served loads after the drain)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const uint64_t total = uint64_t(m) * uint64_t(shared ? A.batch : 1);
      const uint64_t per = (total + NG - 1) / NG;
      const uint64_t lo = min(total, per * uint64_t(g)), hi = min(total, lo + per);
      for (int32_t e0 = 0; e0 < t.dim; e0 += G) {
        const int32_t e = e0 + j;
        const uint64_t at = rbase + uint32_t(t.start) + uint32_t(e);
        const bool active = e < t.dim && at < cnt;
        float acc = 0.f;
        bool has = false;
        if (active) {
          for (uint64_t s0 = lo; s0 < hi; s0 += kLayoutUnroll) {
            float x[kLayoutUnroll];
            bool ok[kLayoutUnroll];
#pragma unroll
            for (int r = 0; r < kLayoutUnroll; ++r) {   // (unconditional loads from clamped indices)
              const uint64_t s_ = min(s0 + uint64_t(r), hi - 1u);
              const uint32_t i = uint32_t(shared ? s_ % m : s_);
              const uint32_t q = __hip_atomic_load(X.cpos + l0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              const int32_t f = X.qf[q];
              const int32_t b = shared ? int32_t(s_ / m) : f - off;
              bool o2;
              x[r] = layout_contribution(A, X, t, og, q, f, b, e, &o2);
              ok[r] = o2 && s0 + uint64_t(r) < hi;
            }
#pragma unroll
            for (int r = 0; r < kLayoutUnroll; ++r)
              if (ok[r]) {
                acc = has ? acc + x[r] : x[r];
                has = true;
              }
          }
        }
        s_part[g][j] = acc;
        // (has is the same for the lanes of a group that are active; lane 0 of a group is active
        // whenever any lane is: e0 + 0 < t.dim, and the bound check only cuts a row's tail)
        if (j == 0) s_has[g] = (active && has) ? 1u : 0u;
        __syncthreads();
        if (g == 0 && active) {
          float a = __hip_atomic_load(mat + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (int gg = 0; gg < NG; ++gg)
            if (s_has[gg]) a = a + s_part[gg][j];
          mat[at] = a;
        }
        __syncthreads();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (this slice's stores before the next one's loads)
      __syncthreads();
    }
    __syncthreads();
  }
}

}  // namespace mhte
#endif  // MHTE_LAYOUT_KERNELS_H_
