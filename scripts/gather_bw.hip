// Practical HBM ceiling for the engine's access patterns on MI355X: random chunks of S bytes out of
// a footprint of F bytes, read (gather), read + streamed write (lookup-like) and read-modify-write
// (update-like), with as many independent chunks in flight as the chip holds.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_bw scripts/gather_bw.hip && /tmp/gather_bw
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e = (x);                                                       \
    if (e != hipSuccess) {                                                    \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                  \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode 0: gather (sum into a register, one store per lane at the end)
// mode 1: gather + streaming store of the chunk to out[i]
// mode 2: read-modify-write in place
// LPC lanes per chunk (chunk = LPC * 16 bytes), UNR chunks per lane group in flight
template <int LPC, int UNR, int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const uint32_t* __restrict__ idx, uint64_t n,
                                                     f32x4* __restrict__ buf, f32x4* __restrict__ out) {
  const uint64_t tid = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const uint64_t grp = tid / LPC;
  const int j = tid % LPC;
  const uint64_t ngroups = (n + UNR - 1) / UNR;
  f32x4 acc = {0, 0, 0, 0};
  for (uint64_t g = grp; g < ngroups; g += uint64_t(gridDim.x) * blockDim.x / LPC) {
    f32x4 v[UNR];
    uint64_t off[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const uint64_t i = g * UNR + u;
      off[u] = uint64_t(idx[i < n ? i : 0]) * LPC + j;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u)
      if (MODE <= 2) v[u] = buf[off[u]];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const uint64_t i = g * UNR + u;
      if (MODE == 0) {
        acc += v[u];
      } else if (MODE == 1) {
        if (i < n) __builtin_nontemporal_store(v[u], &out[i * LPC + j]);
      } else if (MODE == 2) {
        v[u] += 1.0f;
        if (i < n) buf[off[u]] = v[u];
      } else if (MODE == 3) {   // scattered write only (streaming stores)
        if (i < n) __builtin_nontemporal_store(acc, &buf[off[u]]);
      } else if (MODE == 4) {   // scattered write only (plain stores)
        if (i < n) buf[off[u]] = acc;
      } else {                  // sequential write only
        if (i < n) __builtin_nontemporal_store(acc, &out[i * LPC + j]);
      }
    }
  }
  if (MODE == 0 && acc.x == 12345.678f) out[tid] = acc;
}

template <int LPC, int UNR, int MODE>
static float run(const uint32_t* idx, uint64_t n, f32x4* buf, f32x4* out, int blocks) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  gather_kernel<LPC, UNR, MODE><<<blocks, 256>>>(idx, n, buf, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 3; ++r) gather_kernel<LPC, UNR, MODE><<<blocks, 256>>>(idx, n, buf, out);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 3;
}

int main(int argc, char** argv) {
  const uint64_t footprint = (argc > 1 ? strtoull(argv[1], 0, 10) : 32ull) << 30;  // GiB
  const uint64_t n = 1ull << 22;  // chunks per launch
  f32x4* buf;
  f32x4* out;
  uint32_t* idx;
  CK(hipMalloc(&buf, footprint));
  CK(hipMemset(buf, 0, footprint));
  CK(hipMalloc(&out, n * 512));
  CK(hipMalloc(&idx, n * 4));
  std::vector<uint32_t> h(n);
  printf("footprint %llu GiB, %llu random chunks per launch\n", (unsigned long long)(footprint >> 30),
         (unsigned long long)n);
  printf("%-6s %-8s %-4s %-8s %10s %10s\n", "chunkB", "mode", "unr", "blocks", "us", "GB/s");
#define CASE(LPC, UNR, MODE, NAME, BLK)                                                        \
  {                                                                                            \
    const uint64_t chunks = footprint / (LPC * 16);                                            \
    uint64_t x = 88172645463325252ull;                                                         \
    for (uint64_t i = 0; i < n; ++i) {                                                         \
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;                                                 \
      h[i] = uint32_t(x % chunks);                                                             \
    }                                                                                          \
    CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));                                \
    const float ms = run<LPC, UNR, MODE>(idx, n, buf, out, BLK);                               \
    const double bytes = double(n) * LPC * 16 * ((MODE == 1 || MODE == 2) ? 2 : 1) + double(n) * 4;           \
    printf("%-6d %-8s %-4d %-8d %10.1f %10.1f\n", LPC * 16, NAME, UNR, BLK, ms * 1e3,          \
           bytes / ms / 1e6);                                                                  \
  }
  for (int blk : {2048, 8192}) {
    CASE(4, 4, 3, "scatNT", blk)
    CASE(8, 4, 3, "scatNT", blk)
    CASE(16, 4, 3, "scatNT", blk)
    CASE(32, 2, 3, "scatNT", blk)
    CASE(4, 4, 4, "scat", blk)
    CASE(8, 4, 4, "scat", blk)
    CASE(16, 4, 4, "scat", blk)
    CASE(4, 4, 5, "seqNT", blk)
    CASE(16, 4, 5, "seqNT", blk)
    CASE(4, 2, 0, "gather", blk)
    CASE(4, 4, 0, "gather", blk)
    CASE(8, 2, 0, "gather", blk)
    CASE(8, 4, 0, "gather", blk)
    CASE(16, 2, 0, "gather", blk)
    CASE(16, 4, 0, "gather", blk)
    CASE(32, 2, 0, "gather", blk)
    CASE(4, 4, 1, "copy", blk)
    CASE(8, 4, 1, "copy", blk)
    CASE(16, 2, 1, "copy", blk)
    CASE(16, 4, 1, "copy", blk)
    CASE(4, 4, 2, "rmw", blk)
    CASE(8, 4, 2, "rmw", blk)
    CASE(16, 4, 2, "rmw", blk)
    CASE(32, 2, 2, "rmw", blk)
  }
  return 0;
}
