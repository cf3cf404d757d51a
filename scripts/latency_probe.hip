// Dependent random-access latency on MI355X as a function of the footprint touched: one wavefront
// chases a pseudo-random chain of 64-byte lines (each address depends on the data just loaded),
// so the time per hop is the full miss latency including address translation.  Also measured:
// the same chain with 256 independent wavefronts per CU resident (latency under load).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/latency_probe scripts/latency_probe.hip && /tmp/latency_probe
// Measurement aid for DESIGN.md §4 (why a step is bound by dependent round trips); not product code.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void chase(const uint64_t* __restrict__ base, uint64_t nlines, int hops,
                      unsigned long long* out_cycles, uint64_t* sink, uint64_t seed_stride) {
  const uint64_t wave = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  uint64_t idx = (wave * seed_stride + 12345) % nlines;
  // warm: nothing.  The chain: next = hash(loaded value ^ idx) % nlines; memory holds its own index.
  const long long t0 = wall_clock64();
  uint64_t acc = 0;
  for (int h = 0; h < hops; ++h) {
    const uint64_t v = base[idx * 8];             // one 8-byte load of a 64-byte line
    acc += v;
    uint64_t x = (v ^ (idx * 0x9E3779B97F4A7C15ull)) + h;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    idx = x % nlines;
  }
  const long long t1 = wall_clock64();
  if ((threadIdx.x & 63) == 0) out_cycles[wave] = (unsigned long long)(t1 - t0);
  if (acc == 0x1234567) *sink = acc;
}

__global__ void fill(uint64_t* base, uint64_t nwords) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
  for (uint64_t k = i; k < nwords; k += stride) base[k] = k * 0x2545F4914F6CDD1Dull;
}

int main() {
  const int hops = 200;
  const double sizes_gb[] = {0.25, 1, 4, 16, 64, 128};
  uint64_t* sink;
  CK(hipMalloc(&sink, 8));
  printf("| footprint GB | waves | us per dependent 64-B miss (median over waves) | min | max |\n|---|---|---|---|---|\n");
  for (double gb : sizes_gb) {
    const uint64_t bytes = uint64_t(gb * (1ull << 30));
    uint64_t* buf = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("| %.2f | alloc failed |\n", gb); continue; }
    fill<<<4096, 256>>>(buf, bytes / 8);
    CK(hipDeviceSynchronize());
    for (int load = 0; load < 2; ++load) {
      const int blocks = load ? 256 * 8 : 1;      // 1 wavefront, or 8 x 256-thread workgroups per CU
      const int threads = load ? 256 : 64;
      const int waves = blocks * threads / 64;
      unsigned long long* cyc;
      CK(hipMalloc(&cyc, sizeof(unsigned long long) * waves));
      chase<<<blocks, threads>>>(buf, bytes / 64, hops, cyc, sink, 7919);   // warm the kernel
      chase<<<blocks, threads>>>(buf, bytes / 64, hops, cyc, sink, 104729);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> h(waves);
      CK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost));
      std::sort(h.begin(), h.end());
      const double k = 1.0 / 100.0 / hops;        // 100 MHz wall clock -> us per hop
      printf("| %.2f | %d | %.3f | %.3f | %.3f |\n", gb, waves, h[waves / 2] * k, h[0] * k, h[waves - 1] * k);
      CK(hipFree(cyc));
    }
    CK(hipFree(buf));
  }
  return 0;
}
