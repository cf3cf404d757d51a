// Development build of the dense tower alone (seconds instead of the library's five minutes):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o monolith_amd/libmhte_gemm_dev.so scripts/dbg/gemm_dev.hip
// exports the mhte_dense_mlp_* entry points over the same kernels and host code with minimal stand-ins
// for what mhte.hip provides; MHTE_DENSE_LIBRARY=<path> makes monolith_amd.dense_mlp load it.  A/B only.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/monolith_amd_hash_table.h"
#include "../../monolith_amd/csrc/mhte_gemm_kernels.h"

namespace mhte {
struct Error : std::runtime_error {
  mhte_status code;
  Error(mhte_status c, const std::string& m) : std::runtime_error(m), code(c) {}
};
#define HIP_OK(x)                                                                       \
  do {                                                                                  \
    hipError_t e__ = (x);                                                               \
    if (e__ != hipSuccess) throw Error(MHTE_INTERNAL, std::string(hipGetErrorString(e__))); \
  } while (0)
template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  void reserve(size_t n) {
    if (n <= cap) return;
    if (p) (void)hipFree(p);
    HIP_OK(hipMalloc(&p, n * sizeof(T)));
    cap = n;
  }
};
enum { kTagGemm = 19 };
#define LAUNCH_HOT(TAG, KERNEL, GRID, BLOCK, ST, ...) (KERNEL)<<<dim3(GRID), dim3(BLOCK), 0, (ST)>>>(__VA_ARGS__)
static hipStream_t S(void* s) { return static_cast<hipStream_t>(s); }
static thread_local std::string g_last_error;
template <class F>
static mhte_status guard(F&& f) {
  try { f(); return MHTE_OK; }
  catch (const Error& e) { g_last_error = e.what(); return e.code; }
  catch (const std::exception& e) { g_last_error = e.what(); return MHTE_INTERNAL; }
}
}  // namespace mhte
#include "../../monolith_amd/csrc/mhte_gemm_host.h"
struct mhte_dense_mlp { mhte::DenseMlp m; };
using namespace mhte;
extern "C" {
const char* mhte_last_error(void) { return g_last_error.c_str(); }
mhte_status mhte_dense_mlp_create(const int32_t* widths, int32_t n_widths, int64_t max_batch, int32_t gpu_ordinal,
                                  mhte_dense_mlp** out) {
  return guard([&] {
    HIP_OK(hipSetDevice(gpu_ordinal));
    std::unique_ptr<mhte_dense_mlp> m(new mhte_dense_mlp);
    m->m.create(widths, n_widths, max_batch, gpu_ordinal);
    *out = m.release();
  });
}
void mhte_dense_mlp_destroy(mhte_dense_mlp* m) { if (m) { (void)hipDeviceSynchronize(); delete m; } }
mhte_status mhte_dense_mlp_set_params(mhte_dense_mlp* m, int32_t layer, const float* w, const float* b, void* st) {
  return guard([&] { m->m.set_params(layer, w, b, S(st)); });
}
mhte_status mhte_dense_mlp_get_params(mhte_dense_mlp* m, int32_t layer, float* w, float* b, void* st) {
  return guard([&] { m->m.get_params(layer, w, b, S(st)); });
}
mhte_status mhte_dense_mlp_forward(mhte_dense_mlp* m, const float* x, int64_t batch, float* y, void* st) {
  return guard([&] { m->m.forward(x, batch, y, S(st)); });
}
mhte_status mhte_dense_mlp_backward(mhte_dense_mlp* m, const float* dy, float* dx, float lr, void* st) {
  return guard([&] { m->m.backward(dy, dx, lr, S(st)); });
}
}
