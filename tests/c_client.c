/* A plain C99 client of libmhte.so: includes the public header, links the library and walks the
 * MultiHashTable op sequence a TF shim would issue — create, assign, lookup, optimize, save,
 * restore into a second table, lookup-entry, feature-stat — through the C ABI only (device memory
 * through the HIP runtime's C API).  Built by tests/test_abi.py with gcc (no hipcc, no C++);
 * run on the GPU box by tests/test_boundary_gpu.py.  Exit code 0 = every check passed. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "monolith_amd_hash_table.h"

#define CHECK(cond, ...)                         \
  do {                                           \
    if (!(cond)) {                               \
      fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); \
      fprintf(stderr, __VA_ARGS__);              \
      fprintf(stderr, "\n");                     \
      return 1;                                  \
    }                                            \
  } while (0)
#define OK(call) CHECK((call) == MHTE_OK, "%s -> %s", #call, mhte_last_error())
#define HIP(call) CHECK((call) == hipSuccess, "%s", #call)

int main(int argc, char** argv) {
  const char* base = argc > 1 ? argv[1] : "/tmp/mhte_c_client_ckpt";
  CHECK(mhte_abi_version() == MHTE_ABI_VERSION, "ABI %d vs header %d", mhte_abi_version(), MHTE_ABI_VERSION);

  mhte_segment_config seg;
  memset(&seg, 0, sizeof(seg));
  seg.dim_size = 4;
  seg.opt_type = MHTE_OPT_SGD;
  seg.init_type = MHTE_INIT_ZEROS;
  mhte_table_config cfg[2];
  memset(cfg, 0, sizeof(cfg));
  cfg[0].name = "user";
  cfg[0].n_segments = 1;
  cfg[0].segments = &seg;
  cfg[1].name = "item";
  cfg[1].n_segments = 1;
  cfg[1].segments = &seg;
  mhte_multi_table* t = NULL;
  OK(mhte_multi_table_create(cfg, 2, 0, "c_client", &t));
  CHECK(mhte_num_tables(t) == 2 && strcmp(mhte_table_name(t, 0), "item") == 0, "tables sorted by name");
  CHECK(mhte_multi_table_is_initialized("c_client") == 1 && mhte_multi_table_find("c_client") == t, "registry");
  CHECK(mhte_multi_table_is_initialized("nobody") == 0, "registry miss");

  /* ragged batch: item {7, 8}, user {9}  (tables in sorted-name order) */
  const int64_t h_id[3] = {7, 8, 9};
  const int64_t split[3] = {0, 2, 3};
  const float h_val[12] = {1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3};
  const float h_grad[12] = {1, 2, 3, 4, 1, 1, 1, 1, -1, -1, -1, -1};
  const float lr[2] = {0.5f, 0.25f};
  int64_t* d_id;
  float *d_val, *d_out;
  HIP(hipMalloc((void**)&d_id, sizeof(h_id)));
  HIP(hipMalloc((void**)&d_val, sizeof(h_val)));
  HIP(hipMalloc((void**)&d_out, sizeof(h_val)));
  HIP(hipMemcpy(d_id, h_id, sizeof(h_id), hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_val, h_val, sizeof(h_val), hipMemcpyHostToDevice));
  OK(mhte_assign(t, d_id, split, 3, d_val, 12, 100, 0, NULL));
  HIP(hipMemcpy(d_val, h_grad, sizeof(h_grad), hipMemcpyHostToDevice));
  OK(mhte_optimize(t, d_id, split, 3, d_val, 12, lr, 2, 101, 0, 0, NULL));
  OK(mhte_lookup(t, d_id, split, 3, d_out, 12, NULL));
  float got[12];
  HIP(hipMemcpy(got, d_out, sizeof(got), hipMemcpyDeviceToHost));
  for (int i = 0; i < 12; ++i) {
    const float want = h_val[i] - (i < 8 ? 0.5f : 0.25f) * h_grad[i];
    CHECK(fabsf(got[i] - want) < 1e-6f, "row element %d: %f vs %f", i, got[i], want);
  }
  /* a wrong id_split is InvalidArgument with the reference's message shape */
  CHECK(mhte_lookup(t, d_id, split, 2, d_out, 12, NULL) == MHTE_INVALID_ARGUMENT, "error convention");

  OK(mhte_multi_table_save(t, base, 1, NULL));
  char names[256];
  uint64_t counts[8];
  int32_t n = 0;
  OK(mhte_feature_stat(base, names, sizeof(names), counts, 8, &n));
  CHECK(n == 2 && strcmp(names, "item") == 0 && counts[0] == 2 && counts[1] == 1, "feature stat");

  mhte_multi_table* t2 = NULL;
  OK(mhte_multi_table_create(cfg, 2, 0, "c_client_restored", &t2));
  OK(mhte_multi_table_restore(t2, base, NULL));
  HIP(hipMemset(d_out, 0, sizeof(got)));
  OK(mhte_lookup(t2, d_id, split, 3, d_out, 12, NULL));
  float got2[12];
  HIP(hipMemcpy(got2, d_out, sizeof(got2), hipMemcpyDeviceToHost));
  CHECK(memcmp(got, got2, sizeof(got)) == 0, "restored rows differ");
  int64_t sz = -1;
  OK(mhte_table_size(t2, 0, &sz, NULL));
  CHECK(sz == 2, "restored size %lld", (long long)sz);

  char entries[1024];
  int64_t offs[4], need = 0;
  OK(mhte_lookup_entry(t2, d_id, split, 3, entries, sizeof(entries), offs, &need, NULL));
  CHECK(offs[0] == 0 && offs[3] == need && need > 0 && offs[1] > 0, "entry offsets");

  mhte_multi_table_destroy(t2);
  mhte_multi_table_destroy(t);
  CHECK(mhte_multi_table_is_initialized("c_client") == 0, "registry after destroy");
  hipFree(d_id);
  hipFree(d_val);
  hipFree(d_out);
  printf("c_client ok\n");
  return 0;
}
