// CPU test driver: the engine's fp16 stochastic rounding (monolith_amd/csrc/mhte_core.h, one source
// for host and device) on (value, draw) pairs read as raw floats from a file; prints the result
// bits.  tests/test_stochastic_rounding.py compares them with the reference's own function
// (oracle/_ref/libmonolith_ref_sr.so) and with the C restatement.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "mhte_core.h"

int main(int argc, char** argv) {
  if (argc != 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 3;
  int32_t n = 0;
  if (fread(&n, 4, 1, f) != 1) return 4;
  std::vector<float> v(size_t(n) * 2);
  if (fread(v.data(), 4, v.size(), f) != v.size()) return 5;
  fclose(f);
  for (int32_t i = 0; i < n; ++i) {
    const float r = mhte::stochastic_round(v[size_t(i) * 2], v[size_t(i) * 2 + 1]);
    uint32_t b;
    memcpy(&b, &r, 4);
    printf("%08x\n", b);
  }
  // and a few draws: in [0, 1), the same for the same inputs, different across occurrences
  const float x = 0.123f;
  const float d0 = mhte::sr_draw(&x, x, 1000u, 0u), d1 = mhte::sr_draw(&x, x, 1000u, 1u);
  if (!(d0 >= 0.f && d0 < 1.f && d1 >= 0.f && d1 < 1.f) || d0 == d1 || d0 != mhte::sr_draw(&x, x, 1000u, 0u)) return 6;
  return 0;
}
