// CPU-side unit driver for the serial displacement logic in monolith_amd/csrc/mhte_core.h
// (compiled by tests/test_core_host.py with g++ -DMHTE_HOST_ONLY).  It inserts ids one at a time
// through serial_insert_slot() into a pre-sized bucket array and prints bucket*4+slot of every id,
// which the test compares with the reference map's placement (tests/golden/placement_seq.npz).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mhte_core.h"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const unsigned hp = std::atoi(argv[1]);
  FILE* f = std::fopen(argv[2], "rb");
  if (!f) return 3;
  std::vector<int64_t> ids;
  int64_t v;
  while (std::fread(&v, sizeof(v), 1, f) == 1) ids.push_back(v);
  std::fclose(f);
  std::vector<mhte::Bucket> b(size_t(1) << hp);
  for (auto& x : b)
    for (int s = 0; s < mhte::kSlots; ++s) {
      x.key[s] = mhte::kEmptyKey;
      x.row[s] = mhte::kNoRow;
      x.ts[s] = 0;
    }
  std::vector<mhte::BfsSlot> q(mhte::kMaxCuckooCount);
  mhte::CuckooRecord path[mhte::kMaxBfsPathLen];
  for (size_t i = 0; i < ids.size(); ++i) {
    long long pos = mhte::serial_insert_slot(b.data(), hp, ids[i], q.data(), path);
    if (pos < 0) {
      std::printf("FAIL %zu\n", i);
      return 1;
    }
    b[pos >> 2].row[pos & 3] = (uint32_t)i;
  }
  // final position of every id (displacement may have moved earlier ones)
  for (size_t bi = 0; bi < b.size(); ++bi)
    for (int s = 0; s < mhte::kSlots; ++s)
      if (b[bi].key[s] != mhte::kEmptyKey)
        std::printf("%lld %zu\n", (long long)b[bi].key[s], bi * 4 + s);
  return 0;
}
