// TEST INFRASTRUCTURE (oracle/_ref build only).  Shadows the reference header of the same
// path, which drags in protobuf-generated code.  cuckoohash_map.hpp:740-773 (partial_dump)
// only uses the two PODs below (reference: embedding_hash_table_interface.h:89-97).
#pragma once
#include <cstdint>
namespace monolith {
namespace hash_table {
class EmbeddingHashTableInterface {
 public:
  struct DumpShard {
    int idx;
    int total;
    int64_t limit = 1LL << 61;
  };
  struct DumpIterator {
    int64_t offset = 0;
  };
};
}  // namespace hash_table
}  // namespace monolith
