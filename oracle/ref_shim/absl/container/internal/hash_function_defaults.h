// TEST INFRASTRUCTURE (oracle/_ref build only).  Shadow of the absl header that
// /root/reference/.../cuckoohash/cuckoohash_map.hpp:42 includes.  The reference map hashes
// with absl::container_internal::hash_default_hash<Key> (cuckoohash_map.hpp:64-70), whose
// value is ASLR-seeded and whose source is not under /root/reference.
// This shim substitutes the engine's documented fixed 64-bit mixer (murmur3 fmix64) so
// that bucket indices produced by the reference map are reproducible and comparable with
// the HIP engine's placement.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
namespace absl {
namespace container_internal {
template <class K>
struct hash_default_hash {
  size_t operator()(const K& k) const {
    uint64_t h = static_cast<uint64_t>(k);
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdULL;
    h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ULL;
    h ^= h >> 33;
    return static_cast<size_t>(h);
  }
};
template <class K>
using hash_default_eq = std::equal_to<K>;
}  // namespace container_internal
}  // namespace absl
