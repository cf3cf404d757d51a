// TEST INFRASTRUCTURE (oracle/_ref build only).  Shadow of absl/hash/hash.h for
// /root/reference/.../hash_filter/hash_filter.h:191 (`absl::Hash<FID>()(fid)`): absl's hash is
// seeded per process (and its source is not under /root/reference), so a filter's placement is not
// reproducible in the reference itself.  The shim substitutes the engine's documented slot hash —
// murmur3 fmix64 of (fid ^ 0x5bd1e995), csrc/mhte_kernels.h filter_home — so that which probe
// window an id falls into, and with it every aliasing of 12-bit signatures, is comparable.
#pragma once
#include <cstddef>
#include <cstdint>
namespace absl {
template <class T>
struct Hash {
  size_t operator()(const T& v) const {
    uint64_t h = static_cast<uint64_t>(v) ^ 0x5bd1e995ULL;
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdULL;
    h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ULL;
    h ^= h >> 33;
    return static_cast<size_t>(h);
  }
};
}  // namespace absl
