// TEST INFRASTRUCTURE (oracle/_ref build only).  Shadow of the protoc-generated header (no protoc /
// protobuf in this image): plain structs with the accessor names the hash-filter sources call
// (hash_filter.cc:33-78, sliding_hash_filter.cc:158-212) for the three messages of
// embedding_hash_table.proto:112-137.
#pragma once
#include <cstdint>
#include <vector>
namespace google {
namespace protobuf {}
}  // namespace google
namespace monolith {
namespace hash_table {
#define MHTE_SHIM_FIELD(T, name)            \
 private:                                   \
  T name##_ = T();                          \
                                            \
 public:                                    \
  T name() const { return name##_; }        \
  void set_##name(T v) { name##_ = v; }
class SlidingHashFilterMetaDump {
  MHTE_SHIM_FIELD(uint32_t, split_num)
  MHTE_SHIM_FIELD(uint32_t, max_forward_step)
  MHTE_SHIM_FIELD(uint32_t, max_backward_step)
  MHTE_SHIM_FIELD(uint32_t, max_step)
  MHTE_SHIM_FIELD(uint32_t, head)
  MHTE_SHIM_FIELD(uint32_t, head_increment)
  MHTE_SHIM_FIELD(uint64_t, failure_count)
};
class HashFilterSplitMetaDump {
  MHTE_SHIM_FIELD(uint64_t, failure_count)
  MHTE_SHIM_FIELD(uint64_t, total_size)
  MHTE_SHIM_FIELD(uint64_t, num_elements)
  MHTE_SHIM_FIELD(double, fill_rate)
 private:
  SlidingHashFilterMetaDump sliding_;

 public:
  const SlidingHashFilterMetaDump& sliding_hash_filter_meta() const { return sliding_; }
  SlidingHashFilterMetaDump* mutable_sliding_hash_filter_meta() { return &sliding_; }
};
class HashFilterSplitDataDump {
  MHTE_SHIM_FIELD(uint32_t, offset)
 private:
  std::vector<uint32_t> data_;

 public:
  void add_data(uint32_t v) { data_.push_back(v); }
  int data_size() const { return int(data_.size()); }
  uint32_t data(int i) const { return data_[size_t(i)]; }
  void clear_data() { data_.clear(); }
};
#undef MHTE_SHIM_FIELD
}  // namespace hash_table
}  // namespace monolith
