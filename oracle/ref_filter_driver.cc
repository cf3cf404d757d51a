// TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.
//
// Thin C driver around the REFERENCE's own hash-filter sources, compiled in place from
// /root/reference (nothing is copied into this repo):
//   * monolith/native_training/runtime/hash_filter/sliding_hash_filter.{h,cc}
//   * monolith/native_training/runtime/hash_filter/hash_filter.{h,cc}, filter.h, types.h
// with the shadow headers in oracle/ref_shim/ (absl StrFormat / Hash / Span / algorithm, glog, the
// protoc-generated embedding_hash_table.pb.h as plain structs, embedding_hash_table_interface.h).
// The one substitution with an effect is absl::Hash -> the engine's documented slot hash (absl's is
// seeded per process): see ref_shim/absl/hash/hash.h.  Output: oracle/_ref/libmonolith_ref_filter.so.
//
// What is the reference's code here: SlidingHashFilter::add / get / ShouldBeFiltered, the window
// (head, look-ahead, backward scan, async_clear), HashFilter<uint16_t>::find (probe sequence, 12-bit
// signatures, 4-bit saturating counts), Save / Restore of the splits.
//
// Only tests/ may load this library.
#include <cassert>   // (hash_filter.h uses assert without including it: the absl headers it shadows did)
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <vector>

// (hash_filter.cc first: its explicit specialisations of Save / Restore must precede the first use)
#include "monolith/native_training/runtime/hash_filter/hash_filter.cc"
#include "monolith/native_training/runtime/hash_filter/sliding_hash_filter.cc"

using monolith::hash_filter::SlidingHashFilter;
using monolith::hash_table::HashFilterSplitDataDump;
using monolith::hash_table::HashFilterSplitMetaDump;

extern "C" {

void* rf_create(uint64_t capacity, int split_num) { return new SlidingHashFilter(size_t(capacity), split_num); }
void rf_destroy(void* f) { delete static_cast<SlidingHashFilter*>(f); }
uint32_t rf_add(void* f, uint64_t fid, uint32_t count) { return static_cast<SlidingHashFilter*>(f)->add(fid, count); }
uint32_t rf_get(void* f, uint64_t fid) { return static_cast<SlidingHashFilter*>(f)->get(fid); }
int rf_should_be_filtered(void* f, int64_t fid, int64_t count, int64_t threshold) {
  return static_cast<SlidingHashFilter*>(f)->ShouldBeFiltered(fid, count, threshold, nullptr) ? 1 : 0;
}
void rf_add_many(void* f, const uint64_t* fid, const uint32_t* count, int64_t n, uint32_t* old) {
  SlidingHashFilter* s = static_cast<SlidingHashFilter*>(f);
  for (int64_t i = 0; i < n; ++i) old[i] = s->add(fid[i], count[i]);
}
void rf_get_many(void* f, const uint64_t* fid, int64_t n, uint32_t* out) {
  SlidingHashFilter* s = static_cast<SlidingHashFilter*>(f);
  for (int64_t i = 0; i < n; ++i) out[i] = s->get(fid[i]);
}
uint64_t rf_estimated_total_element(void* f) { return static_cast<SlidingHashFilter*>(f)->estimated_total_element(); }
uint64_t rf_failure_count(void* f) { return static_cast<SlidingHashFilter*>(f)->failure_count(); }
uint64_t rf_split_num(void* f) { return static_cast<SlidingHashFilter*>(f)->split_num(); }

// Save(split): meta -> meta_out[11] = {failure_count, total_size, num_elements, fill_rate * 1e6,
// split_num, max_forward_step, max_backward_step, max_step, head, head_increment, sliding failure
// count}; data -> the split's words in order (returns how many; at most cap are stored)
int64_t rf_save_split(void* f, int split, uint64_t* meta_out, uint32_t* data, int64_t cap) {
  SlidingHashFilter* s = static_cast<SlidingHashFilter*>(f);
  int64_t n = 0;
  s->Save(
      split,
      [&](HashFilterSplitMetaDump m) {
        const auto& sl = m.sliding_hash_filter_meta();
        const uint64_t v[11] = {m.failure_count(), m.total_size(), m.num_elements(), uint64_t(m.fill_rate() * 1e6 + 0.5),
                                sl.split_num(), sl.max_forward_step(), sl.max_backward_step(), sl.max_step(), sl.head(),
                                sl.head_increment(), sl.failure_count()};
        memcpy(meta_out, v, sizeof(v));
      },
      [&](HashFilterSplitDataDump d) {
        for (int i = 0; i < d.data_size(); ++i) {
          const int64_t at = int64_t(d.offset()) + i;
          if (at < cap) data[at] = d.data(i);
          if (at + 1 > n) n = at + 1;
        }
      });
  return n;
}
// Restore(split) from the same representation; returns 0, or 1 when the reference's validation throws
int rf_restore_split(void* f, int split, const uint64_t* meta, const uint32_t* data, int64_t n) {
  SlidingHashFilter* s = static_cast<SlidingHashFilter*>(f);
  try {
    int64_t at = 0;
    s->Restore(
        split,
        [&](HashFilterSplitMetaDump* m) {
          m->set_failure_count(meta[0]);
          m->set_total_size(meta[1]);
          m->set_num_elements(meta[2]);
          m->set_fill_rate(double(meta[3]) / 1e6);
          auto* sl = m->mutable_sliding_hash_filter_meta();
          sl->set_split_num(uint32_t(meta[4]));
          sl->set_max_forward_step(uint32_t(meta[5]));
          sl->set_max_backward_step(uint32_t(meta[6]));
          sl->set_max_step(uint32_t(meta[7]));
          sl->set_head(uint32_t(meta[8]));
          sl->set_head_increment(uint32_t(meta[9]));
          sl->set_failure_count(meta[10]);
          return true;
        },
        [&](HashFilterSplitDataDump* d) {
          if (at >= n) return false;
          d->clear_data();
          d->set_offset(uint32_t(at));
          const int64_t e = at + 10000 < n ? at + 10000 : n;
          for (; at < e; ++at) d->add_data(data[at]);
          return true;
        });
    return 0;
  } catch (const std::exception&) {
    return 1;
  }
}

}  // extern "C"
