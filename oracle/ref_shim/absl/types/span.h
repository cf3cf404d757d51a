// TEST INFRASTRUCTURE (oracle/_ref build only).  abseil is not in the reference tree; the reference
// headers compiled in place need of absl::Span only what is below (hash_filter.h:23 includes the
// header and uses nothing; optimizer_interface.h / stochastic_rounding.h pass spans and call size()
// and operator[]).
#pragma once
#include <cstddef>
namespace absl {
template <typename T>
class Span {
 public:
  Span() : p_(nullptr), n_(0) {}
  Span(T* p, size_t n) : p_(p), n_(n) {}
  template <typename C, typename = decltype(((C*)nullptr)->data()), typename = decltype(((C*)nullptr)->size())>
  Span(C& c) : p_(c.data()), n_(c.size()) {}   // a container (dc_optimizer.cc:41 passes a std::vector)
  template <typename U>
  Span(const Span<U>& o) : p_(o.data()), n_(o.size()) {}   // Span<float> -> Span<const float>
  T* data() const { return p_; }
  size_t size() const { return n_; }
  T& operator[](size_t i) const { return p_[i]; }
  T* begin() const { return p_; }
  T* end() const { return p_ + n_; }
  Span subspan(size_t pos) const { return Span(p_ + pos, n_ - pos); }

 private:
  T* p_;
  size_t n_;
};
}  // namespace absl
