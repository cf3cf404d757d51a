// TEST INFRASTRUCTURE.  The reference's fp16 stochastic rounding compiled IN PLACE
// (runtime/hash_table/optimizer/stochastic_rounding.{h,cc} + third_party/half_sourceforge_net/
// half.hpp; nothing copied): stochastic_round(vf, p) itself, and the decorator around a plain
// w -= lr * g optimizer defined here, so that tests can check the restatement (oracle/mhte_oracle.c
// mo_stochastic_round, csrc/mhte_core.h stochastic_round) value for value, and what the decorator
// does to a weight vector: every weight becomes one of the two binary16 neighbours of the inner
// optimizer's result, drawn from the header's multiply-with-carry generator (thread-local, seeded
// {0, 1}: reproducible on one thread).
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "monolith/native_training/runtime/hash_table/optimizer/stochastic_rounding.h"
#include "monolith/native_training/runtime/hash_table/optimizer/stochastic_rounding.cc"

namespace {
using monolith::hash_table::OptimizerDump;
using monolith::hash_table::OptimizerInterface;
class PlainSgd : public OptimizerInterface {   // (the inner optimizer: not reference code)
 public:
  explicit PlainSgd(int dim) : dim_(dim) {}
  int64_t SizeBytes() const override { return 0; }
  int64_t UncompressedSizeBytes() const override { return 0; }
  std::string DebugString() const override { return "plain_sgd"; }
  int DimSize() const override { return dim_; }
  int SliceSize() const override { return 1; }
  void Init(void*) const override {}
  void Optimize(void*, absl::Span<float> num, absl::Span<const float> grad,
                absl::Span<const float> learning_rates, const int64_t = 0) const override {
    for (size_t i = 0; i < num.size(); ++i) num[i] = num[i] - learning_rates[0] * grad[i];
  }
  OptimizerDump Save(const void*) const override { return OptimizerDump(); }
  void Restore(void*, OptimizerDump) const override {}

 private:
  int dim_;
};
}  // namespace

extern "C" {
float ref_stochastic_round(float vf, float p) { return monolith::hash_table::stochastic_round(vf, p); }

// one Optimize() of the decorated optimizer on num[0..n) (in place)
void ref_sr_decorated_sgd(float* num, const float* grad, int n, float lr) {
  monolith::hash_table::StochasticRoundingFloat16OptimizerDecorator opt(std::make_unique<PlainSgd>(n));
  opt.Optimize(nullptr, absl::Span<float>(num, size_t(n)), absl::Span<const float>(grad, size_t(n)),
               absl::Span<const float>(&lr, 1));
}
}
