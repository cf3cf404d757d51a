// TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.
//
// The reference's per-optimizer sources compiled IN PLACE from /root/reference (nothing copied):
//   monolith/native_training/runtime/hash_table/optimizer/
//     {sgd,adagrad,ftrl,adam,amsgrad,adadelta,momentum,rmsprop,moving_average,group_adagrad,
//      batch_softmax}_optimizer.cc  (+ dc_optimizer.cc, optimizer_decorator.h)
// against the protobuf stand-in oracle/ref_shim/.../optimizer.pb.h (accessor names only).  The C
// entry points below create an optimizer through the reference's own factory function and run its
// own Init() / Optimize() on caller-held buffers, so tests/test_oracle.py can check the restatement
// (oracle/mhte_oracle.c) Optimize() call by Optimize() call, bit for bit.
// Output: oracle/_ref/libmonolith_ref_opt.so (-ffp-contract=off, the scalar arithmetic the engine
// follows) and libmonolith_ref_opt_avx.so (-mavx2 -mfma as .bazelrc:63-68 builds it: Adagrad then
// takes avx_utils.h's vector path, and the compiler may contract a*b+c).
//
// Parameter vector p[] = the oracle's per-segment layout (oracle/mhte_oracle.h, mo_segment.p):
//   adagrad {initial_accumulator_value, weight_decay_factor}
//   ftrl {initial_accumulator_value, beta, l1, l2}          momentum {momentum, wd, use_nesterov}
//   adadelta {averaging_ratio, epsilon, wd}                 rmsprop / v2 {momentum, wd, conf learning_rate}
//   adam / amsgrad {beta1, beta2, epsilon, wd, use_nesterov}
//   moving_average {momentum}     group_adagrad {initial_accumulator_value, beta, l2, wd}
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "monolith/native_training/runtime/hash_table/optimizer/sgd_optimizer.cc"
#include "monolith/native_training/runtime/hash_table/optimizer/adagrad_optimizer.cc"
#include "monolith/native_training/runtime/hash_table/optimizer/ftrl_optimizer.cc"
#include "monolith/native_training/runtime/hash_table/optimizer/adam_optimizer.cc"
#include "monolith/native_training/runtime/hash_table/optimizer/amsgrad_optimizer.cc"
#include "monolith/native_training/runtime/hash_table/optimizer/adadelta_optimizer.cc"
#include "monolith/native_training/runtime/hash_table/optimizer/momentum_optimizer.cc"
#include "monolith/native_training/runtime/hash_table/optimizer/rmsprop_optimizer.cc"
#include "monolith/native_training/runtime/hash_table/optimizer/moving_average_optimizer.cc"
#include "monolith/native_training/runtime/hash_table/optimizer/group_adagrad_optimizer.cc"
#include "monolith/native_training/runtime/hash_table/optimizer/batch_softmax_optimizer.cc"
#include "monolith/native_training/runtime/hash_table/optimizer/dc_optimizer.cc"

namespace {
namespace ht = monolith::hash_table;
enum {  // oracle/mhte_oracle.h MO_OPT_*
  kSgd = 0, kAdagrad = 1, kFtrl = 2, kMomentum = 3, kAdadelta = 4, kRmsprop = 5, kRmspropV2 = 6,
  kAdam = 7, kAmsgrad = 8, kMovingAverage = 9, kBatchSoftmax = 10, kGroupAdagrad = 11
};

std::unique_ptr<ht::OptimizerInterface> make(int opt, int dim, const float* p) {
  switch (opt) {
    case kSgd: { ht::SgdOptimizerConfig c; c.set_dim_size(dim); return ht::NewSgdOptimizer(c); }
    case kAdagrad: {
      ht::AdagradOptimizerConfig c; c.set_dim_size(dim);
      c.set_initial_accumulator_value(p[0]); c.set_weight_decay_factor(p[1]);
      return ht::NewAdagradOptimizer(c);
    }
    case kFtrl: {
      ht::FtrlOptimizerConfig c; c.set_dim_size(dim);
      c.set_initial_accumulator_value(p[0]); c.set_beta(p[1]);
      c.set_l1_regularization_strength(p[2]); c.set_l2_regularization_strength(p[3]);
      return ht::NewFtrlOptimizer(c);
    }
    case kMomentum: {
      ht::MomentumOptimizerConfig c; c.set_dim_size(dim);
      c.set_momentum(p[0]); c.set_weight_decay_factor(p[1]); c.set_use_nesterov(p[2] != 0.f);
      return ht::NewMomentumOptimizer(c);
    }
    case kAdadelta: {
      ht::AdadeltaOptimizerConfig c; c.set_dim_size(dim);
      c.set_averaging_ratio(p[0]); c.set_epsilon(p[1]); c.set_weight_decay_factor(p[2]);
      return ht::NewAdadeltaOptimizer(c);
    }
    case kRmsprop: {
      ht::RmspropOptimizerConfig c; c.set_dim_size(dim);
      c.set_momentum(p[0]); c.set_weight_decay_factor(p[1]); c.set_learning_rate(p[2]);
      return ht::NewRmspropOptimizer(c);
    }
    case kRmspropV2: {
      ht::RmspropV2OptimizerConfig c; c.set_dim_size(dim);
      c.set_momentum(p[0]); c.set_weight_decay_factor(p[1]); c.set_learning_rate(p[2]);
      return ht::NewRmspropV2Optimizer(c);
    }
    case kAdam: {
      ht::AdamOptimizerConfig c; c.set_dim_size(dim);
      c.set_beta1(p[0]); c.set_beta2(p[1]); c.set_epsilon(p[2]); c.set_weight_decay_factor(p[3]);
      c.set_use_nesterov(p[4] != 0.f);
      return ht::NewAdamOptimizer(c);
    }
    case kAmsgrad: {
      ht::AmsgradOptimizerConfig c; c.set_dim_size(dim);
      c.set_beta1(p[0]); c.set_beta2(p[1]); c.set_epsilon(p[2]); c.set_weight_decay_factor(p[3]);
      c.set_use_nesterov(p[4] != 0.f);
      return ht::NewAmsgradOptimizer(c);
    }
    case kMovingAverage: {
      ht::MovingAverageOptimizerConfig c; c.set_dim_size(dim); c.set_momentum(p[0]);
      return ht::NewMovingAverageOptimizer(c);
    }
    case kBatchSoftmax: {
      ht::BatchSoftmaxOptimizerConfig c; c.set_dim_size(dim);
      return ht::NewBatchSoftmaxOptimizer(c);
    }
    case kGroupAdagrad: {
      ht::GroupAdaGradOptimizerConfig c; c.set_dim_size(dim);
      c.set_initial_accumulator_value(p[0]); c.set_beta(p[1]);
      c.set_l2_regularization_strength(p[2]); c.set_weight_decay_factor(p[3]);
      return ht::NewGroupAdaGradOptimizer(c);
    }
  }
  return nullptr;
}
struct Handle {
  std::unique_ptr<ht::OptimizerInterface> opt;
  int dim;
};
}  // namespace

extern "C" {
void* ref_opt_new(int opt, int dim, const float* p) {
  try {
    auto o = make(opt, dim, p);
    if (!o) return nullptr;
    return new Handle{std::move(o), dim};
  } catch (...) {
    return nullptr;
  }
}
void ref_opt_free(void* h) { delete static_cast<Handle*>(h); }
int64_t ref_opt_ctx_bytes(void* h) { return static_cast<Handle*>(h)->opt->SizeBytes(); }
int ref_opt_slice_size(void* h) { return static_cast<Handle*>(h)->opt->SliceSize(); }
void ref_opt_init(void* h, float* ctx) { static_cast<Handle*>(h)->opt->Init(ctx); }
// one Optimize() on num[0..dim) with its context (in place)
int ref_opt_optimize(void* h, float* ctx, float* num, const float* grad, float lr, int64_t global_step) {
  Handle* H = static_cast<Handle*>(h);
  try {
    H->opt->Optimize(ctx, absl::Span<float>(num, size_t(H->dim)),
                     absl::Span<const float>(grad, size_t(H->dim)), absl::Span<const float>(&lr, 1),
                     global_step);
  } catch (...) {
    return 1;
  }
  return 0;
}
// Save() then Restore() into a fresh context: what a checkpoint round trip does to the state
int ref_opt_save_restore(void* h, const float* ctx, float* ctx_out) {
  Handle* H = static_cast<Handle*>(h);
  try {
    ht::OptimizerDump d = H->opt->Save(ctx);
    H->opt->Restore(ctx_out, d);
  } catch (...) {
    return 1;
  }
  return 0;
}
// DcOptimizer (dc_optimizer.cc:28-43) around the reference's own SGD: one OptimizeWithLatestValue()
int ref_dc_sgd(float* num, const float* grad, const float* latest, int dim, float lr, float lambda_) {
  try {
    ht::SgdOptimizerConfig sc;
    sc.set_dim_size(dim);
    ht::DcOptimizerConfig dc;
    dc.set_dim_size(dim);
    dc.set_lambda_(lambda_);
    auto opt = ht::NewDcOptimizer(dc, ht::NewSgdOptimizer(sc));
    std::vector<float> lv(latest, latest + dim);
    opt->OptimizeWithLatestValue(nullptr, absl::Span<float>(num, size_t(dim)),
                                 absl::Span<const float>(grad, size_t(dim)), absl::Span<const float>(&lr, 1),
                                 absl::Span<float>(lv.data(), size_t(dim)), 0);
  } catch (...) {
    return 1;
  }
  return 0;
}
}
