// TEST INFRASTRUCTURE (oracle/_ref).  Stand-in for the protobuf-generated optimizer.pb.h, which is
// not in the reference tree (its build generates it from optimizer.proto; there is no protoc in this
// image).  Plain structs with the accessor names protoc would generate for the fields the
// reference's optimizer sources touch (field names, types and defaults: optimizer.proto:17-252), so
// that runtime/hash_table/optimizer/*_optimizer.cc, optimizer_interface.h and stochastic_rounding.h
// compile IN PLACE.  Nothing here restates reference behaviour: the arithmetic under test is the
// reference's own .cc files.
#ifndef ORACLE_REF_SHIM_OPTIMIZER_PB_H_
#define ORACLE_REF_SHIM_OPTIMIZER_PB_H_
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
namespace monolith {
namespace hash_table {

#define PB_FIELD(T, name, def)                 \
  T name##__ = def;                            \
  T name() const { return name##__; }          \
  void set_##name(T v) { name##__ = v; }
#define PB_REPEATED(T, name)                                   \
  std::vector<T> name##__;                                     \
  T name(int i) const { return name##__[size_t(i)]; }          \
  void add_##name(T v) { name##__.push_back(v); }              \
  int name##_size() const { return int(name##__.size()); }
#define PB_MESSAGE(T, name)                          \
  T name##__;                                        \
  const T& name() const { return name##__; }         \
  T* mutable_##name() { return &name##__; }

struct AdagradOptimizerConfig {
  PB_FIELD(int32_t, dim_size, 0) PB_FIELD(float, learning_rate, 0.001f)
  PB_FIELD(float, initial_accumulator_value, 0.1f) PB_FIELD(int32_t, hessian_compression_times, 1)
  PB_FIELD(float, weight_decay_factor, 0.f) PB_FIELD(int64_t, warmup_steps, 0)
};
struct AdagradOptimizerDump { PB_REPEATED(float, norm) };
struct SgdOptimizerConfig {
  PB_FIELD(int32_t, dim_size, 0) PB_FIELD(float, learning_rate, 0.01f) PB_FIELD(int64_t, warmup_steps, 0)
};
struct SgdOptimizerDump {};
struct FtrlOptimizerConfig {
  PB_FIELD(int32_t, dim_size, 0) PB_FIELD(float, learning_rate, 0.01f) PB_FIELD(float, beta, 0.f)
  PB_FIELD(float, initial_accumulator_value, 0.1f) PB_FIELD(float, l1_regularization_strength, 0.f)
  PB_FIELD(float, l2_regularization_strength, 0.f) PB_FIELD(int64_t, warmup_steps, 0)
};
struct FtrlOptimizerDump { PB_REPEATED(float, zero) PB_REPEATED(float, norm) };
struct GroupAdaGradOptimizerConfig {
  PB_FIELD(int32_t, dim_size, 0) PB_FIELD(float, learning_rate, 0.01f) PB_FIELD(float, beta, 0.f)
  PB_FIELD(float, initial_accumulator_value, 0.1f) PB_FIELD(float, l2_regularization_strength, 0.f)
  PB_FIELD(float, weight_decay_factor, 0.f) PB_FIELD(int64_t, warmup_steps, 0)
};
struct GroupAdaGradOptimizerDump { PB_FIELD(float, grad_square_sum, 0.f) };
struct AdadeltaOptimizerConfig {
  PB_FIELD(int32_t, dim_size, 0) PB_FIELD(float, learning_rate, 0.01f)
  PB_FIELD(float, weight_decay_factor, 0.f) PB_FIELD(float, averaging_ratio, 0.9f)
  PB_FIELD(float, epsilon, 0.01f) PB_FIELD(int64_t, warmup_steps, 0)
};
struct AdadeltaOptimizerDump { PB_REPEATED(float, accum) PB_REPEATED(float, accum_update) };
struct AdamOptimizerConfig {
  PB_FIELD(int32_t, dim_size, 0) PB_FIELD(float, learning_rate, 0.01f) PB_FIELD(float, beta1, 0.9f)
  PB_FIELD(float, beta2, 0.99f) PB_FIELD(bool, use_beta1_warmup, false)
  PB_FIELD(float, weight_decay_factor, 0.f) PB_FIELD(bool, use_nesterov, false)
  PB_FIELD(float, epsilon, 0.01f) PB_FIELD(int64_t, warmup_steps, 0)
};
struct AdamOptimizerDump {
  PB_REPEATED(float, m) PB_REPEATED(float, v) PB_FIELD(float, beta1_power, 0.f) PB_FIELD(float, beta2_power, 0.f)
};
struct AmsgradOptimizerConfig {
  PB_FIELD(int32_t, dim_size, 0) PB_FIELD(float, learning_rate, 0.01f) PB_FIELD(float, beta1, 0.9f)
  PB_FIELD(float, beta2, 0.99f) PB_FIELD(float, weight_decay_factor, 0.f)
  PB_FIELD(bool, use_nesterov, false) PB_FIELD(float, epsilon, 0.01f) PB_FIELD(int64_t, warmup_steps, 0)
};
struct AmsgradOptimizerDump {
  PB_REPEATED(float, m) PB_REPEATED(float, v) PB_REPEATED(float, vhat)
  PB_FIELD(float, beta1_power, 0.f) PB_FIELD(float, beta2_power, 0.f)
};
struct MomentumOptimizerConfig {
  PB_FIELD(int32_t, dim_size, 0) PB_FIELD(float, learning_rate, 0.01f)
  PB_FIELD(float, weight_decay_factor, 0.f) PB_FIELD(bool, use_nesterov, false)
  PB_FIELD(float, momentum, 0.9f) PB_FIELD(int64_t, warmup_steps, 0)
};
struct MomentumOptimizerDump { PB_REPEATED(float, n) };
struct MovingAverageOptimizerConfig { PB_FIELD(int32_t, dim_size, 0) PB_FIELD(float, momentum, 0.9f) };
struct BatchSoftmaxOptimizerConfig { PB_FIELD(int32_t, dim_size, 0) PB_FIELD(float, learning_rate, 0.1f) };
struct BatchSoftmaxOptimizerDump { PB_FIELD(int64_t, global_step, 0) };
struct RmspropOptimizerConfig {
  PB_FIELD(int32_t, dim_size, 0) PB_FIELD(float, learning_rate, 0.01f)
  PB_FIELD(float, weight_decay_factor, 0.f) PB_FIELD(float, momentum, 0.9f)
};
struct RmspropOptimizerDump { PB_REPEATED(float, n) };
struct RmspropV2OptimizerConfig {
  PB_FIELD(int32_t, dim_size, 0) PB_FIELD(float, learning_rate, 0.01f)
  PB_FIELD(float, weight_decay_factor, 0.f) PB_FIELD(float, momentum, 0.9f)
};
struct RmspropV2OptimizerDump { PB_REPEATED(float, n) };
struct DcOptimizerConfig { PB_FIELD(int32_t, dim_size, 0) PB_FIELD(float, lambda_, 0.f) };

struct SingleOptimizerDump {
  PB_MESSAGE(AdagradOptimizerDump, adagrad) PB_MESSAGE(SgdOptimizerDump, sgd)
  PB_MESSAGE(FtrlOptimizerDump, ftrl) PB_MESSAGE(AdadeltaOptimizerDump, adadelta)
  PB_MESSAGE(AdamOptimizerDump, adam) PB_MESSAGE(AmsgradOptimizerDump, amsgrad)
  PB_MESSAGE(MomentumOptimizerDump, momentum) PB_MESSAGE(RmspropOptimizerDump, rmsprop)
  PB_MESSAGE(RmspropV2OptimizerDump, rmspropv2) PB_MESSAGE(BatchSoftmaxOptimizerDump, batch_softmax)
  PB_MESSAGE(GroupAdaGradOptimizerDump, group_adagrad)
};
class OptimizerDump {
 public:
  SingleOptimizerDump* add_dump() {
    dump_.emplace_back();
    return &dump_.back();
  }
  const SingleOptimizerDump& dump(int i) const { return dump_[size_t(i)]; }
  int dump_size() const { return int(dump_.size()); }

 private:
  std::vector<SingleOptimizerDump> dump_;
};

#undef PB_FIELD
#undef PB_REPEATED
#undef PB_MESSAGE
}  // namespace hash_table
}  // namespace monolith
#endif  // ORACLE_REF_SHIM_OPTIMIZER_PB_H_
