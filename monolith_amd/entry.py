"""Table / optimizer / initializer configuration, mirroring the call-site contract of
monolith/native_training/entry.py (reference :27-640) without protobuf: the same class names and
constructor arguments, lowered to the flat C structs of include/monolith_amd_hash_table.h instead of
EmbeddingHashTableConfig protos (runtime/hash_table/embedding_hash_table.proto:23-95).

What is present: every per-row optimizer of the reference — SGD / Adagrad / FTRL (the fused training step's
BASIC kernel instances), momentum, adadelta, rmsprop, adam, amsgrad, moving average, batch softmax (its FULL
instances) and the whole-segment group-lasso adagrad (op-level kernels, fused optimize, the sharded step's
owner side) —, zeros / ones /
constants initializers, cuckoo table config, per-feature-slot expire times and occurrence
thresholds.  Asking for anything else raises (no silent downgrade)."""
import dataclasses
from typing import Any, Dict, List, Optional, Sequence

from monolith_amd import _lib


class Optimizer:
  opt_type = None

  def params(self) -> Sequence[float]:
    return ()


class StochasticRoundingFloat16OptimizerWrapper(Optimizer):
  """reference entry.py:42-50: sets OptimizerConfig.stochastic_rounding_float16 on the wrapped
  optimizer's config — the segment's weights are stochastically rounded to binary16 values after
  every update (MHTE_OPT_FLAG_STOCHASTIC_ROUNDING_FP16)."""

  def __init__(self, optimizer: Optimizer):
    if isinstance(optimizer, StochasticRoundingFloat16OptimizerWrapper) or optimizer.opt_type is None:
      raise ValueError("StochasticRoundingFloat16OptimizerWrapper wraps a concrete optimizer")
    self._optimizer = optimizer
    self.opt_type = optimizer.opt_type | _lib.OPT_FLAG_STOCHASTIC_ROUNDING_FP16

  def params(self):
    return self._optimizer.params()

  def __getattr__(self, name):   # learning_rate, warmup_steps, ... of the wrapped optimizer
    return getattr(self.__dict__["_optimizer"], name)


class SgdOptimizer(Optimizer):
  """reference entry.py:54-74; proto default learning_rate 0.01 (optimizer.proto:50-54)."""
  opt_type = _lib.OPT_SGD

  def __init__(self, learning_rate=None):
    self.learning_rate = 0.01 if learning_rate is None else learning_rate


class AdagradOptimizer(Optimizer):
  """reference entry.py:77-112; proto defaults lr 0.001, initial_accumulator_value 0.1,
  weight_decay_factor 0 (optimizer.proto:19-26)."""
  opt_type = _lib.OPT_ADAGRAD

  def __init__(self, learning_rate=None, initial_accumulator_value=None,
               hessian_compression_times=1, warmup_steps=0, weight_decay_factor=0.0,
               avx_semantics=False):
    # avx_semantics (an extension; not a field of the reference's config): the update of the
    # reference AS ITS .bazelrc:63-68 BUILDS IT — avx_utils.h:96-119, fused multiply-adds and, in
    # every block of 8 elements, the weight step taken with the raw gradient (differs from the
    # baseline loop whenever weight_decay_factor != 0) — bit for bit; default: the baseline loop
    self.avx_semantics = bool(avx_semantics)
    # accepted and carried like the reference's config field (optimizer.proto:23); the
    # reference's open-source runtime never reads it (adagrad_optimizer.cc has no sketching
    # code path), so — as there — it does not change the update
    self.hessian_compression_times = hessian_compression_times
    self.learning_rate = 0.001 if learning_rate is None else learning_rate
    self.initial_accumulator_value = (0.1 if initial_accumulator_value is None else
                                      initial_accumulator_value)
    self.weight_decay_factor = weight_decay_factor
    self.warmup_steps = warmup_steps

  def params(self):
    return (self.initial_accumulator_value, self.weight_decay_factor, 1.0 if self.avx_semantics else 0.0)


class FtrlOptimizer(Optimizer):
  """reference entry.py:365-392; proto defaults lr 0.01, beta 0, initial_accumulator_value 0.1,
  l1 = l2 = 0 (optimizer.proto:59-67)."""
  opt_type = _lib.OPT_FTRL

  def __init__(self, learning_rate=None, initial_accumulator_value=None, beta=None, warmup_steps=0,
               l1_regularization=None, l2_regularization=None):
    self.learning_rate = 0.01 if learning_rate is None else learning_rate
    self.initial_accumulator_value = (0.1 if initial_accumulator_value is None else
                                      initial_accumulator_value)
    self.beta = 0.0 if beta is None else beta
    self.l1_regularization_strength = 0.0 if l1_regularization is None else l1_regularization
    self.l2_regularization_strength = 0.0 if l2_regularization is None else l2_regularization
    self.warmup_steps = warmup_steps

  def params(self):
    return (self.initial_accumulator_value, self.beta, self.l1_regularization_strength,
            self.l2_regularization_strength)


class MomentumOptimizer(Optimizer):
  """reference entry.py MomentumOptimizer; proto defaults lr 0.01, momentum 0.9, weight decay 0,
  use_nesterov false (optimizer.proto:156-163).  Rides the fused training-step kernels too (their FULL instances)."""
  opt_type = _lib.OPT_MOMENTUM

  def __init__(self, learning_rate=None, weight_decay_factor=0.0, use_nesterov=False, momentum=None,
               warmup_steps=0):
    self.learning_rate = 0.01 if learning_rate is None else learning_rate
    self.weight_decay_factor = weight_decay_factor
    self.use_nesterov = use_nesterov
    self.momentum = 0.9 if momentum is None else momentum
    self.warmup_steps = warmup_steps

  def params(self):
    return (self.momentum, self.weight_decay_factor, 1.0 if self.use_nesterov else 0.0)


class AdadeltaOptimizer(Optimizer):
  """proto defaults lr 0.01, averaging_ratio 0.9, epsilon 0.01, weight decay 0
  (optimizer.proto:104-111).  Rides the fused training-step kernels too (their FULL instances)."""
  opt_type = _lib.OPT_ADADELTA

  def __init__(self, learning_rate=None, weight_decay_factor=0.0, averaging_ratio=None, epsilon=None,
               warmup_steps=0):
    self.learning_rate = 0.01 if learning_rate is None else learning_rate
    self.weight_decay_factor = weight_decay_factor
    self.averaging_ratio = 0.9 if averaging_ratio is None else averaging_ratio
    self.epsilon = 0.01 if epsilon is None else epsilon
    self.warmup_steps = warmup_steps

  def params(self):
    return (self.averaging_ratio, self.epsilon, self.weight_decay_factor)


class RmspropOptimizer(Optimizer):
  """proto defaults lr 0.01, weight decay 0, momentum 0.9 (optimizer.proto:186-191).  v1 steps
  with the CONFIG's learning rate (rmsprop_optimizer.cc:66); ``v2=True`` is RmspropV2
  (:127-144), which uses the op's learning-rate input.  Rides the fused training-step kernels too (their FULL instances)."""

  def __init__(self, learning_rate=None, weight_decay_factor=0.0, momentum=None, v2=False):
    self.opt_type = _lib.OPT_RMSPROPV2 if v2 else _lib.OPT_RMSPROP
    self.learning_rate = 0.01 if learning_rate is None else learning_rate
    self.weight_decay_factor = weight_decay_factor
    self.momentum = 0.9 if momentum is None else momentum

  def params(self):
    return (self.momentum, self.weight_decay_factor, self.learning_rate)


class AdamOptimizer(Optimizer):
  """proto defaults lr 0.01, beta1 0.9, beta2 0.99, weight decay 0, use_nesterov false,
  epsilon 0.01 (optimizer.proto:137-146); ``amsgrad=True`` is AmsgradOptimizer (same config
  fields, :118-128).  Rides the fused training-step kernels too (their FULL instances)."""

  def __init__(self, learning_rate=None, beta1=None, beta2=None, weight_decay_factor=0.0,
               use_nesterov=False, epsilon=None, warmup_steps=0, amsgrad=False):
    self.opt_type = _lib.OPT_AMSGRAD if amsgrad else _lib.OPT_ADAM
    self.learning_rate = 0.01 if learning_rate is None else learning_rate
    self.beta1 = 0.9 if beta1 is None else beta1
    self.beta2 = 0.99 if beta2 is None else beta2
    self.weight_decay_factor = weight_decay_factor
    self.use_nesterov = use_nesterov
    self.epsilon = 0.01 if epsilon is None else epsilon
    self.warmup_steps = warmup_steps

  def params(self):
    return (self.beta1, self.beta2, self.epsilon, self.weight_decay_factor,
            1.0 if self.use_nesterov else 0.0)


class MovingAverageOptimizer(Optimizer):
  """reference entry.py:247-256; proto default momentum 0.9 (optimizer.proto:169-172).  No state and
  no learning rate: w <- momentum w + (1 - momentum) g.  Rides the fused training-step kernels too (their FULL instances)."""
  opt_type = _lib.OPT_MOVING_AVERAGE
  learning_rate = 0.0

  def __init__(self, momentum=0.9):
    self.momentum = momentum

  def params(self):
    return (self.momentum,)


class BatchSoftmaxOptimizer(Optimizer):
  """reference entry.py:207-223; proto default learning_rate 0.1 (optimizer.proto:174-177).  A
  one-float segment holding the moving average of the steps between two occurrences of the id
  (https://research.google/pubs/pub48840/); reads the ops' ``global_step``.  Rides the fused training-step kernels too (their FULL instances)."""
  opt_type = _lib.OPT_BATCH_SOFTMAX

  def __init__(self, learning_rate=None):
    self.learning_rate = 0.1 if learning_rate is None else learning_rate


class AdaGradWithGroupLassoOptimizer(Optimizer):
  """reference entry.py:310-331 (GroupAdaGradOptimizerConfig, optimizer.proto:90-98: lr 0.01,
  beta 0, initial_accumulator_value 0.1, l2 0, weight decay 0).  The one whole-segment optimizer: the
  op-level kernels, mhte_fused_optimize and the owner side of the id-sharded step apply it; the pipelined and
  multi-table fused steps do not."""
  opt_type = _lib.OPT_GROUP_ADAGRAD

  def __init__(self, learning_rate=None, beta=None, initial_accumulator_value=None,
               l2_regularization=None, weight_decay_factor=0.0, warmup_steps=0):
    self.learning_rate = 0.01 if learning_rate is None else learning_rate
    self.beta = 0.0 if beta is None else beta
    self.initial_accumulator_value = (0.1 if initial_accumulator_value is None else
                                      initial_accumulator_value)
    self.l2_regularization_strength = 0.0 if l2_regularization is None else l2_regularization
    self.weight_decay_factor = weight_decay_factor
    self.warmup_steps = warmup_steps

  def params(self):
    return (self.initial_accumulator_value, self.beta, self.l2_regularization_strength,
            self.weight_decay_factor)


class Initializer:
  init_type = None
  value = 0.0


class ZerosInitializer(Initializer):
  """reference entry.py:404-411"""
  init_type = _lib.INIT_ZEROS


class OnesInitializer(Initializer):
  """runtime/hash_table/initializer (ones)"""
  init_type = _lib.INIT_ONES


class ConstantsInitializer(Initializer):
  """reference entry.py:414-423"""
  init_type = _lib.INIT_CONSTANT

  def __init__(self, constant: float):
    self.value = float(constant)


class RandomUniformInitializer(Initializer):
  """reference entry.py:426-440 (proto defaults minval -0.05, maxval 0.05,
  initializer_config.proto).  Draws are counter-based per element on the device; the reference's
  generator is thread-local and unseeded, so only the distribution is comparable."""
  init_type = _lib.INIT_RANDOM_UNIFORM

  def __init__(self, minval: float = -0.05, maxval: float = 0.05):
    self.value = float(minval)
    self.value2 = float(maxval)


class Fp32Compressor:
  """reference entry.py:505-511 — training rows are fp32; the serving-side compressors are out of
  scope."""


@dataclasses.dataclass
class Segment:
  """EntryConfig.Segment (embedding_hash_table.proto:23-43)."""
  dim_size: int
  initializer: Initializer
  optimizer: Optimizer


def CombineAsSegment(dim_size: int, initializer: Initializer, optimizer: Optimizer,
                     compressor: Any = None) -> Segment:
  """reference entry.py:514-537"""
  if compressor is not None and not isinstance(compressor, Fp32Compressor):
    raise NotImplementedError("only Fp32Compressor rows are on the MI355X hot path")
  if not isinstance(initializer, Initializer) or initializer.init_type is None:
    raise NotImplementedError("initializer %r is not supported" % (initializer,))
  if not isinstance(optimizer, Optimizer) or optimizer.opt_type is None:
    raise NotImplementedError("optimizer %r is not supported" % (optimizer,))
  return Segment(dim_size=int(dim_size), initializer=initializer, optimizer=optimizer)


@dataclasses.dataclass
class SlotExpireTimeConfig:
  """embedding_hash_table.proto:54-64: per-feature-slot TTL in days, default 36500."""
  default_expire_time: int = 36500
  slot_expire_times: Dict[int, int] = dataclasses.field(default_factory=dict)


@dataclasses.dataclass
class SlotOccurrenceThresholdConfig:
  """embedding_hash_table.proto:98-110: per-feature-slot admission thresholds of the hash filter,
  default 0 (= admit at once)."""
  default_occurrence_threshold: int = 0
  slot_occurrence_thresholds: Dict[int, int] = dataclasses.field(default_factory=dict)


@dataclasses.dataclass
class EmbeddingHashTableConfig:
  """embedding_hash_table.proto:70-95 (the fields the hot path reads)."""
  segments: List[Segment] = dataclasses.field(default_factory=list)
  initial_capacity: int = 1
  slot_expire_time_config: SlotExpireTimeConfig = dataclasses.field(
      default_factory=SlotExpireTimeConfig)
  slot_occurrence_threshold_config: SlotOccurrenceThresholdConfig = dataclasses.field(
      default_factory=SlotOccurrenceThresholdConfig)
  enable_feature_eviction: bool = False
  feature_evict_every_n_hours: int = 240
  # MI355X extensions
  reserve_rows: int = 0
  max_load_factor: float = 0.0

  @property
  def dim_size(self):
    return sum(s.dim_size for s in self.segments)


class CuckooHashTableConfig:
  """reference entry.py:549-563"""

  def __init__(self, initial_capacity=1, feature_evict_every_n_hours=0, reserve_rows=0,
               max_load_factor=0.0):
    self._initial_capacity = initial_capacity
    self._feature_evict_every_n_hours = feature_evict_every_n_hours
    self._reserve_rows = reserve_rows
    self._max_load_factor = max_load_factor

  def mutate_table(self, table_config: EmbeddingHashTableConfig):
    table_config.initial_capacity = self._initial_capacity
    table_config.reserve_rows = self._reserve_rows
    table_config.max_load_factor = self._max_load_factor
    if self._feature_evict_every_n_hours > 0:
      table_config.enable_feature_eviction = True
      table_config.feature_evict_every_n_hours = self._feature_evict_every_n_hours


class HashTableConfigInstance:
  """reference entry.py:566-628: a table config + one learning-rate fn/value per segment."""

  def __init__(self, table_config: EmbeddingHashTableConfig, learning_rate_fns: List[Any],
               extra_restore_names=None):
    self._table_config = table_config
    self._learning_rate_fns = list(learning_rate_fns)
    self.extra_restore_names = list(extra_restore_names or [])

  @property
  def table_config(self):
    return self._table_config

  @property
  def learning_rate_fns(self):
    return self._learning_rate_fns

  def call_learning_rate_fns(self) -> List[float]:
    if not self._learning_rate_fns:
      raise Exception("Learning_rate_fns must be not empty.")
    return [float(fn() if callable(fn) else fn) for fn in self._learning_rate_fns]


def make_table_config(segments: Sequence[Segment], hash_table_config: Optional[CuckooHashTableConfig]
                      = None, slot_expire_time_config: Optional[SlotExpireTimeConfig] = None,
                      learning_rates: Optional[Sequence[float]] = None,
                      slot_occurrence_threshold_config: Optional[SlotOccurrenceThresholdConfig] = None
                      ) -> HashTableConfigInstance:
  """Convenience: what the reference's test helpers build by hand (multi_hash_table_ops_test.py
  :30-48) — segments + cuckoo config -> HashTableConfigInstance with one lr per segment."""
  tc = EmbeddingHashTableConfig(segments=list(segments))
  (hash_table_config or CuckooHashTableConfig()).mutate_table(tc)
  if slot_expire_time_config is not None:
    tc.slot_expire_time_config = slot_expire_time_config
  if slot_occurrence_threshold_config is not None:
    tc.slot_occurrence_threshold_config = slot_occurrence_threshold_config
  if learning_rates is None:
    learning_rates = [s.optimizer.learning_rate for s in segments]
  return HashTableConfigInstance(tc, list(learning_rates))
