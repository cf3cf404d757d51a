/* monolith_amd_hash_table.h — C ABI of the MI355X-native MultiHashTable engine (libmhte.so).
 *
 * Drop-in boundary for Monolith's MultiHashTable TensorFlow custom ops.  Every entry point below
 * names the reference interface it replaces; paths are relative to
 * /root/reference/monolith/native_training/runtime/ ("RT/") or .../native_training/ ("NT/").
 * The reference-side binding (a TF OpKernel shim, and the ctypes binding used here) is shown in
 * INTEGRATION.md.
 *
 * Conventions
 *   - Every function returns an mhte_status; on failure mhte_last_error() (thread-local) holds a
 *     message.  Codes are TensorFlow's error codes, because the reference maps engine exceptions to
 *     errors::InvalidArgument / errors::ResourceExhausted
 *     (RT/ops/embedding_hash_table_tf_bridge.cc:132-134,365-367).
 *   - "dev" pointers are HIP device pointers on the table's GPU; "host" pointers are ordinary
 *     host memory.  Shape-determining inputs (row splits, slot sizes, learning rates, scalars) are
 *     host memory — the TF kernel registers them as HostMemory inputs.
 *   - `stream` is a hipStream_t passed as void*.  All work is enqueued on it and is ordered with
 *     respect to earlier calls on the same stream (the reference chains ops through the returned
 *     resource handle, RT/ops/multi_hash_table_update_op.cc:87; here stream order is that chain).
 *     A table must be driven from one stream at a time.
 *   - Tables of a MultiHashTable are ordered by sorted name, as NT/multi_hash_table_ops.py:83 does.
 */
#ifndef MONOLITH_AMD_HASH_TABLE_H_
#define MONOLITH_AMD_HASH_TABLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t mhte_status;
enum {
  MHTE_OK = 0,
  MHTE_INVALID_ARGUMENT = 3,    /* tensorflow::error::INVALID_ARGUMENT */
  MHTE_NOT_FOUND = 5,
  MHTE_RESOURCE_EXHAUSTED = 8,  /* tensorflow::error::RESOURCE_EXHAUSTED */
  MHTE_FAILED_PRECONDITION = 9,
  MHTE_INTERNAL = 13,
  MHTE_UNAVAILABLE = 14         /* no HIP device / runtime error */
};

const char* mhte_last_error(void);
/* ABI version of this header; mhte_abi_version() must return the same value. */
#define MHTE_ABI_VERSION 17
int32_t mhte_abi_version(void);

/* ---- configuration (flat C form of RT/hash_table/embedding_hash_table.proto) --------------- */
enum {                                                                     /* optimizer.proto:210-229 */
  MHTE_OPT_SGD = 0, MHTE_OPT_ADAGRAD = 1, MHTE_OPT_FTRL = 2,
  /* every per-element optimizer below rides the fused training-step kernels too (the single-table,
     multi-table and sharded steps: their FULL instantiations); only GROUP_ADAGRAD — one step needs
     the whole segment — stays on the op-level kernels (mhte_optimize & co) */
  MHTE_OPT_MOMENTUM = 3, MHTE_OPT_ADADELTA = 4, MHTE_OPT_RMSPROP = 5, MHTE_OPT_RMSPROPV2 = 6,
  MHTE_OPT_ADAM = 7, MHTE_OPT_AMSGRAD = 8, MHTE_OPT_MOVING_AVERAGE = 9,
  MHTE_OPT_BATCH_SOFTMAX = 10, /* dim_size 1; uses the ops' global_step argument */
  MHTE_OPT_GROUP_ADAGRAD = 11, /* AdaGradWithGroupLasso: one step needs the whole segment */
  /* OR-ed into opt_type: OptimizerConfig.stochastic_rounding_float16 (optimizer.proto:228) — the
     StochasticRoundingFloat16OptimizerDecorator (optimizer/stochastic_rounding.h:27-59): after every
     Optimize() each weight of the segment becomes one of its two binary16 neighbours, the upper one
     with probability (w - down) / (up - down).  The rounding function is the reference's, value for
     value; its draws there come from a thread-local generator consumed in call order, here from a
     counter-based hash of (element address, unrounded value, update_time, occurrence).  Not with
     GROUP_ADAGRAD (INVALID_ARGUMENT). */
  MHTE_OPT_FLAG_STOCHASTIC_ROUNDING_FP16 = 0x100
};
enum { MHTE_INIT_ZEROS = 0, MHTE_INIT_ONES = 1, MHTE_INIT_CONSTANT = 2, /* initializer_config.proto */
       MHTE_INIT_RANDOM_UNIFORM = 3 /* uniform in [init_value, init_value2): a counter-based draw per
                                       element; the reference's thread-local mt19937
                                       (initializer/random_uniform_initializer.cc:31-37) is not
                                       reproducible, so parity is distributional */ };

/* EntryConfig.Segment (embedding_hash_table.proto:23-43) */
typedef struct {
  int32_t dim_size;
  int32_t opt_type;     /* MHTE_OPT_* */
  float opt_params[8];  /* ADAGRAD:  {initial_accumulator_value, weight_decay_factor, avx_form}
                                     avx_form != 0 (extension): the update of the reference as its
                                     .bazelrc:63-68 builds it, avx_utils.h:96-119 — fused
                                     multiply-adds and, inside blocks of 8 elements, the weight
                                     step taken with the raw gradient (:112); default 0: the
                                     baseline loop (:29-38).  They differ when
                                     weight_decay_factor != 0
                           FTRL:     {initial_accumulator_value, beta, l1, l2}
                           MOMENTUM: {momentum, weight_decay_factor, use_nesterov}
                           ADADELTA: {averaging_ratio, epsilon, weight_decay_factor}
                           RMSPROP / RMSPROPV2: {momentum, weight_decay_factor, learning_rate of the
                                     config (v1 ignores the op's learning-rate input,
                                     rmsprop_optimizer.cc:66)}
                           ADAM / AMSGRAD: {beta1, beta2, epsilon, weight_decay_factor, use_nesterov}
                           MOVING_AVERAGE: {momentum}     BATCH_SOFTMAX: none
                           GROUP_ADAGRAD: {initial_accumulator_value, beta,
                                     l2_regularization_strength, weight_decay_factor} */
  int32_t init_type;    /* MHTE_INIT_* */
  float init_value;     /* ConstantsInitializerConfig.constant; RandomUniform: minval */
  float init_value2;    /* RandomUniformInitializerConfig.maxval */
} mhte_segment_config;

/* EmbeddingHashTableConfig (embedding_hash_table.proto:70-95) + SlotExpireTimeConfig (:54-64) */
typedef struct {
  const char* name;
  int32_t n_segments;
  const mhte_segment_config* segments;
  uint64_t initial_capacity;     /* slots; hashpower = reserve_calc(initial_capacity),
                                    cuckoohash_map.hpp:2114-2121; proto default 1 */
  uint64_t reserve_rows;         /* MI355X extension: rows of HBM slab to allocate up front (0 = grow
                                    on demand in 2^20-row slabs) */
  float max_load_factor;         /* MI355X extension: proactive doubling threshold, 0 -> 0.5 */
  int64_t default_expire_days;   /* SlotExpireTimeConfig.default_expire_time; 0 (unset) -> 36500, the
                                    proto default; negative -> a TTL of 0 days */
  int32_t n_slot_expire;
  const int64_t* expire_slots;   /* host */
  const int32_t* expire_days;    /* host */
  /* SlotOccurrenceThresholdConfig (embedding_hash_table.proto:98-110): an id that is not in the
     table yet is admitted once the attached hash filter has seen it this many times; <= 0: always */
  int32_t default_occurrence_threshold;
  int32_t n_slot_occurrence;
  const int64_t* occurrence_slots;      /* host */
  const int32_t* occurrence_thresholds; /* host */
  /* EmbeddingHashTableConfig.enable_feature_eviction / feature_evict_every_n_hours (:84-87): the
     reference's bridge runs a thread per table that wakes every 10 s and evicts once that many
     hours have passed (RT/ops/embedding_hash_table_tf_bridge.cc:73-104).  A scan that rewrites
     buckets has to be ordered with the table's other work, so here the same check rides on the
     update entry points (Optimize / FusedOptimize / the step's backward) and the scan is enqueued on
     their stream. */
  int32_t enable_feature_eviction;
  int32_t feature_evict_every_n_hours;  /* <= 0 -> 240 */
} mhte_table_config;

typedef struct mhte_multi_table mhte_multi_table;
typedef struct mhte_hash_filter mhte_hash_filter;   /* the admission filter resource, below */

/* CreateMonolithMultiHashTable (RT/ops/multi_hash_table_op.cc:44-112).  `configs` may be in any
 * order; tables are stored sorted by name.  device = HIP device ordinal. */
mhte_status mhte_multi_table_create(const mhte_table_config* configs, int32_t n_tables,
                                    int32_t device, const char* shared_name,
                                    mhte_multi_table** out);
void mhte_multi_table_destroy(mhte_multi_table* t);
/* CreateMonolithMultiHashTable taking its `config` input as is: the serialized
 * MultiEmbeddingHashTableConfig (RT/hash_table/embedding_hash_table.proto:93-96; names / configs of
 * different length -> InvalidArgument as RT/ops/multi_hash_table_op.cc:50-53).  `filter` (may be NULL)
 * is the filter_handle input; its occurrence thresholds (mhte_hash_filter_create_from_proto) become
 * the tables'.  reserve_rows / max_load_factor: the MI355X extensions of mhte_table_config, applied to
 * every table (0: defaults).  learning_rates_out (optional, host float[cap]) receives the configs' own
 * per-segment learning rates in table order (the ops take the live values as an input). */
mhte_status mhte_multi_table_create_from_proto(const void* config, int64_t config_len,
                                               mhte_hash_filter* filter, uint64_t reserve_rows,
                                               float max_load_factor, int32_t device,
                                               const char* shared_name, float* learning_rates_out,
                                               int32_t learning_rates_cap, mhte_multi_table** out);
/* ReadMonolithMultiHashTable / IsHashTableInitialized (RT/ops/multi_hash_table_op.cc:137-166): the
 * live table created under `shared_name`, or NULL / 0. */
mhte_multi_table* mhte_multi_table_find(const char* shared_name);
int32_t mhte_multi_table_is_initialized(const char* shared_name);
int32_t mhte_num_tables(const mhte_multi_table* t);
const char* mhte_table_name(const mhte_multi_table* t, int32_t i);
int32_t mhte_table_dim(const mhte_multi_table* t, int32_t i);        /* dim_size() */
int32_t mhte_table_slice_size(const mhte_multi_table* t, int32_t i); /* slice_size() */
int32_t mhte_table_index(const mhte_multi_table* t, const char* name); /* -1 when absent */
const char* mhte_shared_name(const mhte_multi_table* t);

/* MonolithMultiHashTableLookup (RT/ops/multi_hash_table_lookup_op.cc:33-89,200-209).
 * id [dev, id_split[n_split-1]], id_split [host, n_tables+1], embedding [dev, sum n_t*dim_t].
 * Absent ids produce zeros and are not inserted. */
mhte_status mhte_lookup(mhte_multi_table* t, const int64_t* id, const int64_t* id_split,
                        int64_t n_split, float* embedding, int64_t embedding_len, void* stream);

/* flags for the mutating ops */
enum {
  MHTE_IDS_UNIQUE = 1,   /* caller guarantees ids are distinct within each table segment (they come
                            from unique_key_with_value_and_offset / FusedReorderByIndices); skips
                            the in-op grouping pass */
  MHTE_SUM_DUPLICATES = 2 /* enable_grad_accumulation / enable_dedup: add duplicates' gradients,
                            then ONE optimizer step (RT/ops/embedding_hash_table_tf_bridge.cc:270-310).
                            Default: one optimizer step per occurrence, in order
                            (cuckoo_embedding_hash_table.cc:229-236). */
};

/* MonolithMultiHashTableOptimize (RT/ops/multi_hash_table_update_op.cc:47-100).
 * learning_rate [host, sum slice_size_t]; update_time seconds (stored as uint32). */
mhte_status mhte_optimize(mhte_multi_table* t, const int64_t* id, const int64_t* id_split,
                          int64_t n_split, const float* value, int64_t value_len,
                          const float* learning_rate, int64_t n_learning_rate, int64_t update_time,
                          int64_t global_step, int32_t flags, void* stream);
/* MonolithMultiHashTableAssign (:106-145) */
mhte_status mhte_assign(mhte_multi_table* t, const int64_t* id, const int64_t* id_split,
                        int64_t n_split, const float* value, int64_t value_len, int64_t update_time,
                        int32_t flags, void* stream);
/* MonolithMultiHashTableAssignAdd (:147-190) */
mhte_status mhte_assign_add(mhte_multi_table* t, const int64_t* id, const int64_t* id_split,
                            int64_t n_split, const float* value, int64_t value_len,
                            int64_t update_time, int32_t flags, void* stream);
/* MonolithMultiHashTableReinitialize (:192-245).  id_status [dev, n]: -1 unknown table (not an
 * error), 0 inserted, 1 existed.  `now` replaces absl::Now() so runs are reproducible; pass 0 to
 * use the wall clock. */
mhte_status mhte_reinitialize(mhte_multi_table* t, const char* table_name, const int64_t* id,
                              int64_t n, int32_t* id_status, int64_t now, void* stream);

/* ComputeFusedOffsets (RT/hash_table/utils.h:29-61), host arithmetic. */
mhte_status mhte_compute_fused_offsets(const mhte_multi_table* t, const int32_t* fused_slot_size,
                                       int32_t num_of_shards, int32_t* id_offsets,
                                       int32_t* embedding_offsets, int32_t* embedding_splits,
                                       int64_t* total_ids, int64_t* total_embeddings);
/* MonolithMultiHashTableFusedLookup (RT/ops/multi_hash_table_lookup_op.cc:128-197,229-255).
 * ids [dev], fused_slot_size [host, num_of_shards*n_tables] (shard-major, table-minor);
 * embeddings [dev, total_embeddings from mhte_compute_fused_offsets];
 * embedding_splits [host, num_of_shards], id_offsets / embedding_offsets [host, N*T+1].
 * ONE launch over the N*T segments (the reference loops over the tables inside Shard() over the
 * shards, :150-196) when the model has <= 32 tables whose rows are whole float4s (up to 256 floats)
 * or of any layout up to 64 floats; otherwise table by table. */
mhte_status mhte_fused_lookup(mhte_multi_table* t, const int64_t* ids,
                              const int32_t* fused_slot_size, int32_t num_of_shards,
                              int64_t req_time, float* embeddings, int64_t embeddings_len,
                              int32_t* embedding_splits, int32_t* id_offsets,
                              int32_t* embedding_offsets, void* stream);
/* MonolithMultiHashTableFusedOptimize (RT/ops/multi_hash_table_update_op.cc:247-325).
 * enable_grad_accumulation == (flags & MHTE_SUM_DUPLICATES).
 * With MHTE_IDS_UNIQUE (ids distinct inside every segment, which is what FusedReorderByIndices
 * hands over) the N*T segments are ONE upsert launch + one displacement launch; without it every
 * segment is grouped and applied on its own. */
mhte_status mhte_fused_optimize(mhte_multi_table* t, const int64_t* ids,
                                const int32_t* fused_slot_size, const float* id_grads,
                                int64_t id_grads_len, const int32_t* id_offsets,
                                const int32_t* grad_offsets, const float* learning_rates,
                                int64_t n_learning_rates, int64_t req_time, int64_t global_step,
                                int32_t num_of_shards, int32_t flags, void* stream);

/* MonolithMultiHashTableLookupEntry (RT/ops/multi_hash_table_lookup_op.cc:91-118,214-223): the
 * serialized EntryDump (embedding_hash_table.proto:45-50) of every id — what Save writes for it —
 * and an empty string for an id the table does not hold (cuckoo_embedding_hash_table.cc:174-182).
 * Strings are host data in TF: entries [host, cap bytes] receives them back to back,
 * entry_offsets [host, n + 1] where each starts.  *needed = total bytes; cap too small ->
 * MHTE_INVALID_ARGUMENT with *needed set.  Synchronises. */
mhte_status mhte_lookup_entry(mhte_multi_table* t, const int64_t* id, const int64_t* id_split,
                              int64_t n_split, char* entries, int64_t cap, int64_t* entry_offsets,
                              int64_t* needed, void* stream);
/* MonolithHashTableSaveAsTensor (RT/ops/hash_table/misc_ops.cc:46-94; op :96-109): one table's entries
 * as serialized EntryDump strings, `limit` at a time, resumable — the table iterated in Save's order
 * (cuckoohash_map.hpp:740-773 partial_dump: bucket range of shard `shard_idx` of `num_shards`, resumed
 * `offset` slots into it; the count is checked after an entry is taken, so limit <= 0 still yields one).
 * *new_offset is the op's first output: behind the last entry when the limit stopped the scan, one
 * bucket past the shard's range when the range ran out (call again until *n_entries == 0).
 * entries / entry_offsets [host]: as mhte_lookup_entry (offsets_cap >= limit + 1 values).  Synchronises. */
mhte_status mhte_table_save_as_tensor(mhte_multi_table* t, int32_t table, int32_t shard_idx,
                                      int32_t num_shards, int64_t limit, int64_t offset, int64_t* new_offset,
                                      char* entries, int64_t cap, int64_t* entry_offsets,
                                      int64_t offsets_cap, int64_t* n_entries, int64_t* needed, void* stream);
/* MonolithHashTableLookupGradient (RT/ops/hash_table_lookup_op.cc:110-147; op :254-270): the gradient of
 * a lookup gathered back per (batch row, id) pair of a sparse id tensor:
 *   out_ids[i] = id_values[i],  out_grads[i, :] = input_grads[id_indices[i * index_cols], :]
 * id_indices [dev, n x index_cols] (column 0 = the batch row), id_values [dev, n], input_grads
 * [dev, n_rows x dim] -> out_ids [dev, n], out_grads [dev, n x dim].  A row outside [0, n_rows) is
 * InvalidArgument (its output row is zeros; the reference indexes unchecked).  Synchronises. */
mhte_status mhte_lookup_gradient(const int64_t* id_indices, int64_t n, int64_t index_cols,
                                 const int64_t* id_values, const float* input_grads, int64_t n_rows,
                                 int32_t dim, int64_t* out_ids, float* out_grads, void* stream);
/* MonolithMultiHashTableFeatureStat (RT/ops/multi_hash_table_save_restore_ops.cc:424-497,522-530):
 * entries per table name summed over the .meta sidecars of a checkpoint.  names [host, names_cap
 * bytes]: the names, each NUL-terminated, sorted; counts [host, cap]. */
mhte_status mhte_feature_stat(const char* basename, char* names, int64_t names_cap, uint64_t* counts,
                              int32_t cap, int32_t* n_out);
/* Test hook for the eviction cadence: moves the library's clock forward by `seconds`. */
void mhte_advance_clock_for_testing(double seconds);

/* ---- introspection / maintenance ----------------------------------------------------------- */
/* Size() (RT/hash_table/embedding_hash_table_interface.h); synchronises the stream. */
mhte_status mhte_table_size(mhte_multi_table* t, int32_t table, int64_t* size, void* stream);
/* Contains() for a batch: out [dev, n] 0/1 */
mhte_status mhte_table_contains(mhte_multi_table* t, int32_t table, const int64_t* id, int64_t n,
                                int32_t* out, void* stream);
/* Evict(max_update_time) (cuckoo_embedding_hash_table.cc:251-264).  max_update_time < 0 uses the
 * table's own fuzzy max of all update_times seen (tf_bridge.cc:262-263). */
mhte_status mhte_table_evict(mhte_multi_table* t, int32_t table, int64_t max_update_time,
                             void* stream);
/* Table geometry: hashpower, rows allocated, lookup hits since last call. Synchronises. */
typedef struct {
  int64_t size;
  int32_t hashpower;
  int64_t rows_allocated;
  int64_t lookup_hits;
  int64_t dropped;      /* ids dropped because displacement failed (0 unless max_load_factor ~ 1) */
  int64_t evicted;
  int64_t max_update_ts;
  int64_t bytes_buckets;
  int64_t bytes_rows;
} mhte_table_stats;
mhte_status mhte_table_get_stats(mhte_multi_table* t, int32_t table, mhte_table_stats* out,
                                 void* stream);
/* Lookup hit counting (the `lookup_fid_hit_rate` metric of RT/ops/multi_hash_table_lookup_op.cc
 * :81-87, emitted by the reference for serving tables only).  Off by default: it costs one global
 * atomic per wavefront. */
mhte_status mhte_table_set_count_hits(mhte_multi_table* t, int32_t table, int32_t enable);
/* Dump in bucket-major order (Save's iteration order, cuckoohash_map.hpp:740-773).
 * All outputs dev, capacity `cap` entries; rows may be NULL; row_floats = dim + optimizer state.
 * Returns the number of entries in *n_out. Synchronises. */
mhte_status mhte_table_dump(mhte_multi_table* t, int32_t table, int64_t cap, int64_t* ids,
                            int64_t* positions, uint32_t* ts, float* rows, int64_t* n_out,
                            void* stream);
int32_t mhte_table_row_floats(const mhte_multi_table* t, int32_t i);

/* ---- the step right after the exchange: gather per merged slot, its gradient, ragged reductions --
 * MonolithFusedGatherEmbeddingsByInput (+Gradient) (RT/ops/map_id_to_embedding.cu.cc:31-118,
 * NT/distribution_ops.py:671-686): outputs[i][j, :] = fused_embeddings[offsets[i][j] + 0 .. dims[i]);
 * gradient: fused_grad (zero-filled, fused_len floats) += grads[i][j, :] * scale at the same offsets
 * (float atomics, as the reference's GpuAtomicAdd).  offsets / outputs / grads: HOST arrays of
 * n_inputs device pointers; n, dims: host arrays.
 * MonolithReduceSum / ReduceMean / ReduceSquareNorm (RT/ops/reduce_op.cc:29-125,
 * NT/embedding_combiners.py:41-102): out[b, :] over the rows i with indices[i] == b; mode 0 sum,
 * 1 mean, 2 sqrt of the sum of squares.  indices_sorted != 0 (the layout sparse/ragged inputs
 * have): sequential in-order accumulation, bit-identical to the reference; otherwise atomics. */
mhte_status mhte_fused_gather_embeddings_by_input(const float* fused_embeddings, int32_t n_inputs,
                                                  const int32_t* const* offsets, const int64_t* n,
                                                  const int32_t* dims, float* const* outputs,
                                                  void* stream);
mhte_status mhte_fused_gather_embeddings_by_input_gradient(float* fused_grad, int64_t fused_len,
                                                           int32_t n_inputs,
                                                           const float* const* grads,
                                                           const int32_t* const* offsets,
                                                           const int64_t* n, const int32_t* dims,
                                                           float scale, void* stream);
mhte_status mhte_reduce_rows(const int64_t* indices, const float* values, int64_t n, int32_t dim,
                             int64_t batch, int32_t mode, int32_t indices_sorted, float* out,
                             void* stream);

/* MonolithEmbeddingToLayout / MonolithEmbeddingToLayoutGrad (RT/ops/fused_embedding_to_layout.cc
 * :1037-1066; CUDA kernels RT/ops/fused_embedding_to_layout.cu.cc:96-199,337-...; GatherEmb /
 * ScatterGrad RT/ops/fused_embedding_to_layout.h:204-346): pools the looked-up embeddings of every
 * feature into the dense model's input tensors, and scatters the tensors' gradients back.
 *   embeddings[M]      HOST array of M device pointers: matrix m is [rows_m, emb_row_floats[m]],
 *                      emb_len[m] floats in all (the op's embeddings_list, one per shard x sub-table)
 *   fid_offset         [dev u64, n_fid] (matrix index << 32 | row) of every fid occurrence (:50-54)
 *   feature_offset     [dev i32, n_feature] first fid of every feature instance
 *   nfl_offset         [dev u32, n_nfl] first feature instance of every named feature list; bit 31:
 *                      the feature is shared by all rows of the batch (:56-60)
 *   slices             HOST: the slices of all layouts, layouts in sorted-name order (the op sorts
 *                      them, fused_embedding_to_layout.cc:283), slices in configuration order —
 *                      SliceConfig / OutConfig / FeatureConfig of idl/matrix/proto/example.proto
 *                      :176-221 flattened: feature_idx = the feature's index among the sorted feature
 *                      names (:273-279); pooling 0 SUM, 1 MEAN, 3 FIRSTN (max_sequence_length rows);
 *                      out_type 0 CONCAT, 1 STACK, 2 ADDN, 3 NONE; out_index / out_offset /
 *                      out_row_floats place the slice in its output tensor [batch, out_row_floats]
 *   outputs            HOST array of device pointers, zero-filled first (rows without fids stay 0)
 * SUM / MEAN walk a feature's fids in order (the reference's loop: bit-identical sums); the slices
 * of an ADDN layout are added in configuration order (the reference's CPU order; its CUDA path uses
 * float atomics).  The gradient op zero-fills embeddings_grad and adds, per embedding row, in the
 * order the reference's CPU kernel does (slices in configuration order, batch rows ascending, fids
 * in list order): its sequential fp32 sums bit for bit, no float atomics; a row with more than 1 024
 * contributions to one slice sums 64 contiguous ranges of that sequence and adds the range sums in
 * order (same bits on every run; fp32 re-association against the sequential sum).
 * MHTE_POOL_ATOMICS=1 in the environment selects the reference's CUDA form (float atomics, arrival
 * order).  Any number of matrices and outputs (beyond 64 / 32 their pointers are uploaded per call).
 * flags: MHTE_LAYOUT_ONE_FID_UNIQUE_ROWS — the caller states that every feature instance has exactly
 * one fid and no embedding row is referenced twice (the per-occurrence rows of a lookup feeding a
 * one-id-per-feature model): slices on float4 boundaries then move as float4 copies and the gradient
 * is a plain store instead of a float atomic per element. */
enum { MHTE_LAYOUT_ONE_FID_UNIQUE_ROWS = 1 };
typedef struct {
  int32_t feature_idx, start, dim;
  int32_t pooling, max_sequence_length;
  int32_t out_type, out_index, out_offset, out_row_floats;
} mhte_layout_slice;
mhte_status mhte_embedding_to_layout(const float* const* embeddings, const int32_t* emb_row_floats,
                                     const int64_t* emb_len, int32_t n_emb, const uint64_t* fid_offset,
                                     int64_t n_fid, const int32_t* feature_offset, int64_t n_feature,
                                     const uint32_t* nfl_offset, int32_t n_nfl, int32_t batch_size,
                                     const mhte_layout_slice* slices, int32_t n_slices,
                                     float* const* outputs, const int64_t* output_len, int32_t n_outputs,
                                     int32_t flags, void* stream);
mhte_status mhte_embedding_to_layout_grad(float* const* embeddings_grad, const int32_t* emb_row_floats,
                                          const int64_t* emb_len, int32_t n_emb, const uint64_t* fid_offset,
                                          int64_t n_fid, const int32_t* feature_offset, int64_t n_feature,
                                          const uint32_t* nfl_offset, int32_t n_nfl, int32_t batch_size,
                                          const mhte_layout_slice* slices, int32_t n_slices,
                                          const float* const* tensors_grad, const int64_t* tensor_len,
                                          int32_t n_tensors, int32_t flags, void* stream);

/* Admission filter: the reference's SlidingHashFilter (RT/hash_filter/sliding_hash_filter.{h,cc};
 * created by HashFilterOp, RT/ops/hash_filter_op.cc:47-81; the `filter_handle` input of
 * CreateMonolithMultiHashTable, RT/ops/multi_hash_table_op.cc:104-112), in HBM, shared by the tables of
 * a MultiHashTable: max(split_num, 5) counting filters (4-bit saturating counts, HashFilter,
 * RT/hash_filter/hash_filter.h:33-214) of capacity / (split_num - 1) elements each as a sliding window —
 * counts are taken forward in the head split, an id's older count is looked up backward, the window
 * moves on when the head split is full.  While attached, Optimize / Assign (and the training step's
 * apply) drop an id that is not in the table yet and has been seen fewer times than its feature slot's
 * occurrence threshold (RT/ops/embedding_hash_table_tf_bridge.cc:182-185,300-321); AssignAdd consults
 * the filter without the Contains guard, as the reference's multi-table path does (:230-232).  The
 * window moves between launches, not inside one.
 * mhte_hash_filter_save / _restore: MonolithHashFilterSave / Restore (RT/ops/hash_filter_save_op.cc,
 * hash_filter_restore_op.cc): one TFRecord file per split, <basename>-%05d-of-%05d, a
 * HashFilterSplitMetaDump then HashFilterSplitDataDump records (embedding_hash_table.proto:112-137);
 * restore checks the geometry like SlidingHashFilter::RestoreMetaDump.  (Slot positions depend on the
 * hash function — the reference's absl::Hash is unpinned — so files are interchangeable between
 * instances of this engine, not with the reference.)  Both synchronise. */
mhte_status mhte_hash_filter_create(uint64_t capacity, int32_t split_num, int32_t device,
                                    mhte_hash_filter** out);
/* the same with the filter op's `config` attr: a serialized SlotOccurrenceThresholdConfig
 * (embedding_hash_table.proto:100-110).  Tables created from a proto with this filter attached take
 * their occurrence thresholds from it. */
mhte_status mhte_hash_filter_create_from_proto(uint64_t capacity, int32_t split_num,
                                               const void* config, int64_t config_len,
                                               int32_t device, mhte_hash_filter** out);
/* MonolithProbabilisticFilter (RT/ops/hash_filter_op.cc:81-110, RT/hash_filter/probabilistic_filter.cc):
 * no counts are kept — an id the table does not hold is admitted with probability count / threshold
 * per consultation (equal_probability != 0: 1 - (1 - p)^count, p = 1 - 0.05^(1/threshold)); an id the
 * table holds is never filtered.  The reference draws from a time-seeded thread-local xorshift, so
 * parity is the admission rate; here a counter-based generator keyed by (seed, id, update launch,
 * occurrence) — seed 0: taken from the clock.  config: serialized SlotOccurrenceThresholdConfig
 * (may be NULL).  get returns 15, save / restore are no-ops (split_num() == 0), as in the reference. */
mhte_status mhte_hash_filter_create_probabilistic(int32_t equal_probability, uint64_t seed,
                                                  const void* config, int64_t config_len,
                                                  int32_t device, mhte_hash_filter** out);
void mhte_hash_filter_destroy(mhte_hash_filter* f);
mhte_status mhte_multi_table_set_filter(mhte_multi_table* t, mhte_hash_filter* f /* NULL detaches */);
/* Filter::estimated_total_element / failure_count / split_num (RT/hash_filter/filter.h:31-36):
 * out[0] = head, [1] = head_increment, [2] = failure_count (adds that found no usable slot),
 * [3] = number of splits S, [4 .. 4+S) = elements per split; cap >= 4 + S.  Synchronises. */
mhte_status mhte_hash_filter_stats(mhte_hash_filter* f, int64_t* out, int32_t cap, void* stream);
mhte_status mhte_hash_filter_save(mhte_hash_filter* f, const char* basename, void* stream);
mhte_status mhte_hash_filter_restore(mhte_hash_filter* f, const char* basename, void* stream);
/* Filter::get: the seen count of ids (0..15), out [dev u32, n] */
mhte_status mhte_hash_filter_get(mhte_hash_filter* f, const int64_t* id, int64_t n, uint32_t* out,
                                 void* stream);

/* MonolithMultiHashTableSave / Restore (RT/ops/multi_hash_table_save_restore_ops.cc:115-263,
 * 266-420) in the reference's on-disk format, so checkpoints are interchangeable:
 *   <basename>-%05d-of-%05d       TFRecord + SNAPPY stream of EntryDump (embedding_hash_table.proto
 *                                 :45-50: id, num, opt {one SingleOptimizerDump per segment},
 *                                 last_update_ts_sec), tables in sorted-name order, each shard a
 *                                 contiguous bucket range (cuckoohash_map.hpp:740-773)
 *   <basename>.meta-%05d-of-%05d  TFRecord stream of MultiHashTableMetadata {table_name, num_entries}
 * Save drops rows expired relative to the table's max_update_ts (:203-211; per-slot TTLs of the
 * table config).  nshards < 0: min(4, max(1, total_size / 1e6)) (:240-248).
 * Restore upserts every entry of every table it knows (whole row + the entry's own timestamp),
 * skips tables it does not know, reports a short or corrupted shard as an error.  Both synchronise
 * the stream; host-side work is proportional to the table. */
mhte_status mhte_multi_table_save(mhte_multi_table* t, const char* basename, int32_t nshards,
                                  void* stream);
mhte_status mhte_multi_table_restore(mhte_multi_table* t, const char* basename, void* stream);

/* MonolithHashTableSave / MonolithHashTableRestore for ONE table (RT/ops/hash_table_save_op.cc:72-200,
 * RT/ops/hash_table_restore_op.cc:63-160): the single-table layout the reference's exported models
 * carry (model_export/testdata/saved_model/ps_N/.../assets/MonolithHashTable_*):
 *   <basename>-%05d-of-%05d   UNCOMPRESSED TFRecord stream of EntryDump, one file per shard, no .meta
 * Save: nshards < 0 = min(4, max(1, size / 1e6)); each shard a contiguous bucket range, expired
 * rows left out, written under a temporary name and renamed.  Restore validates the shard set
 * (RT/ops/file_utils.cc:34-78), CLEARS the table (restore_op.cc:88), then upserts every record of
 * every shard (whole row + the entry's own timestamp; an entry without last_update_ts_sec gets 0).
 * mhte_table_clear: EmbeddingHashTableInterface::Clear (cuckoo_embedding_hash_table.cc:322-326).
 * All three synchronise the stream. */
mhte_status mhte_table_save(mhte_multi_table* t, int32_t table, const char* basename, int32_t nshards,
                            void* stream);
mhte_status mhte_table_restore(mhte_multi_table* t, int32_t table, const char* basename, void* stream);
mhte_status mhte_table_clear(mhte_multi_table* t, int32_t table, void* stream);

/* ---- caller-side dedup / packing ops on the device ------------------------------------------ */
typedef struct mhte_dedup_ws mhte_dedup_ws;
mhte_status mhte_dedup_ws_create(int32_t device, mhte_dedup_ws** out);
void mhte_dedup_ws_destroy(mhte_dedup_ws* ws);

/* Device form of the per-table dedup inside MonolithUniqueKeyWithValueAndOffset
 * (RT/ops/unique_mapping_ops.cc:82-114) and FusedReorderByIndices (RT/ops/fused_reorder_by_indices.cc
 * :52-60): unique ids in FIRST-OCCURRENCE order, the unique index of every position, and each
 * unique id's occurrence positions in occurrence order (CSR).
 *   ids [dev,n] -> unique_ids [dev, cap n], inverse [dev u32, n], seg_off [dev u32, n+1],
 *   seg_pos [dev u32, n], n_unique_dev [dev u32, 1].
 * n_unique_host (optional, host): filled after synchronising the stream. */
mhte_status mhte_unique(mhte_dedup_ws* ws, const int64_t* ids, int64_t n, int64_t* unique_ids,
                        uint32_t* inverse, uint32_t* seg_off, uint32_t* seg_pos,
                        uint32_t* n_unique_dev, int64_t* n_unique_host, void* stream);
/* Dedup for the fused training step: same key set and occurrence lists as mhte_unique, but the
 * numbering of the unique ids is unspecified (it follows atomic arrival order, like the iteration
 * order of the absl::flat_hash_map the reference dedups with, RT/ops/unique_mapping_ops.cc:82-114
 * keeps insertion order only because it also stores a vector).  Saves the two scan launches that
 * first-occurrence numbering needs.  list_start / list_end [dev u32, n]: positions of unique id u
 * are seg_pos[list_start[u] .. list_end[u]); lists longer than 32 positions are in ascending
 * position order, shorter ones in any order (mhte_table_sum_optimize_n orders them itself). */
mhte_status mhte_unique_unordered(mhte_dedup_ws* ws, const int64_t* ids, int64_t n,
                                  int64_t* unique_ids, uint32_t* inverse, uint32_t* list_start,
                                  uint32_t* list_end, uint32_t* seg_pos, uint32_t* n_unique_dev,
                                  int64_t* n_unique_host, void* stream);
/* FillWithOffsetMap in gather form (RT/ops/unique_mapping_ops.cc:225-242):
 * out[p,:] = src[index[p],:] for p < n. */
mhte_status mhte_gather_rows(const float* src, const uint32_t* index, int64_t n, int32_t dim,
                             float* out, void* stream);
/* FillWithOffsetMapGradient (RT/ops/unique_mapping_ops.cc:307-324): out[u,:] = sum over the
 * occurrence list of u of grads[p,:].  exact_order != 0: strictly sequential sum (bit-exact with
 * the reference); 0: windowed deterministic sum (fp32 re-association only). */
mhte_status mhte_segment_sum(mhte_dedup_ws* ws, const float* grads, const uint32_t* inverse,
                             const uint32_t* seg_off, const uint32_t* seg_pos,
                             const uint32_t* n_unique_dev, int64_t n, int32_t dim, float* out,
                             int32_t exact_order, void* stream);
/* Lookup / optimize of ONE table taking the id count from device memory (n_dev may be NULL):
 * lets dedup -> lookup -> ... -> optimize run without a host round trip. */
mhte_status mhte_table_lookup_n(mhte_multi_table* t, int32_t table, const int64_t* id,
                                int64_t n_max, const uint32_t* n_dev, float* embedding,
                                void* stream);
mhte_status mhte_table_optimize_n(mhte_multi_table* t, int32_t table, const int64_t* id,
                                  int64_t n_max, const uint32_t* n_dev, const float* value,
                                  const float* learning_rate, int64_t n_learning_rate,
                                  int64_t update_time, int64_t global_step, int32_t flags,
                                  void* stream);

/* Fused backward of one table: MonolithFillWithOffsetMapGradient (RT/ops/unique_mapping_ops.cc
 * :284-329) followed by MonolithMultiHashTableOptimize (RT/ops/multi_hash_table_update_op.cc:47-100)
 * on the unique ids, as ONE launch: out-of-order duplicates are summed per unique id and the
 * optimizer is applied once per id, without materialising the summed gradients.
 *   ws              the workspace whose most recent mhte_unique / mhte_unique_unordered produced
 *                   unique_ids / inverse / lists for this batch of n ids
 *                   (MHTE_FAILED_PRECONDITION otherwise)
 *   list_start/end  per unique id, its positions are seg_pos[list_start[u] .. list_end[u]);
 *                   after mhte_unique pass (seg_off, seg_off + 1)
 *   grads           [dev, n, dim] gradient of every occurrence
 *   grad_unique     [dev, n_max, dim] scratch (summed gradients of ids that take the displacement
 *                   slow path; all unique ids when the row is too wide for the fused kernel)
 *   flags           MHTE_EXACT_ORDER: every list is summed strictly in occurrence order (bit-exact
 *                   with the reference); otherwise lists of > 32 occurrences are summed in
 *                   256-entry chunks in a fixed association (deterministic; fp32 re-association
 *                   only).  MHTE_DEFER_SLOWPATH: do not launch the displacement pass for ids whose
 *                   two buckets are full; see mhte_table_finish_pending. */
enum { MHTE_EXACT_ORDER = 1, MHTE_DEFER_SLOWPATH = 2 };
mhte_status mhte_table_sum_optimize_n(mhte_multi_table* t, int32_t table, mhte_dedup_ws* ws,
                                      const int64_t* unique_ids, int64_t n_max,
                                      const uint32_t* n_unique_dev, const float* grads,
                                      const uint32_t* inverse, const uint32_t* list_start,
                                      const uint32_t* list_end, const uint32_t* seg_pos, int64_t n,
                                      float* grad_unique, const float* learning_rate,
                                      int64_t n_learning_rate, int64_t update_time,
                                      int64_t global_step, int32_t flags, void* stream);
/* Displacement (BFS) pass for the ids a MHTE_DEFER_SLOWPATH call left queued.  Every other entry
 * point on the table runs it first if it is still outstanding, so calling it is optional; callers
 * that time kernels, or that update several tables back to back, call it once at the end. */
mhte_status mhte_table_finish_pending(mhte_multi_table* t, int32_t table, void* stream);

/* Pipelined training step of one table: TWO launches per step, the dedup of the NEXT batch (it
 * depends on the ids only — the reference prefetches it too, NT/distributed_ps_sync.py:199-203)
 * riding in them, different workgroups doing the jobs side by side on one queue:
 *   mhte_table_step_forward   lookup of id[n] -> embedding (as mhte_table_lookup_n)
 *                             + dedup of id_next[n_next] into ws_next (one phase: per-workgroup
 *                               occurrence runs, see csrc/mhte_step_kernels.h)
 *                             + the displacement pass of the previous update (one wavefront; the
 *                               lookup workgroups wait for it only when it has work)
 *   mhte_table_step_backward  duplicate-gradient sum + upsert + optimizer of the batch held by ws
 *                             (MonolithFillWithOffsetMapGradient + MonolithMultiHashTableOptimize,
 *                             RT/ops/unique_mapping_ops.cc:284-329, multi_hash_table_update_op.cc
 *                             :47-100) + the heavy-list work items of the batch held by ws_next
 * Batches of 1..65 536 ids.  The dedup's device-side results are the unique ids (unspecified
 * order, like the iteration order of the flat_hash_map the reference dedups with) and their count;
 * the occurrence lists stay in the workspace in the step's own run format.
 *   ws / ws_next   two workspaces used alternately; ws_next may be NULL (no following batch)
 *   ws_cur         (step_forward, optional) the workspace that holds THIS batch's run dedup: the
 *                  launch then also reserves, with one counter bump per workgroup, the row handles
 *                  of the ids its update will have to insert (the update's own reservations all
 *                  hit one counter word and pace that launch); NULL: the update reserves itself
 *   flags          MHTE_EXACT_ORDER: every list is summed strictly in occurrence order (bit-exact
 *                  with the reference); otherwise lists of > 32 occurrences are summed as a fixed
 *                  tree over position ranges (deterministic; fp32 re-association only)
 * The first batch of a pipeline is deduplicated by mhte_step_dedup.  The displacement pass left
 * by step_backward is run by the next step_forward, or by any other call on the table
 * (mhte_table_finish_pending).  The table row must satisfy mhte_table_fused_backward_ok. */
mhte_status mhte_step_dedup(mhte_dedup_ws* ws, const int64_t* id, int64_t n, int64_t* unique_ids,
                            uint32_t* n_unique_dev, void* stream);
/* Sender side of the id-sharded step (NT/distributed_ps_sync.py:95-490), on the batch held by ws
 * (mhte_step_dedup):
 *   mhte_shard_partition  FusedReorderByIndices' shard-major packing (RT/ops/fused_reorder_by_indices.cc
 *                         :75-123; shard = floormod(id, num_shards), NT/distributed_ps.py:289) of ids
 *                         that are already unique: send_ids [dev, n_max] shard-major, send_pos
 *                         [dev u32, n_max] position of ids[u] in it, counts [dev u32, num_shards].
 *                         The id count comes from device memory (n_dev).  num_shards <= 64.
 *   mhte_step_scatter     out[p, :] = rows[idx(u), :] for every occurrence p of unique index u
 *                         (MonolithFillWithOffsetMap, RT/ops/unique_mapping_ops.cc:204-268)
 *   mhte_step_sum         out[idx(u), :] = sum over the occurrences p of u of grads[p, :], in
 *                         occurrence order for lists of <= 32, a fixed tree otherwise
 *                         (MonolithFillWithOffsetMapGradient, :284-329)
 * idx(u) = index ? index[u] : u (pass send_pos: rows arrive / gradients leave in send order).
 * dim <= 256. */
mhte_status mhte_shard_partition(mhte_dedup_ws* ws, const int64_t* ids, int64_t n_max,
                                 const uint32_t* n_dev, int32_t num_shards, int64_t* send_ids,
                                 uint32_t* send_pos, uint32_t* counts, void* stream);
mhte_status mhte_step_scatter(mhte_dedup_ws* ws, const float* rows, const uint32_t* index,
                              int32_t dim, float* out, void* stream);
mhte_status mhte_step_sum(mhte_dedup_ws* ws, const float* grads, const uint32_t* index, int32_t dim,
                          float* out, void* stream);
mhte_status mhte_table_step_forward(mhte_multi_table* t, int32_t table, const int64_t* id,
                                    int64_t n, float* embedding, mhte_dedup_ws* ws_next,
                                    const int64_t* id_next, int64_t n_next,
                                    int64_t* unique_ids_next, uint32_t* n_unique_dev_next,
                                    mhte_dedup_ws* ws_cur, void* stream);
mhte_status mhte_table_step_backward(mhte_multi_table* t, int32_t table, mhte_dedup_ws* ws,
                                     mhte_dedup_ws* ws_next, const int64_t* unique_ids,
                                     int64_t n_max, const uint32_t* n_unique_dev,
                                     const float* grads, int64_t n, float* grad_unique,
                                     const float* learning_rate, int64_t n_learning_rate,
                                     int64_t update_time, int64_t global_step, int32_t flags,
                                     void* stream);
/* ... with TWO batches of look-ahead: the run dedup of the batch after the next (id_ahead, into
 * ws_ahead, a third workspace) rides in this launch beside the update and the next batch's numbering.
 * The forward launch of a step whose batch was deduplicated this way carries the lookups alone
 * (mhte_table_step_forward with ws_next NULL): ~4 us less per step at 65 536 ids.  The reference's
 * pipeline prefetches the same way, one stage queue per step of look-ahead
 * (NT/distributed_ps_sync.py:199-203,270-275).  ws_ahead NULL: exactly mhte_table_step_backward. */
mhte_status mhte_table_step_backward_ahead(mhte_multi_table* t, int32_t table, mhte_dedup_ws* ws,
                                           mhte_dedup_ws* ws_next, const int64_t* unique_ids,
                                           int64_t n_max, const uint32_t* n_unique_dev,
                                           const float* grads, int64_t n, float* grad_unique,
                                           const float* learning_rate, int64_t n_learning_rate,
                                           int64_t update_time, int64_t global_step, int32_t flags,
                                           mhte_dedup_ws* ws_ahead, const int64_t* id_ahead,
                                           int64_t n_ahead, int64_t* unique_ids_ahead,
                                           uint32_t* n_unique_dev_ahead, void* stream);

/* Pipelined training step over ALL tables of a MultiHashTable: what a MonolithModel does per step
 * with MonolithMultiHashTableLookup (RT/ops/multi_hash_table_lookup_op.cc:33-89) and
 * MonolithMultiHashTableOptimize (RT/ops/multi_hash_table_update_op.cc:47-100) on the ragged
 * (id, id_split) batch of its T feature tables (NT/multi_hash_table_ops.py:349-413,
 * NT/multi_type_hash_table.py:253-303), as ONE forward and ONE backward launch for every table
 * together (+ a displacement launch that usually finds nothing to do) instead of 2 T launches:
 *   mhte_multi_step_forward   embedding[sum n_t*dim_t] = rows of id (layout of mhte_lookup: tables in
 *                             sorted-name order, per OCCURRENCE, absent ids -> zeros, no insert)
 *                             + the run dedup of the NEXT batch (id_next, id_split_next; NULL: none)
 *   mhte_multi_step_backward  value[sum n_t*dim_t] = gradient of every occurrence of the forward
 *                             batch: per table duplicate-gradient sum in occurrence order
 *                             (MonolithFillWithOffsetMapGradient, RT/ops/unique_mapping_ops.cc:284-329)
 *                             + upsert + one optimizer step per distinct id, + the numbering of the
 *                             next batch's distinct ids
 * prefetched != 0: the caller states that (id, id_split) is the batch the previous forward call
 * received as id_next (its dedup is picked up; MHTE_FAILED_PRECONDITION if there is none of that
 * shape); 0: the batch is deduplicated now (two more launches), any batch deduplicated ahead is
 * dropped.  Up to max_batch_per_table (<= 65 536) ids per table and step; empty tables are fine.
 * Every table must satisfy mhte_table_fused_backward_ok with segment boundaries on 4-float
 * multiples; embedding / value 16-byte aligned.  flags: MHTE_EXACT_ORDER as in
 * mhte_table_step_backward.  learning_rate [host, sum slice_size_t] as in mhte_optimize.
 * mhte_multi_step_unique_counts: distinct ids per table of the forward batch (host int64[T]);
 * synchronises. */
typedef struct mhte_multi_step mhte_multi_step;
mhte_status mhte_multi_step_create(mhte_multi_table* t, int64_t max_batch_per_table,
                                   mhte_multi_step** out);
void mhte_multi_step_destroy(mhte_multi_step* s);
mhte_status mhte_multi_step_forward(mhte_multi_step* s, const int64_t* id, const int64_t* id_split,
                                    int64_t n_split, float* embedding, int64_t embedding_len,
                                    const int64_t* id_next, const int64_t* id_split_next,
                                    int64_t n_split_next, int32_t prefetched, void* stream);
mhte_status mhte_multi_step_backward(mhte_multi_step* s, const float* value, int64_t value_len,
                                     const float* learning_rate, int64_t n_learning_rate,
                                     int64_t update_time, int64_t global_step, int32_t flags,
                                     void* stream);
mhte_status mhte_multi_step_unique_counts(mhte_multi_step* s, int64_t* counts, void* stream);

/* Id-sharded training step over ALL tables, one process per GPU: the reference's sync-training
 * exchange (NT/distributed_ps_sync.py:95-287 lookup, :289-490 apply_gradients; shard =
 * floormod(id, world), NT/distributed_ps.py:289; packing RT/ops/fused_reorder_by_indices.cc:75-123)
 * driven from C++ (RCCL send / recv groups, or direct peer stores) on the caller's stream.  Rank r's mhte_multi_table
 * holds the ids it owns of every table.  Per step a rank deduplicates its ragged batch, packs the
 * distinct ids into one block per peer that carries its own per-table counts (no size exchange),
 * and the ranks trade: id blocks -> owners look the rows up
 * (no insert) -> row blocks back -> scatter to the occurrences; gradients: per-id sums in
 * occurrence order -> row-shaped blocks to the owners -> every owner applies the peers' blocks one
 * after the other in rank order (the reference's default of one optimizer application per
 * sender).  Three exchanges per step, all tables in each.  Any number of tables (tables x world <=
 * 65535): the kernels of a stage take 32 tables per launch, the exchanges carry all of them.
 *   ids_per_peer_table  id slots per (peer, table) block; 0 (default) = max_batch: a block can hold
 *                       the whole batch, so no step can overflow one and no id is ever dropped —
 *                       the reference's all-to-all is variable-sized.  A smaller value is an explicit
 *                       choice of fixed-size blocks that cross the links whole: a step in which one
 *                       table sends more distinct ids than that to one peer hands those ids zero
 *                       rows, drops their gradients and makes the next call (or
 *                       mhte_shard_step_check, which must then be called before the step's results
 *                       are used) return MHTE_RESOURCE_EXHAUSTED.
 *   unique_id           128 bytes from mhte_shard_unique_id on one rank, distributed by the
 *                       launcher: the step creates its RCCL communicator (collective: every rank
 *                       calls create).  NULL with world == 1: no communicator, the exchange is the
 *                       identity.  NULL with world > 1: the ranks live in this process on one device
 *                       and are driven together through mhte_shard_group_* (device copies stand in
 *                       for the links; tests).
 * RCCL transport: with the default capacity the row / gradient exchanges move only the occupied
 * part of every (peer, table) segment; the counts are the id blocks' headers, copied to pinned host
 * memory behind the id exchange (a step ahead of their use for a batch that was prepared ahead).
 * With an explicit ids_per_peer_table: whole fixed-size blocks, nothing known to the host.
 * MHTE_SHARD_EXACT=0 / 1 in the environment at creation overrides the choice.
 *
 * Peer-store transport (mhte_shard_step_create_ipc): every rank maps every other rank's receive
 * WINDOW (hipIpcGetMemHandle / hipIpcOpenMemHandle — across xGMI between devices, or between
 * processes sharing one device) and an exchange is one copy kernel per rank that stores the
 * occupied part of every (peer, table) segment straight into the peers' windows, sized ON THE
 * DEVICE from the id blocks' headers: exact-size exchanges with no count on the host, no size
 * exchange and no staging copy.  Flow control is two words per (channel, peer) in the windows
 * (ready-to-receive credit, block-arrived sequence number); a consumer launch is preceded by a
 * one-wavefront wait for exactly the peers it needs, so an owner applies a peer's gradient block as
 * soon as that block has landed.  Waits are bounded (MHTE_SHARD_TIMEOUT_MS, default 30 000): a peer
 * that never arrives makes mhte_shard_step_check / the next call return MHTE_UNAVAILABLE instead of
 * hanging the queue.  Creation is three calls: create_ipc (allocates the window), ipc_handle (128
 * bytes for the launcher to gather from every rank), ipc_connect (all ranks' handles, rank-major);
 * ipc_selftest moves a data pattern through every pair of windows, three rounds over the same
 * addresses, checked on the receiving side (collective, synchronises; MHTE_UNAVAILABLE: use RCCL).
 * forward / backward arguments are those of mhte_multi_step_* for this rank's batch; `prefetched`
 * and the next batch must agree across the ranks (the calls are collective).  A table with an
 * occurrence filter is filtered on its OWNER: every id of a sender's block that the owner does not
 * hold asks the owner's filter with count 1 (ids of a block are distinct), as the reference's fused
 * optimize does per (sender, id); the filter's window moves between senders.  Tables with the
 * whole-segment optimizer (GroupAdaGrad) are accepted: the sender side does not look at the
 * optimizer, the owner applies such a table's blocks with an instance of its own.  global_step
 * reaches the optimizers. */
typedef struct mhte_shard_step mhte_shard_step;
mhte_status mhte_shard_unique_id(void* out128);
mhte_status mhte_shard_step_create(mhte_multi_table* t, int64_t max_batch_per_table, int32_t rank,
                                   int32_t world, int64_t ids_per_peer_table, const void* unique_id,
                                   mhte_shard_step** out);
void mhte_shard_step_destroy(mhte_shard_step* s);
mhte_status mhte_shard_step_create_ipc(mhte_multi_table* t, int64_t max_batch_per_table, int32_t rank,
                                       int32_t world, int64_t ids_per_peer_table,
                                       mhte_shard_step** out);
mhte_status mhte_shard_step_ipc_handle(mhte_shard_step* s, void* out128);
mhte_status mhte_shard_step_ipc_connect(mhte_shard_step* s, const void* handles, int32_t n_handles);
mhte_status mhte_shard_step_ipc_selftest(mhte_shard_step* s, void* stream);
mhte_status mhte_shard_step_forward(mhte_shard_step* s, const int64_t* id, const int64_t* id_split,
                                    int64_t n_split, float* embedding, int64_t embedding_len,
                                    const int64_t* id_next, const int64_t* id_split_next,
                                    int64_t n_split_next, int32_t prefetched, void* stream);
mhte_status mhte_shard_step_backward(mhte_shard_step* s, const float* value, int64_t value_len,
                                     const float* learning_rate, int64_t n_learning_rate,
                                     int64_t update_time, int64_t global_step, void* stream);
/* mode 1 (or MHTE_SHARD_OVERLAP=1 at creation): what the NEXT batch needs and the tables do not — its
 * run dedup, the numbering + owner packing of its distinct ids, the id exchange (peer-store
 * transport) — is enqueued by mhte_shard_step_forward on a stream of the step's own and runs beside
 * whatever the caller enqueues between forward and backward (layout -> dense model -> layout
 * gradient); backward then carries the gradient sums only.  Results are the same as mode 0 (nothing
 * stale is read: the reference pipelines the same stages with its prefetch queues,
 * NT/distributed_ps_sync.py:199-203,270-275).  The caller's next-batch ids must stay valid until the
 * following forward call. */
mhte_status mhte_shard_step_set_overlap(mhte_shard_step* s, int32_t mode);
/* bits 16 (or MHTE_SHARD_GRAD_FP16=1 at creation): the gradient exchange carries fp16 — the sender
 * rounds its per-id sums to nearest even, the owner widens them before its update — half the link
 * bytes of exchange 3.  A numerics change the reference offers as an option (`grad_flat` cast to
 * tf.float16 around the gradient all-to-all, NT/distributed_ps_sync.py:47,334-337); every rank of the
 * world must make the same choice, before its first backward.  bits 32: fp32 (default). */
mhte_status mhte_shard_step_set_grad_bits(mhte_shard_step* s, int32_t bits);
/* on != 0: this rank's per-id gradient sums are the reference's SEQUENTIAL sums for every duplicate list
 * (MonolithFillWithOffsetMapGradient adds an id's gradients in occurrence order,
 * RT/ops/unique_mapping_ops.cc:284-329) — lists of up to 32 occurrences are summed that way in any mode; with
 * this the heavy ones are too (one more launch in front of the sums: a workgroup streams a list's rows through LDS,
 * one wavefront adds them in order), instead of a fixed tree within 1e-6 of it.  Every owner's rows are then the
 * reference's bit for bit (its senders applied in rank order, distributed_ps_sync.py:357-479).  A per-rank
 * choice; takes effect at the next backward. */
mhte_status mhte_shard_step_set_exact_order(mhte_shard_step* s, int32_t on);
/* waits for the stream, then reports a block overflow of the steps enqueued so far */
mhte_status mhte_shard_step_check(mhte_shard_step* s, void* stream);
/* distinct ids per table of this rank's forward batch (host int64[T]); synchronises */
mhte_status mhte_shard_step_unique_counts(mhte_shard_step* s, int64_t* counts, void* stream);
/* info[0] = id slots per (peer, table), [1] = bytes of one id block, [2] = bytes of one row block,
 * [3] = transport: 0 identity, 1 RCCL, 2 in-process group, 3 peer stores into fine-grained windows,
 * 4 peer stores into plain device memory (MHTE_SHARD_WINDOW=coarse) */
mhte_status mhte_shard_step_info(mhte_shard_step* s, int64_t info[4]);
/* out[0] / out[1] = kernel launches + exchanges (a peer-store push, a sync, an RCCL send / recv group; in
 * an in-process group one per exchange) the step's last forward / backward call enqueued for this rank.
 * With the one-launch owner update (the shape of MonolithMultiHashTableFusedOptimize,
 * RT/ops/multi_hash_table_update_op.cc:247-308: every shard's segments in one op) the count does not
 * depend on the world size; per 32 tables of the model. */
mhte_status mhte_shard_step_launches(mhte_shard_step* s, int32_t out[2]);
/* What the step's last forward + backward put on the wire through send / recv pairs (the RCCL transport and the
 * in-process group, whose device copies stand for the pairs; 0 for identity / peer stores):
 * out[0] = pairs in total, [1] = exchanges, [2] = the most pairs any one exchange took, [3] = times the HOST
 * waited for the id blocks' counts inside the two calls.  The exact-size form (RCCL with whole-batch blocks)
 * packs the occupied part of a peer's table segments back to back: one pair per peer and exchange — [2] <=
 * world (+ world when the next batch's id headers ride in the gradient exchange's group), not world x T —
 * and a batch that was prepared a step ahead finds its counts on the host: [3] = 0.  The reference moves
 * every exchange as ONE all-to-all-v per tensor (native_training/distributed_ps_sync.py:131-159,357-479). */
mhte_status mhte_shard_step_wire_stats(mhte_shard_step* s, int64_t out[4]);
/* out[0] = ncclCommCount, out[1] = ncclCommUserRank of the step's own RCCL communicator (0, 0 when the
 * step has none: identity, peer stores, in-process group) — a launcher checks out[0] == world on
 * every rank before it trusts a multi-GPU number */
mhte_status mhte_shard_step_comm_ranks(mhte_shard_step* s, int32_t out[2]);
/* all `n` ranks of a world living in this process (created with unique_id NULL, world n): the same
 * step, with arrays of per-rank arguments */
mhte_status mhte_shard_group_forward(mhte_shard_step** steps, int32_t n, const int64_t* const* id,
                                     const int64_t* const* id_split, int64_t n_split,
                                     float* const* embedding, const int64_t* embedding_len,
                                     const int64_t* const* id_next,
                                     const int64_t* const* id_split_next, int64_t n_split_next,
                                     int32_t prefetched, void* stream);
mhte_status mhte_shard_group_backward(mhte_shard_step** steps, int32_t n, const float* const* value,
                                      const int64_t* value_len, const float* learning_rate,
                                      int64_t n_learning_rate, int64_t update_time,
                                      int64_t global_step, void* stream);

/* != 0 when table i fits the fused training step (row of <= 256 floats, or <= 64 when segment
 * boundaries are not multiples of 4 floats; no whole-segment optimizer): 1 = SGD / Adagrad / FTRL, 2 =
 * any per-element optimizer (Momentum, Adadelta, RMSProp, Adam, AMSGrad, MovingAverage, BatchSoftmax:
 * the step kernels' FULL instantiations).  0, and 2 for the unpipelined mhte_table_sum_optimize_n:
 * segment sum + optimize, which needs the ordered mhte_unique. */
int32_t mhte_table_fused_backward_ok(const mhte_multi_table* t, int32_t i);

/* MonolithUniqueKeyWithValueAndOffset (RT/ops/unique_mapping_ops.cc:51-155), one table at a time:
 * value_offset[q] = value_base + position*dim for the occurrence lists, value_offset_split[u] =
 * split_base + list start.  Inputs are the outputs of mhte_unique. */
mhte_status mhte_value_offsets(const uint32_t* seg_off, const uint32_t* seg_pos,
                               const uint32_t* n_unique_dev, int64_t n, int64_t value_base,
                               int64_t dim, int64_t split_base, int64_t* value_offset,
                               int64_t* value_offset_split, void* stream);

/* MonolithFillWithOffsetMap / MonolithFillWithOffsetMapGradient for ONE table
 * (RT/ops/unique_mapping_ops.cc:204-329).  pos [dev,n] indexes the unique-key list whose float
 * offsets into the flat buffer are offset_map[split[pos] .. split[pos+1]); value / backprop_grad
 * are [n, dim].  offsets_vec4 != 0 asserts that every offset is a multiple of 4 floats.
 * The reference's `pos < map size` InvalidArgument check is the caller's responsibility here. */
mhte_status mhte_fill_with_offset_map(const int64_t* pos, int64_t n, const float* value,
                                      const int64_t* value_offset_map,
                                      const int64_t* value_offset_map_split, int32_t dim,
                                      int32_t offsets_vec4, float* value_buffer, void* stream);
mhte_status mhte_fill_with_offset_map_gradient(const int64_t* pos, int64_t n, const float* grad,
                                               const int64_t* grad_offset_map,
                                               const int64_t* grad_offset_map_split, int32_t dim,
                                               int32_t offsets_vec4, float* backprop_grad,
                                               void* stream);

/* ---- the dense tower downstream of the embedding path -------------------------------------------
 * The ranking MLP that consumes fused_embedding_to_layout's output (native_training/layers/mlp.py:
 * Dense + ReLU stack ending in one logit; its gradient feeds fused_embedding_to_layout_grad): bf16
 * GEMMs on the matrix cores (csrc/mhte_gemm_kernels.h), fp32 master weights and accumulation, SGD
 * inside backward.  widths[n]: input width, hidden widths, 1 (the last layer has ONE output); every
 * width but the last and every batch are multiples of 128 (the GEMM tile).  Layer l in [0, n - 2]:
 * weight [widths[l+1]][widths[l]] row-major (torch.nn.Linear's layout), bias [widths[l+1]].
 *   forward   x [dev, batch x widths[0] fp32] -> y [dev, batch fp32]; keeps the activations
 *   backward  dy [dev, batch fp32] (the loss gradient at the logit) -> SGD step with learning_rate on
 *             every layer, dx [dev, batch x widths[0] fp32] when not NULL
 * One stream at a time; forward and backward of a step on the same stream. */
typedef struct mhte_dense_mlp mhte_dense_mlp;
mhte_status mhte_dense_mlp_create(const int32_t* widths, int32_t n_widths, int64_t max_batch,
                                  int32_t gpu_ordinal, mhte_dense_mlp** out);
void mhte_dense_mlp_destroy(mhte_dense_mlp* m);
mhte_status mhte_dense_mlp_set_params(mhte_dense_mlp* m, int32_t layer, const float* weight,
                                      const float* bias, void* stream);
mhte_status mhte_dense_mlp_get_params(mhte_dense_mlp* m, int32_t layer, float* weight, float* bias,
                                      void* stream);
mhte_status mhte_dense_mlp_forward(mhte_dense_mlp* m, const float* x, int64_t batch, float* y,
                                   void* stream);
mhte_status mhte_dense_mlp_backward(mhte_dense_mlp* m, const float* dy, float* dx,
                                    float learning_rate, void* stream);

/* ---- measurement aid (no reference counterpart) ---------------------------------------------
 * Kernel-exact timing of the hot kernels for bench.py's `roofline`: after mhte_profile_arm(n) the
 * next n launches of the step kernels made by the calling thread go through
 * hipExtLaunchKernelGGL, whose start/stop HIP events carry the kernel's own begin/end timestamps
 * on its queue (the interval rocprofv3 --kernel-trace reports).  mhte_profile_read disarms,
 * waits for the recorded launches and returns per launch the kernel tag and the duration in
 * microseconds.  Tags: 1 lookup_kernel, 2 sum_apply_kernel, 6 slowpath_kernel, 7 dd_* / rd_*
 * (dedup on its own), 8 upsert_kernel, 9 step_fwd_kernel, 10 step_bwd_kernel, 19 gemm_nt_bf16_kernel.
 * Not for use inside a stream capture. */
mhte_status mhte_profile_arm(int32_t n);
mhte_status mhte_profile_read(int32_t cap, int32_t* kernel_tag, float* usec, int32_t* n_out);
/* Per-wavefront timeline of the step kernels, for finding what bounds a launch.  Between
 * mhte_trace_begin(dev_buf, cap) and mhte_trace_end every launch of a step / lookup / fused-backward
 * kernel made by the calling thread writes, per wavefront w of the launch, eight uint64 words at
 * dev_buf[8 * (offset + w)]: {begin, end} on the 100 MHz wall clock, the role the wavefront played
 * and up to five intermediate time marks (0 = not reached); roles: the role the wavefront
 * played (3 run dedup, 4 displacement pass, 5 lookup, 6 heavy work list, 7 item workgroup of the
 * apply, 8 id-major workgroup of the apply).
 * dev_buf [dev, 8 * cap_records uint64].  mhte_trace_end stops tracing and returns, per traced
 * launch, the kernel tag (as above), grid and block size and `offset`. */
mhte_status mhte_trace_begin(void* dev_buf, int64_t cap_records);
mhte_status mhte_trace_end(int32_t cap, int32_t* kernel_tag, int32_t* grid, int32_t* block,
                           int64_t* offset, int32_t* n_out);

#ifdef __cplusplus
}
#endif
#endif /* MONOLITH_AMD_HASH_TABLE_H_ */
