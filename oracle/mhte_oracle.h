/* TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.
 *
 * CPU restatement ("oracle") of the reference's collisionless embedding-table hot path, in
 * plain C.  Every function cites the reference file:line it follows (paths relative to
 * /root/reference/monolith/native_training/runtime/).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product (monolith_amd/) never does.
 *
 * Pinning (see oracle/README.md and tests/test_oracle_*.py):
 *   - table mechanics + physical placement: checked against oracle/_ref (the reference's own
 *     cuckoohash_map.hpp compiled with the same fixed hash) on seeded workloads;
 *   - arithmetic: checked against the reference's avx_utils.h (oracle/_ref) and the golden
 *     values of adagrad_optimizer_test.cc:32-88, sgd tests, hash_table_ops_test.py,
 *     multi_hash_table_ops_test.py, distribution_ops_test.py (tests/golden/).
 *   - physical placement vs an absl::Hash-seeded reference run: PARITY UNPINNED (absl source
 *     is not under /root/reference and its hash is ASLR-seeded).
 */
#ifndef MHTE_ORACLE_H_
#define MHTE_ORACLE_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MO_OPT_SGD = 0, MO_OPT_ADAGRAD = 1, MO_OPT_FTRL = 2, MO_OPT_MOMENTUM = 3, MO_OPT_ADADELTA = 4,
       MO_OPT_RMSPROP = 5, MO_OPT_RMSPROPV2 = 6, MO_OPT_ADAM = 7, MO_OPT_AMSGRAD = 8,
       MO_OPT_MOVING_AVERAGE = 9, MO_OPT_BATCH_SOFTMAX = 10, MO_OPT_GROUP_ADAGRAD = 11 };
enum { MO_INIT_ZEROS = 0, MO_INIT_ONES = 1, MO_INIT_CONSTANT = 2 };

/* One EntryConfig.Segment (hash_table/embedding_hash_table.proto:23-43). */
typedef struct {
  int32_t dim;
  int32_t opt;       /* MO_OPT_* */
  float p[8];        /* adagrad: p[0]=initial_accumulator_value p[1]=weight_decay_factor
                                 p[2]!=0: the reference's AVX form (avx_utils.h:96-119)
                        ftrl:    p[0]=initial_accumulator_value p[1]=beta p[2]=l1 p[3]=l2 */
  int32_t init;      /* MO_INIT_* */
  float init_value;  /* constant initializer value */
} mo_segment;

typedef struct mo_table mo_table;

float mo_half_up(float vf);     /* nearest binary16 value >= vf (as float) */
float mo_half_down(float vf);   /* nearest binary16 value <= vf */
float mo_stochastic_round(float vf, float p);  /* stochastic_rounding.h:27-40 */
uint64_t mo_hash(int64_t key);                       /* engine's fixed hash (fmix64) */
uint8_t mo_partial(uint64_t hash);                   /* cuckoohash_map.hpp:860-869 */
uint64_t mo_alt_index(int hp, uint8_t partial, uint64_t index); /* :882-888 */

mo_table* mo_table_new(int32_t nseg, const mo_segment* segs, uint64_t initial_capacity);
void mo_table_free(mo_table* t);
int32_t mo_dim(const mo_table* t);
int32_t mo_row_floats(const mo_table* t);
int32_t mo_slice_size(const mo_table* t);
int64_t mo_size(const mo_table* t);
int32_t mo_hashpower(const mo_table* t);
int32_t mo_contains(const mo_table* t, int64_t id);
/* returns bucket*4+slot or -1 */
int64_t mo_locate(const mo_table* t, int64_t id);

int64_t mo_lookup(const mo_table* t, const int64_t* ids, int64_t n, float* out);
void mo_assign(mo_table* t, const int64_t* ids, int64_t n, const float* values, int64_t update_time);
void mo_assign_add(mo_table* t, const int64_t* ids, int64_t n, const float* values,
                   int64_t update_time);
void mo_reinitialize(mo_table* t, const int64_t* ids, int64_t n, int32_t* status, int64_t now);
/* lrs: one float per segment (SliceSize), optimizer_combination.cc:63-72 */
void mo_optimize(mo_table* t, const int64_t* ids, int64_t n, const float* grads, const float* lrs,
                 int64_t update_time, int64_t global_step);
void mo_set_ttl(mo_table* t, int64_t default_days, int32_t n, const int64_t* slots,
                const int32_t* days);
void mo_evict(mo_table* t, int64_t max_update_time);
/* dump in bucket-major, slot-minor order (cuckoohash_map.hpp:740-773). rows may be NULL. */
int64_t mo_dump(const mo_table* t, int64_t cap, int64_t* ids, int64_t* positions, uint32_t* ts,
                float* rows);

/* ---- optimizer arithmetic on raw buffers (optimizer/test_utils.h style) ---- */
void mo_sgd(float* num, const float* grad, int64_t len, float lr);
void mo_adagrad(float* num, float* norm, const float* grad, int64_t len, float lr, float wd);
/* avx_utils.h:96-119: the AVX2/FMA form (blocks of 8 fused, raw gradient in the weight step) */
void mo_adagrad_avx(float* num, float* norm, const float* grad, int64_t len, float lr, float wd);

/* ---- caller-side dedup / packing ops ---- */
/* ops/unique_mapping_ops.cc:51-155.  key_split has T+1 entries; dims has T entries.
 * Outputs: unique_key (cap n), unique_key_split (T+1), value_offset (n),
 * value_offset_split (cap n+1).  Returns number of unique keys; *value_buffer_len receives
 * the flat value-buffer length (sum n_t*dims_t). */
int64_t mo_unique_key_with_value_and_offset(const int64_t* key, const int64_t* key_split,
                                            int32_t T, const int32_t* dims, int64_t* unique_key,
                                            int64_t* unique_key_split, int64_t* value_offset,
                                            int64_t* value_offset_split,
                                            int64_t* value_buffer_len);
/* ops/unique_mapping_ops.cc:204-268 */
int32_t mo_fill_with_offset_map(const int64_t* pos, const int64_t* pos_split, int32_t T,
                                const int32_t* dims, const float* value, int64_t value_len,
                                const int64_t* value_offset_map, int64_t value_offset_map_len,
                                const int64_t* value_offset_map_split, float* value_buffer);
/* ops/unique_mapping_ops.cc:284-329 ; bgrad has sum dims_t*(pos_split[t+1]-pos_split[t]) */
int32_t mo_fill_with_offset_map_gradient(const int64_t* pos, const int64_t* pos_split, int32_t T,
                                         const int32_t* dims, const float* grad,
                                         const int64_t* grad_offset_map,
                                         int64_t grad_offset_map_len,
                                         const int64_t* grad_offset_map_split, float* bgrad);
/* hash_table/utils.h:29-61 */
void mo_compute_fused_offsets(const int32_t* slot_size_vec, const int32_t* table_dims,
                              int32_t num_tables, int32_t num_shards, int32_t* key_offsets,
                              int32_t* emb_offsets, int32_t* keys_per_table, int32_t* emb_splits,
                              int32_t* total_keys, int32_t* total_embs);
/* ops/fused_reorder_by_indices.cc:38-123.  inputs: M id vectors given as one concatenated
 * array + input_split (M+1).  Outputs: output (cap total), shard_sizes (N),
 * sharded_slot_sizes (N*M), emb_offset_sz (M), fused_emb_offset (total).  Returns the number
 * of deduped ids written to `output`. */
int64_t mo_fused_reorder_by_indices(const int64_t* input, const int64_t* input_split, int32_t M,
                                    int32_t num_shards, const int32_t* slot_embedding_dims,
                                    int32_t rank0_empty, int64_t* output, int32_t* shard_sizes,
                                    int32_t* sharded_slot_sizes, int32_t* emb_offset_sz,
                                    int32_t* fused_emb_offset);

#ifdef __cplusplus
}
#endif
#endif /* MHTE_ORACLE_H_ */
