/* Measurement aid (bench.py `timing_ms_per_step.eager_cpp`): K pipelined training steps of ONE table
 * enqueued by a plain C loop over the public C ABI — mhte_table_step_forward / mhte_table_step_backward,
 * two calls per step, no hipGraph — i.e. what a TF OpKernel pair
 * (MonolithMultiHashTableLookup / MonolithMultiHashTableOptimize, RT/ops/multi_hash_table_lookup_op.cc:33-89,
 * multi_hash_table_update_op.cc:47-100) would pay per call when the framework enqueues from C++.
 * C99, includes only include/monolith_amd_hash_table.h, links libmhte.so; built by
 * monolith_amd._lib.build_library into monolith_amd/libmhte_eager.so.  Not part of the op surface. */
#include <stddef.h>
#include <stdint.h>

#include "monolith_amd_hash_table.h"

/* the ABI version of the header this helper was compiled against (the binding refuses a stale helper) */
int32_t mhte_eager_abi_version(void) { return MHTE_ABI_VERSION; }

/* Steps [lo, hi) of a stream of batches laid out back to back in `ids` (batch s = ids + s * n).
 *   ws / unique_ids / n_unique_dev   three dedup workspaces and their result buffers, used in rotation
 *   cur          the slot that holds the run dedup of batch `lo` (made by the previous step or by
 *                mhte_step_dedup)
 *   grads        n_grads gradient buffers [n, dim], rotated: step s reads grads[s % n_grads]
 * Every step: forward = lookup of batch s + run dedup of batch s + 1 into the next slot; backward =
 * update of batch s + numbering / probe of batch s + 1.  Returns the slot that holds batch `hi`'s run
 * dedup, or -(1 + status) on the first failing call. */
int32_t mhte_eager_step_loop(mhte_multi_table* t, int32_t table, mhte_dedup_ws* const* ws,
                             int64_t* const* unique_ids, uint32_t* const* n_unique_dev, int32_t cur,
                             const int64_t* ids, int64_t n, int64_t lo, int64_t hi, float* embedding,
                             const float* const* grads, int32_t n_grads, float* grad_unique,
                             const float* learning_rate, int64_t n_learning_rate, int64_t update_time0,
                             int32_t flags, void* stream) {
  for (int64_t s = lo; s < hi; ++s) {
    const int32_t nxt = (cur + 1) % 3;
    mhte_status st = mhte_table_step_forward(t, table, ids + s * n, n, embedding, ws[nxt], ids + (s + 1) * n, n,
                                             unique_ids[nxt], n_unique_dev[nxt], NULL, stream);
    if (st != MHTE_OK) return -(1 + (int32_t)st);
    st = mhte_table_step_backward(t, table, ws[cur], ws[nxt], unique_ids[cur], n, n_unique_dev[cur],
                                  grads[s % n_grads], n, grad_unique, learning_rate, n_learning_rate,
                                  update_time0 + s, 0, flags, stream);
    if (st != MHTE_OK) return -(1 + (int32_t)st);
    cur = nxt;
  }
  return cur;
}
