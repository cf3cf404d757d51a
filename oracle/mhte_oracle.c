/* TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.  See mhte_oracle.h for scope and pinning.
 *
 * Plain-C, single-threaded restatement of the reference CPU path.  Paths below are relative to
 * /root/reference/monolith/native_training/runtime/.
 */
#include "mhte_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SLOTS 4               /* hash_table/cuckoohash/cuckoohash_config.hpp:26 */
#define MAX_BFS_PATH_LEN 5    /* hash_table/cuckoohash/cuckoohash_map.hpp:1432 */
#define MAX_CUCKOO_COUNT 682  /* 2*(4^5-1)/3, cuckoohash_map.hpp:1705-1709 */
#define ROWS_PER_BLOCK 4096
#define SEC_PER_DAY (24 * 60 * 60)

/* ------------------------------------------------------------------ hash */
/* The engine's documented fixed hash (murmur3 fmix64); stands in for absl::Hash
 * (cuckoohash_map.hpp:64-70), which is ASLR-seeded and not under /root/reference. */
uint64_t mo_hash(int64_t key) {
  uint64_t h = (uint64_t)key;
  h ^= h >> 33;
  h *= 0xff51afd7ed558ccdULL;
  h ^= h >> 33;
  h *= 0xc4ceb9fe1a85ec53ULL;
  h ^= h >> 33;
  return h;
}
/* cuckoohash_map.hpp:860-869 */
uint8_t mo_partial(uint64_t hash) {
  uint32_t h32 = (uint32_t)hash ^ (uint32_t)(hash >> 32);
  uint16_t h16 = (uint16_t)h32 ^ (uint16_t)(h32 >> 16);
  uint8_t h8 = (uint8_t)h16 ^ (uint8_t)(h16 >> 8);
  return h8;
}
static uint64_t hashmask(int hp) { return ((uint64_t)1 << hp) - 1; }
/* cuckoohash_map.hpp:873-875 */
static uint64_t index_hash(int hp, uint64_t hv) { return hv & hashmask(hp); }
/* cuckoohash_map.hpp:882-888 */
uint64_t mo_alt_index(int hp, uint8_t partial, uint64_t index) {
  uint64_t nonzero_tag = (uint64_t)partial + 1;
  return (index ^ (nonzero_tag * 0xc6a4a7935bd1e995ULL)) & hashmask(hp);
}

/* ------------------------------------------------------------------ table */
typedef struct {
  int64_t key[SLOTS];
  uint32_t row[SLOTS];
  uint32_t ts[SLOTS];
  uint8_t occ[SLOTS];
} bucket_t;

struct mo_table {
  int32_t nseg;
  mo_segment seg[8];
  int32_t state_off[8]; /* float offset of each segment's optimizer ctx within the row */
  int32_t dim;          /* sum of segment dims */
  int32_t row_floats;
  int hp;
  bucket_t* b;
  int64_t size;
  float** blocks;
  int64_t nblocks;
  uint32_t next_row;
  int64_t default_ttl_days;
  int32_t n_ttl;
  int64_t ttl_slot[256];
  int32_t ttl_days[256];
};

static int32_t state_floats(const mo_segment* s) {
  switch (s->opt) {
    case MO_OPT_SGD: return 0;               /* optimizer/sgd_optimizer.cc:28 */
    case MO_OPT_ADAGRAD: return s->dim;      /* optimizer/adagrad_optimizer.cc:30-32 */
    case MO_OPT_FTRL: return 2 * s->dim;     /* optimizer/ftrl_optimizer.cc:31-33 */
    case MO_OPT_MOMENTUM: return s->dim;     /* optimizer/momentum_optimizer.cc:30-32 */
    case MO_OPT_ADADELTA: return 2 * s->dim; /* optimizer/adadelta_optimizer.cc:30-32 */
    case MO_OPT_RMSPROP:
    case MO_OPT_RMSPROPV2: return s->dim;    /* optimizer/rmsprop_optimizer.cc:33-35,106-108 */
    case MO_OPT_ADAM: return 2 * s->dim + 2;     /* optimizer/adam_optimizer.cc:30-32 */
    case MO_OPT_AMSGRAD: return 3 * s->dim + 2;  /* optimizer/amsgrad_optimizer.cc:30-32 */
    case MO_OPT_MOVING_AVERAGE: return 0;        /* optimizer/moving_average_optimizer.cc:29 */
    case MO_OPT_BATCH_SOFTMAX: return 2;         /* optimizer/batch_softmax_optimizer.cc:33: one int64 */
    case MO_OPT_GROUP_ADAGRAD: return 1;         /* optimizer/group_adagrad_optimizer.cc:31: one float */
    default: return 0;
  }
}

/* cuckoohash_map.hpp:2114-2121 */
static int reserve_calc(uint64_t n) {
  uint64_t buckets = (n + SLOTS - 1) / SLOTS;
  int blog2 = 0;
  while (((uint64_t)1 << blog2) < buckets) ++blog2;
  return blog2;
}

mo_table* mo_table_new(int32_t nseg, const mo_segment* segs, uint64_t initial_capacity) {
  if (nseg < 1 || nseg > 8) return NULL;
  mo_table* t = (mo_table*)calloc(1, sizeof(mo_table));
  t->nseg = nseg;
  int32_t dim = 0;
  for (int i = 0; i < nseg; ++i) {
    t->seg[i] = segs[i];
    dim += segs[i].dim;
  }
  t->dim = dim;
  /* row = float num[dim] | opt ctx of seg0 | opt ctx of seg1 ... (entry_accessor.cc:113-114,
   * optimizer_combination.cc:56-60) */
  int32_t off = dim;
  for (int i = 0; i < nseg; ++i) {
    t->state_off[i] = off;
    off += state_floats(&segs[i]);
  }
  t->row_floats = off;
  t->hp = reserve_calc(initial_capacity);
  t->b = (bucket_t*)calloc((size_t)1 << t->hp, sizeof(bucket_t));
  t->default_ttl_days = 36500; /* hash_table/embedding_hash_table.proto:63 */
  return t;
}

void mo_table_free(mo_table* t) {
  if (!t) return;
  for (int64_t i = 0; i < t->nblocks; ++i) free(t->blocks[i]);
  free(t->blocks);
  free(t->b);
  free(t);
}

int32_t mo_dim(const mo_table* t) { return t->dim; }
int32_t mo_row_floats(const mo_table* t) { return t->row_floats; }
int32_t mo_slice_size(const mo_table* t) { return t->nseg; } /* SliceSize()=1 per optimizer */
int64_t mo_size(const mo_table* t) { return t->size; }
int32_t mo_hashpower(const mo_table* t) { return t->hp; }

static float* row_ptr(const mo_table* t, uint32_t r) {
  return t->blocks[r / ROWS_PER_BLOCK] + (size_t)(r % ROWS_PER_BLOCK) * t->row_floats;
}
/* PACKED rows come from a slab allocator that never frees individual rows
 * (cuckoo_embedding_hash_table.cc:56-73, allocator/block_allocator.h:126-). */
static uint32_t alloc_row(mo_table* t) {
  uint32_t r = t->next_row++;
  while ((uint64_t)t->nblocks * ROWS_PER_BLOCK <= r) {
    t->blocks = (float**)realloc(t->blocks, sizeof(float*) * (t->nblocks + 1));
    t->blocks[t->nblocks++] = (float*)malloc(sizeof(float) * ROWS_PER_BLOCK * t->row_floats);
  }
  return r;
}

/* cuckoohash_map.hpp:1234-1263 (is_simple(): partial tags are not compared for int64 keys) */
static int try_read_from_bucket(const bucket_t* b, int64_t key) {
  for (int i = 0; i < SLOTS; ++i) {
    if (!b->occ[i]) continue;
    if (b->key[i] == key) return i;
  }
  return -1;
}
static int64_t find_pos(const mo_table* t, int64_t key) {
  uint64_t hv = mo_hash(key);
  uint64_t i1 = index_hash(t->hp, hv);
  uint64_t i2 = mo_alt_index(t->hp, mo_partial(hv), i1);
  int s = try_read_from_bucket(&t->b[i1], key);
  if (s != -1) return (int64_t)(i1 * SLOTS + s);
  s = try_read_from_bucket(&t->b[i2], key);
  if (s != -1) return (int64_t)(i2 * SLOTS + s);
  return -1;
}
int32_t mo_contains(const mo_table* t, int64_t id) { return find_pos(t, id) >= 0; }
int64_t mo_locate(const mo_table* t, int64_t id) { return find_pos(t, id); }

/* cuckoohash_map.hpp:1398-1418: returns 0 if duplicate (slot = its index), else 1 and slot =
 * LAST empty slot or -1. */
static int try_find_insert_bucket(const bucket_t* b, int* slot, int64_t key) {
  *slot = -1;
  for (int i = 0; i < SLOTS; ++i) {
    if (b->occ[i]) {
      if (b->key[i] == key) {
        *slot = i;
        return 0;
      }
    } else {
      *slot = i;
    }
  }
  return 1;
}

typedef struct {
  uint64_t bucket;
  uint16_t pathcode;
  int8_t depth;
} b_slot;

/* cuckoohash_map.hpp:1725-1762 */
static b_slot slot_search(const mo_table* t, uint64_t i1, uint64_t i2) {
  static b_slot q[MAX_CUCKOO_COUNT];
  int first = 0, last = 0;
  b_slot a = {i1, 0, 0}, c = {i2, 1, 0};
  q[last++] = a;
  q[last++] = c;
  while (first != last) {
    b_slot x = q[first++];
    const bucket_t* b = &t->b[x.bucket];
    int starting_slot = x.pathcode % SLOTS;
    for (int i = 0; i < SLOTS; ++i) {
      uint16_t slot = (uint16_t)((starting_slot + i) % SLOTS);
      if (!b->occ[slot]) {
        x.pathcode = (uint16_t)(x.pathcode * SLOTS + slot);
        return x;
      }
      uint8_t partial = mo_partial(mo_hash(b->key[slot]));
      if (x.depth < MAX_BFS_PATH_LEN - 1) {
        b_slot y;
        y.bucket = mo_alt_index(t->hp, partial, x.bucket);
        y.pathcode = (uint16_t)(x.pathcode * SLOTS + slot);
        y.depth = (int8_t)(x.depth + 1);
        q[last++] = y;
      }
    }
  }
  b_slot fail = {0, 0, -1};
  return fail;
}

typedef struct {
  uint64_t bucket;
  int slot;
  uint64_t hash;
  uint8_t partial;
} cuckoo_rec;

/* cuckoohash_map.hpp:1508-1561 */
static int cuckoopath_search(const mo_table* t, cuckoo_rec* path, uint64_t i1, uint64_t i2) {
  b_slot x = slot_search(t, i1, i2);
  if (x.depth == -1) return -1;
  for (int i = x.depth; i >= 0; --i) {
    path[i].slot = x.pathcode % SLOTS;
    x.pathcode /= SLOTS;
  }
  path[0].bucket = (x.pathcode == 0) ? i1 : i2;
  {
    const bucket_t* b = &t->b[path[0].bucket];
    if (!b->occ[path[0].slot]) return 0;
    path[0].hash = mo_hash(b->key[path[0].slot]);
    path[0].partial = mo_partial(path[0].hash);
  }
  for (int i = 1; i <= x.depth; ++i) {
    path[i].bucket = mo_alt_index(t->hp, path[i - 1].partial, path[i - 1].bucket);
    const bucket_t* b = &t->b[path[i].bucket];
    if (!b->occ[path[i].slot]) return i;
    path[i].hash = mo_hash(b->key[path[i].slot]);
    path[i].partial = mo_partial(path[i].hash);
  }
  return x.depth;
}

/* cuckoohash_map.hpp:1569-1636 (single-threaded: the validity re-checks cannot fail) */
static int cuckoopath_move(mo_table* t, cuckoo_rec* path, int depth) {
  if (depth == 0) {
    return !t->b[path[0].bucket].occ[path[0].slot];
  }
  while (depth > 0) {
    cuckoo_rec* from = &path[depth - 1];
    cuckoo_rec* to = &path[depth];
    bucket_t* fb = &t->b[from->bucket];
    bucket_t* tb = &t->b[to->bucket];
    if (tb->occ[to->slot] || !fb->occ[from->slot] ||
        mo_hash(fb->key[from->slot]) != from->hash) {
      return 0;
    }
    tb->key[to->slot] = fb->key[from->slot];
    tb->row[to->slot] = fb->row[from->slot];
    tb->ts[to->slot] = fb->ts[from->slot];
    tb->occ[to->slot] = 1;
    fb->occ[from->slot] = 0;
    depth--;
  }
  return 1;
}

/* cuckoohash_map.hpp:1850-1894 */
static void move_bucket(const bucket_t* oldb, int old_hp, bucket_t* newb, int new_hp,
                        uint64_t old_ind) {
  const bucket_t* ob = &oldb[old_ind];
  uint64_t new_ind = old_ind + ((uint64_t)1 << old_hp);
  int new_slot = 0;
  for (int s = 0; s < SLOTS; ++s) {
    if (!ob->occ[s]) continue;
    uint64_t hv = mo_hash(ob->key[s]);
    uint8_t p = mo_partial(hv);
    uint64_t old_i = index_hash(old_hp, hv);
    uint64_t old_a = mo_alt_index(old_hp, p, old_i);
    uint64_t new_i = index_hash(new_hp, hv);
    uint64_t new_a = mo_alt_index(new_hp, p, new_i);
    uint64_t dst_b;
    int dst_s;
    if ((old_ind == old_i && new_i == new_ind) || (old_ind == old_a && new_a == new_ind)) {
      dst_b = new_ind;
      dst_s = new_slot++;
    } else {
      dst_b = old_ind;
      dst_s = s;
    }
    bucket_t* d = &newb[dst_b];
    d->key[dst_s] = ob->key[s];
    d->row[dst_s] = ob->row[s];
    d->ts[dst_s] = ob->ts[s];
    d->occ[dst_s] = 1;
  }
}

/* cuckoohash_map.hpp:1768-1848 (eager form of the lazy migration; same final placement) */
static void fast_double(mo_table* t) {
  int new_hp = t->hp + 1;
  bucket_t* nb = (bucket_t*)calloc((size_t)1 << new_hp, sizeof(bucket_t));
  uint64_t n_old = (uint64_t)1 << t->hp;
  for (uint64_t i = 0; i < n_old; ++i) move_bucket(t->b, t->hp, nb, new_hp, i);
  free(t->b);
  t->b = nb;
  t->hp = new_hp;
}

/* cuckoohash_map.hpp:573-590,1280-1380: find-or-insert.  Returns position; *inserted=1 when the
 * key was absent (slot reserved, key written, row NOT yet allocated). */
static int64_t upsert_pos(mo_table* t, int64_t key, int* inserted) {
  uint64_t hv = mo_hash(key);
  uint8_t partial = mo_partial(hv);
  for (;;) {
    uint64_t i1 = index_hash(t->hp, hv);
    uint64_t i2 = mo_alt_index(t->hp, partial, i1);
    int res1, res2;
    if (!try_find_insert_bucket(&t->b[i1], &res1, key)) {
      *inserted = 0;
      return (int64_t)(i1 * SLOTS + res1);
    }
    if (!try_find_insert_bucket(&t->b[i2], &res2, key)) {
      *inserted = 0;
      return (int64_t)(i2 * SLOTS + res2);
    }
    uint64_t ib;
    int is;
    if (res1 != -1) {
      ib = i1;
      is = res1;
    } else if (res2 != -1) {
      ib = i2;
      is = res2;
    } else {
      /* run_cuckoo, cuckoohash_map.hpp:1445-1497 */
      cuckoo_rec path[MAX_BFS_PATH_LEN];
      int done = 0;
      for (;;) {
        int depth = cuckoopath_search(t, path, i1, i2);
        if (depth < 0) break;
        if (cuckoopath_move(t, path, depth)) {
          done = 1;
          break;
        }
      }
      if (!done) {
        fast_double(t); /* failure_table_full -> cuckoo_fast_double, :1296-1299 */
        continue;
      }
      ib = path[0].bucket;
      is = path[0].slot;
    }
    bucket_t* b = &t->b[ib];
    b->key[is] = key;
    b->occ[is] = 1;
    b->row[is] = 0;
    b->ts[is] = 0;
    t->size++;
    *inserted = 1;
    return (int64_t)(ib * SLOTS + is);
  }
}

/* entry_accessor.cc:158-162: initializer then optimizer Init. */
static void init_row(const mo_table* t, float* row) {
  int32_t w = 0;
  for (int i = 0; i < t->nseg; ++i) {
    const mo_segment* s = &t->seg[i];
    float v = 0.f;
    if (s->init == MO_INIT_ONES) v = 1.f;
    if (s->init == MO_INIT_CONSTANT) v = s->init_value;
    for (int k = 0; k < s->dim; ++k) row[w + k] = v;
    float* st = row + t->state_off[i];
    if (s->opt == MO_OPT_GROUP_ADAGRAD) {
      st[0] = s->p[0]; /* group_adagrad_optimizer.cc:45-48 */
    } else if (s->opt == MO_OPT_ADAGRAD) {
      for (int k = 0; k < s->dim; ++k) st[k] = s->p[0]; /* adagrad_optimizer.cc:47-52 */
    } else if (s->opt == MO_OPT_FTRL) {
      for (int k = 0; k < s->dim; ++k) { /* ftrl_optimizer.cc:44-51: norm | zero */
        st[k] = s->p[0];
        st[s->dim + k] = 0.f;
      }
    } else if (s->opt == MO_OPT_ADAM || s->opt == MO_OPT_AMSGRAD) {
      /* adam_optimizer.cc:43-54 / amsgrad_optimizer.cc: vectors 0, powers = beta1, beta2 */
      int nv = s->opt == MO_OPT_ADAM ? 2 : 3;
      for (int k = 0; k < nv * s->dim; ++k) st[k] = 0.f;
      st[nv * s->dim] = s->p[0];
      st[nv * s->dim + 1] = s->p[1];
    } else {
      /* momentum / adadelta / rmsprop / batch softmax: Init() zeroes the context */
      for (int k = 0; k < state_floats(s); ++k) st[k] = 0.f;
    }
    w += s->dim;
  }
}

/* UpsertEntry (cuckoo_embedding_hash_table.cc:346-353): returns 1 if inserted; row is
 * initialised before the caller's update when new. */
static int upsert_entry(mo_table* t, int64_t id, bucket_t** pb, int* ps) {
  int inserted;
  int64_t pos = upsert_pos(t, id, &inserted);
  bucket_t* b = &t->b[pos / SLOTS];
  int s = (int)(pos % SLOTS);
  if (inserted) {
    b->row[s] = alloc_row(t);
    init_row(t, row_ptr(t, b->row[s]));
  }
  *pb = b;
  *ps = s;
  return inserted;
}

/* cuckoo_embedding_hash_table.cc:140-171 */
int64_t mo_lookup(const mo_table* t, const int64_t* ids, int64_t n, float* out) {
  int64_t found = 0;
  for (int64_t i = 0; i < n; ++i) {
    int64_t pos = find_pos(t, ids[i]);
    float* dst = out + i * t->dim;
    if (pos >= 0) {
      const bucket_t* b = &t->b[pos / SLOTS];
      memcpy(dst, row_ptr(t, b->row[pos % SLOTS]), sizeof(float) * t->dim);
      ++found;
    } else {
      memset(dst, 0, sizeof(float) * t->dim);
    }
  }
  return found;
}

/* :186-203 */
void mo_assign(mo_table* t, const int64_t* ids, int64_t n, const float* values,
               int64_t update_time) {
  for (int64_t i = 0; i < n; ++i) {
    bucket_t* b;
    int s;
    upsert_entry(t, ids[i], &b, &s);
    b->ts[s] = (uint32_t)update_time; /* entry_defs.h:36-38 */
    memcpy(row_ptr(t, b->row[s]), values + i * t->dim, sizeof(float) * t->dim);
  }
}

/* :206-213, entry_accessor.cc:179-185 */
void mo_assign_add(mo_table* t, const int64_t* ids, int64_t n, const float* values,
                   int64_t update_time) {
  for (int64_t i = 0; i < n; ++i) {
    bucket_t* b;
    int s;
    upsert_entry(t, ids[i], &b, &s);
    b->ts[s] = (uint32_t)update_time;
    float* row = row_ptr(t, b->row[s]);
    const float* v = values + i * t->dim;
    for (int k = 0; k < t->dim; ++k) row[k] += v[k];
  }
}

/* :215-226 */
void mo_reinitialize(mo_table* t, const int64_t* ids, int64_t n, int32_t* status, int64_t now) {
  for (int64_t i = 0; i < n; ++i) {
    bucket_t* b;
    int s;
    int inserted = upsert_entry(t, ids[i], &b, &s);
    b->ts[s] = (uint32_t)now;
    init_row(t, row_ptr(t, b->row[s]));
    status[i] = !inserted;
  }
}

/* optimizer/sgd_optimizer.cc:42-49 */
void mo_sgd(float* num, const float* grad, int64_t len, float lr) {
  for (int64_t i = 0; i < len; ++i) num[i] -= lr * grad[i];
}
/* optimizer/avx_utils.h:29-38 (BaselineAdagradOptimize) */
void mo_adagrad(float* num, float* norm, const float* grad, int64_t len, float lr, float wd) {
  for (int64_t i = 0; i < len; ++i) {
    float g = grad[i] + wd * num[i];
    norm[i] += g * g;
    float effective_lr = lr / sqrtf(norm[i]);
    num[i] -= effective_lr * g;
  }
}
/* optimizer/avx_utils.h:96-119 (Avx256AdagradOptimize) — what AdagradOptimize IS in the reference as
 * its .bazelrc:63-68 builds it: blocks of 8 elements with fused multiply-adds, the weight step taken
 * with the RAW gradient (:112, not grad + wd * w); the len % 8 tail falls back to the baseline loop
 * (:115-117) as that build compiles it.  Opt-in (segment p[2] != 0). */
void mo_adagrad_avx(float* num, float* norm, const float* grad, int64_t len, float lr, float wd) {
  int64_t i = 0;
  for (; i + 8 <= len; i += 8) {
    for (int k = 0; k < 8; ++k) {
      float ug = fmaf(wd, num[i + k], grad[i + k]);
      float nn = fmaf(ug, ug, norm[i + k]);
      norm[i + k] = nn;
      float eff = lr / sqrtf(nn);
      num[i + k] = fmaf(-eff, grad[i + k], num[i + k]);
    }
  }
  /* the tail: BaselineAdagradOptimize's formula (the weight step uses g = grad + wd * w), with the
   * multiply-adds the -mfma build contracts (gcc and clang both fuse each of its three statements) */
  for (; i < len; ++i) {
    float g = fmaf(wd, num[i], grad[i]);
    norm[i] = fmaf(g, g, norm[i]);
    float eff = lr / sqrtf(norm[i]);
    num[i] = fmaf(-eff, g, num[i]);
  }
}
/* optimizer/ftrl_optimizer.cc:56-75 (including the std::signbit quirk) */
static void mo_ftrl(float* num, float* norm, float* zero, const float* grad, int64_t len,
                    float lr, float beta, float l1, float l2) {
  for (int64_t i = 0; i < len; ++i) {
    float norm_new = norm[i] + grad[i] * grad[i];
    float sigma = (sqrtf(norm_new) - sqrtf(norm[i])) / lr;
    zero[i] += (grad[i] - sigma * num[i]);
    norm[i] = norm_new;
    num[i] = (fabsf(zero[i]) > l1)
                 ? lr * ((signbit(zero[i]) ? 1.f : 0.f) * l1 - zero[i]) /
                       (sqrtf(norm[i]) + beta + l2 * lr)
                 : 0.0f;
  }
}

/* optimizer/momentum_optimizer.cc:50-71; p = {momentum, weight_decay_factor, use_nesterov} */
static void mo_momentum(float* num, float* n, const float* grad, int64_t len, float lr,
                        const float* p) {
  for (int64_t i = 0; i < len; ++i) {
    float dx = lr * (grad[i] + p[1] * num[i]);
    float new_n = n[i];
    float new_w = num[i];
    if (p[2] != 0.f) {
      float prev_n = new_n;
      new_n = p[0] * new_n - dx;
      new_w += -p[0] * prev_n + (1 + p[0]) * new_n;
    } else {
      new_n = p[0] * new_n - dx;
      new_w += new_n;
    }
    n[i] = new_n;
    num[i] = new_w;
  }
}
/* optimizer/adadelta_optimizer.cc:51-72; p = {averaging_ratio, epsilon, weight_decay_factor} */
static void mo_adadelta(float* num, float* accum, float* accum_update, const float* grad,
                        int64_t len, float lr, const float* p) {
  for (int64_t i = 0; i < len; ++i) {
    float cur_grad = grad[i] + p[2] * num[i];
    float new_accum = accum[i] * p[0] + cur_grad * cur_grad * (1 - p[0]);
    float update = sqrtf(accum_update[i] + p[1]) / sqrtf(new_accum + p[1]) * cur_grad;
    float new_w = num[i] - update * lr;
    float new_accum_update = accum_update[i] * p[0] + update * update * (1 - p[0]);
    num[i] = new_w;
    accum[i] = new_accum;
    accum_update[i] = new_accum_update;
  }
}
/* optimizer/rmsprop_optimizer.cc:54-72 (v1: conf learning rate p[2]) and :127-144 (v2: lrs[0]);
 * p = {momentum, weight_decay_factor, conf learning_rate} */
static void mo_rmsprop(float* num, float* n, const float* grad, int64_t len, float lr,
                       const float* p, int v2) {
  for (int64_t i = 0; i < len; ++i) {
    float new_n = n[i];
    float new_w = num[i];
    double dx = grad[i] + (double)p[1] * new_w;
    if (v2) {
      new_n = (double)p[0] * new_n + dx * dx;
    } else {
      new_n = (double)p[0] * new_n + (1 - (double)p[0]) * dx * dx;
    }
    double eta = (double)(v2 ? lr : p[2]) / (sqrtf(new_n) + 1);
    new_w -= eta * dx;
    n[i] = new_n;
    num[i] = new_w;
  }
}
/* optimizer/adam_optimizer.cc:56-86, amsgrad_optimizer.cc (vhat != NULL);
 * p = {beta1, beta2, epsilon, weight_decay_factor, use_nesterov}; ctx = m | v | [vhat |] powers.
 * The reference calls `sqrt` UNQUALIFIED (adam_optimizer.cc:64,74,76; amsgrad_optimizer.cc:66,77,79)
 * after <cmath> only: that is ::sqrt(double) (libstdc++'s <cmath> puts the float overload in std::
 * alone), so the effective learning rate and the quotient of every element are formed in double and
 * rounded to float once, at the assignment — restated so here. */
static void mo_adam(float* num, float* ctx, const float* grad, int64_t len, float lr0,
                    const float* p, int amsgrad) {
  float* m = ctx;
  float* v = m + len;
  float* vhat = amsgrad ? v + len : NULL;
  float* pw = (amsgrad ? vhat : v) + len;
  float lr = (float)((double)lr0 * sqrt((double)(1 - pw[1])) / (double)(1 - pw[0]));
  for (int64_t i = 0; i < len; ++i) {
    float cur_grad = grad[i] + p[3] * num[i];
    float new_m = m[i] + (cur_grad - m[i]) * (1 - p[0]);
    float new_v = v[i] + (cur_grad * cur_grad - v[i]) * (1 - p[1]);
    float den_v = new_v;
    if (amsgrad) {
      den_v = vhat[i] > new_v ? vhat[i] : new_v;
      vhat[i] = den_v;
    }
    float new_w = num[i];
    if (p[4] != 0.f) {
      float numr = (cur_grad * (1 - p[0]) + p[0] * new_m) * lr;
      new_w = (float)((double)new_w - (double)numr / (sqrt((double)den_v) + (double)p[2]));
    } else {
      float numr = new_m * lr;
      new_w = (float)((double)new_w - (double)numr / (sqrt((double)den_v) + (double)p[2]));
    }
    num[i] = new_w;
    m[i] = new_m;
    v[i] = new_v;
  }
  pw[0] *= p[0];
  pw[1] *= p[1];
}

/* optimizer/group_adagrad_optimizer.cc:50-93; p = {initial_accumulator_value, beta,
 * l2_regularization_strength, weight_decay_factor} */
static void mo_group_adagrad(float* num, float* grad_square_sum, const float* grad, int64_t len,
                             float effective_lr, const float* p) {
  const float beta = p[1], l2 = p[2], wd = p[3];
  float max_grad_square = 0.f;
  float* g_decayed = (float*)malloc(sizeof(float) * (size_t)(len > 0 ? len : 1));
  for (int64_t i = 0; i < len; ++i) {
    float g = grad[i] + wd * num[i];
    if (g * g > max_grad_square) max_grad_square = g * g;
    g_decayed[i] = g;
  }
  *grad_square_sum = *grad_square_sum + max_grad_square;
  float lr = effective_lr / (beta + sqrtf(*grad_square_sum));
  float z_norm = 0.f;
  for (int64_t i = 0; i < len; ++i) {
    num[i] = g_decayed[i] - num[i] / lr;
    z_norm += num[i] * num[i];
  }
  z_norm = sqrtf(z_norm);
  if (z_norm < l2) {
    for (int64_t i = 0; i < len; ++i) num[i] = 0;
  } else {
    float coeffi = -lr * (z_norm - l2) / z_norm;
    for (int64_t i = 0; i < len; ++i) num[i] = coeffi * num[i];
  }
  free(g_decayed);
}

/* optimizer/moving_average_optimizer.cc:44-52 */
static void mo_moving_average(float* num, const float* grad, int64_t len, const float* p) {
  const float momentum = p[0];
  for (int64_t i = 0; i < len; ++i) {
    float new_w = momentum * num[i] + (1 - momentum) * grad[i];
    num[i] = new_w;
  }
}

/* optimizer/batch_softmax_optimizer.cc:52-63 (dim_size 1; ctx = int64 A, the id's last step) */
static void mo_batch_softmax(float* num, float* ctx, float alpha, int64_t global_step) {
  int64_t A;
  memcpy(&A, ctx, sizeof(A));
  num[0] = (1 - alpha) * num[0] + alpha * (float)(global_step - A);
  A = global_step;
  memcpy(ctx, &A, sizeof(A));
}

/* :229-247 + entry_accessor.cc:187-195 + optimizer_combination.cc:63-72 */
void mo_optimize(mo_table* t, const int64_t* ids, int64_t n, const float* grads, const float* lrs,
                 int64_t update_time, int64_t global_step) {
  for (int64_t i = 0; i < n; ++i) {
    bucket_t* b;
    int s;
    upsert_entry(t, ids[i], &b, &s);
    b->ts[s] = (uint32_t)update_time;
    float* row = row_ptr(t, b->row[s]);
    const float* g = grads + i * t->dim;
    int32_t w = 0;
    for (int k = 0; k < t->nseg; ++k) {
      const mo_segment* sg = &t->seg[k];
      float* st = row + t->state_off[k];
      if (sg->opt == MO_OPT_SGD) {
        mo_sgd(row + w, g + w, sg->dim, lrs[k]);
      } else if (sg->opt == MO_OPT_ADAGRAD) {
        if (sg->p[2] != 0.f) mo_adagrad_avx(row + w, st, g + w, sg->dim, lrs[k], sg->p[1]);
        else mo_adagrad(row + w, st, g + w, sg->dim, lrs[k], sg->p[1]);
      } else if (sg->opt == MO_OPT_FTRL) {
        mo_ftrl(row + w, st, st + sg->dim, g + w, sg->dim, lrs[k], sg->p[1], sg->p[2], sg->p[3]);
      } else if (sg->opt == MO_OPT_MOMENTUM) {
        mo_momentum(row + w, st, g + w, sg->dim, lrs[k], sg->p);
      } else if (sg->opt == MO_OPT_ADADELTA) {
        mo_adadelta(row + w, st, st + sg->dim, g + w, sg->dim, lrs[k], sg->p);
      } else if (sg->opt == MO_OPT_RMSPROP || sg->opt == MO_OPT_RMSPROPV2) {
        mo_rmsprop(row + w, st, g + w, sg->dim, lrs[k], sg->p, sg->opt == MO_OPT_RMSPROPV2);
      } else if (sg->opt == MO_OPT_ADAM || sg->opt == MO_OPT_AMSGRAD) {
        mo_adam(row + w, st, g + w, sg->dim, lrs[k], sg->p, sg->opt == MO_OPT_AMSGRAD);
      } else if (sg->opt == MO_OPT_MOVING_AVERAGE) {
        mo_moving_average(row + w, g + w, sg->dim, sg->p);
      } else if (sg->opt == MO_OPT_BATCH_SOFTMAX) {
        mo_batch_softmax(row + w, st, lrs[k], global_step);
      } else if (sg->opt == MO_OPT_GROUP_ADAGRAD) {
        mo_group_adagrad(row + w, st, g + w, sg->dim, lrs[k], sg->p);
      }
      w += sg->dim;
    }
  }
}

void mo_set_ttl(mo_table* t, int64_t default_days, int32_t n, const int64_t* slots,
                const int32_t* days) {
  t->default_ttl_days = default_days;
  t->n_ttl = n > 256 ? 256 : n;
  for (int i = 0; i < t->n_ttl; ++i) {
    t->ttl_slot[i] = slots[i];
    t->ttl_days[i] = days[i];
  }
}

/* cuckoo_embedding_hash_table.cc:251-264 + cuckoohash_map.hpp:775-799;
 * slot_id_v2: data/training_instance/cc/reader_util.h:36-38 */
void mo_evict(mo_table* t, int64_t max_update_time) {
  uint64_t nb = (uint64_t)1 << t->hp;
  for (uint64_t i = 0; i < nb; ++i) {
    bucket_t* b = &t->b[i];
    for (int s = 0; s < SLOTS; ++s) {
      if (!b->occ[s]) continue;
      int64_t slot = (b->key[s] >> 48) & 0x7fff;
      int64_t ttl = t->default_ttl_days;
      for (int k = 0; k < t->n_ttl; ++k)
        if (t->ttl_slot[k] == slot) ttl = t->ttl_days[k];
      if (max_update_time - (int64_t)b->ts[s] >= ttl * SEC_PER_DAY) {
        b->occ[s] = 0;
        t->size--;
      }
    }
  }
}

int64_t mo_dump(const mo_table* t, int64_t cap, int64_t* ids, int64_t* positions, uint32_t* ts,
                float* rows) {
  uint64_t nb = (uint64_t)1 << t->hp;
  int64_t n = 0;
  for (uint64_t i = 0; i < nb && n < cap; ++i) {
    const bucket_t* b = &t->b[i];
    for (int s = 0; s < SLOTS && n < cap; ++s) {
      if (!b->occ[s]) continue;
      ids[n] = b->key[s];
      positions[n] = (int64_t)(i * SLOTS + s);
      ts[n] = b->ts[s];
      if (rows)
        memcpy(rows + n * t->row_floats, row_ptr(t, b->row[s]), sizeof(float) * t->row_floats);
      ++n;
    }
  }
  return n;
}

/* ------------------------------------------------------------------ tiny int64 -> int64 map */
typedef struct {
  int64_t* k;
  int64_t* v;
  uint8_t* used;
  uint64_t mask;
} imap;
static void imap_init(imap* m, int64_t n) {
  uint64_t cap = 16;
  while (cap < (uint64_t)(2 * n + 2)) cap <<= 1;
  m->k = (int64_t*)malloc(sizeof(int64_t) * cap);
  m->v = (int64_t*)malloc(sizeof(int64_t) * cap);
  m->used = (uint8_t*)calloc(cap, 1);
  m->mask = cap - 1;
}
static void imap_clear(imap* m) { memset(m->used, 0, m->mask + 1); }
static void imap_free(imap* m) {
  free(m->k);
  free(m->v);
  free(m->used);
}
/* returns pointer to the value slot; *fresh=1 if the key was absent (value uninitialised) */
static int64_t* imap_get(imap* m, int64_t key, int* fresh) {
  uint64_t i = mo_hash(key) & m->mask;
  while (m->used[i]) {
    if (m->k[i] == key) {
      *fresh = 0;
      return &m->v[i];
    }
    i = (i + 1) & m->mask;
  }
  m->used[i] = 1;
  m->k[i] = key;
  *fresh = 1;
  return &m->v[i];
}

/* ops/unique_mapping_ops.cc:51-155 */
int64_t mo_unique_key_with_value_and_offset(const int64_t* key, const int64_t* key_split,
                                            int32_t T, const int32_t* dims, int64_t* unique_key,
                                            int64_t* unique_key_split, int64_t* value_offset,
                                            int64_t* value_offset_split,
                                            int64_t* value_buffer_len) {
  int64_t n = key_split[T];
  int64_t nu = 0, vo_n = 0, vos_n = 0, value_off = 0;
  unique_key_split[0] = 0;
  value_offset_split[vos_n++] = 0;
  imap m;
  imap_init(&m, n);
  /* per-key occurrence lists: head/next chains in occurrence order */
  int64_t* uidx_of = (int64_t*)malloc(sizeof(int64_t) * (n + 1));
  int64_t* cnt = (int64_t*)calloc(n + 1, sizeof(int64_t));
  int64_t* off_of_pos = (int64_t*)malloc(sizeof(int64_t) * (n + 1));
  for (int32_t j = 0; j < T; ++j) {
    imap_clear(&m);
    int64_t u0 = nu;
    for (int64_t i = key_split[j]; i < key_split[j + 1]; ++i) {
      int fresh;
      int64_t* v = imap_get(&m, key[i], &fresh);
      if (fresh) {
        *v = nu;
        unique_key[nu++] = key[i];
      }
      uidx_of[i] = *v;
      cnt[*v]++;
      off_of_pos[i] = value_off;
      value_off += dims[j];
    }
    unique_key_split[j + 1] = nu;
    /* emit the offset lists of this table's unique keys, in unique order, each list in
     * occurrence order (InlinedVector push_back order, :109-113,88-94) */
    int64_t* start = (int64_t*)malloc(sizeof(int64_t) * (nu - u0 + 1));
    int64_t acc = vo_n;
    for (int64_t u = u0; u < nu; ++u) {
      start[u - u0] = acc;
      acc += cnt[u];
    }
    for (int64_t i = key_split[j]; i < key_split[j + 1]; ++i) {
      value_offset[start[uidx_of[i] - u0]++] = off_of_pos[i];
    }
    for (int64_t u = u0; u < nu; ++u) value_offset_split[vos_n++] = start[u - u0];
    vo_n = acc;
    free(start);
  }
  free(uidx_of);
  free(cnt);
  free(off_of_pos);
  imap_free(&m);
  *value_buffer_len = value_off;
  return nu;
}

/* ops/unique_mapping_ops.cc:204-268 ; returns 0 ok, nonzero = InvalidArgument */
int32_t mo_fill_with_offset_map(const int64_t* pos, const int64_t* pos_split, int32_t T,
                                const int32_t* dims, const float* value, int64_t value_len,
                                const int64_t* value_offset_map, int64_t value_offset_map_len,
                                const int64_t* value_offset_map_split, float* value_buffer) {
  int64_t value_off = 0;
  int32_t j = 0;
  int64_t n = pos_split[T];
  for (int64_t i = 0; i < n; ++i) {
    while (i == pos_split[j + 1]) ++j;
    if (pos[i] >= value_offset_map_len) return 1;
    int64_t end = value_off + dims[j];
    if (end > value_len) return 2;
    for (int64_t q = value_offset_map_split[pos[i]]; q < value_offset_map_split[pos[i] + 1]; ++q) {
      memcpy(value_buffer + value_offset_map[q], value + value_off, sizeof(float) * dims[j]);
    }
    value_off = end;
  }
  return 0;
}

/* ops/unique_mapping_ops.cc:284-329 */
int32_t mo_fill_with_offset_map_gradient(const int64_t* pos, const int64_t* pos_split, int32_t T,
                                         const int32_t* dims, const float* grad,
                                         const int64_t* grad_offset_map,
                                         int64_t grad_offset_map_len,
                                         const int64_t* grad_offset_map_split, float* bgrad) {
  int64_t bsize = 0;
  for (int32_t j = 0; j < T; ++j) bsize += (int64_t)dims[j] * (pos_split[j + 1] - pos_split[j]);
  for (int64_t i = 0; i < bsize; ++i) bgrad[i] = 0.f;
  int64_t boff = 0;
  int32_t j = 0;
  int64_t n = pos_split[T];
  for (int64_t i = 0; i < n; ++i) {
    while (i == pos_split[j + 1]) ++j;
    if (pos[i] >= grad_offset_map_len) return 1;
    for (int64_t q = grad_offset_map_split[pos[i]]; q < grad_offset_map_split[pos[i] + 1]; ++q) {
      int64_t go = grad_offset_map[q];
      for (int k = 0; k < dims[j]; ++k) bgrad[boff + k] += grad[go + k];
    }
    boff += dims[j];
  }
  return 0;
}

/* hash_table/utils.h:29-61 */
void mo_compute_fused_offsets(const int32_t* slot_size_vec, const int32_t* table_dims,
                              int32_t num_tables, int32_t num_shards, int32_t* key_offsets,
                              int32_t* emb_offsets, int32_t* keys_per_table, int32_t* emb_splits,
                              int32_t* total_keys_out, int32_t* total_embs_out) {
  if (keys_per_table)
    for (int i = 0; i < num_tables; ++i) keys_per_table[i] = 0;
  int total_keys = 0, total_embs = 0, prev_total_emb = 0;
  key_offsets[0] = emb_offsets[0] = 0;
  for (int shard_id = 0; shard_id < num_shards; shard_id++) {
    for (int table_id = 0; table_id < num_tables; table_id++) {
      int idx = num_tables * shard_id + table_id;
      int slot_sz = slot_size_vec[idx];
      int segment_dim = table_dims[table_id] * slot_sz;
      if (keys_per_table) keys_per_table[table_id] += slot_sz;
      total_keys += slot_sz;
      total_embs += segment_dim;
      key_offsets[idx + 1] = key_offsets[idx] + slot_sz;
      emb_offsets[idx + 1] = emb_offsets[idx] + segment_dim;
    }
    emb_splits[shard_id] = total_embs - prev_total_emb;
    prev_total_emb = total_embs;
  }
  *total_keys_out = total_keys;
  *total_embs_out = total_embs;
}

/* ops/fused_reorder_by_indices.cc:38-123 */
int64_t mo_fused_reorder_by_indices(const int64_t* input, const int64_t* input_split, int32_t M,
                                    int32_t N, const int32_t* dims, int32_t rank0_empty,
                                    int64_t* output, int32_t* shard_sizes,
                                    int32_t* sharded_slot_sizes, int32_t* emb_offset_sz,
                                    int32_t* fused_emb_offset) {
  int64_t total = input_split[M];
  /* per (shard,table) id lists */
  int64_t** lists = (int64_t**)calloc((size_t)N * M, sizeof(int64_t*));
  int64_t* lens = (int64_t*)calloc((size_t)N * M, sizeof(int64_t));
  int64_t* within = (int64_t*)malloc(sizeof(int64_t) * (total + 1)); /* ids_sets[m][val] */
  imap m;
  imap_init(&m, total);
  for (int32_t t = 0; t < M; ++t) {
    imap_clear(&m);
    int64_t sz = input_split[t + 1] - input_split[t];
    for (int n = 0; n < N; ++n) lists[n * M + t] = (int64_t*)malloc(sizeof(int64_t) * (sz + 1));
    for (int64_t i = input_split[t]; i < input_split[t + 1]; ++i) {
      int64_t val = input[i];
      int shard = (int)(val % (N - rank0_empty) + rank0_empty); /* shard_func :121-123 */
      int64_t idx = (int64_t)shard * M + t;
      int fresh;
      int64_t* v = imap_get(&m, val, &fresh);
      if (fresh) {
        *v = lens[idx] * dims[t];
        lists[idx][lens[idx]++] = val;
      }
      within[i] = *v;
    }
  }
  for (int n = 0; n < N; ++n) shard_sizes[n] = 0;
  int64_t uniq = 0;
  int32_t emb_offset = 0;
  int32_t* emb_offsets_cm = (int32_t*)malloc(sizeof(int32_t) * N * M);
  for (int n = 0; n < N; ++n) {
    for (int t = 0; t < M; ++t) {
      int idx = n * M + t;
      int64_t sz = lens[idx];
      sharded_slot_sizes[idx] = (int32_t)sz;
      shard_sizes[n] += (int32_t)sz;
      uniq += sz;
      emb_offsets_cm[t * N + n] = emb_offset;
      emb_offset += (int32_t)sz * dims[t];
    }
  }
  int64_t w = 0;
  for (int idx = 0; idx < N * M; ++idx) {
    memcpy(output + w, lists[idx], sizeof(int64_t) * lens[idx]);
    w += lens[idx];
  }
  for (int32_t t = 0; t < M; ++t) {
    emb_offset_sz[t] = (int32_t)(input_split[t + 1] - input_split[t]);
    for (int64_t i = input_split[t]; i < input_split[t + 1]; ++i) {
      int64_t val = input[i];
      int shard = (int)(val % (N - rank0_empty) + rank0_empty);
      fused_emb_offset[i] = (int32_t)(within[i] + emb_offsets_cm[shard + t * N]);
    }
  }
  for (int idx = 0; idx < N * M; ++idx) free(lists[idx]);
  free(lists);
  free(lens);
  free(within);
  free(emb_offsets_cm);
  imap_free(&m);
  return uniq;
}

/* ---------------------------------------------------------------------------------------------
 * fp16 stochastic rounding (runtime/hash_table/optimizer/stochastic_rounding.h:27-40): the two
 * binary16 neighbours of vf — float2half<round_toward_infinity> / <round_toward_neg_infinity> of
 * third_party/half_sourceforge_net/half.hpp, converted back — and the upper one iff
 * p <= (vf - down) / (up - down) in fp32 (0 / 0 for an exactly representable vf is NaN: the lower,
 * equal, neighbour).  Pinned value for value to the reference's own function compiled in place
 * (oracle/ref_sr_driver.cc, tests/test_stochastic_rounding.py).
 * ------------------------------------------------------------------------------------------- */
static uint16_t mo_float_to_half_dir(float f, int toward_pos_inf) {
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = x >> 31, exp = (x >> 23) & 0xffu, man = x & 0x7fffffu;
  const uint16_t hs = (uint16_t)(sign << 15);
  if (exp == 0xffu) return (uint16_t)(hs | (man ? 0x7e00u : 0x7c00u));
  /* inexact results move the MAGNITUDE up when rounding away from zero on this side */
  const int mag_up = toward_pos_inf ? !sign : (int)sign;
  const int e = (int)exp - 127;
  uint32_t h, rem;
  if (exp == 0) {                 /* float zero / subnormal: far below the smallest half */
    h = 0;
    rem = man;
  } else if (e >= 16) {           /* beyond the largest finite half (65504) */
    return (uint16_t)(hs | (mag_up ? 0x7c00u : 0x7bffu));
  } else if (e >= -14) {          /* normal half */
    h = ((uint32_t)(e + 15) << 10) | (man >> 13);
    rem = man & 0x1fffu;
  } else {                        /* subnormal half: units of 2^-24 */
    const uint32_t full = 0x800000u | man;
    const int sh = 13 + (-14 - e);
    if (sh >= 32) {
      h = 0;
      rem = 1;
    } else {
      h = full >> sh;
      rem = full & ((1u << sh) - 1u);
    }
  }
  if (rem && mag_up) h += 1;      /* (a carry out of the mantissa is the next exponent, up to inf) */
  return (uint16_t)(hs | h);
}

static float mo_half_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h >> 15) << 31, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
  uint32_t x;
  if (exp == 0x1fu) {
    x = sign | 0x7f800000u | (man << 13);
  } else if (exp) {
    x = sign | ((exp + 112u) << 23) | (man << 13);
  } else if (man) {               /* subnormal: man * 2^-24 */
    int e = -1;
    uint32_t m = man;
    while (!(m & 0x400u)) {
      m <<= 1;
      ++e;
    }
    x = sign | ((uint32_t)(112 - e) << 23) | ((m & 0x3ffu) << 13);
  } else {
    x = sign;
  }
  float f;
  memcpy(&f, &x, 4);
  return f;
}

float mo_half_up(float vf) { return mo_half_to_float(mo_float_to_half_dir(vf, 1)); }
float mo_half_down(float vf) { return mo_half_to_float(mo_float_to_half_dir(vf, 0)); }

float mo_stochastic_round(float vf, float p) {
  const float up = mo_half_up(vf), down = mo_half_down(vf);
  const float num = vf - down, den = up - down;
  const float frac = num / den;
  return (p <= frac) ? up : down;
}
