/* TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.
 *
 * Plain-C restatement of the reference's occurrence filter, the checker of the device filter
 * (csrc/mhte_kernels.h filter_consult / filter_get_kernel / filter_advance_kernel):
 *   SlidingHashFilter          runtime/hash_filter/sliding_hash_filter.{h,cc}
 *   HashFilter<uint16_t>       runtime/hash_filter/hash_filter.h:84-200 (find :109-127, iterator add
 *                              :39-66, full :129, async_clear :132-136), filter.h:56-57 (count_bit 4)
 * Pinned in tests/test_oracle.py against those very sources compiled in place
 * (oracle/_ref/libmonolith_ref_filter.so, oracle/ref_filter_driver.cc) and against the KATs of
 * sliding_hash_filter_test.cc / hash_filter_test.cc.  The slot hash is the engine's documented one
 * (absl::Hash is seeded per process in the reference): fmix64(fid ^ 0x5bd1e995).
 *
 * One mode has no reference counterpart and says so: `defer_advance` moves the window only when
 * mo_filter_advance_if_full() is called — the device checks between launches (all adds of one launch
 * see one window); with one id per launch the two coincide.
 *
 * Only tests/ may link or load this file (through liboracle.so). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { kCountBit = 4, kMaxCount = 15, kHashFilterMaxStep = 64, kSlidingMaxStep = 16, kMaxForward = 2 };

typedef struct mo_split {
  uint16_t* map;            /* total_size + MAX_STEP words (hash_filter.h:93) */
  uint64_t total_size, capacity, num_elements, failure_count;
} mo_split;

typedef struct mo_filter {
  uint64_t capacity;
  int split_num_arg;        /* split_num_: the constructor argument, unclamped (:30) */
  int nsplit;               /* filters_.size() */
  int max_backward;
  mo_split* s;
  uint32_t head;
  int head_increment;
  uint64_t failure_count;
  int defer_advance;
} mo_filter;

static uint64_t filter_hash(uint64_t fid) {   /* ref_shim/absl/hash/hash.h; csrc filter_home */
  uint64_t h = fid ^ 0x5bd1e995ULL;
  h ^= h >> 33;
  h *= 0xff51afd7ed558ccdULL;
  h ^= h >> 33;
  h *= 0xc4ceb9fe1a85ec53ULL;
  h ^= h >> 33;
  return h;
}
/* HashFilter<uint16_t>::signature (:151): sign_mask = 0xffff >> 4 */
static uint16_t filter_signature(uint64_t fid) { return (uint16_t)(((fid >> 17) | (fid << 15)) & 0x0fffu); }

mo_filter* mo_filter_new(uint64_t capacity, int split_num) {
  mo_filter* f = (mo_filter*)calloc(1, sizeof(mo_filter));
  f->split_num_arg = split_num;
  if (capacity < 300) capacity = 300;        /* sliding_hash_filter.cc:31-33 */
  if (split_num < 5) split_num = 5;
  f->capacity = capacity;
  f->nsplit = split_num;
  f->max_backward = split_num - kMaxForward;
  const uint64_t split_capacity = capacity / (uint64_t)(split_num - kMaxForward + 1);   /* get_split_capacity */
  f->s = (mo_split*)calloc((size_t)split_num, sizeof(mo_split));
  for (int i = 0; i < split_num; ++i) {
    mo_split* sp = &f->s[i];
    sp->capacity = split_capacity;
    sp->total_size = (uint64_t)((double)split_capacity * 1.2);    /* HashFilter(capacity, 1.2) :88 */
    sp->map = (uint16_t*)calloc((size_t)(sp->total_size + kHashFilterMaxStep), sizeof(uint16_t));
  }
  return f;
}
void mo_filter_free(mo_filter* f) {
  if (!f) return;
  for (int i = 0; i < f->nsplit; ++i) free(f->s[i].map);
  free(f->s);
  free(f);
}
void mo_filter_set_defer_advance(mo_filter* f, int on) { f->defer_advance = on; }

/* HashFilter::find(fid, max_step) :109-127 -> word index or -1 */
static int64_t split_find(const mo_split* sp, uint64_t fid, int max_step) {
  const uint16_t sign = filter_signature(fid);
  uint64_t at = filter_hash(fid) % sp->total_size;
  const uint64_t n = sp->total_size + kHashFilterMaxStep;
  int step = 0;
  do {
    const uint16_t w = sp->map[at];
    if (w == 0 || (w >> kCountBit) == sign) return (int64_t)at;
    if (++at == n) at = 0;
  } while (++step < max_step);
  return -1;
}
/* HashFilterIterator::add :39-66 on a valid position */
static uint32_t split_add_at(mo_split* sp, int64_t at, uint64_t fid, uint32_t add_count) {
  if (add_count > kMaxCount) add_count = kMaxCount;
  uint16_t* pv = &sp->map[at];
  if (*pv == 0) {
    if (++sp->num_elements > sp->capacity) sp->num_elements = sp->capacity;
    *pv = (uint16_t)((filter_signature(fid) << kCountBit) + add_count);
    return 0;
  }
  const uint32_t count = *pv & kMaxCount;
  if (count + add_count >= kMaxCount) *pv |= kMaxCount;
  else *pv = (uint16_t)(*pv + add_count);
  return count;
}
static uint32_t prev_i(const mo_filter* f, uint32_t i) { return i == 0 ? (uint32_t)f->nsplit - 1 : i - 1; }
static uint32_t next_i(const mo_filter* f, uint32_t i) { return i == (uint32_t)f->nsplit - 1 ? 0 : i + 1; }

/* bidirectional_find :116-128: split index + position of the first usable slot, or -1 */
static int64_t bidir_find(const mo_filter* f, uint32_t begin, int max_look, uint64_t fid, int exhaust,
                          int backward, uint32_t* split_out) {
  uint32_t idx = begin;
  for (int i = 0; i != max_look; ++i) {
    const int64_t at = split_find(&f->s[idx], fid, kSlidingMaxStep);
    if (at >= 0 && (!exhaust || f->s[idx].map[at] != 0)) {
      *split_out = idx;
      return at;
    }
    idx = backward ? prev_i(f, idx) : next_i(f, idx);
  }
  return -1;
}

void mo_filter_advance_if_full(mo_filter* f) {                      /* :85-89 */
  mo_split* h = &f->s[f->head];
  if (h->num_elements >= h->capacity - 1) {                         /* HashFilter::full() */
    f->head = next_i(f, f->head);
    f->head_increment += 1;
    mo_split* c = &f->s[(f->head + kMaxForward - 1) % (uint32_t)f->nsplit];
    memset(c->map, 0, (size_t)(c->total_size + kHashFilterMaxStep) * sizeof(uint16_t));   /* async_clear */
    c->num_elements = 0;
    c->failure_count = 0;
  }
}

uint32_t mo_filter_add(mo_filter* f, uint64_t fid, uint32_t count) {   /* SlidingHashFilter::add :56-91 */
  uint32_t sp = 0;
  const int64_t cur = bidir_find(f, f->head, kMaxForward, fid, 0, 0, &sp);
  if (cur < 0) {
    f->failure_count += 1;
    return kMaxCount;
  }
  if (f->s[sp].map[cur] != 0) return split_add_at(&f->s[sp], cur, fid, count);
  uint32_t old_count = 0, so = 0;
  const int look = f->head_increment < f->max_backward ? f->head_increment : f->max_backward;
  const int64_t old = bidir_find(f, prev_i(f, f->head), look, fid, 1, 1, &so);
  if (old >= 0) {
    old_count = f->s[so].map[old] & kMaxCount;
    split_add_at(&f->s[sp], cur, fid, old_count + count);
  } else {
    split_add_at(&f->s[sp], cur, fid, count);
  }
  if (!f->defer_advance) mo_filter_advance_if_full(f);
  return old_count;
}

uint32_t mo_filter_get(const mo_filter* f, uint64_t fid) {             /* SlidingHashFilter::get :93-114 */
  uint32_t sp = 0;
  const int64_t cur = bidir_find(f, f->head, kMaxForward, fid, 0, 0, &sp);
  if (cur < 0) return kMaxCount;
  if (f->s[sp].map[cur] != 0) return f->s[sp].map[cur] & kMaxCount;
  const int look = f->head_increment < f->max_backward ? f->head_increment : f->max_backward;
  const int64_t old = bidir_find(f, prev_i(f, f->head), look, fid, 1, 1, &sp);
  return old >= 0 ? (uint32_t)(f->s[sp].map[old] & kMaxCount) : 0u;
}

/* ShouldBeFiltered (sliding_hash_filter.h:50-57) */
int mo_filter_should_be_filtered(mo_filter* f, int64_t fid, int64_t count, int64_t threshold) {
  if (threshold <= 0) return 0;
  return (int64_t)mo_filter_add(f, (uint64_t)fid, (uint32_t)count) < threshold;
}

uint64_t mo_filter_estimated_total_element(const mo_filter* f) {
  uint64_t n = 0;
  for (int i = 0; i < f->nsplit; ++i) n += f->s[i].num_elements;
  return n;
}
/* out[0..3] = head, head_increment, failure_count, nsplit; out[4 + i] = num_elements of split i */
void mo_filter_state(const mo_filter* f, uint64_t* out) {
  out[0] = f->head;
  out[1] = (uint64_t)f->head_increment;
  out[2] = f->failure_count;
  out[3] = (uint64_t)f->nsplit;
  for (int i = 0; i < f->nsplit; ++i) out[4 + i] = f->s[i].num_elements;
}
uint64_t mo_filter_split_words(const mo_filter* f, int split, uint32_t* out, uint64_t cap) {
  const mo_split* sp = &f->s[split];
  const uint64_t n = sp->total_size + kHashFilterMaxStep;
  for (uint64_t i = 0; i < n && i < cap; ++i) out[i] = sp->map[i];
  return n;
}
