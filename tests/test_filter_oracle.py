"""CPU suite: the occurrence-filter checker (oracle/mhte_filter_oracle.c, the restatement the GPU
tests compare the device filter with) pinned to
  * the reference's own sources compiled in place — runtime/hash_filter/sliding_hash_filter.cc +
    hash_filter.{h,cc} -> oracle/_ref/libmonolith_ref_filter.so (oracle/ref_filter_driver.cc; absl::Hash
    replaced by the engine's fixed slot hash, everything else the reference's code) — add by add,
    word by word, through windows that go round several times, probe failures, Save / Restore;
  * the KATs of sliding_hash_filter_test.cc:27-41,43-62,101-111 and hash_filter_test.cc (test_simple,
    test_count, compare_to_unordered_map's conflict rate, SkipZeroThresholdFeatures).
"""
import numpy as np
import pytest

import oracle as O

needs_ref = pytest.mark.skipif(not O.ref_filter_available(),
                               reason="oracle/_ref/libmonolith_ref_filter.so not built (no /root/reference here)")


def window_sequence(seed=9, n=1000):
  """the drifting working set of tests/test_boundary_gpu.py's window test"""
  rng = np.random.default_rng(seed)
  universe = (rng.integers(1, 2**40, n // 2 + 400).astype(np.int64) | (1 << 48))
  return [int(universe[k // 2 + int(rng.integers(0, 60))]) for k in range(n)], universe


@needs_ref
@pytest.mark.parametrize("capacity,split_num", [(300, 5), (300, 3), (700, 7), (50, 10)])
def test_restatement_equals_the_compiled_reference_through_moving_windows(capacity, split_num):
  ref, mod = O.RefSlidingFilter(capacity, split_num), O.SlidingFilter(capacity, split_num)
  seq, _ = window_sequence(n=4000)
  rng = np.random.default_rng(1)
  for i, fid in enumerate(seq):
    c = int(rng.integers(1, 5)) if i % 7 else 20          # (counts above max_count are clamped)
    assert ref.add(fid, c) == mod.add(fid, c), i
    if i % 250 == 249:
      probe = np.unique(np.array(seq[:i + 1], dtype=np.int64))
      got = [mod.get(int(x)) for x in probe]
      np.testing.assert_array_equal(ref.get_many(probe.astype(np.uint64)), got)
  st = mod.state()
  assert st["head_increment"] > len(st["num_elements"])              # the window went round
  assert ref.estimated_total_element() == mod.estimated_total_element()
  assert ref.failure_count() == st["failure_count"]
  for sp in range(len(st["num_elements"])):
    meta, words = ref.save_split(sp)
    np.testing.assert_array_equal(words, mod.split_words(sp))
    assert (meta["num_elements"], meta["head"], meta["head_increment"]) == (
        st["num_elements"][sp], st["head"], st["head_increment"])
    assert meta["split_num"] == split_num                              # the argument, unclamped (:30,160)
    assert words.max() < 1 << 16                                      # HashFilter<uint16_t>


@needs_ref
def test_probe_failures_agree():
  """Ids that all hash into one neighbourhood exhaust the 16 probe positions of both look-ahead
  splits: add returns max_count, failure_count moves (:64-67)."""
  cap = 3000
  ref, mod = O.RefSlidingFilter(cap, 5), O.SlidingFilter(cap, 5)
  total = int((cap // 4) * 1.2)
  L = O.lib()
  # ids whose home slot falls in a 4-slot neighbourhood
  # (ids with varied bits 17..28: the 12-bit signature is fid bits 17..28, hash_filter.h:151)
  ids, rng = [], np.random.default_rng(4)
  while len(ids) < 80:
    x = int(rng.integers(1 << 20, 1 << 40))
    if (int(L.mo_hash(x ^ 0x5bd1e995)) % total) < 4:
      ids.append(x)
  res_r = [ref.add(i, 1) for i in ids]
  res_m = [mod.add(i, 1) for i in ids]
  assert res_r == res_m and 15 in res_r
  assert ref.failure_count() == mod.state()["failure_count"] > 0


@needs_ref
def test_restore_into_the_reference():
  """What the restatement holds, handed to the reference's Restore split by split, answers like the
  restatement; the reference's validation refuses another geometry (RestoreMetaDump :189-204)."""
  mod = O.SlidingFilter(300, 5)
  seq, _ = window_sequence(n=700)
  for fid in seq:
    mod.add(fid, 2)
  st = mod.state()
  ref = O.RefSlidingFilter(300, 5)
  for sp in range(5):
    meta = {"failure_count": 0, "total_size": 90, "num_elements": st["num_elements"][sp], "fill_rate_e6": 1200000,
            "split_num": 5, "max_forward_step": 2, "max_backward_step": 3, "max_step": 16, "head": st["head"],
            "head_increment": st["head_increment"], "sliding_failure_count": st["failure_count"]}
    assert ref.restore_split(sp, meta, mod.split_words(sp))
  probe = np.unique(np.array(seq, dtype=np.int64))
  np.testing.assert_array_equal(ref.get_many(probe.astype(np.uint64)), [mod.get(int(x)) for x in probe])
  for fid in seq[-100:]:
    assert ref.add(fid, 1) == mod.add(fid, 1)
  bad = dict(meta, split_num=7)
  assert not O.RefSlidingFilter(300, 5).restore_split(0, bad, mod.split_words(0))


def _both():
  out = [O.SlidingFilter]
  if O.ref_filter_available():
    out.append(O.RefSlidingFilter)
  return out


@pytest.mark.parametrize("cls", _both(), ids=lambda c: c.__name__)
def test_reference_kats(cls):
  # sliding_hash_filter_test.cc:27-41 test_simple
  for key_num in (1, 3, 100):
    f = cls(key_num, 10)
    for i in range(16):
      assert f.add(1, 1) == i
    assert f.add(1, 1) == 15
  # :43-62 test_count
  f = cls(1000000, 10)
  rng = np.random.default_rng(0)
  keys = rng.integers(0, 2**31, 11)
  f.add(int(keys[0]), 2)
  assert f.estimated_total_element() == 1
  for k in keys[1:]:
    f.add(int(k), 2)
  assert f.estimated_total_element() == 11
  f2 = cls(1000000, 10)
  for c in (1, 20, 1):
    f2.add(10000002961562801052, c)
  assert f2.estimated_total_element() == 1 and f2.get(10000002961562801052) == 15
  # :101-111 SkipZeroThresholdFeatures
  f = cls(1000000, 10)
  for i in range(5):
    assert not f.should_be_filtered(i, 1, 0)
    assert f.should_be_filtered(i * 2, 1, 1)
  assert f.estimated_total_element() == 5


@pytest.mark.parametrize("keys,capacity,expected", [(1000000, 1000000, 0.00908), (1000000, 500000, 0.50)])
def test_conflict_rate_kat(keys, capacity, expected):
  """sliding_hash_filter_test.cc:64-99 compare_to_unordered_map: 12-bit signatures over 16 probe
  positions alias — the reference EXPECTS 0.9 % of the counts to be off at this load (± half), 50 %
  when the window holds half the keys.  (A wider signature would fail this KAT from below.)"""
  f = O.SlidingFilter(capacity, 10)
  rng = np.random.default_rng(capacity)
  nums = rng.integers(0, 2**31 - 1, keys)
  counter = {}
  L, h = f._L, f._h  # pylint: disable=protected-access
  import ctypes as C
  for n in nums.tolist():
    c = counter.get(n, 0)
    if c < 14:
      counter[n] = c + 2
      L.mo_filter_add(h, C.c_uint64(n), C.c_uint32(2))
  wrong = sum(1 for k, v in counter.items() if L.mo_filter_get(h, C.c_uint64(k)) != v)
  rate = wrong / len(counter)
  assert abs(rate - expected) <= expected / 2, rate
  assert f.state()["failure_count"] < len(counter) / 10000
