"""GPU parity suite (-m gpu): the HIP engine, driven through the C ABI (libmhte.so) by the
MultiHashTable mirror, against
  * the reference's own known-answer tests (multi_hash_table_ops_test.py, hash_table_ops_test.py,
    embedding_hash_table_test.h, distribution_ops.py docstrings),
  * the committed golden fixtures generated from the reference's code (tests/golden/),
  * the CPU oracle (oracle/) on seeded random op sequences,
  * size-independent properties at BASELINE.json's full batch size.
Bars: bit-exact for ids, sizes, hit counts, status codes, dedup order / offsets, placement;
fp32 rows bit-exact wherever the summation order is the reference's, else |diff| <= 1e-5.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle as O  # noqa: E402
from monolith_amd import _lib, entry, synthetic as S  # noqa: E402
from monolith_amd import distribution_ops as D  # noqa: E402
from monolith_amd.fused_step import SparseStep  # noqa: E402
from monolith_amd.multi_hash_table_ops import MultiHashTable, Ragged  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-5  # north_star: fp32 embedding values within 1e-5
# The bar actually applied to rows whose gradient list was summed as a fixed tree instead of
# sequentially (lists of > 32 occurrences outside MHTE_EXACT_ORDER): |diff| <= 5e-7 + 1e-5 |expected|
# (measured 2.1e-7 on the 12 000-occurrence lists of a 65 536-id Zipf batch; the rows are 1e-4..1e-2
# in magnitude, so a bare atol of 1e-5 would be a 1-10 % relative bar — VERDICT r1).  Everything
# else is compared bit for bit.
RTOL_TREE, ATOL_TREE = 1e-5, 5e-7
# Gradient SUMS themselves (segment sum, sender-side sum) and rows of one id that fills a whole
# batch: up to 65 536 terms of ~1e-2, sum of magnitudes ~1e2: re-association moves them by a few
# 1e-6 (measured <= 2.2e-6); fp32 bound eps * log2(n) * sum|g| ~ 1e-4.
ATOL_SUM = 4e-6

_counter = [0]


def _name():
  _counter[0] += 1
  return "t%d" % _counter[0]


def dev(x, dtype=None):
  t = torch.as_tensor(np.asarray(x))
  if dtype is not None:
    t = t.to(dtype)
  return t.cuda()


def ids_t(x):
  return dev(np.asarray(x, dtype=np.int64))


def val_t(x):
  return dev(np.asarray(x, dtype=np.float32))


def sgd_cfg(dim, lr=1.0, **kw):
  """test_utils.generate_test_hash_table_config(dim): SGD, zeros init, lr 1."""
  return entry.make_table_config(
      [entry.CombineAsSegment(dim, entry.ZerosInitializer(), entry.SgdOptimizer(lr))],
      entry.CuckooHashTableConfig(**kw))


def adagrad_cfg(dim, lr=0.001, init_acc=0.1, wd=0.0, **kw):
  return entry.make_table_config([
      entry.CombineAsSegment(dim, entry.ZerosInitializer(),
                             entry.AdagradOptimizer(lr, init_acc, weight_decay_factor=wd))
  ], entry.CuckooHashTableConfig(**kw))


def make(configs):
  return MultiHashTable.from_configs(configs, name_suffix=_name())


# =============================================================================== reference KATs
def test_lookup_assign_add_reinitialize_kat():
  # multi_hash_table_ops_test.py:52-99
  mt = make({"slot0": sgd_cfg(1), "not_used": sgd_cfg(2), "slot1": sgd_cfg(2), "slot2": sgd_cfg(2)})
  assert mt.table_names == ("not_used", "slot0", "slot1", "slot2")
  mt = mt.assign_add({
      "slot0": (ids_t([0]), val_t([[1]])),
      "slot1": (ids_t([1]), val_t([[2, 2]])),
      "slot2": (ids_t([2, 3]), val_t([[4, 4], [8, 8]])),
  })
  v = mt.lookup({"slot0": ids_t([0]), "slot1": ids_t([1]), "slot2": ids_t([2, 3])})
  assert v["slot0"].cpu().tolist() == [[1]]
  assert v["slot1"].cpu().tolist() == [[2, 2]]
  assert v["slot2"].cpu().tolist() == [[4, 4], [8, 8]]
  mt, st1 = mt.reinitialize("slot2", ids_t([1, 2, 3]), now=123)
  mt, st2 = mt.reinitialize("slot3", ids_t([1, 2, 3]), now=123)
  v = mt.lookup({"slot0": ids_t([0]), "slot1": ids_t([1]), "slot2": ids_t([1, 2, 3])})
  assert v["slot0"].cpu().tolist() == [[1]]
  assert v["slot1"].cpu().tolist() == [[2, 2]]
  assert v["slot2"].cpu().tolist() == [[0, 0], [0, 0], [0, 0]]
  assert st1.cpu().tolist() == [0, 1, 1]
  assert st2.cpu().tolist() == [-1, -1, -1]
  assert mt.size("slot2") == 3 and mt.size("not_used") == 0


def test_apply_gradients_kat():
  # multi_hash_table_ops_test.py:101-127
  mt = make({"slot0": sgd_cfg(1), "slot1": sgd_cfg(2)})
  mt = mt.apply_gradients({
      "slot0": (ids_t([0]), val_t([[2.0]])),
      "slot1": (ids_t([1, 2]), val_t([[1.0, 3.0], [2.0, 4.0]])),
  }, global_step=0)
  v = mt.lookup({"slot0": ids_t([0]), "slot1": ids_t([1, 2])})
  assert v["slot0"].cpu().tolist() == [[-2]]
  assert v["slot1"].cpu().tolist() == [[-1, -3], [-2, -4]]


def test_lookup_miss_zero_no_insert_and_negative_ids():
  # embedding_hash_table_test.h:41-58
  mt = make({"a": sgd_cfg(3)})
  v = mt.lookup({"a": ids_t([5, -10, 7])})["a"]
  assert not v.any().item() and mt.size("a") == 0
  mt.assign_add({"a": (ids_t([-10]), val_t([[2.5, 2.5, 2.5]]))})
  assert mt.lookup({"a": ids_t([-10])})["a"].cpu().tolist() == [[2.5, 2.5, 2.5]]
  assert mt.size("a") == 1


def test_duplicate_ids_apply_sequentially_kat():
  # hash_table_ops_test.py:134-148: ids [0,0,1], grad -1, lr .1 -> [[.2],[.1]]
  mt = make({"a": sgd_cfg(1, lr=0.1)})
  mt.apply_gradients({"a": (ids_t([0, 0, 1]), val_t([[-1.0], [-1.0], [-1.0]]))})
  got = mt.lookup({"a": ids_t([0, 1])})["a"].cpu().numpy()
  exp = O.Table(O.segment(1, O.OPT_SGD), 1)
  exp.optimize([0, 0, 1], [[-1.0], [-1.0], [-1.0]], [0.1])
  np.testing.assert_array_equal(got, exp.lookup([0, 1])[0])
  np.testing.assert_allclose(got, [[0.2], [0.1]], atol=1e-7)


def test_duplicate_ids_adagrad_sequential_vs_accumulated():
  # cuckoo_embedding_hash_table.cc:229-236 (one step per occurrence) vs tf_bridge.cc:270-310
  ids = [7, 7, 9, 7]
  g = np.float32([[1, 2], [3, 4], [5, 6], [7, 8]])
  seq = O.Table(O.segment(2, O.OPT_ADAGRAD, p=(0.1, 0.0)), 1)
  seq.optimize(ids, g, [0.5], 10)
  acc = O.Table(O.segment(2, O.OPT_ADAGRAD, p=(0.1, 0.0)), 1)
  acc.optimize([7, 9], [g[0] + g[1] + g[3], g[2]], [0.5], 10)
  for flags, exp in ((0, seq), (_lib.MHTE_SUM_DUPLICATES, acc)):
    mt = make({"a": adagrad_cfg(2, lr=0.5)})
    r = mt.get_ragged_id({"a": ids_t(ids)})
    C = _lib.C
    _lib.check(mt._lib.mhte_optimize(mt.handle, _lib.vp(r.values),
                                     r.row_splits.ctypes.data_as(C.POINTER(C.c_int64)),
                                     C.c_int64(2), _lib.vp(val_t(g)), C.c_int64(g.size),
                                     mt.learning_rate.ctypes.data_as(C.POINTER(C.c_float)),
                                     C.c_int64(1), C.c_int64(10), C.c_int64(0), C.c_int32(flags),
                                     None))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(mt.lookup({"a": ids_t([7, 9])})["a"].cpu().numpy(),
                                  exp.lookup([7, 9])[0])


def test_fused_lookup_kat():
  # hash_table_ops_test.py:1086-1107 (three tables dims 1,1,2)
  mt = make({"t0": sgd_cfg(1), "t1": sgd_cfg(1), "t2": sgd_cfg(2)})
  mt.assign({"t0": (ids_t([0, 1]), torch.ones(2, 1).cuda()),
             "t1": (ids_t([3, 4]), torch.zeros(2, 1).cuda()),
             "t2": (ids_t([6, 7]), torch.ones(2, 2).cuda())})
  emb, splits, id_off, emb_off, ids = mt.fused_lookup(ids_t([0, 4, 6, 1, 3, 7]),
                                                      [1, 1, 1, 1, 1, 1], num_of_shards=2)
  assert emb.cpu().tolist() == [1, 0, 1, 1, 1, 0, 1, 1]
  assert splits.tolist() == [4, 4]
  assert id_off.tolist() == [0, 1, 2, 3, 4, 5, 6]
  assert emb_off.tolist() == [0, 1, 2, 4, 5, 6, 8]
  assert ids.cpu().tolist() == [0, 4, 6, 1, 3, 7]


def test_fused_optimize_kat():
  # hash_table_ops_test.py:1109-1150
  mt = make({"t0": sgd_cfg(1, lr=0.1), "t1": sgd_cfg(2, lr=0.1)})
  mt.assign({"t0": (ids_t([0, 1]), torch.ones(2, 1).cuda()),
             "t1": (ids_t([3, 4]), torch.zeros(2, 2).cuda())})
  ids = ids_t([0, 4, 1, 3])
  emb, splits, id_off, emb_off, idx = mt.fused_lookup(ids, [1, 1, 1, 1], num_of_shards=2)
  assert emb.cpu().tolist() == [1, 0, 0, 1, 0, 0]
  mt.fused_apply_gradient(ids, idx, [1, 1, 1, 1], val_t([-1, -2, -2, -1, -2, -2]), id_off, emb_off,
                          global_step=0, req_time=0, num_of_shards=2)
  emb, splits, id_off, emb_off, _ = mt.fused_lookup(ids, [1, 1, 1, 1], num_of_shards=2)
  np.testing.assert_allclose(emb.cpu().numpy(), [1.1, 0.2, 0.2, 1.1, 0.2, 0.2], atol=1e-6)
  assert splits.tolist() == [3, 3]
  assert id_off.tolist() == [0, 1, 2, 3, 4]
  assert emb_off.tolist() == [0, 1, 3, 4, 6]


def test_error_convention():
  # multi_hash_table_update_op.cc:34-45,57-58,71-77 -> InvalidArgument
  mt = make({"a": sgd_cfg(2), "b": sgd_cfg(2)})
  bad = Ragged(ids_t([1, 2]), np.array([0, 2], dtype=np.int64))  # one split too few
  with pytest.raises(_lib.InvalidArgumentError):
    mt.raw_apply_gradients(bad, val_t([[1, 1], [1, 1]]))
  ok = mt.get_ragged_id({"a": ids_t([1, 2])})
  with pytest.raises(_lib.InvalidArgumentError):  # value too short
    mt.raw_apply_gradients(ok, val_t([1.0]))
  with pytest.raises(_lib.InvalidArgumentError):
    mt.raw_lookup(bad)


def test_evict_ttl_kat():
  # embedding_hash_table_test.h:282-326
  se = entry.SlotExpireTimeConfig(default_expire_time=14, slot_expire_times={1: 5, 2: 6})
  cfg = entry.make_table_config(
      [entry.CombineAsSegment(1, entry.ZerosInitializer(), entry.SgdOptimizer(1.0))],
      slot_expire_time_config=se)
  mt = make({"a": cfg})
  day = 86400
  f1, f2, f3 = (1 << 48) | 123, (2 << 48) | 123, (3 << 48) | 123
  mt.assign({"a": (ids_t([f1, f2, f3]), val_t([[1], [2], [3]]))}, req_time=1000)
  mt.evict("a", 1000 + 5 * day + 60)
  assert mt.contains("a", ids_t([f1, f2, f3])).cpu().tolist() == [False, True, True]
  assert mt.size("a") == 2
  assert mt.lookup({"a": ids_t([f1, f2])})["a"].cpu().tolist() == [[0], [2]]


def test_special_key_int64_min():
  mt = make({"a": adagrad_cfg(4, lr=0.1)})
  k = -(1 << 63)
  g = np.float32([[1, 2, 3, 4]])
  exp = O.Table(O.segment(4, O.OPT_ADAGRAD, p=(0.1, 0.0)), 1)
  for _ in range(2):
    mt.apply_gradients({"a": (ids_t([k, 5]), val_t(np.repeat(g, 2, 0)))}, req_time=3)
    # the oracle's map has no reserved key
    exp.optimize([k, 5], np.repeat(g, 2, 0), [0.1], 3)
  assert mt.size("a") == 2
  np.testing.assert_array_equal(mt.lookup({"a": ids_t([5, k, 6])})["a"].cpu().numpy(),
                                exp.lookup([5, k, 6])[0])


# =============================================================================== dedup ops
def test_unique_key_with_value_and_offset_docstring_kat():
  # reference distribution_ops.py:101-110
  key = Ragged(ids_t([0, 1, 0, 0]), np.array([0, 3, 4], dtype=np.int64))
  r = D.unique_key_with_value_and_offset(key, [2, 3])
  assert r.unique_key.values.cpu().tolist() == [0, 1, 0]
  assert r.unique_key.row_splits.tolist() == [0, 2, 3]
  assert r.value_offset.cpu().tolist() == [0, 4, 2, 6]
  assert r.value_offset_split.cpu().tolist() == [0, 2, 3, 4]
  assert r.value_buffer.numel() == 9
  # reference distribution_ops.py:133-140
  pos = Ragged(ids_t([0, 1, 2]), np.array([0, 2, 3], dtype=np.int64))
  buf = D.fill_with_offset_map(pos, val_t(np.arange(7)), r.value_offset, r.value_offset_split,
                               r.value_buffer, [2, 3])
  assert buf.cpu().tolist() == [0, 1, 2, 3, 0, 1, 4, 5, 6]
  g = val_t([1, 2, 10, 20, 100, 200, 7, 8, 9])
  bg = D.fill_with_offset_map_gradient(pos, g, r.value_offset, r.value_offset_split, [2, 3])
  assert bg.cpu().tolist() == [101, 202, 10, 20, 7, 8, 9]


@pytest.mark.parametrize("n,universe,dist", [(1, 10, "uniform"), (257, 40, "uniform"),
                                              (5000, 300, "uniform"), (65536, 10**9, "zipf"),
                                              (70001, 3, "uniform")])
def test_unique_matches_oracle_bit_exact(n, universe, dist):
  ids = S.id_batch(3, n, universe, dist)
  if n > 1000:
    ids[17] = -(1 << 63)  # the reserved key must dedup like any other
    ids[n // 2] = -(1 << 63)
  ws = D.DedupWorkspace()
  r = ws.unique(ids_t(ids))
  uk, uks, vo, vos, _ = O.unique_key_with_value_and_offset(ids, [0, n], [1])
  U = r.n_unique
  assert U == uk.size
  np.testing.assert_array_equal(r.unique_ids[:U].cpu().numpy(), uk)
  np.testing.assert_array_equal(r.seg_off[:U + 1].cpu().numpy().astype(np.int64), vos)
  np.testing.assert_array_equal(r.seg_pos.cpu().numpy().astype(np.int64), vo)
  inv = r.inverse.cpu().numpy()
  np.testing.assert_array_equal(uk[inv], ids)


@pytest.mark.parametrize("n,universe,dist", [(1, 10, "uniform"), (257, 40, "uniform"),
                                             (5000, 300, "zipf"), (65536, 10**9, "zipf"),
                                             (300000, 10**5, "zipf")])
def test_unique_unordered_same_sets_and_lists(n, universe, dist):
  """mhte_unique_unordered: unspecified numbering, but the same key set, the same occurrence
  list per key, lists > 32 positions in ascending order (including n > one bitmap chunk)."""
  ids = S.id_batch(3, n, universe, dist)
  if n > 100:
    ids[5] = np.iinfo(np.int64).min  # the reserved key is a legal id
    ids[77] = np.iinfo(np.int64).min
  ws = D.DedupWorkspace()
  for _ in range(2):  # second call runs on the cleaned-after-use scratch
    r = ws.unique_unordered(ids_t(ids), want_host_count=True)
    U = r.n_unique
    uk = np.unique(ids)
    assert U == uk.size
    uids = r.unique_ids[:U].cpu().numpy()
    np.testing.assert_array_equal(np.sort(uids), uk)
    inv = r.inverse.cpu().numpy()
    np.testing.assert_array_equal(uids[inv], ids)
    st, en = r.seg_off[:U].cpu().numpy().astype(np.int64), r.list_end[:U].cpu().numpy().astype(np.int64)
    pos = r.seg_pos.cpu().numpy()
    order = np.argsort(st)
    assert st[order][0] == 0 and en[order][-1] == n
    np.testing.assert_array_equal(st[order][1:], en[order][:-1])  # the lists tile [0, n)
    for u in list(range(min(U, 50))) + list(np.argsort(st - en)[:20]):
      lst = pos[st[u]:en[u]]
      np.testing.assert_array_equal(np.sort(lst), np.flatnonzero(ids == uids[u]))
      if lst.size > 32:
        assert np.all(np.diff(lst.astype(np.int64)) > 0)


def test_unique_empty():
  ws = D.DedupWorkspace()
  r = ws.unique(torch.empty(0, dtype=torch.int64, device="cuda"))
  assert r.n_unique == 0


@pytest.mark.parametrize("dim", [1, 8, 32, 64, 100])
def test_segment_sum_exact_and_windowed(dim):
  n = 20000
  ids = S.id_batch(5, n, 10**6, "zipf")
  g = S.grad_batch(5, n, dim)
  ws = D.DedupWorkspace()
  r = ws.unique(ids_t(ids))
  U = r.n_unique
  uk, uks, vo, vos, _ = O.unique_key_with_value_and_offset(ids, [0, n], [dim])
  exp = O.fill_with_offset_map_gradient(np.arange(U), [0, U], g.ravel(), vo, vos,
                                        [dim]).reshape(U, dim)
  got_exact = ws.segment_sum(val_t(g), r, dim, exact_order=True)[:U].cpu().numpy()
  np.testing.assert_array_equal(got_exact, exp)  # same order as the reference -> bit exact
  got_fast = ws.segment_sum(val_t(g), r, dim, exact_order=False)[:U].cpu().numpy()
  np.testing.assert_allclose(got_fast, exp, rtol=RTOL_TREE, atol=ATOL_SUM)
  # deterministic run to run
  again = ws.segment_sum(val_t(g), r, dim, exact_order=False)[:U].cpu().numpy()
  np.testing.assert_array_equal(got_fast, again)
  # forward scatter
  src = val_t(np.random.default_rng(0).standard_normal((U, dim)))
  out = ws.gather_rows(src, r.inverse, n, dim).cpu().numpy()
  np.testing.assert_array_equal(out, src.cpu().numpy()[r.inverse.cpu().numpy()])


# =============================================================================== golden fixtures
@pytest.mark.parametrize("name", ["sgd_d8_uniform", "adagrad_d16_zipf", "adagrad_d64_zipf"])
@pytest.mark.parametrize("exact", [True, False])
def test_training_loop_matches_reference_fixture(name, exact):
  z = np.load(os.path.join(GOLD, "table_%s.npz" % name))
  dim, opt = int(z["dim"]), int(z["opt"])
  lr = float(z["lr"])
  cfg = sgd_cfg(dim, lr) if opt == O.OPT_SGD else adagrad_cfg(dim, lr, float(z["init_acc"]),
                                                               float(z["wd"]))
  mt = make({"emb": cfg})
  batch = int(z["batch"])
  step = SparseStep(mt, "emb", batch, exact_order=exact)
  for s in range(int(z["steps"])):
    ids = S.id_batch(s, batch, int(z["universe"]), str(z["dist"]))
    g = S.grad_batch(s, batch, dim)
    emb = step.forward(ids_t(ids))
    assert step.n_unique() == int(z["n_unique"][s])
    e = emb.cpu().numpy()
    if exact:
      np.testing.assert_array_equal(e[:64], z["step_emb_first"][s])
    else:
      np.testing.assert_allclose(e[:64], z["step_emb_first"][s], rtol=RTOL_TREE, atol=ATOL_TREE)
    step.backward(val_t(g), S.update_time(s))
  assert mt.size("emb") == int(z["size"])
  final = mt.lookup({"emb": ids_t(z["probe_ids"])})["emb"].cpu().numpy()
  if exact:
    np.testing.assert_array_equal(final, z["final_rows"])
  else:
    np.testing.assert_allclose(final, z["final_rows"], rtol=RTOL_TREE, atol=ATOL_TREE)
  st = mt.stats("emb")
  assert st.dropped == 0 and st.rows_allocated == int(z["size"])


def test_sequential_placement_matches_reference_fixture():
  """id -> (bucket, slot) is bit-identical to the reference map (same fixed hash) when ids arrive
  one per op into a pre-sized table, including keys placed by BFS displacement."""
  z = np.load(os.path.join(GOLD, "placement_seq.npz"))
  mt = make({"a": sgd_cfg(4, initial_capacity=int(z["cap"]), max_load_factor=1.0)})
  ids_d, vals_d = ids_t(z["ids"]), val_t(z["vals"])
  for i in range(z["ids"].size):
    mt.assign({"a": (ids_d[i:i + 1], vals_d[i:i + 1])}, req_time=100 + i)
  ids, pos, ts, rows = mt.dump("a")
  assert mt.stats("a").hashpower == 10
  np.testing.assert_array_equal(ids.cpu().numpy(), z["dump_ids"])
  np.testing.assert_array_equal(pos.cpu().numpy(), z["dump_pos"])
  np.testing.assert_array_equal(ts.cpu().numpy().astype(np.uint32), z["dump_ts"])
  np.testing.assert_array_equal(rows.cpu().numpy(), z["dump_rows"])


# =============================================================================== oracle, random ops
def _check_placement_valid(mt, name, hp):
  ids, pos, _, _ = mt.dump(name, with_rows=False)
  ids, pos = ids.cpu().numpy(), pos.cpu().numpy()
  L = O.lib()
  for k, p in zip(ids[:20000], pos[:20000]):
    hv = L.mo_hash(int(k))
    i1 = hv & ((1 << hp) - 1)
    i2 = L.mo_alt_index(hp, L.mo_partial(hv), i1)
    assert (p >> 2) in (i1, i2), (k, p)
  assert len(np.unique(ids)) == ids.size


@pytest.mark.parametrize("kind", ["sgd8", "adagrad32", "adagrad_wd5", "ftrl3", "multiseg"])
def test_random_op_sequence_matches_oracle(kind):
  rng = np.random.default_rng({"sgd8": 1, "adagrad32": 2, "adagrad_wd5": 3, "ftrl3": 4, "multiseg": 5}[kind])
  if kind == "sgd8":
    segs_o = [O.segment(8, O.OPT_SGD)]
    segs_e = [entry.CombineAsSegment(8, entry.ZerosInitializer(), entry.SgdOptimizer(0.05))]
    lrs = [0.05]
  elif kind == "adagrad32":
    segs_o = [O.segment(32, O.OPT_ADAGRAD, p=(0.1, 0.0), init=O.INIT_CONSTANT, init_value=0.25)]
    segs_e = [entry.CombineAsSegment(32, entry.ConstantsInitializer(0.25),
                                     entry.AdagradOptimizer(0.02, 0.1))]
    lrs = [0.02]
  elif kind == "adagrad_wd5":
    segs_o = [O.segment(5, O.OPT_ADAGRAD, p=(0.2, 0.01), init=O.INIT_ONES)]
    segs_e = [entry.CombineAsSegment(5, entry.OnesInitializer(),
                                     entry.AdagradOptimizer(0.02, 0.2, weight_decay_factor=0.01))]
    lrs = [0.02]
  elif kind == "ftrl3":
    segs_o = [O.segment(3, O.OPT_FTRL, p=(0.1, 1.0, 0.01, 0.02))]
    segs_e = [entry.CombineAsSegment(3, entry.ZerosInitializer(),
                                     entry.FtrlOptimizer(0.03, 0.1, 1.0, 0, 0.01, 0.02))]
    lrs = [0.03]
  else:  # bias FTRL + vector Adagrad, distributed_ps_test.py:480-505
    segs_o = [O.segment(1, O.OPT_FTRL, p=(0.1, 1.0, 0.0, 0.0)),
              O.segment(16, O.OPT_ADAGRAD, p=(0.1, 0.0)), O.segment(4, O.OPT_SGD)]
    segs_e = [entry.CombineAsSegment(1, entry.ZerosInitializer(), entry.FtrlOptimizer(0.03, 0.1, 1.0)),
              entry.CombineAsSegment(16, entry.ZerosInitializer(), entry.AdagradOptimizer(0.02, 0.1)),
              entry.CombineAsSegment(4, entry.ZerosInitializer(), entry.SgdOptimizer(0.5))]
    lrs = [0.03, 0.02, 0.5]
  dim = sum(s.dim for s in segs_o)
  ot = O.Table(segs_o, 1)
  mt = make({"a": entry.make_table_config(segs_e, learning_rates=lrs)})
  mt.set_count_hits("a")
  universe = rng.integers(-2**62, 2**62, 40000)
  for step in range(14):
    n = int(rng.integers(1, 9000))
    ids = rng.choice(universe, n)  # with duplicates
    v = (rng.standard_normal((n, dim)) * 0.5).astype(np.float32)
    k = step % 5
    if k == 0:
      ot.assign(ids, v, 100 + step)
      mt.assign({"a": (ids_t(ids), val_t(v))}, req_time=100 + step)
    elif k == 1:
      ot.assign_add(ids, v, 100 + step)
      mt.assign_add({"a": (ids_t(ids), val_t(v))}, req_time=100 + step)
    elif k in (2, 3):
      ot.optimize(ids, v, lrs, 100 + step)
      mt.apply_gradients({"a": (ids_t(ids), val_t(v))}, req_time=100 + step)
    else:
      exp_st = ot.reinitialize(ids[:700], 55)
      _, st = mt.reinitialize("a", ids_t(ids[:700]), now=55)
      np.testing.assert_array_equal(st.cpu().numpy(), exp_st)
    assert mt.size("a") == ot.size(), step
    probe = rng.choice(universe, 3000)
    e, hits = ot.lookup(probe)
    got = mt.lookup({"a": ids_t(probe)})["a"].cpu().numpy()
    np.testing.assert_array_equal(got, e, err_msg="step %d" % step)
    assert mt.stats("a").lookup_hits == hits
  # full state incl. optimizer ctx and timestamps, compared as a key-sorted dump
  ids, pos, ts, rows = mt.dump("a")
  o_ids, o_pos, o_ts, o_rows = ot.dump()
  a = np.argsort(ids.cpu().numpy())
  b = np.argsort(o_ids)
  np.testing.assert_array_equal(ids.cpu().numpy()[a], o_ids[b])
  np.testing.assert_array_equal(ts.cpu().numpy().astype(np.uint32)[a], o_ts[b])
  np.testing.assert_array_equal(rows.cpu().numpy()[a], o_rows[b])
  _check_placement_valid(mt, "a", mt.stats("a").hashpower)


def test_growth_from_capacity_one_and_bulk_unique_insert():
  n = 300000
  rng = np.random.default_rng(1)
  ids = rng.permutation(np.arange(1, 4 * n, 4, dtype=np.int64) * 7919)[:n]
  v = rng.standard_normal((n, 8)).astype(np.float32)
  mt = make({"a": sgd_cfg(8)})
  half = n // 2
  mt.assign({"a": (ids_t(ids[:half]), val_t(v[:half]))})
  mt.assign({"a": (ids_t(ids[half:]), val_t(v[half:]))})  # forces doublings of a non-empty table
  assert mt.size("a") == n
  st = mt.stats("a")
  assert st.dropped == 0 and st.size <= 0.5 * (4 << st.hashpower)
  got = mt.lookup({"a": ids_t(ids)})["a"].cpu().numpy()
  np.testing.assert_array_equal(got, v)
  _check_placement_valid(mt, "a", st.hashpower)


def test_slow_path_displacement_at_high_load():
  # max_load_factor 0.95 in a pre-sized table forces full buckets -> deferred ids -> BFS kernel
  cap = 1 << 14
  n = int(cap * 0.90)
  rng = np.random.default_rng(2)
  ids = rng.integers(1, 2**60, n)
  v = rng.standard_normal((n, 4)).astype(np.float32)
  mt = make({"a": sgd_cfg(4, initial_capacity=cap, max_load_factor=0.95)})
  for lo in range(0, n, 2048):
    mt.assign({"a": (ids_t(ids[lo:lo + 2048]), val_t(v[lo:lo + 2048]))})
  st = mt.stats("a")
  assert st.hashpower == 12 and st.dropped == 0
  assert mt.size("a") == len(np.unique(ids))
  got = mt.lookup({"a": ids_t(ids)})["a"].cpu().numpy()
  # later duplicates of an id win (sequential assign), so compare via the oracle
  ot = O.Table(O.segment(4, O.OPT_SGD), cap)
  ot.assign(ids, v)
  np.testing.assert_array_equal(got, ot.lookup(ids)[0])
  _check_placement_valid(mt, "a", 12)


# =============================================================================== fused backward
def _oracle_step(ot, ids, g, dim, lr, t):
  n = ids.size
  uk, _, vo, vos, _ = O.unique_key_with_value_and_offset(ids, [0, n], [dim])
  gu = O.fill_with_offset_map_gradient(np.arange(uk.size), [0, uk.size], g.ravel(), vo, vos,
                                       [dim]).reshape(-1, dim)
  ot.optimize(uk, gu, [lr], t)
  return uk


@pytest.mark.parametrize("dim,opt", [(4, "sgd"), (8, "adagrad"), (13, "adagrad"), (32, "sgd"),
                                     (64, "adagrad"), (100, "adagrad"), (200, "sgd"),
                                     (256, "adagrad"), (300, "adagrad")])
@pytest.mark.parametrize("exact", [True, False])
def test_fused_backward_matches_oracle(dim, opt, exact):
  """mhte_table_sum_optimize_n (one launch: duplicate-gradient sum + upsert + optimizer) against
  FillWithOffsetMapGradient + Optimize of the oracle, for every lanes-per-id shape (dim 4..256),
  the wide-row fallback (dim 300), lists that cross window-block boundaries (Zipf head keys) and
  ragged batch sizes."""
  n = 30001
  lr = 0.05
  cfg = sgd_cfg(dim, lr) if opt == "sgd" else adagrad_cfg(dim, lr, 0.1)
  mt = make({"emb": cfg})
  step = SparseStep(mt, "emb", n, exact_order=exact)
  ot = O.Table(O.segment(dim, O.OPT_SGD if opt == "sgd" else O.OPT_ADAGRAD, p=(0.1, 0.0)), 1)
  seen = set()
  for s_ in range(3):
    ids = S.id_batch(100 + s_, n, 5 * 10**4, "zipf")
    if s_ == 1:
      ids[::3] = ids[0]        # one list with n/3 occurrences, interleaved with everything else
    g = S.grad_batch(s_, n, dim)
    step.forward(ids_t(ids))
    step.backward(val_t(g), S.update_time(s_))
    _oracle_step(ot, ids, g, dim, lr, S.update_time(s_))
    seen.update(ids.tolist())
  allids = np.fromiter(seen, dtype=np.int64)
  got = mt.lookup({"emb": ids_t(allids)})["emb"].cpu().numpy()
  exp = ot.lookup(allids)[0]
  if exact:
    np.testing.assert_array_equal(got, exp)
  else:
    np.testing.assert_allclose(got, exp, rtol=RTOL_TREE, atol=ATOL_TREE)
  assert mt.size("emb") == len(seen) == ot.size()


def test_fused_backward_equals_unfused_and_is_deterministic():
  n, dim = 65536, 64
  rows = []
  for fused in (True, False, True):
    mt = make({"emb": adagrad_cfg(dim, 0.01, 0.1)})
    step = SparseStep(mt, "emb", n, fused_backward=fused)
    for s_ in range(2):
      ids = S.id_batch(7 + s_, n, 10**9, "zipf")
      step.forward(ids_t(ids))
      step.backward(val_t(S.grad_batch(s_, n, dim)), S.update_time(s_))
    probe = np.unique(np.concatenate([S.id_batch(7 + s_, n, 10**9, "zipf") for s_ in range(2)]))
    rows.append(mt.lookup({"emb": ids_t(probe)})["emb"].cpu().numpy())
  np.testing.assert_array_equal(rows[0], rows[2])            # run-to-run identical
  np.testing.assert_allclose(rows[0], rows[1], rtol=RTOL_TREE, atol=ATOL_TREE)


def test_pipelined_step_matches_oracle_and_unpipelined():
  """forward(ids, next_ids=...): the dedup of the next batch rides in the two launches of the
  current one (mhte_table_step_forward / _backward; run dedup + work items,
  csrc/mhte_step_kernels.h).  Within 1e-5 of the oracle and of the unpipelined path (heavy lists
  are summed as a different — fixed — tree), run-to-run identical, and with exact order bit-exact
  with the oracle."""
  n, dim, steps = 20000, 32, 5
  batches = [S.id_batch(40 + s_, n, 10**5, "zipf") for s_ in range(steps + 1)]
  batches[2][::2] = batches[2][1]  # a list of n/2 occurrences
  ids_dev = [ids_t(b) for b in batches]
  probe = np.unique(np.concatenate(batches[:steps]))
  out = {}
  for mode in ("pipelined", "plain", "pipelined_exact", "pipelined_again"):
    mt = make({"emb": adagrad_cfg(dim, 0.01, 0.1)})
    step = SparseStep(mt, "emb", n, exact_order=(mode == "pipelined_exact"))
    for s_ in range(steps):
      nxt = ids_dev[s_ + 1] if mode != "plain" else None
      emb = step.forward(ids_dev[s_], next_ids=nxt)
      if s_ == 3:
        first = emb.cpu().numpy().copy()
      step.backward(val_t(S.grad_batch(s_, n, dim)), S.update_time(s_))
      assert step.n_unique() == np.unique(batches[s_]).size
    out[mode] = (mt.lookup({"emb": ids_t(probe)})["emb"].cpu().numpy(), first, mt.size("emb"))
  ot = O.Table(O.segment(dim, O.OPT_ADAGRAD, p=(0.1, 0.0)), 1)
  for s_ in range(steps):
    if s_ == 3:
      exp_first = ot.lookup(batches[3])[0]
    _oracle_step(ot, batches[s_], S.grad_batch(s_, n, dim), dim, 0.01, S.update_time(s_))
  exp = ot.lookup(probe)[0]
  np.testing.assert_array_equal(out["pipelined"][0], out["pipelined_again"][0])
  np.testing.assert_array_equal(out["pipelined"][1], out["pipelined_again"][1])
  # (rows reserved by the build role's probe are consumed or returned: the key count is exact)
  assert out["pipelined"][2] == probe.size and out["plain"][2] == probe.size
  np.testing.assert_allclose(out["pipelined"][0], out["plain"][0], rtol=RTOL_TREE, atol=ATOL_TREE)
  np.testing.assert_allclose(out["pipelined"][1], out["plain"][1], rtol=RTOL_TREE, atol=ATOL_TREE)
  np.testing.assert_allclose(out["pipelined"][0], exp, rtol=RTOL_TREE, atol=ATOL_TREE)
  np.testing.assert_array_equal(out["pipelined_exact"][0], exp)
  np.testing.assert_array_equal(out["pipelined_exact"][1], exp_first)
  assert out["pipelined"][2] == out["plain"][2] == ot.size() == probe.size


@pytest.mark.parametrize("n,kind", [(1, "one"), (33, "same"), (1000, "uniform"), (1025, "same"),
                                    (5000, "pairs"), (20000, "same"), (65536, "zipf"),
                                    (65536, "same"), (4097, "special")])
@pytest.mark.parametrize("exact", [False, True])
def test_pipelined_step_run_dedup_edge_shapes(n, kind, exact):
  """Run dedup + work items on the shapes that stress them: a single id, lists exactly at the
  light/heavy boundary, one id filling whole dedup workgroups (runs of 1024, 64 work items), batch
  sizes that are not multiples of 1024, the maximum batch, and kEmptyKey (INT64_MIN) as an id.
  Every list light enough to be summed sequentially must be bit-exact; the rest within 1e-5."""
  dim, steps = 16, 3
  rng = np.random.default_rng(n * 7 + len(kind))
  def batch(s_):
    if kind == "one":
      return np.array([77 + s_], dtype=np.int64)
    if kind == "same":
      b = np.full(n, 4242, dtype=np.int64)
      b[-1] = 4243 + s_       # and one single occurrence at the very end
      return b
    if kind == "uniform":
      return rng.integers(1, 500, n).astype(np.int64)
    if kind == "pairs":
      h = rng.integers(1, 2**60, n // 2)
      return np.concatenate([h, h]).astype(np.int64)
    if kind == "special":
      b = rng.integers(1, 3000, n).astype(np.int64)
      b[::7] = np.iinfo(np.int64).min
      return b
    return S.id_batch(90 + s_, n, 10**9, "zipf")
  batches = [batch(s_) for s_ in range(steps + 1)]
  dev = [ids_t(b) for b in batches]
  mt = make({"emb": adagrad_cfg(dim, 0.05, 0.1)})
  ot = O.Table(O.segment(dim, O.OPT_ADAGRAD, p=(0.1, 0.0)), 1)
  step = SparseStep(mt, "emb", n, exact_order=exact)
  for s_ in range(steps):
    g = (S.grad_batch(s_, n, dim) * 10).astype(np.float32)
    emb = step.forward(dev[s_], next_ids=dev[s_ + 1])
    exp_emb = ot.lookup(batches[s_])[0]
    if exact:
      np.testing.assert_array_equal(emb.cpu().numpy(), exp_emb)
    else:
      np.testing.assert_allclose(emb.cpu().numpy(), exp_emb, rtol=RTOL_TREE, atol=ATOL_SUM)
    step.backward(val_t(g), S.update_time(s_))
    _oracle_step(ot, batches[s_], g, dim, 0.05, S.update_time(s_))
    assert step.n_unique() == np.unique(batches[s_]).size
  probe = np.unique(np.concatenate(batches[:steps]))
  got = mt.lookup({"emb": ids_t(probe)})["emb"].cpu().numpy()
  exp = ot.lookup(probe)[0]
  if exact:
    np.testing.assert_array_equal(got, exp)
  else:
    np.testing.assert_allclose(got, exp, rtol=RTOL_TREE, atol=ATOL_SUM)
    # ids that occur at most 32 times in every batch are summed sequentially: bit-exact
    cnt_max = {}
    for b in batches[:steps]:
      u, c = np.unique(b, return_counts=True)
      for k, v in zip(u.tolist(), c.tolist()):
        cnt_max[k] = max(cnt_max.get(k, 0), v)
    light = np.array([cnt_max[k] <= 32 for k in probe.tolist()])
    np.testing.assert_array_equal(got[light], exp[light])
  assert mt.size("emb") == probe.size == ot.size()


def test_fused_backward_slow_path_at_high_load():
  # full buckets -> deferred ids keep their summed gradient in grad_unique for slowpath_kernel
  cap, dim, n = 1 << 13, 8, 3000
  mt = make({"a": adagrad_cfg(dim, 0.1, 0.1, initial_capacity=cap, max_load_factor=0.97)})
  ot = O.Table(O.segment(dim, O.OPT_ADAGRAD, p=(0.1, 0.0)), cap)
  step = SparseStep(mt, "a", n)
  rng = np.random.default_rng(11)
  seen = set()
  for s_ in range(3):
    ids = rng.integers(1, 2**60, n)
    ids[n // 2:] = ids[:n - n // 2]  # every id twice
    g = S.grad_batch(s_, n, dim)
    step.forward(ids_t(ids))
    step.backward(val_t(g), S.update_time(s_))
    _oracle_step(ot, ids, g, dim, 0.1, S.update_time(s_))
    seen.update(ids.tolist())
  st = mt.stats("a")
  assert st.dropped == 0 and st.hashpower == 11 and st.size > 0.5 * cap
  allids = np.fromiter(seen, dtype=np.int64)
  np.testing.assert_array_equal(mt.lookup({"a": ids_t(allids)})["a"].cpu().numpy(),
                                ot.lookup(allids)[0])


@pytest.mark.parametrize("exact", [True, False])
def test_two_batches_of_lookahead_run_dedup_in_the_backward_launch(exact):
  """forward(ids, next_ids, ahead_ids): the run dedup of the batch TWO steps on rides in the backward
  launch (rd_dedup4_role: 256-thread workgroups, four positions per thread, the same scratch / run
  format bit for bit), the forward launches carry lookups only, three workspaces rotate.  Same
  results as one batch of look-ahead and as the oracle; Zipf batches with heavy lists, a batch size
  that is not a multiple of 1024, the special key, a pipeline that is dropped and restarted."""
  n, dim, steps = 33333, 32, 6
  rng = np.random.default_rng(5)
  batches = [S.id_batch(400 + s_, n, 10**6, "zipf") for s_ in range(steps + 2)]
  batches[3][::3] = batches[3][2]                       # a list of n / 3 occurrences
  batches[4][5::11] = np.iinfo(np.int64).min           # the key that lives in the side slot
  dev = [ids_t(b) for b in batches]
  grads = [S.grad_batch(s_, n, dim) for s_ in range(steps)]
  rows = {}
  for mode in ("one", "two", "two_restart"):
    mt = make({"emb": adagrad_cfg(dim, 0.01, 0.1)})
    step = SparseStep(mt, "emb", n, exact_order=exact)
    for s_ in range(steps):
      ahead = dev[s_ + 2] if mode != "one" else None
      if mode == "two_restart" and s_ == 3:
        step = SparseStep(mt, "emb", n, exact_order=exact)     # what was prepared ahead is dropped
      emb = step.forward(dev[s_], next_ids=dev[s_ + 1], ahead_ids=ahead)
      if s_ == 4:
        first = emb.cpu().numpy().copy()
      step.backward(val_t(grads[s_]), S.update_time(s_))
      assert step.n_unique() == np.unique(batches[s_]).size
    probe = np.unique(np.concatenate(batches[:steps]))
    rows[mode] = (mt.lookup({"emb": ids_t(probe)})["emb"].cpu().numpy(), first, mt.size("emb"))
    assert rows[mode][2] == probe.size
  np.testing.assert_array_equal(rows["one"][0], rows["two"][0])
  np.testing.assert_array_equal(rows["one"][1], rows["two"][1])
  np.testing.assert_array_equal(rows["one"][0], rows["two_restart"][0])
  ot = O.Table(O.segment(dim, O.OPT_ADAGRAD, p=(0.1, 0.0)), 1)
  for s_ in range(steps):
    _oracle_step(ot, batches[s_], grads[s_], dim, 0.01, S.update_time(s_))
  exp = ot.lookup(probe)[0]
  if exact:
    np.testing.assert_array_equal(rows["two"][0], exp)
  else:
    np.testing.assert_allclose(rows["two"][0], exp, rtol=RTOL_TREE, atol=ATOL_SUM)


def test_pipelined_step_slow_path_at_high_load():
  """Pipelined step at load factor ~0.97: many ids find both buckets full, so the displacement
  pass has work in every step.  It runs as one wavefront of the NEXT forward launch and the lookup
  workgroups gate on it (a looked-up id may be one it just placed): the forward outputs and the
  final rows must still be the oracle's, bit for bit."""
  cap, dim, n, steps = 1 << 13, 8, 3000, 4
  mt = make({"a": adagrad_cfg(dim, 0.1, 0.1, initial_capacity=cap, max_load_factor=0.97)})
  ot = O.Table(O.segment(dim, O.OPT_ADAGRAD, p=(0.1, 0.0)), cap)
  step = SparseStep(mt, "a", n, exact_order=True)
  rng = np.random.default_rng(12)
  batches = []
  for s_ in range(steps + 1):
    ids = rng.integers(1, 2**60, n)
    ids[n // 2:] = ids[:n - n // 2]          # every id twice
    if s_ > 0:
      ids[:n // 4] = batches[-1][:n // 4]    # a quarter of the previous batch again: the lookup
    batches.append(ids)                      # right after an update sees what that update inserted
  dev = [ids_t(b) for b in batches]
  seen = set()
  for s_ in range(steps):
    g = S.grad_batch(s_, n, dim)
    emb = step.forward(dev[s_], next_ids=dev[s_ + 1])
    np.testing.assert_array_equal(emb.cpu().numpy(), ot.lookup(batches[s_])[0])
    step.backward(val_t(g), S.update_time(s_))
    _oracle_step(ot, batches[s_], g, dim, 0.1, S.update_time(s_))
    seen.update(batches[s_].tolist())
  st = mt.stats("a")
  assert st.dropped == 0 and st.hashpower == 11 and st.size > 0.5 * cap
  allids = np.fromiter(seen, dtype=np.int64)
  np.testing.assert_array_equal(mt.lookup({"a": ids_t(allids)})["a"].cpu().numpy(),
                                ot.lookup(allids)[0])


@pytest.mark.parametrize("n,dim,shards,dist", [(30000, 16, 8, "zipf"), (65536, 64, 2, "zipf"),
                                               (5000, 8, 64, "uniform"), (1, 4, 3, "uniform")])
def test_sender_side_partition_scatter_sum(n, dim, shards, dist):
  """mhte_shard_partition / mhte_step_scatter / mhte_step_sum (the sharded step's sender side):
  shard-major packing by floormod(id, N) (fused_reorder_by_indices.cc:75-123), rows of the unique
  ids scattered to every occurrence (FillWithOffsetMap) and per-id gradient sums in send order
  (FillWithOffsetMapGradient) — against numpy."""
  from tests.torch_sharded_step import HipBackend
  mt = make({"emb": adagrad_cfg(dim)})
  be = HipBackend(mt, "emb")
  ids = S.id_batch(77, n, 10**5 if dist == "zipf" else 10**9, dist)
  if n > 10:
    ids[3] = -5                       # floormod of a negative id
  g = S.grad_batch(77, n, dim)
  uids_t, nu_t = be.dedup(ids_t(ids))
  send_ids_t, send_pos_t, counts_t = be.partition(uids_t, nu_t, shards)
  U = int(nu_t.item())
  uids = uids_t.cpu().numpy()[:U]
  assert sorted(uids.tolist()) == sorted(np.unique(ids).tolist())
  send_ids = send_ids_t.cpu().numpy()[:U]
  send_pos = send_pos_t.cpu().numpy()[:U].astype(np.int64)
  counts = counts_t.cpu().numpy().astype(np.int64)
  np.testing.assert_array_equal(counts, np.bincount(np.mod(uids, shards), minlength=shards))
  assert sorted(send_pos.tolist()) == list(range(U))                 # a permutation
  np.testing.assert_array_equal(send_ids[send_pos], uids)
  np.testing.assert_array_equal(np.mod(send_ids, shards), np.repeat(np.arange(shards), counts))
  # scatter
  index = {int(k): i for i, k in enumerate(uids)}
  inv = np.array([index[int(x)] for x in ids], dtype=np.int64)
  rows = np.random.default_rng(5).standard_normal((U, dim)).astype(np.float32)
  out = be.scatter(val_t(rows), send_pos_t, n).cpu().numpy()
  np.testing.assert_array_equal(out, rows[send_pos[inv]])
  # sum: sequential in occurrence order for lists of <= 32, 1e-5 otherwise
  gs = be.sum(val_t(g), send_pos_t).cpu().numpy()[:U]
  exp = np.zeros((U, dim), np.float32)
  for p in range(n):
    exp[send_pos[inv[p]]] += g[p]
  cnt = np.bincount(inv, minlength=U)
  light = np.zeros(U, bool)
  light[send_pos] = cnt <= 32
  np.testing.assert_array_equal(gs[light], exp[light])
  np.testing.assert_allclose(gs, exp, rtol=RTOL_TREE, atol=ATOL_SUM)


# =============================================================================== remaining optimizers
_OPT_ENTRY = {
    "momentum": lambda: entry.MomentumOptimizer(0.01),
    "adadelta": lambda: entry.AdadeltaOptimizer(0.01),
    "rmsprop": lambda: entry.RmspropOptimizer(0.01),
    "rmspropv2": lambda: entry.RmspropOptimizer(0.01, v2=True),
    "adam": lambda: entry.AdamOptimizer(0.01),
    "amsgrad": lambda: entry.AdamOptimizer(0.01, amsgrad=True),
    "moving_average": lambda: entry.MovingAverageOptimizer(0.9),
    "group_adagrad": lambda: entry.AdaGradWithGroupLassoOptimizer(0.01, beta=1.0, initial_accumulator_value=0.0,
                                                                  l2_regularization=1.0),
}


@pytest.mark.parametrize("name", sorted(_OPT_ENTRY))
def test_remaining_optimizers_kat_and_oracle(name, tmp_path):
  """Momentum / Adadelta / RMSProp (v1, v2) / Adam / AMSGrad in the op-level kernels: the
  reference's own KATs (*_optimizer_test.cc), then a multi-step run with duplicates (one optimizer
  step per occurrence, in order) bit-exact against the oracle's restatement, a table that is not
  mhte_table_fused_backward_ok, and a checkpoint round trip that carries the state."""
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  from test_oracle import OPT_KATS
  kat = [k for k in OPT_KATS if k[0] == name][0]
  _, oopt, p, lr, grads, exp1, exp2, _ = kat
  dim = len(grads)
  mt = make({"t": entry.make_table_config(
      [entry.CombineAsSegment(dim, entry.ZerosInitializer(), _OPT_ENTRY[name]())])})
  g = val_t([grads])
  mt.apply_gradients({"t": (ids_t([7]), g)})
  np.testing.assert_allclose(mt.lookup({"t": ids_t([7])})["t"].cpu().numpy()[0], exp1, rtol=0, atol=1e-6)
  mt.apply_gradients({"t": (ids_t([7]), g)})
  got2 = mt.lookup({"t": ids_t([7])})["t"].cpu().numpy()[0]
  for a, b in zip(got2, exp2):
    if b is not None:
      assert abs(a - b) < 1e-6
  # ---- multi-segment, duplicates, several steps vs the oracle
  d1, d2 = 8, 4
  segs = [entry.CombineAsSegment(d1, entry.ConstantsInitializer(0.25), _OPT_ENTRY[name]()),
          entry.CombineAsSegment(d2, entry.ZerosInitializer(), entry.AdagradOptimizer(0.05, 0.1))]
  mt2 = make({"t": entry.make_table_config(segs)})
  # (2 = fits the fused step through its FULL instantiations; 0 = the whole-segment GroupAdaGrad)
  assert mt2._lib.mhte_table_fused_backward_ok(mt2.handle, 0) == (0 if name == "group_adagrad" else 2)  # pylint: disable=protected-access
  ot = O.Table([O.segment(d1, oopt, p=p, init=O.INIT_CONSTANT, init_value=0.25),
                O.segment(d2, O.OPT_ADAGRAD, p=(0.1, 0.0))], 1)
  rng = np.random.default_rng(len(name))
  for step in range(4):
    ids = rng.integers(0, 300, 1000).astype(np.int64)
    gr = rng.standard_normal((1000, d1 + d2)).astype(np.float32)
    mt2.apply_gradients({"t": (ids_t(ids), val_t(gr))}, req_time=100 + step)
    ot.optimize(ids, gr, [lr, 0.05], 100 + step)
  probe = np.arange(300, dtype=np.int64)
  np.testing.assert_array_equal(mt2.lookup({"t": ids_t(probe)})["t"].cpu().numpy(), ot.lookup(probe)[0])
  # ---- checkpoint round trip keeps the optimizer state (training continues identically)
  base = str(tmp_path / "ck")
  mt2.save(base)
  mt3 = make({"t": entry.make_table_config(segs)})
  mt3.restore(base)
  ids = rng.integers(0, 300, 500).astype(np.int64)
  gr = rng.standard_normal((500, d1 + d2)).astype(np.float32)
  for m_ in (mt2, mt3):
    m_.apply_gradients({"t": (ids_t(ids), val_t(gr))}, req_time=200)
  np.testing.assert_array_equal(mt2.lookup({"t": ids_t(probe)})["t"].cpu().numpy(),
                                mt3.lookup({"t": ids_t(probe)})["t"].cpu().numpy())


@pytest.mark.parametrize("name", sorted(n for n in _OPT_ENTRY if n != "group_adagrad") + ["batch_softmax"])
@pytest.mark.parametrize("how", ["pipelined", "plain", "multi"])
def test_remaining_optimizers_on_the_fused_step(name, how):
  """Round 2 sent tables with these optimizers down the unpipelined op-level path; now the training
  step takes them (FULL instantiations of step_fwd / step_bwd / mstep_bwd): the pipelined two-launch
  step, the unpipelined step, and the multi-table step with a MIX of optimizer families (two backward
  launches, one per family) — duplicate gradients summed in occurrence order, ONE optimizer step per
  distinct id, bit-exact against the oracle with MHTE_EXACT_ORDER."""
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  from test_oracle import OPT_KATS
  from monolith_amd.fused_step import MultiSparseStep
  if name == "batch_softmax":
    oopt, p, lr, mk, d1 = O.OPT_BATCH_SOFTMAX, (), 0.1, (lambda: entry.BatchSoftmaxOptimizer(0.1)), 1
    # (a one-float segment: the row is not whole float4s, so it rides the VEC = 1 kernels)
  else:
    _, oopt, p, lr, _, _, _, _ = [k for k in OPT_KATS if k[0] == name][0]
    mk, d1 = _OPT_ENTRY[name], 16
  d2 = 16
  segs = [entry.CombineAsSegment(d1, entry.ConstantsInitializer(0.25), mk()),
          entry.CombineAsSegment(d2, entry.ZerosInitializer(), entry.AdagradOptimizer(0.05, 0.1))]
  osegs = [O.segment(d1, oopt, p=p, init=O.INIT_CONSTANT, init_value=0.25), O.segment(d2, O.OPT_ADAGRAD, p=(0.1, 0.0))]
  dim, B, steps = d1 + d2, 4096, 5
  rng = np.random.default_rng(len(name) + len(how))
  batches = [(rng.zipf(1.3, B) % 900).astype(np.int64) | (1 << 48) for _ in range(steps + 1)]
  grads = [rng.standard_normal((B, dim)).astype(np.float32) * 0.1 for _ in range(steps)]
  gstep = 1000

  def oracle_step(ot, ids, g, lrs, t, gs):
    uk, _, vo, vos, _ = O.unique_key_with_value_and_offset(ids, [0, ids.size], [g.shape[1]])
    gu = O.fill_with_offset_map_gradient(np.arange(uk.size), [0, uk.size], g.ravel(), vo, vos,
                                         [g.shape[1]]).reshape(-1, g.shape[1])
    ot.optimize(uk, gu, lrs, t, global_step=gs)

  if how == "multi":
    # (d1 = 1, BatchSoftmax: a one-float segment — the table rides the launches with one float per
    # lane, beside the float4 tables)
    cfgs = {"a_full": entry.make_table_config(segs),
            "b_basic": entry.make_table_config([entry.CombineAsSegment(32, entry.ZerosInitializer(),
                                                                       entry.AdagradOptimizer(0.01, 0.1))]),
            "c_full": entry.make_table_config([entry.CombineAsSegment(d1 if d1 % 4 else 16, entry.ZerosInitializer(), mk())])}
    mt = make(cfgs)
    ots = {"a_full": O.Table(osegs, 1), "b_basic": O.Table(O.segment(32, O.OPT_ADAGRAD, p=(0.1, 0.0)), 1),
           "c_full": O.Table(O.segment(d1 if d1 % 4 else 16, oopt, p=p), 1)}
    lrs = {"a_full": [lr, 0.05], "b_basic": [0.01], "c_full": [lr]}
    dims = {"a_full": dim, "b_basic": 32, "c_full": d1 if d1 % 4 else 16}
    names = sorted(cfgs)
    step = MultiSparseStep(mt, B, exact_order=True)
    rag = [mt.get_ragged_id({n: ids_t(batches[s] + k) for k, n in enumerate(names)}) for s in range(steps + 1)]
    for s in range(steps):
      emb = step.forward(rag[s], rag[s + 1])
      views = mt.get_embeddings(rag[s], emb)
      flat = []
      for k, n in enumerate(names):
        ids = batches[s] + k
        np.testing.assert_array_equal(views[n].cpu().numpy(), ots[n].lookup(ids)[0], err_msg="%s step %d" % (n, s))
        g = (np.random.default_rng(100 * s + k).standard_normal((B, dims[n])) * 0.1).astype(np.float32)
        flat.append(g.ravel())
        oracle_step(ots[n], ids, g, lrs[n], 100 + s, gstep + 3 * s)
      step.backward(val_t(np.concatenate(flat)), 100 + s, global_step=gstep + 3 * s)
    for k, n in enumerate(names):
      probe = np.unique(np.concatenate([b + k for b in batches[:steps]]))
      np.testing.assert_array_equal(mt.lookup({n: ids_t(probe)})[n].cpu().numpy(), ots[n].lookup(probe)[0])
    return
  mt = make({"t": entry.make_table_config(segs)})
  ot = O.Table(osegs, 1)
  step = SparseStep(mt, "t", B, exact_order=True)
  dev_ids = [ids_t(b) for b in batches]
  for s in range(steps):
    emb = step.forward(dev_ids[s], next_ids=dev_ids[s + 1] if how == "pipelined" else None)
    np.testing.assert_array_equal(emb.cpu().numpy(), ot.lookup(batches[s])[0], err_msg="step %d" % s)
    step.backward(val_t(grads[s]), 100 + s, global_step=gstep + 3 * s)
    oracle_step(ot, batches[s], grads[s], [lr, 0.05], 100 + s, gstep + 3 * s)
  probe = np.unique(np.concatenate(batches[:steps]))
  np.testing.assert_array_equal(mt.lookup({"t": ids_t(probe)})["t"].cpu().numpy(), ot.lookup(probe)[0])
  assert mt.size("t") == ot.size()


@pytest.mark.parametrize("how", ["op", "pipelined", "plain"])
@pytest.mark.parametrize("dim", [64, 27])
def test_adagrad_avx_form_matches_the_reference_as_built(how, dim):
  """entry.AdagradOptimizer(avx_semantics=True): the reference's AVX2 AdagradOptimize
  (avx_utils.h:96-119 — what its .bazelrc:63-68 build runs), which with weight_decay_factor != 0 is a
  different update from the baseline loop.  Against tests/golden/adagrad_avx_kat.npz (the reference's
  own NewAdagradOptimizer object compiled -mavx2 -mfma, 12 Optimize() calls at wd = 0.1) BIT FOR
  BIT, through the op-level kernel and the fused training step (dim 64: float4 rows; dim 27: three
  fused blocks + a baseline tail on the one-float-per-lane kernels); and against the oracle on a
  Zipf batch with duplicates."""
  z = np.load(os.path.join(GOLD, "adagrad_avx_kat.npz"))
  opt = lambda: entry.AdagradOptimizer(0.05, 0.1, weight_decay_factor=0.1, avx_semantics=True)  # noqa: E731
  mt = make({"t": entry.make_table_config([entry.CombineAsSegment(dim, entry.ZerosInitializer(), opt())])})
  one = np.array([77], np.int64)
  g = z["grad_d%d" % dim]
  step = SparseStep(mt, "t", 1, exact_order=True) if how != "op" else None
  for s in range(g.shape[0]):
    if how == "op":
      mt.apply_gradients({"t": (ids_t(one), val_t(g[s:s + 1]))}, req_time=10 + s)
    else:
      step.forward(ids_t(one), next_ids=ids_t(one) if how == "pipelined" else None)
      step.backward(val_t(g[s:s + 1]), 10 + s)
    got = mt.lookup({"t": ids_t(one)})["t"].cpu().numpy()[0]
    np.testing.assert_array_equal(got, z["num_d%d" % dim][s], err_msg="call %d" % s)
  # a batch with duplicates against the oracle's restatement of the same form
  mt2 = make({"t": entry.make_table_config([entry.CombineAsSegment(dim, entry.ZerosInitializer(), opt())])})
  ot = O.Table([O.segment(dim, O.OPT_ADAGRAD, p=(0.1, 0.1, 1.0))], 1)
  rng = np.random.default_rng(dim)
  B = 2048
  st2 = SparseStep(mt2, "t", B, exact_order=True) if how != "op" else None
  batches = [(rng.zipf(1.3, B) % 500).astype(np.int64) for _ in range(4)]
  for s in range(3):
    gr = (rng.standard_normal((B, dim)) * 0.3).astype(np.float32)
    uk, _, vo, vos, _ = O.unique_key_with_value_and_offset(batches[s], [0, B], [dim])
    gu = O.fill_with_offset_map_gradient(np.arange(uk.size), [0, uk.size], gr.ravel(), vo, vos, [dim]).reshape(-1, dim)
    ot.optimize(uk, gu, [0.05], 50 + s)
    if how == "op":
      mt2.apply_gradients({"t": (ids_t(uk), val_t(gu))}, req_time=50 + s)
    else:
      st2.forward(ids_t(batches[s]), next_ids=ids_t(batches[s + 1]) if how == "pipelined" else None)
      st2.backward(val_t(gr), 50 + s)
  probe = np.unique(np.concatenate(batches[:3]))
  np.testing.assert_array_equal(mt2.lookup({"t": ids_t(probe)})["t"].cpu().numpy(), ot.lookup(probe)[0])


def _half_neighbours(x):
  """(down, up): the two binary16 values around each fp32 x, as fp32 (|x| well inside the range)"""
  h = x.astype(np.float16)
  hf = h.astype(np.float32)
  up = np.where(hf >= x, hf, np.nextafter(h, np.float16(np.inf)).astype(np.float32))
  dn = np.where(hf <= x, hf, np.nextafter(h, np.float16(-np.inf)).astype(np.float32))
  return dn, up


@pytest.mark.parametrize("name", ["sgd", "adagrad", "adam"])
@pytest.mark.parametrize("how", ["op", "pipelined", "plain"])
def test_stochastic_rounding_float16(name, how):
  """OptimizerConfig.stochastic_rounding_float16 (the StochasticRoundingFloat16OptimizerDecorator,
  optimizer/stochastic_rounding.h): after every update the weights of the flagged segment are one of
  the two binary16 neighbours of what the optimizer computed, its state and the other segment are
  what the oracle computes bit for bit, and the rounding is unbiased (the upper neighbour is taken
  with probability (w - down) / (up - down): a z-score over > 30 000 inexact elements).  The rounding function
  itself is pinned to the reference's on the CPU (tests/test_stochastic_rounding.py); its draws are
  the engine's own (the reference's come from a thread-local generator in call order)."""
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  from test_oracle import OPT_KATS
  if name == "sgd":
    oopt, p, lr, mk = O.OPT_SGD, (), 0.05, (lambda: entry.SgdOptimizer(0.05))
  elif name == "adagrad":
    oopt, p, lr, mk = O.OPT_ADAGRAD, (0.1, 0.01), 0.05, (lambda: entry.AdagradOptimizer(0.05, 0.1, weight_decay_factor=0.01))
  else:
    _, oopt, p, lr, _, _, _, _ = [k for k in OPT_KATS if k[0] == name][0]
    mk = _OPT_ENTRY[name]
  d1 = d2 = 16
  segs = [entry.CombineAsSegment(d1, entry.ConstantsInitializer(0.25),
                                 entry.StochasticRoundingFloat16OptimizerWrapper(mk())),
          entry.CombineAsSegment(d2, entry.ZerosInitializer(), entry.AdagradOptimizer(0.05, 0.1))]
  osegs = [O.segment(d1, oopt, p=p, init=O.INIT_CONSTANT, init_value=0.25), O.segment(d2, O.OPT_ADAGRAD, p=(0.1, 0.0))]
  dim, B, steps = d1 + d2, 4096, 4
  rng = np.random.default_rng(5 + len(name) + len(how))
  mt = make({"t": entry.make_table_config(segs)})
  assert mt._lib.mhte_table_fused_backward_ok(mt.handle, 0) == 2    # (the FULL kernel family)
  ot = O.Table(osegs, 1)
  lrs = [lr, 0.05]
  step = SparseStep(mt, "t", B, exact_order=True) if how != "op" else None
  batches = [(rng.zipf(1.3, B) % 1500).astype(np.int64) | (1 << 48) for _ in range(steps + 1)]
  dev_ids = [ids_t(b) for b in batches]
  num = den = 0.0
  n_inexact = 0
  for s in range(steps):
    ids = batches[s]
    g = (rng.standard_normal((B, dim)) * 0.1).astype(np.float32)
    uk = np.unique(ids)
    # the oracle continues from the device's (rounded) weights: Assign overwrites weights only
    have = uk[np.array([ot.contains(int(k)) for k in uk], bool)] if ot.size() else uk[:0]
    if have.size:
      ot.assign(have, mt.lookup({"t": ids_t(have)})["t"].cpu().numpy(), 50)
    uo, _, vo, vos, _ = O.unique_key_with_value_and_offset(ids, [0, ids.size], [dim])
    gu = O.fill_with_offset_map_gradient(np.arange(uo.size), [0, uo.size], g.ravel(), vo, vos, [dim]).reshape(-1, dim)
    ot.optimize(uo, gu, lrs, 100 + s)
    if how == "op":
      mt.table_optimize_n("t", ids_t(uo), None, val_t(gu), np.array(lrs, np.float32), 100 + s, 0)
    else:
      step.forward(dev_ids[s], next_ids=dev_ids[s + 1] if how == "pipelined" else None)
      step.backward(val_t(g), 100 + s)
    got = mt.lookup({"t": ids_t(uk)})["t"].cpu().numpy()
    exp = ot.lookup(uk)[0]
    np.testing.assert_array_equal(got[:, d1:], exp[:, d1:])            # the unflagged segment
    dn, up = _half_neighbours(exp[:, :d1])
    w = got[:, :d1]
    assert ((w == dn) | (w == up)).all(), "step %d" % s
    assert (w.astype(np.float16).astype(np.float32) == w).all()
    inexact = up > dn
    frac = (exp[:, :d1] - dn)[inexact] / (up - dn)[inexact]
    num += float(((w == up)[inexact].astype(np.float64) - frac).sum())
    den += float((frac * (1 - frac)).sum())
    n_inexact += int(inexact.sum())
  # optimizer state of the flagged segment: untouched by the rounding.  (Adam's row keeps its two
  # powers in a float4 here and in two floats in the oracle: its state is checked through the steps
  # above — a state that had drifted would take the next step's weights off the neighbour pair.)
  di, _, _, drows = mt.dump("t")
  oi, _, _, orows = ot.dump()
  do, oo = np.argsort(di.cpu().numpy()), np.argsort(oi)
  np.testing.assert_array_equal(di.cpu().numpy()[do], oi[oo])
  if drows.shape[1] == orows.shape[1]:
    np.testing.assert_array_equal(drows.cpu().numpy()[do][:, dim:], orows[oo][:, dim:])
  else:
    assert name == "adam"
  assert n_inexact > 30000 and abs(num) / np.sqrt(den) < 5.0, (num, den, n_inexact)


def test_batch_softmax_optimizer(tmp_path):
  """BatchSoftmaxOptimizer (batch_softmax_optimizer.cc:52-63): the reference's KAT, then several
  steps with duplicates and a growing global_step bit-exact against the oracle (a one-float
  segment next to an Adagrad one), and a checkpoint round trip that carries the int64 step."""
  mt = make({"t": entry.make_table_config(
      [entry.CombineAsSegment(1, entry.ZerosInitializer(), entry.BatchSoftmaxOptimizer(0.1))])})
  mt.apply_gradients({"t": (ids_t([7]), val_t([[2.0]]))}, global_step=1)
  assert mt.lookup({"t": ids_t([7])})["t"].cpu().numpy()[0][0] == np.float32(0.1)
  segs = [entry.CombineAsSegment(1, entry.ZerosInitializer(), entry.BatchSoftmaxOptimizer(0.1)),
          entry.CombineAsSegment(4, entry.ZerosInitializer(), entry.AdagradOptimizer(0.05, 0.1))]
  mt2 = make({"t": entry.make_table_config(segs)})
  ot = O.Table([O.segment(1, O.OPT_BATCH_SOFTMAX), O.segment(4, O.OPT_ADAGRAD, p=(0.1, 0.0))], 1)
  rng = np.random.default_rng(11)
  step = (1 << 33) + 5   # (the stored step is a full int64)
  for k in range(5):
    ids = rng.integers(0, 200, 700).astype(np.int64)
    gr = rng.standard_normal((700, 5)).astype(np.float32)
    step += int(rng.integers(1, 50))
    mt2.apply_gradients({"t": (ids_t(ids), val_t(gr))}, global_step=step, req_time=100 + k)
    ot.optimize(ids, gr, [0.1, 0.05], 100 + k, global_step=step)
  probe = np.arange(200, dtype=np.int64)
  np.testing.assert_array_equal(mt2.lookup({"t": ids_t(probe)})["t"].cpu().numpy(), ot.lookup(probe)[0])
  base = str(tmp_path / "ck")
  mt2.save(base)
  mt3 = make({"t": entry.make_table_config(segs)})
  mt3.restore(base)
  ids = rng.integers(0, 200, 300).astype(np.int64)
  gr = rng.standard_normal((300, 5)).astype(np.float32)
  for m_ in (mt2, mt3):
    m_.apply_gradients({"t": (ids_t(ids), val_t(gr))}, global_step=step + 17, req_time=200)
  np.testing.assert_array_equal(mt2.lookup({"t": ids_t(probe)})["t"].cpu().numpy(),
                                mt3.lookup({"t": ids_t(probe)})["t"].cpu().numpy())
  with pytest.raises(_lib.MhteError):
    make({"bad": entry.make_table_config(
        [entry.CombineAsSegment(2, entry.ZerosInitializer(), entry.BatchSoftmaxOptimizer(0.1))])})


# =============================================================================== gather + pooling
def test_fused_gather_embeddings_by_input_golden():
  """distribution_ops_test.py:221-249 (forward, with its 12345x / 11777x tiling) and :251-304
  (gradient, 888x tiling, rtol 1e-7 * 888)."""
  fused = val_t([1.1, 1.2, 1.3, 2.1, 2.2, 3.1, 3.2, 3.3, 4.1, 4.2, 4.3, 5.1, 5.2, 6.1, 6.2, 6.3,
                 7.1, 7.2, 8.1, 8.2, 9.1, 9.2])
  scale = (12345, 11777)
  offs = [torch.tensor([13, 0, 5, 13, 8, 13] * scale[0], dtype=torch.int32).cuda(),
          torch.tensor([16, 18, 11, 11, 16, 20, 3] * scale[1], dtype=torch.int32).cuda()]
  outs = D.fused_gather_embeddings_by_input(fused, offs, [3, 2])
  exp0 = np.array([[6.1, 6.2, 6.3], [1.1, 1.2, 1.3], [3.1, 3.2, 3.3], [6.1, 6.2, 6.3],
                   [4.1, 4.2, 4.3], [6.1, 6.2, 6.3]] * scale[0], np.float32)
  exp1 = np.array([[7.1, 7.2], [8.1, 8.2], [5.1, 5.2], [5.1, 5.2], [7.1, 7.2], [9.1, 9.2],
                   [2.1, 2.2]] * scale[1], np.float32)
  np.testing.assert_array_equal(outs[0].cpu().numpy(), exp0)
  np.testing.assert_array_equal(outs[1].cpu().numpy(), exp1)
  k = 888
  grads = [val_t(np.array([[1.1, 1.2, 1.3], [2.1, 2.2, 2.3], [3.1, 3.2, 3.3], [4.1, 4.2, 4.3],
                           [5.1, 5.2, 5.3], [6.1, 6.2, 6.3]] * k, np.float32)),
           val_t(np.array([[1.4, 1.5], [2.4, 2.5], [3.4, 3.5], [4.4, 4.5], [5.4, 5.5], [6.4, 6.5],
                           [7.4, 7.5]] * k, np.float32))]
  eo = [torch.tensor([13, 0, 5, 13, 8, 13] * k, dtype=torch.int32).cuda(),
        torch.tensor([16, 18, 11, 11, 16, 20, 3] * k, dtype=torch.int32).cuda()]
  out = D.fused_gather_embeddings_by_input_gradient(22, grads, eo, [3, 2]).cpu().numpy()
  exp = np.array([2.1, 2.2, 2.3, 7.4, 7.5, 3.1, 3.2, 3.3, 5.1, 5.2, 5.3, 7.8, 8.0, 11.3, 11.6, 11.9,
                  6.8, 7.0, 2.4, 2.5, 6.4, 6.5]) * k
  np.testing.assert_allclose(out, exp, rtol=1e-7 * k)
  out2 = D.fused_gather_embeddings_by_input_gradient(22, grads, eo, [3, 2], scale=0.5).cpu().numpy()
  np.testing.assert_allclose(out2, exp * 0.5, rtol=1e-7 * k)


@pytest.mark.parametrize("sorted_", [True, False])
def test_reduce_sum_mean_sqrtn(sorted_):
  """distribution_ops_test.py:306-352 goldens, then a ragged batch against the reference's
  sequential accumulation (reduce_op.cc:29-125): bit-exact for sorted indices AND for indices in
  any order (rows of one output are grouped and added in index order: no atomics)."""
  idx = torch.tensor([[0], [0], [1]], dtype=torch.int64).cuda()
  np.testing.assert_array_equal(
      D.reduce_mean(idx, val_t([[4, 4], [2, 2], [1, 1]]), [2], sorted_).cpu().numpy(), [[3, 3], [1, 1]])
  np.testing.assert_array_equal(
      D.reduce_sum(idx, val_t([[1, 1], [2, 2], [4, 4]]), [2], sorted_).cpu().numpy(), [[3, 3], [4, 4]])
  np.testing.assert_allclose(
      D.reduce_sqrtn(idx, val_t([[3, 3], [4, 4], [2, 2]]), [2], sorted_).cpu().numpy(), [[5, 5], [2, 2]])
  rng = np.random.default_rng(9)
  batch, dim = 500, 24
  lens = rng.integers(0, 9, batch)
  lens[7] = 0
  ind = np.repeat(np.arange(batch), lens)
  vals = rng.standard_normal((ind.size, dim)).astype(np.float32)
  if not sorted_:
    perm = rng.permutation(ind.size)
    ind, vals = ind[perm], vals[perm]
  for mode, fn in ((0, D.reduce_sum), (1, D.reduce_mean), (2, D.reduce_sqrtn)):
    exp = np.zeros((batch, dim), np.float32)
    cnt = np.zeros(batch, np.int64)
    for i, b in enumerate(ind):
      exp[b] = exp[b] + (vals[i] * vals[i] if mode == 2 else vals[i])
      cnt[b] += 1
    with np.errstate(divide="ignore", invalid="ignore"):
      if mode == 1:
        exp = exp * (np.float32(1.0) / cnt.astype(np.float32))[:, None]
      if mode == 2:
        exp = np.sqrt(exp)
    got = fn(torch.from_numpy(ind[:, None]).cuda(), val_t(vals), [batch], sorted_).cpu().numpy()
    np.testing.assert_array_equal(got, exp)          # incl. the NaN row of an empty mean


def test_fused_gather_gradient_is_sequential_and_repeatable():
  """The gradient of FusedGatherEmbeddingsByInput without float atomics: rows that share an offset are
  added in row order (each scaled first, map_id_to_embedding.cu.cc:98-107) — equal to a sequential
  fp32 loop bit for bit, and the same bits on every run (the reference's GpuAtomicAdd adds in arrival
  order)."""
  rng = np.random.default_rng(21)
  dims, n_rows = [8, 16, 4], [30000, 20000, 5000]
  slots = [400, 300, 50]                     # distinct rows per input: heavy duplication
  base, offs_h, grads_h = 0, [], []
  for d, n, k in zip(dims, n_rows, slots):
    offs_h.append((base + rng.integers(0, k, n) * d).astype(np.int32))
    grads_h.append(rng.standard_normal((n, d)).astype(np.float32))
    base += k * d
  exp = np.zeros(base, np.float32)
  sc = np.float32(0.37)
  for o, g, d in zip(offs_h, grads_h, dims):
    acc = {}
    for j in range(o.size):
      a = acc.get(int(o[j]))
      t = g[j] * sc
      acc[int(o[j])] = t if a is None else a + t
    for off, v in acc.items():
      exp[off:off + d] = np.float32(0) + v
  offs = [torch.from_numpy(o).cuda() for o in offs_h]
  grads = [val_t(g) for g in grads_h]
  outs = [D.fused_gather_embeddings_by_input_gradient(base, grads, offs, dims, scale=float(sc)).cpu().numpy()
          for _ in range(3)]
  # (a sum starts from 0: 0 + first addend)
  np.testing.assert_array_equal(outs[0], exp)
  assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


@pytest.mark.parametrize("form", ["sort", "dedup"])
def test_pooling_gradient_grouping_forms_at_size(form, monkeypatch):
  """The two groupings behind the deterministic pooling gradients (AuxWs::group_sorted): this repo's stable
  radix sort of (key, position) pairs (csrc/mhte_group_kernels.h; several tiles per pass, two and three
  passes, one key holding a fifth of the rows) and the list-building dedup (MHTE_GROUP_DD=1, read per call).
  Both must give the reference's sequential sums bit for bit (reduce_op.cc:46-49;
  map_id_to_embedding.cu.cc:98-107 with the addends in row order)."""
  if form == "dedup":
    monkeypatch.setenv("MHTE_GROUP_DD", "1")
  else:
    monkeypatch.delenv("MHTE_GROUP_DD", raising=False)
  rng = np.random.default_rng(5)
  # unsorted reduce: 70 000 outputs (17 + 1 key bits: two passes), 200 000 rows, Zipf-like skew
  batch, dim, n = 70000, 8, 200000
  ind = np.minimum((rng.pareto(0.9, n) * 3).astype(np.int64), batch - 1)
  ind[rng.random(n) < 0.2] = 4242            # one output takes a fifth of the rows
  ind = (ind * 7919) % batch
  vals = rng.standard_normal((n, dim)).astype(np.float32)
  exp = np.zeros((batch, dim), np.float32)
  np.add.at(exp, ind, vals)                  # (unbuffered: row after row, in index order)
  got = D.reduce_sum(torch.from_numpy(ind[:, None]).cuda(), val_t(vals), [batch], False).cpu().numpy()
  np.testing.assert_array_equal(got, exp)
  # gather gradient: a fused buffer of 2^22 floats (22 + 1 key bits: three passes), 150 000 rows
  dims, n_rows, slots = [8, 4], [90000, 60000], [200000, 300000]
  base, offs_h, grads_h = 0, [], []
  for d, nr, k in zip(dims, n_rows, slots):
    o = rng.integers(0, k, nr)
    o[rng.random(nr) < 0.2] = 17
    offs_h.append((base + o * d).astype(np.int32))
    grads_h.append(rng.standard_normal((nr, d)).astype(np.float32))
    base += k * d
  fused_len = 1 << 22
  assert base <= fused_len
  sc = np.float32(0.37)
  exp = np.zeros(fused_len, np.float32)
  for o, g, d in zip(offs_h, grads_h, dims):
    np.add.at(exp, (o[:, None].astype(np.int64) + np.arange(d)[None, :]).ravel(), (g * sc).ravel())
  offs = [torch.from_numpy(o).cuda() for o in offs_h]
  grads = [val_t(g) for g in grads_h]
  outs = [D.fused_gather_embeddings_by_input_gradient(fused_len, grads, offs, dims, scale=float(sc)).cpu().numpy()
          for _ in range(2)]
  np.testing.assert_array_equal(outs[0], exp)
  assert np.array_equal(outs[0], outs[1])


def test_pooling_gradient_grouping_beyond_the_register_held_tiles():
  """csrc/mhte_group_kernels.h keeps a wavefront's 1 024 keys of a tile in registers; beyond 512 tiles (2 M
  keys) a wavefront walks its keys in ROUNDS of that shape (`GsPass::rounds` > 1: the keys are read again for the
  walk).  2.6 M unsorted rows into 100 000 outputs, one output holding a tenth of them: the reference's
  sequential sums (reduce_op.cc:46-49), bit for bit."""
  rng = np.random.default_rng(11)
  batch, dim, n = 100000, 4, 2600000
  ind = rng.integers(0, batch, n)
  ind[rng.random(n) < 0.1] = 777
  vals = rng.standard_normal((n, dim)).astype(np.float32)
  exp = np.zeros((batch, dim), np.float32)
  np.add.at(exp, ind, vals)
  got = D.reduce_sum(torch.from_numpy(ind[:, None]).cuda(), val_t(vals), [batch], False).cpu().numpy()
  np.testing.assert_array_equal(got, exp)


def test_pooling_gradient_grouping_many_keys_and_wide_keys():
  """The grouping's two shapes the size test above does not reach (csrc/mhte_group_kernels.h): more than 2 M
  keys — a wavefront then walks its share of a tile in ROUNDS of 1 024 keys instead of holding it in registers —
  and keys of 31 bits — four passes of 8 bits, the (key, position) words ping-ponging twice.  Same checks:
  the reference's sequential sums bit for bit (reduce_op.cc:46-49; map_id_to_embedding.cu.cc:98-107)."""
  rng = np.random.default_rng(11)
  # 2.6 M unsorted rows of dim 4 into 300 000 outputs (19 + 1 key bits: two passes, two rounds per wavefront)
  batch, dim, n = 300000, 4, 2600000
  ind = rng.integers(0, batch, n)
  ind[rng.random(n) < 0.1] = 77
  vals = rng.standard_normal((n, dim)).astype(np.float32)
  exp = np.zeros((batch, dim), np.float32)
  np.add.at(exp, ind, vals)
  got = D.reduce_sum(torch.from_numpy(ind[:, None]).cuda(), val_t(vals), [batch], False).cpu().numpy()
  np.testing.assert_array_equal(got, exp)
  del exp, got, vals
  # a fused buffer of 2^30 + 64 floats (4 GB): offsets are 31-bit keys, 31 + 1 bits = four passes
  fused_len = (1 << 30) + 64
  d, nr = 4, 120000
  o = rng.integers(0, fused_len // d - 1, nr).astype(np.int64)
  o[rng.random(nr) < 0.05] = (1 << 28) + 5          # one destination far up takes 5 % of the rows
  o[:3] = [0, fused_len // d - 1, (1 << 27) + 1]    # both ends of the buffer
  offs = (o * d).astype(np.int64)
  g = rng.standard_normal((nr, d)).astype(np.float32)
  sc = np.float32(1.5)
  acc = {}
  gs_ = g * sc
  for j in range(nr):
    a = acc.get(int(offs[j]))
    acc[int(offs[j])] = (np.float32(0) + gs_[j]) if a is None else a + gs_[j]
  out = D.fused_gather_embeddings_by_input_gradient(fused_len, [val_t(g)], [torch.from_numpy(offs.astype(np.int32)).cuda()],
                                                    [d], scale=float(sc))
  keys = np.sort(np.fromiter(acc.keys(), dtype=np.int64))
  # (read back through views of 2^27 floats: torch's advanced indexing of a tensor beyond 2^31 bytes raises a
  # hardware exception on this stack (HSA_STATUS_ERROR_EXCEPTION 0x1016 in its index kernel) and is not what is under test)
  CH = 1 << 27
  got = np.empty((keys.size, d), np.float32)
  for c0 in range(0, fused_len, CH):
    sel = np.nonzero((keys >= c0) & (keys < c0 + CH))[0]
    if sel.size:
      view = out[c0:min(fused_len, c0 + CH)]
      idx = torch.from_numpy(((keys[sel] - c0)[:, None] + np.arange(d)[None, :]).ravel()).cuda()
      got[sel] = view[idx].cpu().numpy().reshape(-1, d)
  np.testing.assert_array_equal(got, np.stack([acc[int(k)] for k in keys]))
  # nothing else was written: the touched floats carry the whole sum of absolute values
  assert float(out.abs().sum(dtype=torch.float64)) == pytest.approx(float(np.abs(got.astype(np.float64)).sum()), rel=1e-12)


# =============================================================================== admission + eviction
def _filtered_cfg(dim, opt, default_thr, slot_thr=None, **kw):
  return entry.make_table_config(
      [entry.CombineAsSegment(dim, entry.ZerosInitializer(), opt)],
      slot_occurrence_threshold_config=entry.SlotOccurrenceThresholdConfig(default_thr, slot_thr or {}),
      **kw)


def test_hash_filter_golden_sequence():
  """hash_table_ops_test.py:223-260: SGD lr 0.1, occurrence_threshold 3, ids [0, 0, 1], gradient -1
  per occurrence, one optimizer step per occurrence: after each of four applies the rows are
  [[0],[0]] -> [[.1],[0]] -> [[.3],[0]] -> [[.5],[.1]]."""
  from monolith_amd.multi_hash_table_ops import HashFilter
  flt = HashFilter(capacity=1000)
  mt = MultiHashTable.from_configs({"t": _filtered_cfg(1, entry.SgdOptimizer(0.1), 3)},
                                   name_suffix=_name(), hash_filter=flt)
  expected = [[[0], [0]], [[0.1], [0]], [[0.3], [0]], [[0.5], [0.1]]]
  for i in range(4):
    mt.apply_gradients({"t": (ids_t([0, 0, 1]), val_t([[-1], [-1], [-1]]))})
    got = mt.lookup({"t": ids_t([0, 1])})["t"].cpu().numpy()
    np.testing.assert_allclose(got, np.array(expected[i], np.float32), rtol=0, atol=1e-7)
  # id 0 stopped consulting the filter once it was in the table (4 consultations), id 1 asked 4 times
  np.testing.assert_array_equal(flt.get(ids_t([0, 1, 2])).cpu().numpy(), [4, 4, 0])
  assert mt.size("t") == 2


def test_hash_filter_modes_thresholds_and_saturation():
  """(a) enable_dedup semantics (MHTE_SUM_DUPLICATES): one consultation with the occurrence count
  (tf_bridge.cc:300-310); (b) per-slot thresholds, <= 0 admits at once (hash_filter.h:158-165);
  (c) the multi-table AssignAdd consults the filter WITHOUT the Contains guard (:230-232);
  (d) counts saturate at 15 (filter.h:56-57)."""
  from monolith_amd.multi_hash_table_ops import HashFilter
  flt = HashFilter(capacity=1000)
  cfg = _filtered_cfg(2, entry.SgdOptimizer(1.0), 4, {7: 2, 9: 0})
  mt = MultiHashTable.from_configs({"t": cfg}, name_suffix=_name(), hash_filter=flt)
  a, b, c = 5, (7 << 48) | 5, (9 << 48) | 5
  ids = ids_t([a, b, c, a, b, a])                # a x3 (thr 4), b x2 (thr 2), c x1 (thr 0)
  g = val_t(np.ones((6, 2), np.float32))
  lrs = np.array([1.0], np.float32)
  # (a) dedup: a seen 0 < 4 -> dropped (count 3); b seen 0 < 2 -> dropped (count 2); c admitted
  mt.table_optimize_n("t", ids, None, g, lrs, 100, flags=_lib.MHTE_SUM_DUPLICATES)
  assert mt.contains("t", ids_t([a, b, c])).cpu().tolist() == [False, False, True]
  np.testing.assert_array_equal(flt.get(ids_t([a, b, c])).cpu().numpy(), [3, 2, 0])
  # again: a seen 3 < 4 -> dropped (count 6); b seen 2 >= 2 -> admitted with the summed gradient
  mt.table_optimize_n("t", ids, None, g, lrs, 101, flags=_lib.MHTE_SUM_DUPLICATES)
  assert mt.contains("t", ids_t([a, b, c])).cpu().tolist() == [False, True, True]
  np.testing.assert_array_equal(mt.lookup({"t": ids_t([b, c])})["t"].cpu().numpy(),
                                [[-2, -2], [-2, -2]])
  # third time: a seen 6 >= 4 -> admitted
  mt.table_optimize_n("t", ids, None, g, lrs, 102, flags=_lib.MHTE_SUM_DUPLICATES)
  np.testing.assert_array_equal(mt.lookup({"t": ids_t([a])})["t"].cpu().numpy(), [[-3, -3]])
  np.testing.assert_array_equal(flt.get(ids_t([a])).cpu().numpy(), [9])
  # (c) AssignAdd: a is in the table but keeps asking the filter; a fresh id x: seen 0,1,2,3 dropped
  x = 77
  for k in range(5):
    mt.assign_add({"t": (ids_t([x]), val_t([[1.0, 1.0]]))})
    assert mt.contains("t", ids_t([x])).cpu().tolist() == [k >= 4]
  # (d) saturation
  for _ in range(20):
    mt.assign_add({"t": (ids_t([x]), val_t([[0.0, 0.0]]))})
  np.testing.assert_array_equal(flt.get(ids_t([x])).cpu().numpy(), [15])


def test_pipelined_step_with_hash_filter_matches_model():
  """The training step consults the filter once per distinct id with its occurrence count (the
  step sums duplicates first = enable_dedup).  Against a host model: dict of saturating counts +
  oracle table that only sees the admitted ids."""
  from monolith_amd.multi_hash_table_ops import HashFilter
  n, dim, steps, thr = 8000, 8, 6, 3
  flt = HashFilter(capacity=200000)
  mt = MultiHashTable.from_configs({"emb": _filtered_cfg(dim, entry.AdagradOptimizer(0.05, 0.1), thr)},
                                   name_suffix=_name(), hash_filter=flt)
  ot = O.Table(O.segment(dim, O.OPT_ADAGRAD, p=(0.1, 0.0)), 1)
  batches = [S.id_batch(300 + s_, n, 3000, "zipf") for s_ in range(steps + 1)]
  dev = [ids_t(b) for b in batches]
  step = SparseStep(mt, "emb", n, exact_order=True)
  seen, present = {}, set()
  for s_ in range(steps):
    g = S.grad_batch(s_, n, dim)
    emb = step.forward(dev[s_], next_ids=dev[s_ + 1])
    np.testing.assert_array_equal(emb.cpu().numpy(), ot.lookup(batches[s_])[0])
    step.backward(val_t(g), S.update_time(s_))
    uk, _, vo, vos, _ = O.unique_key_with_value_and_offset(batches[s_], [0, n], [dim])
    gu = O.fill_with_offset_map_gradient(np.arange(uk.size), [0, uk.size], g.ravel(), vo, vos,
                                         [dim]).reshape(-1, dim)
    cnt = dict(zip(*np.unique(batches[s_], return_counts=True)))
    keep = []
    for k_, i in enumerate(uk.tolist()):
      if i in present:
        keep.append(k_)
        continue
      c0 = seen.get(i, 0)
      seen[i] = min(15, c0 + min(15, int(cnt[i])))
      if c0 >= thr:
        keep.append(k_)
        present.add(i)
    ot.optimize(uk[keep], gu[keep], [0.05], S.update_time(s_))
  probe = np.unique(np.concatenate(batches[:steps]))
  np.testing.assert_array_equal(mt.lookup({"emb": ids_t(probe)})["emb"].cpu().numpy(),
                                ot.lookup(probe)[0])
  assert mt.size("emb") == len(present) == ot.size() and 0 < len(present) < probe.size
  np.testing.assert_array_equal(flt.get(ids_t(probe)).cpu().numpy(),
                                [seen.get(int(i), 0) for i in probe])


def test_feature_eviction_cadence_and_ttl():
  """tf_bridge.cc:73-104 + cuckoo_embedding_hash_table.cc:251-264: every feature_evict_every_n_hours
  the table drops rows with max_update_time - ts >= ttl(slot) days; the check rides on the update
  calls.  TTL values of embedding_hash_table_test.h:282-326."""
  day = 86400
  ttl = entry.SlotExpireTimeConfig(default_expire_time=14, slot_expire_times={1: 5, 2: 6})
  cfg = entry.make_table_config(
      [entry.CombineAsSegment(1, entry.ZerosInitializer(), entry.SgdOptimizer(1.0))],
      entry.CuckooHashTableConfig(feature_evict_every_n_hours=2), slot_expire_time_config=ttl)
  mt = make({"t": cfg})
  adv = _lib.lib().mhte_advance_clock_for_testing   # (the cadence lives in the library)
  t0 = 1_600_000_000
  fids = [(1 << 48) | 123, (2 << 48) | 456, 789]
  mt.assign_add({"t": (ids_t(fids), val_t([[1.0], [2.0], [3.0]]))}, req_time=t0)
  g = val_t([[0.0]])
  adv(3600.0)                # one hour: nothing is due
  mt.apply_gradients({"t": (ids_t([55]), g)}, req_time=t0 + 5 * day + 60)
  assert mt.size("t") == 4
  adv(3600.0 + 11)           # two hours passed: the scan runs on the next update call
  mt.apply_gradients({"t": (ids_t([55]), g)}, req_time=t0 + 5 * day + 61)
  assert mt.contains("t", ids_t(fids + [55])).cpu().tolist() == [False, True, True, True]
  assert mt.stats("t").evicted == 1


# =============================================================================== checkpoints
def test_save_restore_reference_scenario(tmp_path):
  """multi_hash_table_ops_test.py:129-170: save {slot0, slot1, slot2}, restore into
  {slot0, slot2, slot3}: known tables come back, unknown ones are skipped, missing ones stay empty."""
  t0 = make({"slot0": sgd_cfg(1), "slot1": sgd_cfg(2), "slot2": sgd_cfg(2)})
  t0.assign_add({"slot0": (ids_t([0, 1]), val_t([[1], [2]])),
                 "slot1": (ids_t([2, 3, 4, 5]), val_t([[1, 2], [2, 3], [3, 4], [4, 5]])),
                 "slot2": (ids_t([6, 7, 8, 9, 10]), val_t([[1, 1], [2, 2], [3, 3], [4, 4], [5, 5]]))})
  base = str(tmp_path / "test_save_restore" / "table")
  t0.save(base)
  t1 = make({"slot0": sgd_cfg(1), "slot2": sgd_cfg(2), "slot3": sgd_cfg(3)})
  t1.restore(base)
  got = t1.lookup({"slot0": ids_t([0, 1]), "slot2": ids_t([6, 7, 8, 9, 10])})
  np.testing.assert_array_equal(got["slot0"].cpu().numpy(), [[1], [2]])
  np.testing.assert_array_equal(got["slot2"].cpu().numpy(), [[1, 1], [2, 2], [3, 3], [4, 4], [5, 5]])
  assert t1.size("slot3") == 0 and t1.size("slot0") == 2 and t1.size("slot2") == 5


@pytest.mark.parametrize("nshards", [1, 3])
def test_checkpoint_files_are_the_reference_format_and_roundtrip(tmp_path, nshards):
  """Save -> the files parse with an independent reader (TFRecord framing + masked crc32c, TF's
  snappy block stream, EntryDump / MultiHashTableMetadata through the protobuf runtime) into
  exactly the table's content, bucket-range sharded; restore into a fresh table reproduces every
  row, optimizer state and timestamp, and training continues bit-identically."""
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  import ckpt_proto as P
  segs = [entry.CombineAsSegment(1, entry.ZerosInitializer(),
                                 entry.FtrlOptimizer(0.03, 0.1, 1.0, 0, 0.01, 0.02)),
          entry.CombineAsSegment(8, entry.ConstantsInitializer(0.5), entry.AdagradOptimizer(0.05, 0.1)),
          entry.CombineAsSegment(3, entry.ZerosInitializer(), entry.SgdOptimizer(0.1))]
  cfgs = {"b_multi": entry.make_table_config(segs), "a_sgd": sgd_cfg(2, 0.5)}
  mt = make(cfgs)
  rng = np.random.default_rng(3)
  n = 5000
  ids = np.unique(rng.integers(-2**62, 2**62, n))
  ids[0] = np.iinfo(np.int64).min                       # the side-slot key travels too
  for step in range(3):
    mt.apply_gradients(
        {"b_multi": (ids_t(ids), val_t(rng.standard_normal((ids.size, 12)).astype(np.float32))),
         "a_sgd": (ids_t(ids[:ids.size // 2]),
                   val_t(rng.standard_normal((ids.size // 2, 2)).astype(np.float32)))},
        req_time=1_700_000_000 + step)
  base = str(tmp_path / "ckpt" / "model")
  mt.save(base, nshards=nshards)
  # ---- independent parse
  seen = {"a_sgd": {}, "b_multi": {}}
  for sh in range(nshards):
    meta = P.unframe(open("%s.meta-%05d-of-%05d" % (base, sh, nshards), "rb").read())
    data = P.unframe(P.read_tf_snappy(open("%s-%05d-of-%05d" % (base, sh, nshards), "rb").read()))
    p = 0
    names = []
    for m in meta:
      md = P.MultiHashTableMetadata.FromString(m)
      names.append(md.table_name)
      for rec in data[p:p + md.num_entries]:
        e = P.EntryDump.FromString(rec)
        assert e.id not in seen[md.table_name]
        seen[md.table_name][e.id] = e
      p += md.num_entries
    assert names == ["a_sgd", "b_multi"] and p == len(data)   # sorted-name order, nothing left over
  d_ids, _, d_ts, d_rows = mt.dump("b_multi")
  d_ids, d_ts, d_rows = d_ids.cpu().numpy(), d_ts.cpu().numpy(), d_rows.cpu().numpy()
  assert len(seen["b_multi"]) == ids.size == d_ids.size + 1   # (dump() leaves the side-slot key out)
  for k, i in enumerate(d_ids.tolist()):
    e = seen["b_multi"][i]
    row = d_rows[k]
    np.testing.assert_array_equal(np.array(e.num, np.float32), row[:12])
    assert e.last_update_ts_sec == int(d_ts[k]) == 1_700_000_002
    assert [d.WhichOneof("type") for d in e.opt.dump] == ["ftrl", "adagrad", "sgd"]
    np.testing.assert_array_equal(np.array(e.opt.dump[0].ftrl.norm, np.float32), row[12:13])
    np.testing.assert_array_equal(np.array(e.opt.dump[0].ftrl.zero, np.float32), row[13:14])
    np.testing.assert_array_equal(np.array(e.opt.dump[1].adagrad.norm, np.float32), row[14:22])
  assert np.iinfo(np.int64).min in seen["b_multi"]
  # ---- restore into a fresh table, then both continue training identically
  mt2 = make(cfgs)
  mt2.restore(base)
  assert mt2.size("b_multi") == ids.size and mt2.size("a_sgd") == ids.size // 2
  probe = {"b_multi": ids_t(ids), "a_sgd": ids_t(ids)}
  a, b = mt.lookup(probe), mt2.lookup(probe)
  for k in probe:
    np.testing.assert_array_equal(a[k].cpu().numpy(), b[k].cpu().numpy())
  g = val_t(rng.standard_normal((ids.size, 12)).astype(np.float32))
  for m_ in (mt, mt2):
    m_.apply_gradients({"b_multi": (ids_t(ids), g)}, req_time=1_700_000_010)
  np.testing.assert_array_equal(mt.lookup(probe)["b_multi"].cpu().numpy(),
                                mt2.lookup(probe)["b_multi"].cpu().numpy())
  r1, r2 = mt.dump("b_multi"), mt2.dump("b_multi")
  o1, o2 = np.argsort(r1[0].cpu().numpy()), np.argsort(r2[0].cpu().numpy())
  np.testing.assert_array_equal(r1[3].cpu().numpy()[o1], r2[3].cpu().numpy()[o2])   # state too
  np.testing.assert_array_equal(r1[2].cpu().numpy()[o1], r2[2].cpu().numpy()[o2])   # timestamps


def test_restore_reads_a_foreign_writer_and_save_expires_rows(tmp_path):
  """(1) A checkpoint written by another implementation — here the Python writer of
  tests/ckpt_proto.py: protobuf-runtime EntryDumps, snappy WITH copy elements — restores;
  (2) Save drops rows whose age reaches their slot's TTL relative to the table's max update time
  (multi_hash_table_save_restore_ops.cc:203-211; embedding_hash_table_test.h:282-326 values)."""
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  import ckpt_proto as P
  base = str(tmp_path / "foreign")
  ents = []
  for i in range(300):
    e = P.EntryDump()
    e.id = 1000 + i
    e.num.extend([float(i), float(-i)])
    e.opt.dump.add().adagrad.norm.extend([0.1 + i, 0.2 + i])
    if i % 2:
      e.last_update_ts_sec = 500 + i
    ents.append(e.SerializeToString())
  md = P.MultiHashTableMetadata(table_name="emb", num_entries=len(ents)).SerializeToString()
  other = P.MultiHashTableMetadata(table_name="zzz_unknown", num_entries=2).SerializeToString()
  raw = b"".join(P.frame(r) for r in ents + ents[:2])
  open(base + "-00000-of-00001", "wb").write(P.write_tf_snappy(raw, block=4096))
  open(base + ".meta-00000-of-00001", "wb").write(P.frame(md) + P.frame(other))
  mt = make({"emb": adagrad_cfg(2, 0.1, 0.1)})
  mt.restore(base)
  assert mt.size("emb") == 300
  d_ids, _, d_ts, d_rows = mt.dump("emb")
  o = np.argsort(d_ids.cpu().numpy())
  rows = d_rows.cpu().numpy()[o]
  i = np.arange(300, dtype=np.float32)
  np.testing.assert_array_equal(rows, np.stack([i, -i, 0.1 + i, 0.2 + i], 1).astype(np.float32))
  np.testing.assert_array_equal(d_ts.cpu().numpy()[o], np.where(np.arange(300) % 2, 500 + np.arange(300), 0))
  # ---- expiry on save
  day = 86400
  ttl = entry.SlotExpireTimeConfig(default_expire_time=14, slot_expire_times={1: 5, 2: 6})
  cfg = entry.make_table_config(
      [entry.CombineAsSegment(1, entry.ZerosInitializer(), entry.SgdOptimizer(1.0))],
      slot_expire_time_config=ttl)
  t = make({"t": cfg})
  t0 = 1_600_000_000
  fids = [(1 << 48) | 123, (2 << 48) | 456, 789]
  t.assign_add({"t": (ids_t(fids), val_t([[1.0], [2.0], [3.0]]))}, req_time=t0)
  t.assign_add({"t": (ids_t([55]), val_t([[9.0]]))}, req_time=t0 + 5 * day + 60)   # moves max_update_ts
  b2 = str(tmp_path / "exp")
  t.save(b2, nshards=1)
  recs = P.unframe(P.read_tf_snappy(open(b2 + "-00000-of-00001", "rb").read()))
  kept = sorted(P.EntryDump.FromString(r).id for r in recs)
  assert kept == sorted([(2 << 48) | 456, 789, 55])      # slot 1 (TTL 5 d) expired, slot 2 / default kept
  assert P.MultiHashTableMetadata.FromString(
      P.unframe(open(b2 + ".meta-00000-of-00001", "rb").read())[0]).num_entries == 3


# =============================================================================== full-size properties
@pytest.mark.parametrize("dim,opt,universe", [(64, "adagrad", 10**9), (32, "sgd", 10**8)])
def test_full_batch_pipelined_step_configs(dim, opt, universe):
  """BASELINE.json configs[2] (1B ids, dim 64, fused Adagrad) and configs[1] (100M ids, dim 32,
  SGD) at the full batch of 65 536 Zipf(1.2) ids through the pipelined two-launch step (the bench
  path): forward rows, unique counts, final rows and table size against the oracle (1e-5; lists of
  <= 32 occurrences bit-exact), and a second run is bit-identical."""
  B, steps = 65536, 3
  batches = [S.id_batch(s_, B, universe, "zipf") for s_ in range(steps + 1)]
  dev = [ids_t(b) for b in batches]
  lr = 0.001 if opt == "adagrad" else 0.01
  def cfg():
    return {"emb": (adagrad_cfg(dim, lr, 0.1, reserve_rows=1 << 20, initial_capacity=1 << 21)
                    if opt == "adagrad" else
                    sgd_cfg(dim, lr, reserve_rows=1 << 20, initial_capacity=1 << 21))}
  ot = O.Table(O.segment(dim, O.OPT_ADAGRAD if opt == "adagrad" else O.OPT_SGD, p=(0.1, 0.0)), 1)
  runs = []
  for rep in range(2):
    mt = make(cfg())
    step = SparseStep(mt, "emb", B)
    for s_ in range(steps):
      g = S.grad_batch(s_, B, dim)
      emb = step.forward(dev[s_], next_ids=dev[s_ + 1])
      if rep == 0:
        np.testing.assert_allclose(emb.cpu().numpy(), ot.lookup(batches[s_])[0], rtol=RTOL_TREE, atol=ATOL_TREE)
      step.backward(val_t(g), S.update_time(s_))
      if rep == 0:
        uk = _oracle_step(ot, batches[s_], g, dim, lr, S.update_time(s_))
        assert step.n_unique() == uk.size
    probe = np.unique(np.concatenate(batches[:steps]))
    runs.append(mt.lookup({"emb": ids_t(probe)})["emb"].cpu().numpy())
    assert mt.size("emb") == probe.size == ot.size()
  exp = ot.lookup(probe)[0]
  np.testing.assert_allclose(runs[0], exp, rtol=RTOL_TREE, atol=ATOL_TREE)
  np.testing.assert_array_equal(runs[0], runs[1])
  cnt_max = {}
  for b in batches[:steps]:
    u, c = np.unique(b, return_counts=True)
    for k_, v in zip(u.tolist(), c.tolist()):
      cnt_max[k_] = max(cnt_max.get(k_, 0), v)
  light = np.array([cnt_max[k_] <= 32 for k_ in probe.tolist()])
  np.testing.assert_array_equal(runs[0][light], exp[light])


def test_full_batch_zipf_step_properties_d64_adagrad():
  """BASELINE.json configs[2] shape: dim 64, Adagrad, Zipf(1.2) over 1e9 ids, batch 65536."""
  B, D_ = 65536, 64
  mt = make({"emb": adagrad_cfg(D_, 0.001, 0.1, reserve_rows=1 << 20, initial_capacity=1 << 21)})
  step = SparseStep(mt, "emb", B)
  ot = O.Table(O.segment(D_, O.OPT_ADAGRAD, p=(0.1, 0.0)), 1)
  seen = set()
  for s in range(3):
    ids = S.id_batch(s, B, 10**9, "zipf")
    g = S.grad_batch(s, B, D_)
    emb = step.forward(ids_t(ids))
    # oracle: same step
    uk, _, vo, vos, _ = O.unique_key_with_value_and_offset(ids, [0, B], [D_])
    np.testing.assert_allclose(emb.cpu().numpy(), ot.lookup(ids)[0], rtol=RTOL_TREE, atol=ATOL_TREE)
    gu = O.fill_with_offset_map_gradient(np.arange(uk.size), [0, uk.size], g.ravel(), vo, vos,
                                         [D_]).reshape(-1, D_)
    ot.optimize(uk, gu, [0.001], S.update_time(s))
    step.backward(val_t(g), S.update_time(s))
    seen.update(ids.tolist())
    assert step.n_unique() == uk.size
  assert mt.size("emb") == len(seen) == ot.size()
  allids = np.fromiter(seen, dtype=np.int64)
  got = mt.lookup({"emb": ids_t(allids)})["emb"].cpu().numpy()
  np.testing.assert_allclose(got, ot.lookup(allids)[0], rtol=RTOL_TREE, atol=ATOL_TREE)
  # ids never seen are zero rows and are not inserted by lookups
  absent = np.setdiff1d(allids[:1000] ^ (1 << 40), allids)
  miss = mt.lookup({"emb": ids_t(absent)})["emb"]
  assert not miss.any().item()
  assert mt.size("emb") == len(seen)


# =============================================================================== ADVICE r1 regressions
def test_pipelined_step_restart_flushes_deferred_ids():
  """A pipeline that ends (next_ids=None) on a nearly full table leaves ids for the displacement
  pass; the pipeline is then restarted with other batches, in the same slot and through an id
  buffer refilled in place.  The deferred updates must land (they read the slot's buffers, which the
  restart rewrites) and a refilled buffer must not be taken for the batch deduplicated ahead."""
  cap, dim, n = 1 << 13, 8, 2400
  mt = make({"a": adagrad_cfg(dim, 0.1, 0.1, initial_capacity=cap, max_load_factor=0.97)})
  ot = O.Table(O.segment(dim, O.OPT_ADAGRAD, p=(0.1, 0.0)), cap)
  step = SparseStep(mt, "a", n, exact_order=True)
  rng = np.random.default_rng(31)

  def batch():
    ids = rng.integers(1, 2**60, n)
    ids[n // 2:] = ids[:n - n // 2]
    return ids

  seen = set()
  t = [0]

  def train(dev_ids, host_ids, nxt):
    g = S.grad_batch(t[0], n, dim)
    emb = step.forward(dev_ids, next_ids=nxt)
    np.testing.assert_array_equal(emb.cpu().numpy(), ot.lookup(host_ids)[0])
    step.backward(val_t(g), S.update_time(t[0]))
    _oracle_step(ot, host_ids, g, dim, 0.1, S.update_time(t[0]))
    seen.update(host_ids.tolist())
    t[0] += 1

  b = [batch() for _ in range(6)]
  d = [ids_t(x) for x in b]
  train(d[0], b[0], d[1])
  train(d[1], b[1], None)          # pipeline ends: deferred ids of this update are outstanding
  train(d[2], b[2], d[3])          # restart: dedups into the slot the deferred pass still reads
  buf = d[3]                       # deduplicated ahead ...
  buf.copy_(d[4])                  # ... then refilled in place with another batch
  train(buf, b[4], None)
  train(d[5], b[5], None)
  st = mt.stats("a")
  assert st.dropped == 0 and st.hashpower == 11 and st.size > 0.5 * cap
  allids = np.fromiter(seen, dtype=np.int64)
  np.testing.assert_array_equal(mt.lookup({"a": ids_t(allids)})["a"].cpu().numpy(),
                                ot.lookup(allids)[0])
  assert mt.size("a") == ot.size()


def test_pipeline_picks_up_slices_of_a_resident_id_array():
  """bench.py trains `ids_all[s]` with `next_ids=ids_all[s + 1]`: every slice is a NEW view object
  of the same memory.  The batch deduplicated ahead must be recognised by memory + version, not
  by Python identity — otherwise every step deduplicates twice (the r2 regression: 34 -> 59 us
  per step, invisible to the parity tests).  Exactly two launches per step."""
  n, dim = 4096, 16
  mt = make({"a": adagrad_cfg(dim, 0.1, 0.1, initial_capacity=1 << 15)})
  step = SparseStep(mt, "a", n)
  ids_all = ids_t(np.stack([S.id_batch(s, n, 50000, "zipf") for s in range(8)]))
  g = val_t(S.grad_batch(0, n, dim))
  for s in range(3):
    step.forward(ids_all[s], next_ids=ids_all[s + 1])
    step.backward(g, S.update_time(s))
  torch.cuda.synchronize()
  _lib.profile_arm(32)
  for s in range(3, 6):
    step.forward(ids_all[s], next_ids=ids_all[s + 1])
    step.backward(g, S.update_time(s))
  torch.cuda.synchronize()
  tags = [t for t, _ in _lib.profile_read()]
  assert tags == ["step_fwd_kernel", "step_bwd_kernel"] * 3, tags


def test_c_loop_enqueues_the_same_steps_as_the_python_calls():
  """bench.py's `eager_cpp` window: K pipelined steps enqueued by a plain C loop over the C ABI
  (csrc/eager_loop.c, gcc, public header only) — the same two launches per step and the same table
  afterwards as the Python calls, and the Python pipeline continues behind it."""
  n, dim, steps = 4096, 16, 7
  ids_all = ids_t(np.stack([S.id_batch(s, n, 50000, "zipf") for s in range(steps + 2)]))
  pool = [val_t(S.grad_batch(s, n, dim)) for s in range(3)]
  mts = [make({"a": adagrad_cfg(dim, 0.1, 0.1, initial_capacity=1 << 15)}) for _ in range(2)]
  st = [SparseStep(mt, "a", n, exact_order=True) for mt in mts]
  for s in range(steps):          # reference: one Python call pair per step
    st[0].forward(ids_all[s], next_ids=ids_all[s + 1])
    st[0].backward(pool[s % 3], S.update_time(s))
  st[1].forward(ids_all[0], next_ids=ids_all[1])
  st[1].backward(pool[0], S.update_time(0))
  torch.cuda.synchronize()
  _lib.profile_arm(32)
  st[1].c_loop(ids_all, 1, steps - 1, pool, S.update_time(0))
  torch.cuda.synchronize()
  tags = [t for t, _ in _lib.profile_read()]
  assert tags == ["step_fwd_kernel", "step_bwd_kernel"] * (steps - 2), tags
  st[1].forward(ids_all[steps - 1], next_ids=ids_all[steps])     # picks the pipeline up
  st[1].backward(pool[(steps - 1) % 3], S.update_time(steps - 1))
  probe = ids_t(np.unique(ids_all[:steps].cpu().numpy()))
  a = mts[0].lookup({"a": probe})["a"]
  b = mts[1].lookup({"a": probe})["a"]
  assert torch.equal(a, b)
  assert mts[0].size("a") == mts[1].size("a")


def test_restore_rejects_a_stale_shard_set(tmp_path):
  """Two saves under one basename with different shard counts leave two sets of files; the
  reference validates the set it globs (ValidateShardedFiles), so restore must fail instead of
  silently taking the smaller, older set."""
  mt = make({"t": sgd_cfg(4)})
  mt.assign({"t": (ids_t(np.arange(1, 2001)), torch.ones(2000, 4).cuda())})
  base = str(tmp_path / "ck")
  mt.save(base, nshards=2)
  mt.assign({"t": (ids_t(np.arange(2001, 3001)), torch.ones(1000, 4).cuda())})
  mt.save(base, nshards=3)
  mt2 = make({"t": sgd_cfg(4)})
  with pytest.raises(_lib.MhteError):
    mt2.restore(base)
  for f in os.listdir(tmp_path):
    if f.endswith("-of-00002"):
      os.remove(os.path.join(tmp_path, f))
  mt2.restore(base)
  assert mt2.size("t") == 3000
  with pytest.raises(_lib.MhteError):
    mt2.restore(str(tmp_path / "absent"))


def test_two_doublings_of_a_2p24_slot_table_mid_pipeline():
  """A table of 2^24 slots filled to just under its load limit doubles while a pipelined step is
  in flight — twice (2^24 -> 2^25 -> 2^26 slots, 17 M keys): every prefilled row survives both
  re-hashes with its value, and the rows the pipelined steps trained around each doubling equal
  the oracle's (sequential sums, bit-exact)."""
  dim, n, lr = 4, 65536, 0.5
  cap = 1 << 24
  mt = make({"a": sgd_cfg(dim, lr, initial_capacity=cap, reserve_rows=18_500_000)})
  step = SparseStep(mt, "a", n, exact_order=True)
  ot = O.Table(O.segment(dim, O.OPT_SGD), 1)
  rng = np.random.default_rng(77)
  filled = [0]

  def prefill(upto):
    while filled[0] < upto:
      m = min(1 << 21, upto - filled[0])
      ids = torch.arange(filled[0] + 1, filled[0] + 1 + m, dtype=torch.int64, device="cuda") | (3 << 48)
      vals = (ids % 97).to(torch.float32).unsqueeze(1).expand(m, dim).contiguous()
      rg = mt.get_ragged_id({"a": ids})
      _lib.check(mt._lib.mhte_assign(mt.handle, _lib.vp(ids),
                                     rg.row_splits.ctypes.data_as(_lib.C.POINTER(_lib.C.c_int64)),
                                     _lib.C.c_int64(2), _lib.vp(vals), _lib.C.c_int64(vals.numel()),
                                     _lib.C.c_int64(0), _lib.C.c_int32(_lib.MHTE_IDS_UNIQUE), None))
      filled[0] += m

  def fresh_batch():
    ids = rng.integers(1, 2**44, n).astype(np.int64) | (5 << 48)
    ids[n // 2:] = ids[:n - n // 2]      # every id twice
    return ids

  trained = []
  t = [0]

  def pipeline(k):
    b = [fresh_batch() for _ in range(k + 1)]
    d = [ids_t(x) for x in b]
    for s_ in range(k):
      g = S.grad_batch(t[0], n, dim)
      emb = step.forward(d[s_], next_ids=d[s_ + 1] if s_ + 1 < k else None)
      np.testing.assert_array_equal(emb.cpu().numpy(), ot.lookup(b[s_])[0])
      step.backward(val_t(g), S.update_time(t[0]))
      _oracle_step(ot, b[s_], g, dim, lr, S.update_time(t[0]))
      trained.append(b[s_])
      t[0] += 1

  limit = cap // 2
  prefill(limit - 3 * (n // 2))            # three steps of new ids below the limit
  assert mt.stats("a").hashpower == 22
  pipeline(6)                               # crosses it: first doubling, mid-pipeline
  assert mt.stats("a").hashpower == 23
  prefill(2 * limit - 9 * (n // 2) - 3 * (n // 2))
  pipeline(6)                               # second doubling
  st = mt.stats("a")
  assert st.hashpower == 24 and st.dropped == 0
  probe = np.unique(np.concatenate(trained))
  np.testing.assert_array_equal(mt.lookup({"a": ids_t(probe)})["a"].cpu().numpy(), ot.lookup(probe)[0])
  sample = rng.integers(1, filled[0] + 1, 200000).astype(np.int64)
  got = mt.lookup({"a": ids_t(sample | (3 << 48))})["a"].cpu().numpy()
  np.testing.assert_array_equal(
      got, np.repeat(((sample | (3 << 48)) % 97).astype(np.float32)[:, None], dim, 1))
  assert mt.size("a") == filled[0] + probe.size


# =============================================================================== TTL / expire-on-save KATs
def _ttl_table(dim, expire_time, lr=1.0):
  """test_utils / hash_table_ops_test.py test_hash_table(dim_size, expire_time): SGD lr 1."""
  return make({"t": entry.make_table_config(
      [entry.CombineAsSegment(dim, entry.ZerosInitializer(), entry.SgdOptimizer(lr))],
      slot_expire_time_config=entry.SlotExpireTimeConfig(default_expire_time=expire_time))})


@pytest.mark.parametrize("how", ["assign_add", "apply_gradients"])
def test_save_restore_with_feature_eviction(how, tmp_path):
  """hash_table_ops_test.py:294-379: rows older than the TTL relative to the table's max update time
  are dropped by Save; [[0],[2],[3]] after assign_add, [[0],[-2],[-3]] after apply_gradients."""
  t = _ttl_table(1, 1)
  max_ts = 10000000
  sec = 24 * 3600
  for i, ts in ((1, max_ts - sec - 1), (2, max_ts - sec + 1), (3, max_ts)):
    if how == "assign_add":
      t.assign_add({"t": (ids_t([i]), val_t([[float(i)]]))}, req_time=ts)
    else:
      t.apply_gradients({"t": (ids_t([i]), val_t([[float(i)]]))}, req_time=ts)
  base = str(tmp_path / how / "table")
  t.save(base)
  t2 = make({"t": sgd_cfg(1)})
  t2.restore(base)
  sign = 1.0 if how == "assign_add" else -1.0
  assert t2.lookup({"t": ids_t([1, 2, 3])})["t"].cpu().tolist() == [[0.0], [sign * 2], [sign * 3]]


@pytest.mark.parametrize("ttl,expect", [(0, [[0.0], [0.0]]), (3600, [[1.0], [2.0]])])
def test_entry_ttl_zero_and_not_zero(ttl, expect, tmp_path):
  """hash_table_ops_test.py:381-411 (ids -1 and 1, default req_time 0)."""
  t = _ttl_table(1, ttl)
  t.assign_add({"t": (ids_t([-1, 1]), val_t([[1.0], [2.0]]))})
  base = str(tmp_path / "ttl" / "table")
  t.save(base)
  t2 = _ttl_table(1, 36500)
  t2.restore(base)
  assert t2.lookup({"t": ids_t([-1, 1])})["t"].cpu().tolist() == expect


def test_entry_ttl_by_slots(tmp_path):
  """hash_table_ops_test.py:413-466: slot 1 expires at once (TTL 0), slot 2 after a day; the
  restored table saves and restores again to the same rows."""
  ttl = entry.SlotExpireTimeConfig(default_expire_time=3600, slot_expire_times={1: 0, 2: 1})
  cfg = lambda: {"t": entry.make_table_config(
      [entry.CombineAsSegment(1, entry.ZerosInitializer(), entry.SgdOptimizer(1.0))],
      slot_expire_time_config=ttl)}
  id_1, id_2 = 1 << 48, 2 << 48
  t = make(cfg())
  t.assign_add({"t": (ids_t([id_1, id_2]), val_t([[1.0], [2.0]]))}, req_time=100)
  base = str(tmp_path / "slots" / "table")
  t.save(base)
  t2 = make(cfg())
  t2.restore(base)
  assert t2.lookup({"t": ids_t([id_1, id_2])})["t"].cpu().tolist() == [[0.0], [2.0]]
  base2 = str(tmp_path / "slots" / "table_new")
  t2.save(base2)
  t3 = _ttl_table(1, 36500)
  t3.restore(base2)
  assert t3.lookup({"t": ids_t([id_1, id_2])})["t"].cpu().tolist() == [[0.0], [2.0]]


def test_workspace_of_a_closed_table_does_not_touch_its_successor():
  """A pipelined step leaves the NEXT batch numbered, with rows reserved in its table (the build
  role's probe).  Dropping that workspace gives the reservations back — to ITS table only: a table
  created after the first was closed may get the same device counter block, and must not be charged
  (seen in scripts/next_rows_bench.py as a size of keys - 2^32)."""
  import gc
  n, dim = 4096, 16
  a, b = S.id_batch(1, n, 10**6, "uniform"), S.id_batch(2, n, 10**6, "uniform")
  for _ in range(3):   # (a few rounds: the allocator decides whether the address repeats)
    mt1 = make({"emb": adagrad_cfg(dim, 0.01, 0.1)})
    step1 = SparseStep(mt1, "emb", n)
    step1.forward(ids_t(a), next_ids=ids_t(b))
    step1.backward(val_t(S.grad_batch(0, n, dim)), S.update_time(0))   # batch b: numbered, rows reserved
    assert mt1.size("emb") == np.unique(a).size
    torch.cuda.synchronize()
    mt1.close()
    mt2 = make({"emb": adagrad_cfg(dim, 0.01, 0.1)})
    del step1          # the old workspace goes after the new table exists
    gc.collect()
    assert mt2.size("emb") == 0
    step2 = SparseStep(mt2, "emb", n)
    step2.forward(ids_t(a), next_ids=ids_t(b))
    step2.backward(val_t(S.grad_batch(0, n, dim)), S.update_time(0))
    assert mt2.size("emb") == np.unique(a).size
    del step2
    gc.collect()
    assert mt2.size("emb") == np.unique(a).size   # (its own dropped batch gave its reservations back)
    mt2.close()


@pytest.mark.parametrize("exact", [False, True])
def test_clear_between_numbering_and_update_voids_the_probe(exact):
  """Clear() (what a restore starts with) resets the row allocator.  A batch that was numbered and
  probed BEFORE it carries row handles of the old numbering — hints, and rows reserved for the ids the
  table lacked; applied with them it would write into rows that the refilled table has handed to
  other ids.  The step notices (the table's serial number moved) and probes again."""
  n, dim = 20000, 16
  b = [S.id_batch(60 + s_, n, 10**5, "zipf") for s_ in range(3)]
  b[1][::3] = b[1][1]   # a heavy list: a work item whose header held a reservation
  dev = [ids_t(x) for x in b]
  mt = make({"emb": adagrad_cfg(dim, 0.01, 0.1)})
  step = SparseStep(mt, "emb", n, exact_order=exact)
  step.forward(dev[0], next_ids=dev[1])
  step.backward(val_t(S.grad_batch(0, n, dim)), S.update_time(0))   # b[1]: numbered, probed, rows reserved
  mt.clear_table("emb")
  fill = np.arange(1, 30001, dtype=np.int64) * 7919 + (1 << 40)    # these get the first row handles now
  vals = np.random.default_rng(3).standard_normal((fill.size, dim)).astype(np.float32)
  mt.assign({"emb": (ids_t(fill), val_t(vals))}, req_time=5)
  ot = O.Table(O.segment(dim, O.OPT_ADAGRAD, p=(0.1, 0.0)), 1)
  ot.assign(fill, vals, 5)
  emb = step.forward(dev[1], next_ids=dev[2])
  np.testing.assert_array_equal(emb.cpu().numpy(), ot.lookup(b[1])[0])
  g = S.grad_batch(1, n, dim)
  step.backward(val_t(g), S.update_time(1))
  _oracle_step(ot, b[1], g, dim, 0.01, S.update_time(1))
  probe = np.unique(np.concatenate([fill, b[1]]))
  got = mt.lookup({"emb": ids_t(probe)})["emb"].cpu().numpy()
  exp = ot.lookup(probe)[0]
  if exact:
    np.testing.assert_array_equal(got, exp)
  else:
    np.testing.assert_allclose(got, exp, rtol=RTOL_TREE, atol=ATOL_TREE)
  np.testing.assert_array_equal(got[np.isin(probe, fill) & ~np.isin(probe, b[1])],
                                exp[np.isin(probe, fill) & ~np.isin(probe, b[1])])   # untouched rows: exact
  assert mt.size("emb") == probe.size == ot.size()
