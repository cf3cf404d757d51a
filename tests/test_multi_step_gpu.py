"""GPU parity (-m gpu) of the multi-table launches (csrc/mhte_mstep_kernels.h) through the C ABI:
  * mhte_multi_step_forward / _backward — every table of a MultiHashTable in one launch pair —
    against the CPU oracle, table by table, on BASELINE.json configs[4]'s shape (26 feature tables
    of dims 16 / 32 / 64);
  * mhte_fused_lookup / mhte_fused_optimize as one launch over the [shard][table] segments against
    the per-table ops and the oracle.
Bars: bit-exact with MHTE_EXACT_ORDER and for ids that occur <= 32 times in a batch; otherwise
|diff| <= 5e-7 + 1e-5 |expected| (measured: 2.1e-7 on the 12 000-occurrence lists of a 65 536-id
Zipf batch, fp32 re-association of the fixed summation tree) (north_star: fp32 rows within 1e-5).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle as O  # noqa: E402
from monolith_amd import _lib, entry, synthetic as S  # noqa: E402
from monolith_amd.fused_step import MultiSparseStep  # noqa: E402
from monolith_amd.multi_hash_table_ops import MultiHashTable, Ragged  # noqa: E402

RTOL, ATOL = 1e-5, 5e-7
_counter = [0]


def _name():
  _counter[0] += 1
  return "ms%d" % _counter[0]


def ids_t(x):
  return torch.as_tensor(np.asarray(x, dtype=np.int64)).cuda()


def val_t(x):
  return torch.as_tensor(np.asarray(x, dtype=np.float32)).cuda()


class Spec:
  """One feature table: entry config for the engine + the same thing for the oracle."""

  def __init__(self, name, segs, slot, **kw):
    # segs: [(dim, "sgd"|"adagrad"|"ftrl", lr)]
    self.name, self.segs, self.slot = name, segs, slot
    self.dim = sum(d for d, _, _ in segs)
    self.kw = kw

  def entry_cfg(self):
    parts = []
    for d, opt, lr in self.segs:
      o = {"sgd": lambda: entry.SgdOptimizer(lr),
           "adagrad": lambda: entry.AdagradOptimizer(lr, 0.1),
           "ftrl": lambda: entry.FtrlOptimizer(lr, 0.1, 1.0, l1_regularization=0.001,
                                                   l2_regularization=0.001),
           # GroupAdaGrad: the one whole-segment optimizer (group_adagrad_optimizer.cc)
           "group": lambda: entry.AdaGradWithGroupLassoOptimizer(lr, beta=1.0, initial_accumulator_value=0.1,
                                                                 l2_regularization=0.001)}[opt]()
      parts.append(entry.CombineAsSegment(d, entry.ZerosInitializer(), o))
    ttl = getattr(self, "ttl_days", None)
    return entry.make_table_config(
        parts, entry.CuckooHashTableConfig(**self.kw),
        slot_expire_time_config=None if ttl is None else entry.SlotExpireTimeConfig(default_expire_time=ttl))

  def oracle_table(self):
    segs = []
    for d, opt, _ in self.segs:
      if opt == "sgd":
        segs.append(O.segment(d, O.OPT_SGD))
      elif opt == "adagrad":
        segs.append(O.segment(d, O.OPT_ADAGRAD, p=(0.1, 0.0)))
      elif opt == "group":
        segs.append(O.segment(d, O.OPT_GROUP_ADAGRAD, p=(0.1, 1.0, 0.001, 0.0)))
      else:
        segs.append(O.segment(d, O.OPT_FTRL, p=(0.1, 1.0, 0.001, 0.001)))
    t = O.Table(segs if len(segs) > 1 else segs[0], int(self.kw.get("initial_capacity", 1)))
    if getattr(self, "ttl_days", None) is not None:
      t.set_ttl(self.ttl_days)
    return t

  def lrs(self):
    return [lr for _, _, lr in self.segs]


def dlrm_specs(n_tables=26, **kw):
  """configs[4]: 26 feature tables, dims cycling 16 / 32 / 64, mostly Adagrad."""
  specs = []
  for i in range(n_tables):
    d = (16, 32, 64)[i % 3]
    if i == 5:
      segs = [(d, "sgd", 0.01)]
    elif i == 7:
      segs = [(4, "ftrl", 0.05), (d - 4, "adagrad", 0.01)]   # bias FTRL + vector Adagrad
    elif i == 11:
      segs = [(d, "ftrl", 0.05)]
    else:
      segs = [(d, "adagrad", 0.01)]
    specs.append(Spec("f%02d" % i, segs, i + 1, **kw))
  return specs


def make(specs):
  return MultiHashTable.from_configs({s.name: s.entry_cfg() for s in specs}, name_suffix=_name())


def ragged_of(specs, mt, per_table_ids):
  return mt.get_ragged_id({s.name: ids_t(per_table_ids[s.name]) for s in specs
                           if per_table_ids.get(s.name) is not None})


def oracle_backward(ot, spec, ids, g):
  n = ids.size
  if n == 0:
    return np.zeros(0, np.int64)
  uk, _, vo, vos, _ = O.unique_key_with_value_and_offset(ids, [0, n], [spec.dim])
  gu = O.fill_with_offset_map_gradient(np.arange(uk.size), [0, uk.size], g.ravel(), vo, vos,
                                       [spec.dim]).reshape(-1, spec.dim)
  return uk, gu


def run_pipeline(specs, batches, grads, exact, sizes=None, prefetch=True):
  """batches[s][name] -> ids; returns (mt, oracle tables, per-step forward outputs checked)."""
  mt = make(specs)
  B = max(max((b[s.name].size for s in specs if b.get(s.name) is not None), default=1) for b in batches)
  step = MultiSparseStep(mt, B, exact_order=exact)
  ots = {s.name: s.oracle_table() for s in specs}
  by_name = sorted(specs, key=lambda s: s.name)
  rag = [ragged_of(specs, mt, b) for b in batches]
  for s_ in range(len(batches) - 1):
    nxt = rag[s_ + 1] if (prefetch and s_ + 2 < len(batches)) else None
    emb = step.forward(rag[s_], nxt)
    views = mt.get_embeddings(rag[s_], emb)
    uc = step.unique_counts()
    flat_g = []
    for k, sp in enumerate(by_name):
      ids = batches[s_].get(sp.name)
      ids = np.zeros(0, np.int64) if ids is None else ids
      exp = ots[sp.name].lookup(ids)[0] if ids.size else np.zeros((0, sp.dim), np.float32)
      got = views[sp.name].cpu().numpy()
      if exact:
        np.testing.assert_array_equal(got, exp, err_msg="forward %s step %d" % (sp.name, s_))
      else:
        np.testing.assert_allclose(got, exp, rtol=RTOL, atol=ATOL)
      assert uc[k] == np.unique(ids).size, (sp.name, s_)
      if ids.size:
        g = grads(s_, sp, ids.size)
        flat_g.append(g.ravel())
        uk, gu = oracle_backward(ots[sp.name], sp, ids, g)
        ots[sp.name].optimize(uk, gu, sp.lrs(), S.update_time(s_))
    step.backward(val_t(np.concatenate(flat_g)), S.update_time(s_))
  return mt, ots, step


def check_final(mt, ots, specs, batches, exact, light_exact=True):
  for sp in specs:
    seen = [b[sp.name] for b in batches[:-1] if b.get(sp.name) is not None and b[sp.name].size]
    if not seen:
      assert mt.size(sp.name) == 0
      continue
    probe = np.unique(np.concatenate(seen))
    got = mt.lookup({sp.name: ids_t(probe)})[sp.name].cpu().numpy()
    exp = ots[sp.name].lookup(probe)[0]
    assert mt.size(sp.name) == probe.size == ots[sp.name].size(), sp.name
    if exact:
      np.testing.assert_array_equal(got, exp, err_msg=sp.name)
      continue
    np.testing.assert_allclose(got, exp, rtol=RTOL, atol=ATOL, err_msg=sp.name)
    if light_exact:
      cnt_max = {}
      for b in seen:
        u, c = np.unique(b, return_counts=True)
        for k_, v in zip(u.tolist(), c.tolist()):
          cnt_max[k_] = max(cnt_max.get(k_, 0), v)
      light = np.array([cnt_max[k_] <= 32 for k_ in probe.tolist()])
      np.testing.assert_array_equal(got[light], exp[light], err_msg=sp.name)


def zipf_batches(specs, steps, B, universe):
  return [{sp.name: S.id_batch(s_ * 131 + sp.slot, B, universe, "zipf", feature_slot=sp.slot)
           for sp in specs} for s_ in range(steps + 1)]


def seeded_grads(s_, sp, n):
  rng = np.random.Generator(np.random.PCG64(S.SEED0 + 10**9 + s_ * 1000 + sp.slot))
  return rng.standard_normal((n, sp.dim), dtype=np.float32) * np.float32(0.01)


# ============================================================== the reference's standard row layout
def bias_slice_specs():
  """NT/feature.py:117-120 (FeatureSlot(has_bias=True): a dim-1 FTRL bias slice in front of the
  vector) as NT/distributed_ps_test.py:480-505 builds it: FTRL(1) + Adagrad(16) and FTRL(1) +
  Adagrad(32) rows — dims 17 / 33, not whole float4s — beside a plain dim-64 table whose slice of the
  flat buffers then does NOT start on a 16-byte boundary, and a dim-16 one that does."""
  return [Spec("a_bias16", [(1, "ftrl", 0.05), (16, "adagrad", 0.01)], 1),
          Spec("b_plain64", [(64, "adagrad", 0.01)], 2),
          Spec("c_bias32", [(1, "ftrl", 0.05), (32, "adagrad", 0.01)], 3),
          Spec("d_plain16", [(16, "adagrad", 0.01)], 4)]


def group_opt_specs():
  """Tables with the whole-segment optimizer GroupAdaGrad beside plain ones: a float4
  row, a bias + group row of 17 floats (one float per lane), a row of two group segments."""
  return [Spec("a_group16", [(16, "group", 0.02)], 1),
          Spec("b_plain32", [(32, "adagrad", 0.01)], 2),
          Spec("c_bias_group", [(1, "ftrl", 0.05), (16, "group", 0.02)], 3),
          Spec("d_two_groups", [(8, "group", 0.02), (24, "group", 0.01)], 4),
          Spec("e_plain64", [(64, "adagrad", 0.01)], 5)]


@pytest.mark.parametrize("exact", [True, False])
def test_multi_step_bias_slice_rows(exact):
  """Rows with a dim-1 bias segment run through mhte_multi_step_* (round 3 rejected them): one float
  per lane for those tables (and for a float4 table whose slice is pushed off its 16-byte boundary by
  an odd-dim neighbour), float4 lanes for the rest, in the SAME launches; bit-exact vs the oracle."""
  specs = bias_slice_specs()
  B, steps = 4099, 4     # (odd batch: b_plain64's slice starts at 4099 * 17 floats)
  batches = zipf_batches(specs, steps, B, 3000)
  mt, ots, step = run_pipeline(specs, batches, seeded_grads, exact)
  check_final(mt, ots, specs, batches, exact)
  step.close()


def test_multi_step_more_than_32_tables():
  """kMaxStepTables = 32 is a per-launch budget (kernel arguments), not a model limit: 40 tables run
  as two chunks per stage."""
  specs = [Spec("t%02d" % i, [((16, 32)[i % 2], "adagrad", 0.01)], i + 1) for i in range(40)]
  batches = zipf_batches(specs, 3, 1024, 500)
  mt, ots, step = run_pipeline(specs, batches, seeded_grads, True)
  check_final(mt, ots, specs, batches, True)
  step.close()


def test_a_model_whose_flat_layout_can_misalign_a_wide_table_is_refused_at_creation():
  """a table of more than 64 floats per row needs float4 lanes, i.e. its slice of the
  flat embedding / gradient buffer on a 16-byte boundary — which a table of odd dim IN FRONT of it (sorted
  by name) breaks for the batch sizes where n x dim is not a multiple of four.  Round 4 threw from the
  middle of such a step (in the sharded step: after the exchanges were enqueued).  Now the model is
  refused when the step is created; the same tables in the other order are fine, with odd batch sizes."""
  from monolith_amd import _lib as L
  from monolith_amd.distributed_ps_sync import ShardedMultiStep
  bad = [Spec("a_bias16", [(1, "ftrl", 0.05), (16, "adagrad", 0.01)], 1), Spec("b_wide128", [(128, "adagrad", 0.01)], 2)]
  mt = make(bad)
  with pytest.raises(L.InvalidArgumentError, match="follows"):
    MultiSparseStep(mt, 1024)
  with pytest.raises(L.InvalidArgumentError, match="follows"):
    ShardedMultiStep(mt, 1024)
  good = [Spec("a_wide128", [(128, "adagrad", 0.01)], 1), Spec("b_bias16", [(1, "ftrl", 0.05), (16, "adagrad", 0.01)], 2)]
  batches = [{"a_wide128": S.id_batch(10 + s, 701, 3000, "zipf", feature_slot=1),
              "b_bias16": S.id_batch(20 + s, 333, 3000, "zipf", feature_slot=2)} for s in range(4)]
  mt, ots, step = run_pipeline(good, batches, seeded_grads, True)
  check_final(mt, ots, good, batches, True)
  step.close()


# =============================================================================== configs[4] shape
@pytest.mark.parametrize("exact", [True, False])
def test_multi_step_26_tables_matches_oracle(exact):
  """26 tables of dims 16 / 32 / 64 (Adagrad, one SGD, one FTRL, one two-segment row), 4096
  Zipf(1.2) ids per table and step — the head ids occur hundreds of times, so every list class
  (single, short, heavy one-item, heavy multi-item) is present — four pipelined steps."""
  specs = dlrm_specs(initial_capacity=1 << 16)
  batches = zipf_batches(specs, 4, 4096, 10**6)
  mt, ots, _ = run_pipeline(specs, batches, seeded_grads, exact)
  check_final(mt, ots, specs, batches, exact)


def test_multi_step_full_batch_26_tables():
  """configs[4] at the full batch: 26 x 65 536 Zipf(1.2) ids per step through one launch pair;
  rows within 1e-5 of the oracle (bit-exact for ids with <= 32 occurrences), sizes equal, and a
  second run is bit-identical (the heavy-list tree depends on positions only)."""
  specs = dlrm_specs(initial_capacity=1 << 19, reserve_rows=1 << 17)
  batches = zipf_batches(specs, 2, 65536, 10**8)
  runs = []
  for rep in range(2):
    mt, ots, _ = run_pipeline(specs, batches, seeded_grads, False)
    if rep == 0:
      check_final(mt, ots, specs, batches, False)
    out = []
    for sp in specs:
      probe = np.unique(np.concatenate([b[sp.name] for b in batches[:-1]]))
      out.append(mt.lookup({sp.name: ids_t(probe)})[sp.name].cpu().numpy())
    runs.append(out)
    mt.close()
  for a, b in zip(*runs):
    np.testing.assert_array_equal(a, b)


# =============================================================================== edge shapes
def test_multi_step_ragged_and_empty_tables():
  """Tables with different batch sizes, one table with no ids at all in some steps, batches of one
  id, the int64 minimum (the key that lives in the side slot)."""
  specs = dlrm_specs(6, initial_capacity=1 << 12)
  rng = np.random.default_rng(5)
  sizes = [1000, 1, 0, 37, 1024, 2049]
  batches = []
  for s_ in range(5):
    b = {}
    for sp, n in zip(specs, sizes):
      n_s = 0 if (sp.name == "f04" and s_ % 2 == 1) else n
      ids = rng.integers(1, 300, n_s).astype(np.int64) | (sp.slot << 48)
      if n_s > 10:
        ids[3] = np.iinfo(np.int64).min
        ids[7] = np.iinfo(np.int64).min
      b[sp.name] = ids
    batches.append(b)
  mt, ots, _ = run_pipeline(specs, batches, seeded_grads, True)
  check_final(mt, ots, specs, batches, True)


def test_multi_step_restart_and_dropped_prefetch():
  """A batch deduplicated ahead and then NOT trained (the caller hands over another one), a
  pipeline that ends (no next batch) and starts again, forward-only steps: every variant must leave
  the same rows as the oracle (restarting a pipeline must not reuse stale state)."""
  specs = dlrm_specs(3, initial_capacity=1 << 10, max_load_factor=0.95)
  mt = make(specs)
  B = 700
  step = MultiSparseStep(mt, B, exact_order=True)
  ots = {s.name: s.oracle_table() for s in specs}
  by_name = sorted(specs, key=lambda s: s.name)
  rng = np.random.default_rng(17)

  def batch():
    return {sp.name: (rng.integers(1, 900, B).astype(np.int64) | (sp.slot << 48)) for sp in specs}

  def train(b, nxt, t):
    r = ragged_of(specs, mt, b)
    emb = step.forward(r, nxt)
    views = mt.get_embeddings(r, emb)
    fg = []
    for sp in by_name:
      np.testing.assert_array_equal(views[sp.name].cpu().numpy(), ots[sp.name].lookup(b[sp.name])[0])
      g = seeded_grads(t, sp, B)
      fg.append(g.ravel())
      uk, gu = oracle_backward(ots[sp.name], sp, b[sp.name], g)
      ots[sp.name].optimize(uk, gu, sp.lrs(), S.update_time(t))
    step.backward(val_t(np.concatenate(fg)), S.update_time(t))

  b0, b1, b2, b3, b4 = (batch() for _ in range(5))
  r1 = ragged_of(specs, mt, b1)
  train(b0, r1, 0)              # b1 deduplicated ahead ...
  train(b2, None, 1)            # ... but b2 is trained: the prefetched dedup is dropped
  train(b3, None, 2)            # pipeline ended, starts again
  r4 = ragged_of(specs, mt, b4)
  step.forward(ragged_of(specs, mt, b1), r4)   # forward only (no backward)
  emb = step.forward(r4, None)                 # prefetched batch picked up, still no backward
  views = mt.get_embeddings(r4, emb)
  for sp in by_name:
    np.testing.assert_array_equal(views[sp.name].cpu().numpy(), ots[sp.name].lookup(b4[sp.name])[0])
  # an id buffer refilled in place must not be taken for the prefetched batch
  r5 = ragged_of(specs, mt, b0)
  step.forward(r4, r5)
  r5.values.copy_(ids_t(np.concatenate([b2[sp.name] for sp in by_name])))
  b5 = dict(b2)
  train_r = step.forward(r5, None)
  views = mt.get_embeddings(r5, train_r)
  fg = []
  for sp in by_name:
    np.testing.assert_array_equal(views[sp.name].cpu().numpy(), ots[sp.name].lookup(b5[sp.name])[0])
    g = seeded_grads(9, sp, B)
    fg.append(g.ravel())
    uk, gu = oracle_backward(ots[sp.name], sp, b5[sp.name], g)
    ots[sp.name].optimize(uk, gu, sp.lrs(), S.update_time(9))
  step.backward(val_t(np.concatenate(fg)), S.update_time(9))
  for sp in specs:
    probe = np.unique(np.concatenate([b[sp.name] for b in (b0, b2, b3)]))
    np.testing.assert_array_equal(mt.lookup({sp.name: ids_t(probe)})[sp.name].cpu().numpy(),
                                  ots[sp.name].lookup(probe)[0])
    assert mt.size(sp.name) == ots[sp.name].size()


def test_multi_step_displacement_and_doubling():
  """Small tables at load 0.97 (ids whose two buckets are full go to the displacement launch in
  every step) and tables that start at capacity 1 and double several times mid-pipeline (the
  device descriptors are re-uploaded)."""
  hi = dlrm_specs(3, initial_capacity=1 << 13, max_load_factor=0.97)
  for sp in hi:
    sp.name = "h" + sp.name
  grow = dlrm_specs(3)  # initial_capacity 1
  for sp in grow:
    sp.slot += 10
  specs = hi + grow
  rng = np.random.default_rng(3)
  n = 2500
  batches = []
  for s_ in range(5):
    b = {}
    for sp in specs:
      ids = rng.integers(1, 2**40, n).astype(np.int64) | (sp.slot << 48)
      ids[n // 2:] = ids[:n - n // 2]
      if s_ > 0:
        ids[:n // 4] = batches[-1][sp.name][:n // 4]
      b[sp.name] = ids
    batches.append(b)
  mt, ots, _ = run_pipeline(specs, batches, seeded_grads, True)
  check_final(mt, ots, specs, batches, True)
  for sp in hi:
    st = mt.stats(sp.name)
    assert st.dropped == 0 and st.hashpower == 11 and st.size > 0.5 * (1 << 13)
    # the displacement pass works on several deferred ids at a time (slowpath_par_role): every key
    # sits in one of ITS two buckets, once
    ids, pos, _, _ = mt.dump(sp.name, with_rows=False)
    ids, pos = ids.cpu().numpy(), pos.cpu().numpy()
    L = O.lib()
    assert ids.size == st.size == np.unique(ids).size
    for k, p_ in zip(ids, pos):
      hv = L.mo_hash(int(k))
      i1 = hv & ((1 << 11) - 1)
      assert (int(p_) >> 2) in (i1, L.mo_alt_index(11, L.mo_partial(hv), i1)), (sp.name, k, p_)
  for sp in grow:
    assert mt.stats(sp.name).hashpower >= 11


def test_hints_do_not_survive_a_table_change_between_forward_and_backward():
  """The forward leaves (row handle, bucket slot) hints for the backward.  An eviction scan, an
  assign that displaces entries or a doubling between the two calls makes them stale
  (Table::mut_epoch): the backward must probe again.  Here half of the looked-up ids expire and are
  evicted, and a burst of assigns doubles the table, after the forward and before the backward."""
  specs = dlrm_specs(4, initial_capacity=1 << 11)
  for sp in specs:
    sp.ttl_days = 1
  by_name = sorted(specs, key=lambda s: s.name)
  mt = make(specs)
  ots = {s.name: s.oracle_table() for s in specs}
  B = 1500
  step = MultiSparseStep(mt, B, exact_order=True)
  rng = np.random.default_rng(11)
  day = 86400
  t0 = S.update_time(0)

  def ids_of(sp, lo, hi):
    return (rng.integers(lo, hi, B).astype(np.int64)) | (sp.slot << 48)

  # step 0 at time t0: ids in [1, 1000); step 1 two days later looks up the same range
  for s_, now in ((0, t0), (1, t0 + 2 * day)):
    batch = {sp.name: ids_of(sp, 1, 1000) for sp in specs}
    rag = ragged_of(specs, mt, batch)
    emb = step.forward(rag, None)
    views = mt.get_embeddings(rag, emb)
    for sp in by_name:
      np.testing.assert_array_equal(views[sp.name].cpu().numpy(), ots[sp.name].lookup(batch[sp.name])[0])
    if s_ == 1:
      # between forward and backward: everything last touched at t0 expires (TTL 1 day) ...
      for sp in by_name:
        mt.evict(sp.name, now)
        ots[sp.name].evict(now)
        # ... and 6 000 fresh ids double the table
        fresh = (np.arange(5000, 11000, dtype=np.int64)) | (sp.slot << 48)
        vals = np.full((fresh.size, sp.dim), 0.5, np.float32)
        mt.assign({sp.name: (ids_t(fresh), val_t(vals))}, req_time=now)
        ots[sp.name].assign(fresh, vals, now)
    flat = []
    for sp in by_name:
      g = seeded_grads(s_, sp, B)
      flat.append(g.ravel())
      uk, gu = oracle_backward(ots[sp.name], sp, batch[sp.name], g)
      ots[sp.name].optimize(uk, gu, sp.lrs(), now)
    step.backward(val_t(np.concatenate(flat)), now)
  for sp in by_name:
    allids = np.concatenate([np.arange(1, 1000, dtype=np.int64), np.arange(5000, 11000, dtype=np.int64)]) | (sp.slot << 48)
    got = mt.lookup({sp.name: ids_t(allids)})[sp.name].cpu().numpy()
    np.testing.assert_array_equal(got, ots[sp.name].lookup(allids)[0], err_msg=sp.name)
    assert mt.size(sp.name) == ots[sp.name].size()
    assert mt.stats(sp.name).evicted > 0


def test_multi_step_errors():
  specs = dlrm_specs(2, initial_capacity=1 << 10)
  mt = make(specs)
  step = MultiSparseStep(mt, 100)
  with pytest.raises(_lib.InvalidArgumentError):   # more ids than the step was created for
    step.forward(mt.get_ragged_id({"f00": ids_t(np.arange(101))}))
  with pytest.raises(_lib.MhteError):              # backward without a forward batch
    step.backward(val_t(np.zeros(16)), 0)
  r = mt.get_ragged_id({"f00": ids_t(np.arange(10))})
  step.forward(r)
  with pytest.raises(_lib.InvalidArgumentError):   # gradient too short
    step.backward(val_t(np.zeros(16)), 0)
  odd = MultiHashTable.from_configs(
      {"w": entry.make_table_config([entry.CombineAsSegment(
          6, entry.ZerosInitializer(), entry.SgdOptimizer(0.1))])}, name_suffix=_name())
  MultiSparseStep(odd, 10).close()    # rows not made of float4s: one float per lane (up to 64 floats)
  wide = MultiHashTable.from_configs(
      {"w": entry.make_table_config([entry.CombineAsSegment(1, entry.ZerosInitializer(), entry.FtrlOptimizer(0.1)),
                                     entry.CombineAsSegment(68, entry.ZerosInitializer(), entry.SgdOptimizer(0.1))])},
      name_suffix=_name())
  with pytest.raises(_lib.InvalidArgumentError):   # ... a row of 69 floats that is not whole float4s does not fit
    MultiSparseStep(wide, 10)


# =============================================================================== fused ops, one launch
@pytest.mark.parametrize("shards,high_load", [(2, False), (8, False), (4, True)])
def test_fused_ops_over_segments_match_per_table_ops(shards, high_load, with_group=False):
  """FusedLookup / FusedOptimize on [shard][table] segments (ids distinct inside a segment, as
  FusedReorderByIndices leaves them): the one-launch segment kernels against the oracle, which
  applies the segments one after the other like the reference's loop."""
  kw = dict(initial_capacity=1 << 12, max_load_factor=0.97) if high_load else dict(initial_capacity=1 << 14)
  specs = dlrm_specs(5, **kw)
  specs[3].segs = [(specs[3].dim, "adagrad", 0.02)]
  if with_group:
    specs = group_opt_specs()
    for sp in specs:
      sp.kw = kw
  mt = make(specs)
  by_name = sorted(specs, key=lambda s: s.name)
  T = len(by_name)
  ots = {s.name: s.oracle_table() for s in specs}
  rng = np.random.default_rng(shards)
  lr_all = [lr for sp in by_name for lr in sp.lrs()]
  mt.set_learning_rate(lr_all)
  for it in range(3):
    # per table unique ids, then bucket them by id % shards (shard-major, table-minor)
    per = {sp.name: np.unique(rng.integers(1, 5000, 1500 if high_load else 900).astype(np.int64) |
                              (sp.slot << 48)) for sp in by_name}
    segs, fss = [], []
    for sh in range(shards):
      for sp in by_name:
        seg = per[sp.name][per[sp.name] % shards == sh]
        segs.append(seg)
        fss.append(seg.size)
    ids = np.concatenate(segs)
    emb, splits, id_off, emb_off, idx = mt.fused_lookup(ids_t(ids), fss, shards)
    emb = emb.cpu().numpy()
    for y, seg in enumerate(segs):
      sp = by_name[y % T]
      exp = ots[sp.name].lookup(seg)[0].ravel()
      np.testing.assert_array_equal(emb[emb_off[y]:emb_off[y + 1]], exp)
      assert id_off[y + 1] - id_off[y] == seg.size
    grads = (rng.standard_normal(emb.size).astype(np.float32) * np.float32(0.05))
    mt.fused_apply_gradient(ids_t(ids), idx, fss, val_t(grads), id_off[:-1], emb_off[:-1],
                            global_step=it, req_time=S.update_time(it), num_of_shards=shards,
                            ids_unique_per_segment=True)
    for y, seg in enumerate(segs):
      sp = by_name[y % T]
      if seg.size:
        ots[sp.name].optimize(seg, grads[emb_off[y]:emb_off[y + 1]].reshape(-1, sp.dim), sp.lrs(),
                              S.update_time(it))
  for sp in by_name:
    ids_, _, ts_, rows_ = mt.dump(sp.name)
    probe = np.sort(ids_.cpu().numpy())
    assert probe.size == ots[sp.name].size()
    np.testing.assert_array_equal(mt.lookup({sp.name: ids_t(probe)})[sp.name].cpu().numpy(),
                                  ots[sp.name].lookup(probe)[0])
    assert mt.stats(sp.name).dropped == 0


@pytest.mark.parametrize("shards,high_load", [(1, False), (3, True)])
def test_fused_optimize_with_whole_segment_optimizer(shards, high_load):
  """GroupAdaGrad tables in FusedOptimize's one-launch segment kernels (round 3: such a model took the
  per-table op-level path): their own kernel instance (seg_upsert_kernel<VW, GROUP>), same launch
  sequence, bit-exact vs the oracle — incl. the displacement pass at a high load factor."""
  test_fused_ops_over_segments_match_per_table_ops(shards, high_load, with_group=True)
