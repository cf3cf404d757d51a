"""TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.

Checkers of fused_embedding_to_layout (csrc/mhte_layout_kernels.h):

  layout_model / layout_grad_model   the op's algorithm restated in numpy — GatherEmb / ScatterGrad
        (runtime/ops/fused_embedding_to_layout.h:204-346) and the layout placement of
        fused_embedding_to_layout.cc:1037-1225: walks the op's OWN inputs (fid_offset,
        feature_offset, nfl_offset with the shared bit).

  reference_forward_case / reference_grad_case   the reference's own tests restated: the input
        generation AND the truth procedure of native_training/fused_embedding_to_layout_test.py
        :176-530 (forward: 199 feature slots over two tables, SHARED lists for even slots, SUM / MEAN /
        FIRSTN pooling, bias = ADDN, vec = CONCAT, ffm1 = STACK, ffm2 / firstN = NONE; the
        expectation is computed per feature from the ORIGINAL fid lists with `pooling()`, :86-113,
        not from the op's offset encoding) and :553-790 (gradient: every output gradient 1, the
        expected gradient of a fid is how often it was pooled — 1 / len for MEAN, the first
        max_sequence_length only for FIRSTN).

tests/test_layout_oracle.py (CPU) pins the model to the truth procedure on those cases; the GPU
tests compare the device with both.  Configuration objects are plain records with the proto's field
names and enum values (idl/matrix/proto/example.proto:176-221); nothing here imports monolith_amd.
"""
import random
import string
from collections import defaultdict, namedtuple

import numpy as np

SUM, MEAN, FIRSTN = 0, 1, 3                 # PoolingType
CONCAT, STACK, ADDN, NONE = 0, 1, 2, 3      # OutType
SHARD_BIT = 1 << 31

SliceConfig = namedtuple("SliceConfig", "feature_name start end")
OutConfig = namedtuple("OutConfig", "slice_configs out_type shape")
FeatureConfig = namedtuple("FeatureConfig", "table pooling_type slice_dims max_sequence_length")
FeatureConfigs = namedtuple("FeatureConfigs", "feature_configs out_configs")


def infer_shape(slices, out_type, max_sequence_length=0):
  """fused_embedding_to_layout_test.py:38-84 -> OutConfig"""
  dims = [s.end - s.start for s in slices]
  if out_type == NONE:
    shape = [[-1, max_sequence_length, d] if max_sequence_length > 0 else [-1, d] for d in dims]
  elif out_type == CONCAT:
    shape = [[-1, max_sequence_length, sum(dims)] if max_sequence_length > 0 else [-1, sum(dims)]]
  elif out_type == STACK:
    assert len(set(dims)) == 1
    shape = [[-1, len(slices), max_sequence_length, dims[0]] if max_sequence_length > 0 else [-1, len(slices), dims[0]]]
  else:
    assert len(set(dims)) == 1
    shape = [[-1, max_sequence_length, dims[0]] if max_sequence_length > 0 else [-1, dims[0]]]
  return OutConfig(list(slices), out_type, shape)


def pooling(pooling_type, in_data, max_length):
  """:86-113, as written there"""
  data = in_data[0:max_length] if (max_length and len(in_data) > max_length) else in_data
  if pooling_type == SUM:
    result = np.zeros_like(data[0])
    for d in data:
      result += d
    return result
  if pooling_type == MEAN:
    result = np.zeros_like(data[0])
    for d in data:
      result += d
    result /= len(data)
    return result
  last_dim = int(data[0].shape[-1])
  result = np.zeros(shape=(max_length, last_dim), dtype=np.float32)
  for i, d in enumerate(data):
    if i < max_length:
      result[i, :] = d
    else:
      break
  return result


def _feature_cfg(cfgs, ps_num):
  """get_feature_cfg :122-175"""
  feature_cfg, table_cfg = {}, {}
  for name, cfg in cfgs.feature_configs.items():
    feature_cfg[name] = {"table_name": cfg.table, "dim_sum": sum(cfg.slice_dims)}
    table_cfg.setdefault(cfg.table, {"feature_list": []})
  table_names = sorted(table_cfg)
  for idx, name in enumerate(table_names):
    table_cfg[name]["table_index"] = idx
  feature_names = sorted(feature_cfg)
  for idx, name in enumerate(feature_names):
    f, t = feature_cfg[name], table_cfg[feature_cfg[name]["table_name"]]
    f["feature_index"], f["table_index"] = idx, t["table_index"]
    f["feature_in_table_index"] = len(t["feature_list"])
    t["feature_list"].append(name)
  pre = 0
  for name in table_names:
    t = table_cfg[name]
    t["feature_count"] = len(t["feature_list"])
    for fn in t["feature_list"]:
      feature_cfg[fn]["pre_output_index"] = pre
      feature_cfg[fn]["table_feature_count"] = t["feature_count"]
    pre += max(t["feature_count"], 1) * ps_num
  return feature_cfg, table_cfg, feature_names, table_names, pre


def _pre_output_offset(shard, f):
  return f["pre_output_index"] + shard * f["table_feature_count"] + f["feature_in_table_index"]


def _test_configs(slots, names, split, max_sequence_length, firstn_start):
  feats, bias, vec, ffm1, ffm2, firstn = {}, [], [], [], [], []
  for slot in slots:
    fn = names(slot)
    if slot >= split[1]:
      feats[fn] = FeatureConfig("table_one", FIRSTN, [1, 4, 16], max_sequence_length)   # (a table with two dims)
      firstn.append(SliceConfig(fn, firstn_start, 21))
      continue
    if slot < split[0]:
      feats[fn] = FeatureConfig("table_one", SUM, [1, 4, 8], 0)
      ffm1.append(SliceConfig(fn, 5, 13))
    else:
      feats[fn] = FeatureConfig("table_two", MEAN, [1, 4, 16], 0)
      ffm2.append(SliceConfig(fn, 5, 21))
    bias.append(SliceConfig(fn, 0, 1))
    vec.append(SliceConfig(fn, 1, 5))
  outs = {"bias": infer_shape(bias, ADDN), "vec": infer_shape(vec, CONCAT), "ffm1": infer_shape(ffm1, STACK),
          "ffm2": infer_shape(ffm2, NONE), "firstN": infer_shape(firstn, NONE, max_sequence_length)}
  return FeatureConfigs(feats, outs)


def reference_forward_case(seed=0, batch_size=256, num_ps=5, slot_count=200, split=(50, 100)):
  """fused_embedding_to_layout_test.py:176-530 (op_version 2, no shard op) ->
  dict(cfgs, batch, embs [per pre-output index: float32 [rows, dim_sum]], fid_offset, feature_offset,
  nfl_offset, expected [tensors in the op's output order])."""
  rnd = random.Random(seed)
  max_sequence_length = 3
  cfgs = _test_configs(range(1, slot_count), lambda s: "fc_slot_%d" % s, split, max_sequence_length, 1)
  feature_cfg, table_cfg, feature_names, _, n_pre = _feature_cfg(cfgs, num_ps)
  fid_offset, feature_offset, nfl_offset = [], [0], [0]
  std_features = defaultdict(list)
  rows = [[] for _ in range(n_pre)]          # per pre-output index: the fids in row order
  dim_of = [0] * n_pre
  fid_to_emb = {}
  for fn in feature_names:
    slot = int(fn.split("fc_slot_")[-1])
    shared = slot % 2 == 0
    f = feature_cfg[fn]
    index2 = [0] * (num_ps * len(table_cfg))

    def make_fids():
      fids = list(set([(slot * 10000) + (i + 1) * 1000 + rnd.randint(1, 9) * 100
                       for i in range(rnd.randint(1, max_sequence_length * 2))]))
      std_features[fn].append(fids)
      for fid in fids:
        idx = fid % num_ps
        full = _pre_output_offset(idx, f)
        index1 = f["table_index"] * num_ps + idx
        fid_offset.append(full << 32 | index2[index1])
        index2[index1] += 1
        rows[full].append(fid)
        dim_of[full] = f["dim_sum"]
        fid_to_emb[fid] = np.array([fid + j for j in range(f["dim_sum"])], dtype=float)
      feature_offset.append(len(fid_offset))

    for _ in range(1 if shared else batch_size):
      make_fids()
    if shared:
      nfl_offset[-1] |= SHARD_BIT
    nfl_offset.append(len(feature_offset) - 1)
  embs = [np.array([fid_to_emb[fid] for fid in r], dtype=np.float32).reshape(len(r), dim_of[i] if r else 1)
          for i, r in enumerate(rows)]
  # ---- the truth procedure (:430-530)
  expected = []
  for ln in sorted(cfgs.out_configs):
    oc = cfgs.out_configs[ln]
    tensors = [np.zeros([batch_size] + sh[1:], dtype=np.float32) for sh in oc.shape]
    off = 0
    for i, sc in enumerate(oc.slice_configs):
      fc = cfgs.feature_configs[sc.feature_name]
      dim = sc.end - sc.start
      ts = tensors[0] if len(oc.shape) == 1 else tensors[i]
      features = std_features[sc.feature_name]
      tmp_addn = np.zeros(ts.shape) if oc.out_type == ADDN else None
      for b in range(batch_size):
        if b < len(features):
          tmp = [fid_to_emb[fid][sc.start:sc.end] for fid in features[b]]
          val = pooling(fc.pooling_type, tmp, fc.max_sequence_length)
          if oc.out_type == CONCAT:
            ts[b, off:off + dim] = val
          elif oc.out_type == STACK:
            ts[b, i, :] = val
          elif oc.out_type == ADDN:
            tmp_addn[b, :] = val
          else:
            ts[b, :] = val
        else:   # shared & copy
          if oc.out_type == CONCAT:
            ts[b, off:off + dim] = ts[b - 1, off:off + dim]
          elif oc.out_type == STACK:
            ts[b, i, :] = ts[b - 1, i, :]
          elif oc.out_type == ADDN:
            tmp_addn[b, :] = tmp_addn[b - 1, :]
          else:
            ts[b, :] = ts[b - 1, :]
      if oc.out_type == ADDN:
        ts += tmp_addn
      if oc.out_type == CONCAT:
        off += dim
    expected += tensors
  # (feature_offset and nfl_offset carry, as in the test, one entry past the last instance / list)
  return {"cfgs": cfgs, "batch": batch_size, "embs": embs,
          "fid_offset": np.array(fid_offset, dtype=np.uint64),
          "feature_offset": np.array(feature_offset, dtype=np.int32),
          "nfl_offset": np.array(nfl_offset, dtype=np.uint32), "expected": expected}


def reference_grad_case(seed=0, batch_size=256, num_ps=3, slot_num=30, split=(10, 20)):
  """fused_embedding_to_layout_test.py:553-790 (op_version 2): every output gradient is 1 ->
  dict(... as the forward case ..., tensors_grad, expected_grads [per pre-output index, [rows, dim]])."""
  rnd = random.Random(seed)
  max_sequence_length = 3
  alphabet = list(string.ascii_lowercase) + ["za", "zb", "zc", "zd"]
  name_of = lambda s: "fc_slot_%s" % alphabet[s - 1]  # noqa: E731
  cfgs = _test_configs(range(1, slot_num), name_of, split, max_sequence_length, 0)
  feature_cfg, _, feature_names, _, n_pre = _feature_cfg(cfgs, num_ps)
  slot_of = {name_of(s): s for s in range(1, slot_num)}
  slot2fid, idx_lists = {}, {}
  for slot in range(1, slot_num):
    fids = list(set([(slot << 48) + rnd.randint(100, 1000000)
                     for _ in range(rnd.randint(batch_size + 1, batch_size + 10))]))
    slot2fid[slot] = fids
    idx_lists[slot] = [list(range(bi, len(fids) - batch_size + 1 + bi)) for bi in range(batch_size)]
  rows = [[] for _ in range(n_pre)]
  dim_of = [1] * n_pre
  where = {}     # slot -> per fid index (pre-output index, row)
  for slot in range(1, slot_num):
    f = feature_cfg[name_of(slot)]
    where[slot] = []
    for fid in slot2fid[slot]:
      full = _pre_output_offset(fid % num_ps, f)
      where[slot].append((full, len(rows[full])))
      rows[full].append(fid)
      dim_of[full] = f["dim_sum"]
  truth = [np.zeros((len(r), 1), dtype=np.float64) for r in rows]   # pooled how often (x 1 / len for MEAN)
  fid_offset, feature_offset, nfl_offset = [], [0], [0]
  # (the op walks the named feature lists in sorted-name order; the test's slot order is that order)
  assert feature_names == [name_of(s) for s in sorted(slot_of.values(), key=name_of)]
  for fn in feature_names:
    slot = slot_of[fn]
    fc = cfgs.feature_configs[fn]
    for bi in range(batch_size):
      lst = idx_lists[slot][bi]
      for i, idx in enumerate(lst):
        full, row = where[slot][idx]
        fid_offset.append(full << 32 | row)
        if fc.pooling_type == FIRSTN and i >= fc.max_sequence_length:
          pass
        elif fc.pooling_type == MEAN:
          truth[full][row, 0] += 1 / len(lst)
        else:
          truth[full][row, 0] += 1
      feature_offset.append(len(fid_offset))
    nfl_offset.append(len(feature_offset) - 1)
  nrng = np.random.default_rng(seed)
  embs = [nrng.uniform(size=(len(r), dim_of[i])).astype(np.float32) for i, r in enumerate(rows)]
  tensors_grad = []
  for ln in sorted(cfgs.out_configs):
    for sh in cfgs.out_configs[ln].shape:
      tensors_grad.append(np.ones([batch_size] + sh[1:], dtype=np.float32))
  # which columns of a fid's row receive gradient: the slices of its feature, once per layout that
  # uses them (:770-790 compares every element of the row with the count where a slice covers it)
  expected = []
  for i, r in enumerate(rows):
    e = np.zeros((len(r), dim_of[i]), dtype=np.float64)
    expected.append(e)
  cover = {}
  for ln in sorted(cfgs.out_configs):
    for sc in cfgs.out_configs[ln].slice_configs:
      c = cover.setdefault(sc.feature_name, np.zeros(sum(cfgs.feature_configs[sc.feature_name].slice_dims)))
      c[sc.start:sc.end] += 1
  for slot in range(1, slot_num):
    c = cover.get(name_of(slot))
    if c is None:
      continue
    for (full, row) in where[slot]:
      expected[full][row, :] = truth[full][row, 0] * c
  return {"cfgs": cfgs, "batch": batch_size, "embs": embs,
          "fid_offset": np.array(fid_offset, dtype=np.uint64),
          "feature_offset": np.array(feature_offset, dtype=np.int32),
          "nfl_offset": np.array(nfl_offset, dtype=np.uint32),
          "tensors_grad": tensors_grad, "expected_grads": expected}


# ------------------------------------------------------------------------------------------- the op
def _walk(fid_offset, feature_offset, nfl_offset, batch, cfgs):
  """(output tensor index, slice, out config, slice index, CONCAT offset, b, [(matrix, row)...])
  for every slice and batch row that has fids — the op's traversal"""
  names = sorted(cfgs.feature_configs)
  n_feature, n_fid, n_nfl = len(feature_offset), len(fid_offset), len(nfl_offset)
  base = 0
  for ln in sorted(cfgs.out_configs):
    oc = cfgs.out_configs[ln]
    off = 0
    for i, sc in enumerate(oc.slice_configs):
      nfl = names.index(sc.feature_name)
      enc = int(nfl_offset[nfl])
      shared, noff = enc >> 31, enc & 0x7fffffff
      nxt = (int(nfl_offset[nfl + 1]) & 0x7fffffff) if nfl < n_nfl - 1 else n_feature
      t_idx = base if len(oc.shape) == 1 else base + i
      for b in range(batch):
        if nxt - noff <= 0:
          continue
        f = noff + (0 if shared else b)
        f0 = int(feature_offset[f])
        f1 = int(feature_offset[f + 1]) if f < n_feature - 1 else n_fid
        rows = [(int(fid_offset[q]) >> 32, int(fid_offset[q]) & 0xffffffff) for q in range(f0, f1)]
        if rows:
          yield t_idx, sc, oc, i, off, b, rows
      if oc.out_type == CONCAT:
        off += sc.end - sc.start
    base += len(oc.shape)


def _view(t, oc, i, off, dim, b):
  if oc.out_type == CONCAT:
    return t[b, off:off + dim]
  if oc.out_type == STACK:
    return t[b, i, :]
  return t[b]


def layout_model(embs, fid_offset, feature_offset, nfl_offset, batch, cfgs):
  """Forward: GatherEmb (fused_embedding_to_layout.h:204-262), fp32 sums in fid order; MEAN scales
  every row by 1 / n before adding (:236-243); ADDN accumulates over its slices."""
  outs = []
  for ln in sorted(cfgs.out_configs):
    for sh in cfgs.out_configs[ln].shape:
      outs.append(np.zeros([batch if d == -1 else d for d in sh], np.float32))
  for t_idx, sc, oc, i, off, b, rows in _walk(fid_offset, feature_offset, nfl_offset, batch, cfgs):
    fc = cfgs.feature_configs[sc.feature_name]
    view = _view(outs[t_idx], oc, i, off, sc.end - sc.start, b)
    if fc.pooling_type == FIRSTN:
      for s_, (i1, i2) in enumerate(rows[:fc.max_sequence_length]):
        view[s_, :] = embs[i1][i2, sc.start:sc.end]
      continue
    acc = None
    for (i1, i2) in rows:
      x = embs[i1][i2, sc.start:sc.end]
      if fc.pooling_type == MEAN:
        x = x / np.float32(len(rows))
      acc = x.copy() if acc is None else acc + x
    if oc.out_type == ADDN:
      view += acc
    else:
      view[:] = acc
  return outs


def layout_grad_model(embs, fid_offset, feature_offset, nfl_offset, batch, cfgs, tensors_grad,
                      acc_dtype=np.float64):
  """Gradient: ScatterGrad (fused_embedding_to_layout.h:264-346) -> gradients of `embs`.
  acc_dtype = float64: the sums' value (the reference's GPU kernel and the product's
  MHTE_POOL_ATOMICS form add in arrival order: compared with a tolerance); float32: every add in
  fp32 in the op's traversal order — slices in configuration order, batch rows ascending, fids in
  list order — which is what the reference's CPU kernel computes and what the product's grouped
  (atomic-free) form reproduces bit for bit."""
  grads = [np.zeros(e.shape, acc_dtype) for e in embs]
  for t_idx, sc, oc, i, off, b, rows in _walk(fid_offset, feature_offset, nfl_offset, batch, cfgs):
    fc = cfgs.feature_configs[sc.feature_name]
    view = _view(tensors_grad[t_idx], oc, i, off, sc.end - sc.start, b)
    for s_, (i1, i2) in enumerate(rows):
      if fc.pooling_type == FIRSTN:
        if s_ < fc.max_sequence_length:
          grads[i1][i2, sc.start:sc.end] += view[s_, :]
      elif fc.pooling_type == MEAN:
        grads[i1][i2, sc.start:sc.end] += view / np.float32(len(rows))
      else:
        grads[i1][i2, sc.start:sc.end] += view
  return grads


# the product's atomic-free gradient (csrc/mhte_layout_kernels.h: kLayoutLight, kLayoutHeavyGroups)
LAYOUT_LIGHT = 1024
LAYOUT_HEAVY_GROUPS = 64


def layout_grad_model_grouped(embs, fid_offset, feature_offset, nfl_offset, batch, cfgs, tensors_grad,
                              light=LAYOUT_LIGHT, groups=LAYOUT_HEAVY_GROUPS):
  """The gradient as the product's grouped form computes it, fp32 add for fp32 add: per embedding
  row and slice (slices in configuration order) the contributions in the op's order (batch rows
  ascending, fids in list order).  A LIGHT row adds them one after the other onto the row — which is
  layout_grad_model(acc_dtype=float32).  A HEAVY row (more than `light` positions of fid_offset name
  it, or a shared list reaches it and positions x batch exceeds `light`) cuts a slice's sequence into
  `groups` contiguous ranges, sums each range in order and adds the range sums to the row in range
  order.  -> (gradients, number of heavy rows)"""
  names = sorted(cfgs.feature_configs)
  n_feature, n_nfl = len(feature_offset), len(nfl_offset)
  fo = [int(x) for x in fid_offset]
  length = {}
  for v in fo:
    k = (v >> 32, v & 0xffffffff)
    length[k] = length.get(k, 0) + 1
  shared_rows = set()
  for nfl in range(n_nfl):
    enc = int(nfl_offset[nfl])
    if not enc >> 31:
      continue
    noff = enc & 0x7fffffff
    nxt = (int(nfl_offset[nfl + 1]) & 0x7fffffff) if nfl < n_nfl - 1 else n_feature
    if nxt - noff <= 0 or noff >= n_feature:
      continue
    f0 = int(feature_offset[noff])
    f1 = int(feature_offset[noff + 1]) if noff < n_feature - 1 else len(fo)
    for q in range(f0, f1):
      shared_rows.add((fo[q] >> 32, fo[q] & 0xffffffff))
  heavy = {k for k, n in length.items() if n > light or (k in shared_rows and n * batch > light)}
  # (row, slice ordinal) -> contributions in order; None = FIRSTN beyond max_sequence_length
  seqs, order = {}, {}
  for t_idx, sc, oc, i, off, b, rows in _walk(fid_offset, feature_offset, nfl_offset, batch, cfgs):
    fc = cfgs.feature_configs[sc.feature_name]
    view = _view(tensors_grad[t_idx], oc, i, off, sc.end - sc.start, b)
    tid = order.setdefault((id(oc), i), len(order))
    for s_, (i1, i2) in enumerate(rows):
      if i1 >= len(embs):
        continue
      if fc.pooling_type == FIRSTN:
        x = view[s_, :].astype(np.float32) if s_ < fc.max_sequence_length else None
      elif fc.pooling_type == MEAN:
        x = (view / np.float32(len(rows))).astype(np.float32)
      else:
        x = np.asarray(view, np.float32)
      seqs.setdefault((i1, i2), {}).setdefault((tid, sc.start, sc.end), []).append(x)
  grads = [np.zeros(e.shape, np.float32) for e in embs]
  for (i1, i2), per_task in seqs.items():
    for (tid, s0, s1), xs in sorted(per_task.items()):
      acc = grads[i1][i2, s0:s1].copy()
      if (i1, i2) not in heavy:
        for x in xs:
          if x is not None:
            acc = acc + x
      else:
        per = (len(xs) + groups - 1) // groups
        for g in range(groups):
          part = None
          for x in xs[g * per:(g + 1) * per]:
            if x is not None:
              part = x.copy() if part is None else part + x
          if part is not None:
            acc = acc + part
      grads[i1][i2, s0:s1] = acc
  return grads, len(heavy & set(seqs))

