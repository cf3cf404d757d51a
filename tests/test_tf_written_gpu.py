"""GPU parity (-m gpu): the reference's own TF-WRITTEN single-table dumps through the product.

tests/golden/tf_written/ holds shard sets TensorFlow's io::RecordWriter wrote in
MonolithHashTableSave (runtime/ops/hash_table_save_op.cc:147-160; copied by make_tf_written.py from
model_export/testdata/saved_model/ps_N/1622716114/assets/).  mhte_table_restore
(MonolithHashTableRestore, runtime/ops/hash_table_restore_op.cc:63-160) reads them into tables of
the layouts they were saved from — dim 33 = bias FTRL(1) + vector SGD(32), dim 17 = FTRL(1) +
SGD(16), dim 16 / 32 SGD — and
  * mhte_lookup_entry returns, for every id, exactly the record bytes of the file;
  * mhte_table_save writes files whose records are the same SET of byte strings (the shard split
    follows this table's bucket ranges, so which shard holds a record may differ), each file a
    TFRecord stream TensorFlow's framing check accepts;
  * restore clears first: what the table held before is gone; a second restore is idempotent.
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ckpt_proto as P  # noqa: E402
from monolith_amd import _lib, entry  # noqa: E402
from monolith_amd.multi_hash_table_ops import MultiHashTable  # noqa: E402

GOLD = os.path.join(HERE, "golden", "tf_written")
with open(os.path.join(GOLD, "manifest.json")) as _f:
  MANIFEST = json.load(_f)
_n = [0]


def table_for(m):
  segs = []
  for kind, dim in m["segments"]:
    opt = (entry.FtrlOptimizer(0.01, 0.1, 1.0) if kind == "ftrl" else entry.SgdOptimizer(0.01))
    segs.append(entry.CombineAsSegment(dim, entry.ZerosInitializer(), opt))
  _n[0] += 1
  return MultiHashTable.from_configs({"t": entry.make_table_config(segs)}, name_suffix="tfw%d" % _n[0])


def file_records(base, total=4):
  out = []
  for i in range(total):
    out += P.unframe(open("%s-%05d-of-%05d" % (base, i, total), "rb").read())
  return out


@pytest.mark.parametrize("m", MANIFEST, ids=lambda m: "dim%d" % m["dim"])
def test_restore_tf_written_shards_and_return_their_bytes(m, tmp_path):
  base = os.path.join(GOLD, m["basename"])
  recs = file_records(base)
  assert len(recs) == sum(m["records_per_shard"]) > 0
  by_id = {}
  for r in recs:
    e = P.EntryDump.FromString(r)
    by_id[e.id] = r
  mt = table_for(m)
  # something the restore must clear away
  junk = torch.arange(1000, 1040, dtype=torch.int64).cuda()
  mt.assign({"t": (junk, torch.ones((40, m["dim"]), device="cuda"))}, req_time=5)
  assert mt.size("t") == 40
  mt.restore_table("t", base)
  assert mt.size("t") == len(by_id)
  ids = np.array(sorted(by_id), dtype=np.int64)
  got = mt.lookup_entry({"t": torch.from_numpy(ids).cuda()})["t"]
  for i, g in zip(ids.tolist(), got):
    assert g == by_id[i], "id %d: entry bytes differ from the TF-written record" % i
  assert mt.lookup_entry({"t": junk})["t"] == [b""] * 40          # cleared
  # the embedding the lookup op returns is the record's `num`
  emb = mt.lookup({"t": torch.from_numpy(ids).cuda()})["t"].cpu().numpy()
  for k, i in enumerate(ids.tolist()):
    np.testing.assert_array_equal(emb[k], np.array(P.EntryDump.FromString(by_id[i]).num, np.float32))
  # save -> the same records (as a set), in files an independent TFRecord reader accepts
  for nshards in (4, 1):
    out = str(tmp_path / ("out%d" % nshards) / "MonolithHashTable_x")
    mt.save_table("t", out, nshards=nshards)
    back = file_records(out, nshards)
    assert sorted(back) == sorted(recs)
    assert not os.path.exists(out + ".meta-00000-of-%05d" % nshards)   # no sidecar in this layout
  # a dump this engine wrote restores the same way (and restore is idempotent)
  mt2 = table_for(m)
  mt2.restore_table("t", str(tmp_path / "out4" / "MonolithHashTable_x"))
  mt2.restore_table("t", str(tmp_path / "out4" / "MonolithHashTable_x"))
  assert mt2.size("t") == len(by_id)
  assert mt2.lookup_entry({"t": torch.from_numpy(ids).cuda()})["t"] == got


def test_all_tf_written_records_restore(tmp_path):
  """All 150 records TensorFlow wrote, layout by layout (the concatenated files as one-shard sets)."""
  import shutil
  for m in MANIFEST:
    base = str(tmp_path / ("all%d" % m["dim"]))
    shutil.copyfile(os.path.join(GOLD, m["all_records_file"]), base + "-00000-of-00001")
    recs = P.unframe(open(base + "-00000-of-00001", "rb").read())
    last = {}
    for r in recs:                       # (the five parameter servers never share an id; later wins anyway)
      last[P.EntryDump.FromString(r).id] = r
    mt = table_for(m)
    mt.restore_table("t", base)
    assert mt.size("t") == len(last)
    ids = np.array(sorted(last), dtype=np.int64)
    got = mt.lookup_entry({"t": torch.from_numpy(ids).cuda()})["t"]
    assert got == [last[i] for i in ids.tolist()]


def test_restore_table_errors(tmp_path):
  m = MANIFEST[0]
  mt = table_for(m)
  with pytest.raises(_lib.MhteError) as ei:      # no files at all
    mt.restore_table("t", str(tmp_path / "nothing"))
  assert ei.value.code == _lib.MHTE_NOT_FOUND
  import shutil
  base = str(tmp_path / "part")
  src = os.path.join(GOLD, m["basename"])
  for i in (0, 1, 3):                            # shard 2 of 4 missing (ValidateShardedFiles :67-72)
    shutil.copyfile("%s-%05d-of-00004" % (src, i), "%s-%05d-of-00004" % (base, i))
  with pytest.raises(_lib.MhteError) as ei:
    mt.restore_table("t", base)
  assert ei.value.code == _lib.MHTE_INVALID_ARGUMENT


def test_tf_written_dim33_table_trains_through_the_sharded_and_multi_table_steps():
  """A table the reference's TensorFlow wrote (dim 33 = bias FTRL(1) + vector SGD(32): its standard
  row layout) is restored and then TRAINED: two updates through the id-sharded step (one rank) and
  through the multi-table step must leave exactly the rows the op-level update (mhte_optimize on the
  per-id gradient sums, duplicates added in occurrence order) leaves on a third restored copy — the
  restored entries' FTRL state included.  Round 3 could restore these tables but had no fused / multi-GPU path for them."""
  from monolith_amd.distributed_ps_sync import ShardedMultiStep
  from monolith_amd.fused_step import MultiSparseStep
  m = [x for x in MANIFEST if x["dim"] == 33][0]
  base = os.path.join(GOLD, m["basename"])
  recs = {}
  for r in file_records(base):
    e = P.EntryDump.FromString(r)
    recs[e.id] = e
  ids_all = np.array(sorted(recs), dtype=np.int64)
  rng = np.random.default_rng(33)
  B = 3 * ids_all.size + 16          # every restored id three times (shuffled) + 16 new ids
  batches = [np.concatenate([rng.permutation(np.tile(ids_all, 3)), rng.integers(1, 2**40, 16)]).astype(np.int64)
             for _ in range(3)]
  # (every id at most 32 times in a batch: its gradients are added in occurrence order on every path —
  # bit-exact; longer lists are a fixed tree, equal to 1e-5)
  assert max(np.unique(b, return_counts=True)[1].max() for b in batches) <= 32
  grads = [(rng.standard_normal((B, 33)) * 0.1).astype(np.float32) for _ in range(2)]
  results = []
  for kind in ("sharded", "multi"):
    mt = table_for(m)
    mt.restore_table("t", base)
    step = ShardedMultiStep(mt, B) if kind == "sharded" else MultiSparseStep(mt, B, exact_order=True)
    rag = [mt.get_ragged_id({"t": torch.from_numpy(b).cuda()}) for b in batches]
    for s in range(2):
      emb = step.forward(rag[s], rag[s + 1])
      got = mt.get_embeddings(rag[s], emb)["t"].cpu().numpy()
      if s == 0:   # the restored rows (new ids: zeros)
        exp0 = np.stack([np.array(recs[i].num, np.float32) if i in recs else np.zeros(33, np.float32)
                         for i in batches[0].tolist()])
        np.testing.assert_array_equal(got, exp0)
      step.backward(torch.from_numpy(grads[s].ravel()).cuda(), 1000 + s)
    step.close()
    results.append((kind, mt.lookup({"t": torch.from_numpy(np.unique(np.concatenate(batches[:2]))).cuda()})["t"].cpu().numpy()))
  # the two fused paths agree bit for bit with each other, and with the op-level path on a third copy
  mt3 = table_for(m)
  mt3.restore_table("t", base)
  for s in range(2):
    u, inv = np.unique(batches[s], return_inverse=True)
    gu = np.zeros((u.size, 33), np.float32)
    order = np.argsort(inv, kind="stable")
    for p_ in order:          # duplicate gradients summed in occurrence order
      gu[inv[p_]] = gu[inv[p_]] + grads[s][p_]
    mt3.apply_gradients({"t": (torch.from_numpy(u).cuda(), torch.from_numpy(gu).cuda())}, req_time=1000 + s)
  ref = mt3.lookup({"t": torch.from_numpy(np.unique(np.concatenate(batches[:2]))).cuda()})["t"].cpu().numpy()
  for kind, rows in results:
    np.testing.assert_array_equal(rows, ref, err_msg=kind)

