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
