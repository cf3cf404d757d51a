"""CPU suite: MonolithMultiHashTableFeatureStat (multi_hash_table_save_restore_ops.cc:424-497) and the
shard-set validation in front of it (ValidateShardedFiles, :323-349) through the C ABI — host code
only, on .meta sidecars written by the independent Python framing (tests/ckpt_proto.py)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import ckpt_proto as P  # noqa: E402
from monolith_amd import _lib  # noqa: E402
from monolith_amd.multi_hash_table_ops import MultiHashTable  # noqa: E402


def _meta(name, n):
  m = P.MultiHashTableMetadata()
  m.table_name = name
  m.num_entries = n
  return m.SerializeToString()


def _write_set(base, total, tables, shards=None):
  for sh in (range(total) if shards is None else shards):
    with open("%s.meta-%05d-of-%05d" % (base, sh, total), "wb") as f:
      f.write(b"".join(P.frame(_meta(n, c + sh)) for n, c in tables))
    open("%s-%05d-of-%05d" % (base, sh, total), "wb").close()


def test_counts_are_summed_over_the_meta_sidecars(tmp_path):
  base = str(tmp_path / "ck")
  _write_set(base, 3, [("user", 5), ("item", 70), ("ctx", 0)])
  # per table: sum over shards of (c + shard); names come back sorted
  got = MultiHashTable.feature_stat(base)
  assert got == {"ctx": 0 + 1 + 2, "item": 210 + 3, "user": 15 + 3}
  assert list(got) == sorted(got)


def test_a_table_missing_from_some_shards_still_counts(tmp_path):
  base = str(tmp_path / "ck")
  _write_set(base, 2, [("a", 4)])
  with open("%s.meta-00001-of-00002" % base, "ab") as f:   # shard 1 also holds table b
    f.write(P.frame(_meta("b", 9)))
  assert MultiHashTable.feature_stat(base) == {"a": 4 + 5, "b": 9}


def test_incomplete_or_mixed_shard_sets_are_errors(tmp_path):
  base = str(tmp_path / "gone")
  with pytest.raises(_lib.MhteError):                        # nothing there
    MultiHashTable.feature_stat(base)
  base = str(tmp_path / "part")
  _write_set(base, 3, [("a", 1)], shards=(0, 2))             # shard 1 of 3 missing
  with pytest.raises(_lib.MhteError):
    MultiHashTable.feature_stat(base)
  base = str(tmp_path / "mixed")
  _write_set(base, 2, [("a", 1)])
  _write_set(base, 3, [("a", 1)])                            # leftovers of another shard count
  with pytest.raises(_lib.MhteError):
    MultiHashTable.feature_stat(base)


def test_a_corrupted_sidecar_is_reported(tmp_path):
  base = str(tmp_path / "ck")
  _write_set(base, 1, [("a", 3), ("b", 4)])
  p = "%s.meta-00000-of-00001" % base
  b = bytearray(open(p, "rb").read())
  b[14] ^= 0x40                                               # inside the first record's data
  open(p, "wb").write(bytes(b))
  with pytest.raises(_lib.MhteError):
    MultiHashTable.feature_stat(base)
