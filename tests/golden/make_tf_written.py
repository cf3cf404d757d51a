"""Copies a few TF-written single-table hash-table dumps the reference holds as test data into
tests/golden/tf_written/ (run in the build container, where /root/reference exists; the fixtures
are committed because the GPU box has no reference tree).

Source: /root/reference/monolith/native_training/model_export/testdata/saved_model/ps_N/1622716114/
assets/MonolithHashTable_<hash>_<k>-%05d-of-00004 — written by TensorFlow's io::RecordWriter in
MonolithHashTableSave (runtime/ops/hash_table_save_op.cc:147-160): uncompressed TFRecord streams of
EntryDump.  Four table kinds occur (dim 33 = FTRL(1) + SGD(32), dim 17 = FTRL(1) + SGD(16), dim 16
and dim 32 SGD); per kind the shard set with the most records is taken, all four shard files of it
(empty shards included: a restore must accept them).  manifest.json lists, per set, the table
layout and the record count per shard, decoded with tests/ckpt_proto.py (protobuf runtime).
all_dim<D>.tfrecord: EVERY non-empty file of that layout across the five parameter servers,
concatenated byte for byte (a concatenation of TFRecord streams is a TFRecord stream): all 150
TF-written records for the codec test, in 4 files instead of 135.
"""
import glob
import json
import os
import shutil
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import ckpt_proto as P  # noqa: E402

SRC = "/root/reference/monolith/native_training/model_export/testdata/saved_model"
DST = os.path.join(HERE, "tf_written")


def records(path):
  d = open(path, "rb").read()
  pos, out = 0, []
  while pos < len(d):
    n = struct.unpack("<Q", d[pos:pos + 8])[0]
    out.append(d[pos + 12:pos + 12 + n])
    pos += 16 + n
  return out


def main():
  best = {}   # layout -> (records, ps, stem)
  every = {}  # layout -> bytes of all its non-empty files
  for ps in range(5):
    adir = os.path.join(SRC, "ps_%d" % ps, "1622716114", "assets")
    stems = sorted({f.rsplit("-", 3)[0] for f in os.listdir(adir) if f.startswith("MonolithHashTable_")})
    for stem in stems:
      files = sorted(glob.glob(os.path.join(adir, stem + "-*-of-*")))
      recs = [r for f in files for r in records(f)]
      if not recs:
        continue
      e = P.EntryDump()
      e.ParseFromString(recs[0])
      layout = (len(e.num), tuple(o.WhichOneof("type") for o in e.opt.dump))
      every[layout] = every.get(layout, b"") + b"".join(open(f, "rb").read() for f in files)
      if layout not in best or len(recs) > best[layout][0]:
        best[layout] = (len(recs), ps, stem)
  if os.path.isdir(DST):
    shutil.rmtree(DST)
  os.makedirs(DST)
  manifest = []
  for (dim, opts), (n, ps, stem) in sorted(best.items()):
    adir = os.path.join(SRC, "ps_%d" % ps, "1622716114", "assets")
    files = sorted(glob.glob(os.path.join(adir, stem + "-*-of-*")))
    name = "ps%d_%s" % (ps, stem.split("_", 1)[1])
    per = []
    for f in files:
      tail = os.path.basename(f)[len(stem):]
      shutil.copyfile(f, os.path.join(DST, name + tail))
      per.append(len(records(f)))
    # segment dims: a bias segment of 1 (FTRL zero/norm have one element) + the vector
    segs = []
    e = P.EntryDump()
    e.ParseFromString(records([f for f, k in zip(files, per) if k][0])[0])
    left = dim
    for i, o in enumerate(e.opt.dump):
      kind = o.WhichOneof("type")
      d = len(o.ftrl.zero) if kind == "ftrl" else (left if i == len(e.opt.dump) - 1 else None)
      segs.append([kind, d])
      left -= d
    cat = "all_dim%d.tfrecord" % dim
    with open(os.path.join(DST, cat), "wb") as f:
      f.write(every[(dim, opts)])
    manifest.append({"basename": name, "dim": dim, "segments": segs, "records_per_shard": per,
                     "source": "ps_%d/1622716114/assets/%s" % (ps, stem),
                     "all_records_file": cat, "all_records": len(records(os.path.join(DST, cat)))})
  with open(os.path.join(DST, "manifest.json"), "w") as f:
    json.dump(manifest, f, indent=1)
  print(json.dumps(manifest, indent=1))


if __name__ == "__main__":
  main()
