"""Measurements for the rows next to the hot path, one JSON object per line + a markdown
table: checkpoint save / restore, TTL eviction scan, admission filter inside the training step,
post-exchange gather (+ gradient), ragged reductions, and the op-level update of every optimizer.

Every kernel here is HBM-streaming or random-row work, so each line carries algorithmic bytes,
GB/s and the fraction of the 8 TB/s HBM peak; the checkpoint lines are host-codec bound (snappy +
protobuf + crc on CPU threads) and say so.  Run on the GPU box:
    python scripts/next_rows_bench.py > gpurun_out/<tag>/next_rows.jsonl
"""
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from monolith_amd import entry, synthetic as S  # noqa: E402
from monolith_amd import distribution_ops as D  # noqa: E402
from monolith_amd.fused_step import SparseStep  # noqa: E402
from monolith_amd.multi_hash_table_ops import HashFilter, MultiHashTable  # noqa: E402

HBM_PEAK = 8000.0  # GB/s
DEV = torch.device("cuda:0")
LINES = []


def emit(name, seconds, alg_bytes=None, **kw):
  rec = {"name": name, "us": round(seconds * 1e6, 2)}
  if alg_bytes is not None:
    gbps = alg_bytes / seconds / 1e9
    rec.update(alg_bytes=int(alg_bytes), GBps=round(gbps, 1), hbm_frac=round(gbps / HBM_PEAK, 4))
  rec.update(kw)
  LINES.append(rec)
  print(json.dumps(rec), flush=True)


def gpu_time(fn, reps=20, warm=3):
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  t = time.perf_counter()
  for _ in range(reps):
    fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t) / reps


def make_table(opt, dim, rows_cap, suffix, hash_filter=None, occ=None, evict_hours=0, ttl_days=None):
  slots = 4
  while slots * 0.5 < rows_cap:
    slots *= 2
  cfg = entry.make_table_config(
      [entry.CombineAsSegment(dim, entry.ZerosInitializer(), opt)],
      entry.CuckooHashTableConfig(initial_capacity=slots, reserve_rows=rows_cap,
                                  feature_evict_every_n_hours=evict_hours),
      slot_expire_time_config=(entry.SlotExpireTimeConfig(default_expire_time=ttl_days)
                               if ttl_days else None),
      slot_occurrence_threshold_config=occ)
  return MultiHashTable.from_configs({"emb": cfg}, name_suffix=suffix, hash_filter=hash_filter)


def fill(mt, n_rows, dim, ts=1000):
  """n_rows distinct ids with non-trivial rows and state."""
  B = 1 << 18
  lr = mt.learning_rate
  for lo in range(0, n_rows, B):
    n = min(B, n_rows - lo)
    ids = (torch.arange(lo, lo + n, dtype=torch.int64, device=DEV) * 2654435761 % (1 << 40)) | (1 << 48)
    g = torch.full((n, dim), 0.01, dtype=torch.float32, device=DEV)
    mt.table_optimize_n("emb", ids, None, g, lr, ts + lo // B, 0)
  torch.cuda.synchronize()


def bench_checkpoint():
  dim, rows = 64, 1 << 21
  mt = make_table(entry.AdagradOptimizer(0.01, 0.1), dim, rows + (1 << 19), "ckpt")
  fill(mt, rows, dim)
  n = mt.size("emb")
  # (tmpfs when there is one: the measurement is of the codec, not of the box's disk)
  tmp = tempfile.mkdtemp(prefix="mhte_ckpt_", dir=os.environ.get(
      "MHTE_CKPT_DIR", "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"))
  try:
    base = os.path.join(tmp, "model")
    t = time.perf_counter()
    mt.save(base, nshards=4)
    torch.cuda.synchronize()
    ts = time.perf_counter() - t
    disk = sum(os.path.getsize(os.path.join(tmp, f)) for f in os.listdir(tmp))
    row_bytes = 4 * 2 * dim + 16
    emit("checkpoint save (2M rows, dim 64 + Adagrad state, 4 shards, TFRecord + snappy + EntryDump)",
         ts, n * row_bytes, rows=n, rows_per_s=round(n / ts), file_bytes=disk,
         bound="per shard, in chunks of 2^18 slots, three stages beside each other: device scan + copy "
               "(under the table lock) | EntryDump + crc32c on 16 host threads | one writer thread; "
               "first save pins its staging buffers")
    mt2 = make_table(entry.AdagradOptimizer(0.01, 0.1), dim, rows + (1 << 19), "ckpt2")
    t = time.perf_counter()
    mt2.restore(base)
    torch.cuda.synchronize()
    tr = time.perf_counter() - t
    assert mt2.size("emb") == n
    emit("checkpoint restore (same files into an empty table)", tr, n * row_bytes, rows=n,
         rows_per_s=round(n / tr), bound="per shard: a reader thread ahead by one 64 MiB stretch "
                                         "(snappy blocks unpacked on 4 threads) | verify + decode on "
                                         "16 threads | duplicate check + upsert, beside each other")
    mt2.close()
    # the same table in 16 shard files (shards are written / read by threads of their own)
    for f in os.listdir(tmp):
      os.unlink(os.path.join(tmp, f))
    t = time.perf_counter()
    mt.save(base, nshards=16)
    torch.cuda.synchronize()
    ts = time.perf_counter() - t
    emit("checkpoint save, 16 shards", ts, n * row_bytes, rows=n, rows_per_s=round(n / ts))
    mt3 = make_table(entry.AdagradOptimizer(0.01, 0.1), dim, rows + (1 << 19), "ckpt3")
    t = time.perf_counter()
    mt3.restore(base)
    torch.cuda.synchronize()
    tr = time.perf_counter() - t
    assert mt3.size("emb") == n
    emit("checkpoint restore, 16 shards", tr, n * row_bytes, rows=n, rows_per_s=round(n / tr))
    # a second save with the staging buffers warm (they are kept by the table)
    for f in os.listdir(tmp):
      os.unlink(os.path.join(tmp, f))
    t = time.perf_counter()
    mt.save(base, nshards=4)
    torch.cuda.synchronize()
    ts = time.perf_counter() - t
    emit("checkpoint save, 4 shards, staging warm", ts, n * row_bytes, rows=n, rows_per_s=round(n / ts))
    mt3.close()
  finally:
    shutil.rmtree(tmp, ignore_errors=True)
  mt.close()


def bench_evict():
  dim, rows = 64, 1 << 22
  mt = make_table(entry.AdagradOptimizer(0.01, 0.1), dim, rows + (1 << 19), "evict", evict_hours=1,
                  ttl_days=1)
  fill(mt, rows, dim, ts=1000)
  st = mt.stats("emb")
  # scan with nothing expired (max_update_time close to the row stamps): pure bucket scan
  t = gpu_time(lambda: mt.evict("emb", 2000), reps=10)
  emit("TTL eviction scan, nothing expired (4M rows resident, 64-B buckets)", t,
       int(st.bytes_buckets), buckets_bytes=int(st.bytes_buckets))
  before = mt.size("emb")
  t0 = time.perf_counter()
  mt.evict("emb", 1000 + 86400 * 2)
  torch.cuda.synchronize()
  t1 = time.perf_counter() - t0
  # (every occupied bucket line is read AND written back: key and row handle of its slots cleared)
  emit("TTL eviction scan, every row expired (buckets read + written back)", t1,
       2 * int(st.bytes_buckets), evicted=before - mt.size("emb"))
  mt.close()


def bench_filter_step():
  B, dim, K = 65536, 64, 60
  ids_all = [torch.from_numpy(S.id_batch(s, B, int(1e9))).to(DEV) for s in range(K + 12)]
  grad = torch.from_numpy(S.grad_batch(0, B, dim)).to(DEV)
  res = {}
  for name, use in (("off", False), ("on", True)):
    flt = HashFilter(capacity=1 << 24, split_num=7) if use else None
    occ = entry.SlotOccurrenceThresholdConfig(default_occurrence_threshold=2) if use else None
    mt = make_table(entry.AdagradOptimizer(0.001, 0.1), dim, (K + 20) * B, "flt_" + name,
                    hash_filter=flt, occ=occ)
    step = SparseStep(mt, "emb", B)

    def run(lo, hi):
      for s in range(lo, hi):
        step.forward(ids_all[s], next_ids=ids_all[s + 1])
        step.backward(grad, S.update_time(s))
    run(0, 10)
    torch.cuda.synchronize()
    t = time.perf_counter()
    run(10, 10 + K)
    torch.cuda.synchronize()
    res[name] = (time.perf_counter() - t) / K
    rows = mt.size("emb")
    emit("pipelined training step, admission filter %s (threshold 2, empty table at start)" % name,
         res[name], None, rows_after=rows, lookups_updates_per_s=round(2 * B / res[name]))
    mt.close()
    if flt is not None:
      flt.close()


def bench_step_optimizers():
  """The pipelined two-launch training step on tables whose optimizer is not SGD / Adagrad / FTRL
  (round 2: such tables took the unpipelined op-level path)."""
  B, dim, K = 65536, 64, 80
  V = int(1e8)
  ids_all = [torch.from_numpy(S.id_batch(s, B, V)).to(DEV) for s in range(K + 12)]
  grad = torch.from_numpy(S.grad_batch(0, B, dim)).to(DEV)
  for name, opt in (("adagrad", entry.AdagradOptimizer(0.001, 0.1)), ("momentum", entry.MomentumOptimizer(0.01)),
                    ("adam", entry.AdamOptimizer(0.01)), ("amsgrad", entry.AdamOptimizer(0.01, amsgrad=True))):
    mt = make_table(opt, dim, (K + 20) * B, "sopt_" + name)
    step = SparseStep(mt, "emb", B)

    def run(lo, hi):
      for s in range(lo, hi):
        step.forward(ids_all[s], next_ids=ids_all[s + 1])
        step.backward(grad, S.update_time(s), global_step=s)
    run(0, 10)
    torch.cuda.synchronize()
    t = time.perf_counter()
    run(10, 10 + K)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / K
    emit("pipelined training step, %s (dim 64, Zipf batch of 65 536, empty table at start)" % name, dt, None,
         lookups_updates_per_s=round(2 * B / dt))
    mt.close()


def bench_gather():
  n_inputs, rows_each, dims = 16, 65536, [16, 32, 64, 32] * 4
  total = sum(rows_each * d for d in dims)
  fused = torch.randn(total, dtype=torch.float32, device=DEV)
  offs, o = [], 0
  rng = np.random.default_rng(1)
  for d in dims:
    base = o + np.arange(rows_each, dtype=np.int64) * d
    rng.shuffle(base)
    offs.append(torch.from_numpy(base.astype(np.int32)).to(DEV))
    o += rows_each * d
  t = gpu_time(lambda: D.fused_gather_embeddings_by_input(fused, offs, dims))
  emit("fused_gather_embeddings_by_input (16 inputs x 65 536 rows, dims 16/32/64)", t,
       2 * 4 * total + 4 * n_inputs * rows_each)
  grads = [torch.randn((rows_each, d), dtype=torch.float32, device=DEV) for d in dims]
  t = gpu_time(lambda: D.fused_gather_embeddings_by_input_gradient(total, grads, offs, dims))
  emit("fused_gather_embeddings_by_input_gradient (same shapes; zero-fill + grouped in-order sums; MHTE_POOL_ATOMICS=1: float atomics)", t,
       3 * 4 * total + 4 * n_inputs * rows_each)


def bench_reduce():
  n, dim, batch = 1 << 20, 64, 65536
  rng = np.random.default_rng(2)
  idx = np.sort(rng.integers(0, batch, n)).astype(np.int64)
  vals = torch.randn((n, dim), dtype=torch.float32, device=DEV)
  idx_t = torch.from_numpy(idx).to(DEV)
  alg = n * dim * 4 + n * 8 + batch * dim * 4
  for name, fn in (("reduce_sum", D.reduce_sum), ("reduce_mean", D.reduce_mean),
                   ("reduce_sqrtn", D.reduce_sqrtn)):
    t = gpu_time(lambda f=fn: f(idx_t, vals, batch, indices_sorted=True))
    emit("%s, sorted indices (1M rows x dim 64 -> 65 536 rows, sequential order)" % name, t, alg)
  perm = torch.randperm(n, device=DEV)
  idx_u, vals_u = idx_t[perm].contiguous(), vals[perm].contiguous()
  t = gpu_time(lambda: D.reduce_sum(idx_u, vals_u, batch, indices_sorted=False))
  emit("reduce_sum, unsorted indices (grouped in-order sums; MHTE_POOL_ATOMICS=1: float atomics)", t, alg + batch * dim * 4)


def bench_layout():
  """fused_embedding_to_layout, the GENERAL form: 8 pooled features (1-4 fids per batch row, Zipf row
  choice: hot rows are referenced from hundreds of batch rows), dim 32 + bias, CONCAT + ADDN."""
  rng = np.random.default_rng(3)
  batch, nfeat, rows, dim = 65536, 8, 1 << 18, 33
  P_, OT, S_ = D.PoolingType, D.OutType, D.SliceConfig
  names = ["f%02d" % i for i in range(nfeat)]
  feats = {n: D.FeatureConfig("t%d" % i, P_.SUM if i % 2 == 0 else P_.MEAN, [1, dim - 1]) for i, n in enumerate(names)}
  outs = {"bias": D.OutConfig([S_(n, 0, 1) for n in names], OT.ADDN, [[-1, 1]]),
          "vec": D.OutConfig([S_(n, 1, dim) for n in names], OT.CONCAT, [[-1, nfeat * (dim - 1)]])}
  cfgs = D.FeatureConfigs(feats, outs)
  fid_offset, feature_offset, nfl_offset = [], [], []
  pos = 0
  for i, n in enumerate(names):
    nfl_offset.append(len(feature_offset))
    cnt = rng.integers(1, 5, batch)
    starts = pos + np.concatenate([[0], np.cumsum(cnt)[:-1]])
    feature_offset.extend(starts.tolist())
    r = (rng.zipf(1.2, int(cnt.sum())) - 1) % rows
    fid_offset.append((np.uint64(i) << np.uint64(32)) | r.astype(np.uint64))
    pos += int(cnt.sum())
  fo_np = np.concatenate(fid_offset)
  fo = torch.tensor(fo_np.view(np.int64)).to(DEV)
  fe = torch.tensor(np.array(feature_offset, dtype=np.int32)).to(DEV)
  nf = torch.tensor(np.array(nfl_offset, dtype=np.uint32).view(np.int32)).to(DEV)
  embs = [torch.randn((rows, dim), dtype=torch.float32, device=DEV) for _ in names]
  n_fid = int(fo_np.size)
  t = gpu_time(lambda: D.fused_embedding_to_layout(embs, fo, fe, nf, batch, cfgs), reps=10)
  out_bytes = batch * (1 + nfeat * (dim - 1)) * 4
  emit("fused_embedding_to_layout, general form (8 pooled features, %d fids, batch 65 536, dim 33)" % n_fid, t,
       n_fid * (8 + dim * 4) + out_bytes)
  outs_t = D.fused_embedding_to_layout(embs, fo, fe, nf, batch, cfgs)
  tg = [torch.randn_like(o) for o in outs_t]
  t = gpu_time(lambda: D.fused_embedding_to_layout_grad(embs, fo, fe, nf, batch, tg, cfgs), reps=10)
  emit("fused_embedding_to_layout gradient, general form (zero-fill + grouped in-order sums; MHTE_POOL_ATOMICS=1: float atomics)", t,
       n_fid * (8 + dim * 4) + out_bytes + nfeat * rows * dim * 4, distinct_rows=int(np.unique(fo_np).size))


def bench_optimizers():
  B, dim = 65536, 64
  opts = [
      ("sgd", entry.SgdOptimizer(0.01)),
      ("adagrad", entry.AdagradOptimizer(0.01, 0.1)),
      ("ftrl", entry.FtrlOptimizer(0.01, 0.1, 1.0, l1_regularization=0.001, l2_regularization=0.001)),
      ("momentum", entry.MomentumOptimizer(0.01)),
      ("adadelta", entry.AdadeltaOptimizer(0.01)),
      ("rmsprop", entry.RmspropOptimizer(0.01)),
      ("rmspropv2", entry.RmspropOptimizer(0.01, v2=True)),
      ("adam", entry.AdamOptimizer(0.01)),
      ("amsgrad", entry.AdamOptimizer(0.01, amsgrad=True)),
  ]
  ids = (torch.arange(B, dtype=torch.int64, device=DEV) * 2654435761 % (1 << 40)) | (1 << 48)
  g = torch.randn((B, dim), dtype=torch.float32, device=DEV) * 0.01
  from monolith_amd import _lib
  for name, opt in opts:
    mt = make_table(opt, dim, 4 * B, "opt_" + name)
    lr = mt.learning_rate
    fn = lambda: mt.table_optimize_n("emb", ids, None, g, lr, 1000, 3, flags=_lib.MHTE_IDS_UNIQUE)  # noqa: E731
    fn()  # inserts
    t = gpu_time(fn)
    state = {"sgd": 0, "adagrad": 1, "ftrl": 2, "momentum": 1, "adadelta": 2, "rmsprop": 1,
             "rmspropv2": 1, "adam": 2, "amsgrad": 3}[name]
    alg = B * (8 + 72 + 4 + 4 * dim + 2 * 4 * dim * (1 + state))
    emit("op-level update of 65 536 resident unique ids, %s (dim 64)" % name, t, alg)
    mt.close()


def main():
  torch.cuda.set_device(0)
  which = sys.argv[1:] or ["optimizers", "gather", "reduce", "evict", "filter", "checkpoint"]
  import gc
  for w in which:
    gc.collect()
    gc.disable()   # (a generation-2 pass of the interpreter is ~40 ms: it would land in a 60-step window)
    {"optimizers": bench_optimizers, "step_optimizers": bench_step_optimizers, "gather": bench_gather,
     "reduce": bench_reduce, "layout": bench_layout,
     "evict": bench_evict, "filter": bench_filter_step, "checkpoint": bench_checkpoint}[w]()
  md = ["| Measurement | time | algorithmic bytes | GB/s | of 8 TB/s | notes |", "|---|---|---|---|---|---|"]
  for r in LINES:
    notes = ", ".join("%s=%s" % (k, v) for k, v in r.items()
                      if k not in ("name", "us", "alg_bytes", "GBps", "hbm_frac"))
    us = r["us"]
    tm = "%.1f us" % us if us < 1e4 else "%.2f s" % (us / 1e6)
    md.append("| %s | %s | %s | %s | %s | %s |" % (
        r["name"], tm, r.get("alg_bytes", ""), r.get("GBps", ""),
        ("%.1f %%" % (100 * r["hbm_frac"])) if "hbm_frac" in r else "", notes))
  out = os.environ.get("NEXT_ROWS_MD")
  if out:
    with open(out, "w") as f:
      f.write("\n".join(md) + "\n")


if __name__ == "__main__":
  main()
