"""CPU suite, world_size 2 over gloo.  What is under test is tests/torch_sharded_step.py — round 1's
torch.distributed form of the sharded step, a TEST HARNESS since round 6 (and bench.py's last-resort
fallback), NOT the C++ step the product ships (csrc/mhte_shard_host.h; that one runs between processes on
the GPU box: tests/test_shard_ipc_gpu.py, and its wire protocol between two gloo processes:
tests/test_shard_protocol_oracle_gloo.py).  Covered here: the sharding rule (fid mod N), the shard-major
stable packing, the four exchanges and the un-permute / scatter of that harness against a single-process
run of the reference semantics, with a CPU stand-in of the local engine built on the test oracle."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DIM = 8
STEPS = 3
BATCH = 600


class OracleBackend:
  """LocalBackend stand-in: oracle table + numpy dedup / packing (reference semantics)."""

  def __init__(self):
    import oracle as O
    self.O = O
    self.dim = DIM
    self.t = O.Table(O.segment(DIM, O.OPT_ADAGRAD, p=(0.1, 0.0)), 1)
    self._inv = [None, None]   # per state slot: the next batch is dispatched beside this one
    self._slot = 0

  def use_slot(self, i):
    self._slot = i & 1

  @property
  def inv(self):
    return self._inv[self._slot]

  @inv.setter
  def inv(self, v):
    self._inv[self._slot] = v

  def dedup(self, ids):
    a = ids.numpy()
    uk, _, _, _, _ = self.O.unique_key_with_value_and_offset(a, [0, a.size], [1])
    index = {int(k): i for i, k in enumerate(uk)}
    self.inv = np.array([index[int(x)] for x in a], dtype=np.int64)
    buf = np.zeros(a.size, np.int64)
    buf[:uk.size] = uk
    return torch.from_numpy(buf), torch.tensor([uk.size], dtype=torch.int32)

  def partition(self, unique_ids, n_unique, num_shards):
    u = int(n_unique.item())
    uk = unique_ids.numpy()[:u]
    shard = np.mod(uk, num_shards)
    order = np.argsort(shard, kind="stable")          # shard-major, first-occurrence order inside
    send_ids = np.zeros(unique_ids.numel(), np.int64)
    send_ids[:u] = uk[order]
    send_pos = np.zeros(unique_ids.numel(), np.int32)
    send_pos[order] = np.arange(u, dtype=np.int32)
    counts = np.bincount(shard, minlength=num_shards).astype(np.int32)
    return torch.from_numpy(send_ids), torch.from_numpy(send_pos), torch.from_numpy(counts)

  def scatter(self, rows, send_pos, n_out):
    return rows[send_pos.long()[self.inv]]

  def sum(self, grads, send_pos):
    g = grads.numpy()
    out = np.zeros((g.shape[0], DIM), np.float32)
    sp = send_pos.numpy()
    for p, u in enumerate(self.inv):                  # occurrence order, like the reference op
      out[sp[u]] += g[p]
    return torch.from_numpy(out)

  def owner_lookup(self, ids):
    e, _ = self.t.lookup(ids.numpy())
    return torch.from_numpy(e)

  def owner_apply(self, ids, grads, update_time, global_step):
    a, g = ids.numpy(), grads.numpy()
    uk, _, _, _, _ = self.O.unique_key_with_value_and_offset(a, [0, a.size], [1])
    index = {int(k): i for i, k in enumerate(uk)}
    acc = np.zeros((uk.size, DIM), np.float32)
    for p, x in enumerate(a):
      acc[index[int(x)]] += g[p]
    self.t.optimize(uk, acc, [0.05], update_time)


def _batch(rank, step):
  rng = np.random.default_rng(1000 * rank + step)
  ids = rng.integers(0, 400, BATCH).astype(np.int64) | (1 << 48)
  g = rng.standard_normal((BATCH, DIM)).astype(np.float32)
  return ids, g


def _worker(rank, world, port, out_dir):
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  from tests.torch_sharded_step import ShardedEmbedding
  be = OracleBackend()
  se = ShardedEmbedding(be)
  embs = []
  batches = [torch.from_numpy(_batch(rank, s)[0]) for s in range(STEPS + 1)]
  for s in range(STEPS):
    _, g = _batch(rank, s)
    # odd steps also prepare the id dispatch of the next batch ahead (prefetch path)
    e = se.lookup(batches[s], next_ids=batches[s + 1] if s % 2 else None)
    embs.append(e.numpy().copy())
    se.apply_gradients(torch.from_numpy(g), 100 + s)
  d_ids, _, _, d_rows = be.t.dump()
  np.savez(os.path.join(out_dir, "rank%d.npz" % rank), embs=np.stack(embs), ids=d_ids, rows=d_rows)
  dist.destroy_process_group()


def test_two_rank_exchange_matches_single_table(tmp_path):
  world = 2
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  port = s.getsockname()[1]
  s.close()
  mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
  import oracle as O
  # single-process reference semantics: one table; per step every rank looks up the pre-update
  # rows, then per id the gradients of all ranks (each rank's duplicates summed first, then ranks
  # in rank order) are accumulated and applied once.
  t = O.Table(O.segment(DIM, O.OPT_ADAGRAD, p=(0.1, 0.0)), 1)
  exp_embs = [[], []]
  for st in range(STEPS):
    per_rank = []
    for r in range(world):
      ids, g = _batch(r, st)
      exp_embs[r].append(t.lookup(ids)[0])
      uk, _, vo, vos, _ = O.unique_key_with_value_and_offset(ids, [0, ids.size], [DIM])
      gu = O.fill_with_offset_map_gradient(np.arange(uk.size), [0, uk.size], g.ravel(), vo, vos,
                                           [DIM]).reshape(-1, DIM)
      per_rank.append((uk, gu))
    allk = np.concatenate([p[0] for p in per_rank])
    allg = np.concatenate([p[1] for p in per_rank])
    uk, _, vo, vos, _ = O.unique_key_with_value_and_offset(allk, [0, allk.size], [DIM])
    acc = O.fill_with_offset_map_gradient(np.arange(uk.size), [0, uk.size], allg.ravel(), vo, vos,
                                          [DIM]).reshape(-1, DIM)
    t.optimize(uk, acc, [0.05], 100 + st)
  got_ids, got_rows = [], []
  for r in range(world):
    z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
    np.testing.assert_array_equal(z["embs"], np.stack(exp_embs[r]))
    assert ((z["ids"] % world) == r).all()  # ownership: fid mod N
    got_ids.append(z["ids"])
    got_rows.append(z["rows"])
  got_ids = np.concatenate(got_ids)
  got_rows = np.concatenate(got_rows)
  e_ids, _, _, e_rows = t.dump()
  a, b = np.argsort(got_ids), np.argsort(e_ids)
  np.testing.assert_array_equal(got_ids[a], e_ids[b])
  np.testing.assert_array_equal(got_rows[a], e_rows[b])


# ------------------------------------------------------------------------------------------------
# GPU box: two ranks sharing the one GPU, HipBackend on both, collectives staged through host memory
# (gloo).  Exercises the HIP sender side (run dedup, shard packing, scatter, sum) and the owner side
# (lookup, fused sum + apply) under the real exchange code; only the transport differs from RCCL.
# ------------------------------------------------------------------------------------------------
import pytest  # noqa: E402

GPU_BATCH = 20000
GPU_DIM = 32
GPU_STEPS = 4


def _gpu_batch(rank, step):
  from monolith_amd import synthetic as S
  ids = S.id_batch(500 + 10 * step + rank, GPU_BATCH, 10**5, "zipf")
  g = S.grad_batch(500 + 10 * step + rank, GPU_BATCH, GPU_DIM)
  return ids, g


def _gpu_worker(rank, world, port, out_dir):
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  torch.cuda.set_device(0)
  from monolith_amd import entry
  from tests.torch_sharded_step import HipBackend, ShardedEmbedding
  from monolith_amd.multi_hash_table_ops import MultiHashTable
  cfg = entry.make_table_config([
      entry.CombineAsSegment(GPU_DIM, entry.ZerosInitializer(), entry.AdagradOptimizer(0.05, 0.1))
  ])
  mt = MultiHashTable.from_configs({"emb": cfg}, name_suffix="shard%d" % rank)
  se = ShardedEmbedding(HipBackend(mt, "emb"))
  embs = []
  batches = [torch.from_numpy(_gpu_batch(rank, s)[0]).cuda() for s in range(GPU_STEPS + 1)]
  for s in range(GPU_STEPS):
    _, g = _gpu_batch(rank, s)
    # the id dispatch of the next batch is prepared on a side stream (prefetch path), except once
    e = se.lookup(batches[s], next_ids=batches[s + 1] if s != 1 else None)
    embs.append(e.cpu().numpy().copy())
    se.apply_gradients(torch.from_numpy(g).cuda(), 100 + s)
  d_ids, _, _, d_rows = mt.dump("emb")
  np.savez(os.path.join(out_dir, "rank%d.npz" % rank), embs=np.stack(embs),
           ids=d_ids.cpu().numpy(), rows=d_rows.cpu().numpy())
  dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_one_gpu_hip_backend_matches_single_table(tmp_path):
  world = 2
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  port = s.getsockname()[1]
  s.close()
  mp.spawn(_gpu_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
  import oracle as O
  t = O.Table(O.segment(GPU_DIM, O.OPT_ADAGRAD, p=(0.1, 0.0)), 1)
  exp_embs = [[], []]
  for st in range(GPU_STEPS):
    per_rank = []
    for r in range(world):
      ids, g = _gpu_batch(r, st)
      exp_embs[r].append(t.lookup(ids)[0])
      uk, _, vo, vos, _ = O.unique_key_with_value_and_offset(ids, [0, ids.size], [GPU_DIM])
      gu = O.fill_with_offset_map_gradient(np.arange(uk.size), [0, uk.size], g.ravel(), vo, vos,
                                           [GPU_DIM]).reshape(-1, GPU_DIM)
      per_rank.append((uk, gu))
    allk = np.concatenate([p[0] for p in per_rank])
    allg = np.concatenate([p[1] for p in per_rank])
    uk, _, vo, vos, _ = O.unique_key_with_value_and_offset(allk, [0, allk.size], [GPU_DIM])
    acc = O.fill_with_offset_map_gradient(np.arange(uk.size), [0, uk.size], allg.ravel(), vo, vos,
                                          [GPU_DIM]).reshape(-1, GPU_DIM)
    t.optimize(uk, acc, [0.05], 100 + st)
  got_ids, got_rows = [], []
  for r in range(world):
    z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
    # heavy lists are summed as a fixed tree (fp32 re-association): 1e-5, north_star's bar
    np.testing.assert_allclose(z["embs"], np.stack(exp_embs[r]), rtol=0, atol=1e-5)
    assert ((z["ids"] % world) == r).all()  # ownership: fid mod N
    got_ids.append(z["ids"])
    got_rows.append(z["rows"])
  got_ids = np.concatenate(got_ids)
  got_rows = np.concatenate(got_rows)
  e_ids, _, _, e_rows = t.dump()
  a, b = np.argsort(got_ids), np.argsort(e_ids)
  np.testing.assert_array_equal(got_ids[a], e_ids[b])
  # embedding values: 1e-5 absolute (north_star's bar); the Adagrad accumulators next to them in
  # the row grow with the squared gradients, so they are held to the same bar relatively
  np.testing.assert_allclose(got_rows[a][:, :GPU_DIM], e_rows[b][:, :GPU_DIM], rtol=0, atol=1e-5)
  np.testing.assert_allclose(got_rows[a][:, GPU_DIM:], e_rows[b][:, GPU_DIM:], rtol=1e-5, atol=1e-5)
