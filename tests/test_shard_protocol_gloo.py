"""CPU suite, world_size 2 over gloo: the wire protocol of the C++ sharded multi-table step
(csrc/mhte_shard_host.h / mhte_shard_kernels.h) restated in numpy and run between two processes —
fixed-capacity peer blocks whose headers carry the per-table counts (no size exchange), one
exchange per direction for ALL tables, row slot s <-> id slot s, owners applying the senders'
blocks in rank order — against a single-process run of the reference semantics on the oracle.  The
block geometry comes from the product (``shard_block_geometry``, which the GPU tests pin to the
library's own numbers); the HIP kernels themselves are checked against the same oracle semantics
in tests/test_shard_step_gpu.py (N ranks in one process on the GPU box)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DIMS = [8, 4, 16]          # three tables
LRS = [0.05, 0.1, 0.02]
STEPS, BATCH = 3, 500


def _tables():
  import oracle as O
  return [O.Table(O.segment(d, O.OPT_ADAGRAD, p=(0.1, 0.0)), 1) for d in DIMS]


def _batch(rank, step, t):
  rng = np.random.default_rng(977 * rank + 31 * step + t)
  n = BATCH if not (rank == 1 and step == 1 and t == 2) else 0      # ragged: an empty table
  ids = (rng.zipf(1.3, n) % 300).astype(np.int64) | ((t + 1) << 48)
  g = rng.standard_normal((n, DIMS[t])).astype(np.float32)
  return ids, g


def _unique_sum(O, ids, g, d):
  if ids.size == 0:
    return np.zeros(0, np.int64), np.zeros((0, d), np.float32), np.zeros(0, np.int64)
  uk, _, vo, vos, _ = O.unique_key_with_value_and_offset(ids, [0, ids.size], [d])
  gu = O.fill_with_offset_map_gradient(np.arange(uk.size), [0, uk.size], g.ravel(), vo, vos,
                                       [d]).reshape(-1, d)
  index = {int(k): i for i, k in enumerate(uk)}
  inv = np.array([index[int(x)] for x in ids], dtype=np.int64)
  return uk, gu, inv


def _exchange(blocks):
  """blocks [world, n] -> block p goes to peer p, row p of the result came from peer p."""
  out = torch.empty_like(blocks)
  dist.all_to_all_single(out, blocks)
  return out


def _worker(rank, world, port, out_dir):
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  import oracle as O
  from monolith_amd.distributed_ps_sync import shard_block_geometry
  geo = shard_block_geometry(DIMS, BATCH, world)
  T, cap = len(DIMS), geo["cap"]
  mine = _tables()                       # the ids this rank owns
  embs = []
  for step in range(STEPS):
    batches = [_batch(rank, step, t) for t in range(T)]
    # ---- sender: dedup, pack the distinct ids into the owners' blocks (counts in the header)
    ids_send = np.zeros((world, geo["ids_block"]), np.int64)
    slot_of, uniq = [], []
    for t, (ids, g) in enumerate(batches):
      uk, gu, inv = _unique_sum(O, ids, g, DIMS[t])
      owner = np.mod(uk, world)
      slot = np.zeros(uk.size, np.int64)
      for u in range(uk.size):
        p = int(owner[u])
        s = int(ids_send[p, t])
        assert s < cap
        ids_send[p, t] = s + 1
        ids_send[p, geo["id_off"][t] + s] = uk[u]
        slot[u] = p * geo["rows_block"] + geo["row_off"][t] + s * DIMS[t]
      slot_of.append(slot)
      uniq.append((uk, gu, inv))
    ids_recv = _exchange(torch.from_numpy(ids_send)).numpy()          # exchange 1: id blocks
    # ---- owner: rows of the received ids (no insert) into the row blocks
    rows_own = np.zeros((world, geo["rows_block"]), np.float32)
    for p in range(world):
      for t in range(T):
        n = int(ids_recv[p, t])
        if n:
          e, _ = mine[t].lookup(ids_recv[p, geo["id_off"][t]:geo["id_off"][t] + n])
          rows_own[p, geo["row_off"][t]:geo["row_off"][t] + n * DIMS[t]] = e.ravel()
    rows_back = _exchange(torch.from_numpy(rows_own)).numpy().ravel()  # exchange 2: rows
    # ---- sender: rows -> occurrences
    for t, (ids, g) in enumerate(batches):
      uk, gu, inv = uniq[t]
      d = DIMS[t]
      ur = np.stack([rows_back[o:o + d] for o in slot_of[t]]) if uk.size else np.zeros((0, d), np.float32)
      embs.append(ur[inv] if ids.size else np.zeros((0, d), np.float32))
    # ---- sender: per-id gradient sums into the row slots; owner: peers applied in rank order
    grad_send = np.zeros(world * geo["rows_block"], np.float32)
    for t in range(T):
      uk, gu, inv = uniq[t]
      for u in range(uk.size):
        grad_send[slot_of[t][u]:slot_of[t][u] + DIMS[t]] = gu[u]
    grad_recv = _exchange(torch.from_numpy(grad_send.reshape(world, -1))).numpy()   # exchange 3
    for p in range(world):
      for t in range(T):
        n = int(ids_recv[p, t])
        if n:
          ids_p = ids_recv[p, geo["id_off"][t]:geo["id_off"][t] + n]
          g_p = grad_recv[p, geo["row_off"][t]:geo["row_off"][t] + n * DIMS[t]].reshape(n, DIMS[t])
          mine[t].optimize(ids_p, g_p, [LRS[t]], 1_700_000_000 + step)
  np.savez(os.path.join(out_dir, "rank%d.npz" % rank), *embs,
           **{"dump%d" % t: np.concatenate([np.sort(mine[t].dump()[0])]) for t in range(T)},
           **{"rows%d" % t: mine[t].lookup(np.sort(mine[t].dump()[0]))[0] for t in range(T)})
  dist.destroy_process_group()


def test_two_ranks_over_gloo_match_the_single_process_reference(tmp_path):
  import oracle as O
  world = 2
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
  # the reference semantics in one process: one table per feature (owners hold disjoint ids)
  ref = _tables()
  T = len(DIMS)
  exp_embs = {r: [] for r in range(world)}
  for step in range(STEPS):
    for r in range(world):
      for t in range(T):
        ids, _ = _batch(r, step, t)
        exp_embs[r].append(ref[t].lookup(ids)[0] if ids.size else np.zeros((0, DIMS[t]), np.float32))
    for r in range(world):                 # senders in rank order, one optimizer application each
      for t in range(T):
        ids, g = _batch(r, step, t)
        uk, gu, _ = _unique_sum(O, ids, g, DIMS[t])
        if uk.size:
          ref[t].optimize(uk, gu, [LRS[t]], 1_700_000_000 + step)
  got = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
  for r in range(world):
    for k, e in enumerate(exp_embs[r]):
      np.testing.assert_array_equal(got[r]["arr_%d" % k], e, err_msg="rank %d embedding %d" % (r, k))
  for t in range(T):
    all_ids = np.sort(ref[t].dump()[0])
    owned = [got[r]["dump%d" % t] for r in range(world)]
    np.testing.assert_array_equal(np.sort(np.concatenate(owned)), all_ids)
    for r in range(world):
      assert np.all(np.mod(owned[r], world) == r)
      np.testing.assert_array_equal(got[r]["rows%d" % t], ref[t].lookup(owned[r])[0])
