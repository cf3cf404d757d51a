"""CPU suite, world_size 2 over gloo — a test OF THE CHECKER, not of the product.

oracle/shard_protocol.py restates the wire protocol of the C++ sharded multi-table step (whole-batch
peer blocks whose headers carry the per-table counts, one exchange per direction for ALL tables, row
slot s <-> id slot s, owners applying the senders' blocks in rank order).  Here that restatement runs
between two processes against a single-process run of the reference semantics on the oracle: it pins
the protocol model the GPU tests compare the product with, and that a two-process gloo rendezvous
works on this image.  The only product symbol involved is ``shard_block_geometry`` (checked equal to
the oracle's ``block_geometry``; the GPU tests pin it to the library's own numbers).

The PRODUCT's step between processes is tests/test_shard_ipc_gpu.py (separate processes on the GPU,
peer stores and — on a node with >= 2 devices — RCCL, against the oracle's replay) and
tests/test_shard_step_gpu.py (N ranks in one process); the product's torch.distributed form of the
step over gloo on CPU tensors is tests/test_sharded_gloo.py."""
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
  from oracle import shard_protocol as SP
  return SP.unique_sum(O, ids, g, d)


def _worker(rank, world, port, out_dir, grad_bits=32):
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  import oracle as O
  from oracle import shard_protocol as SP
  from monolith_amd.distributed_ps_sync import shard_block_geometry
  geo = SP.block_geometry(DIMS, BATCH, world)
  assert geo == shard_block_geometry(DIMS, BATCH, world)   # the product lays its blocks out the same way
  T = len(DIMS)
  mine = _tables()                       # the ids this rank owns

  def exchange(blocks):
    """blocks [world, n] -> block p goes to peer p, row p of the result came from peer p."""
    inp = torch.from_numpy(np.ascontiguousarray(blocks))
    out = torch.empty_like(inp)
    dist.all_to_all_single(out, inp)
    return out.numpy()

  embs = []
  for step in range(STEPS):
    batches = [_batch(rank, step, t) for t in range(T)]
    embs += SP.rank_step(O, mine, DIMS, LRS, geo, world, batches, 1_700_000_000 + step, exchange,
                         grad_bits=grad_bits)
  np.savez(os.path.join(out_dir, "rank%d.npz" % rank), *embs,
           **{"dump%d" % t: np.concatenate([np.sort(mine[t].dump()[0])]) for t in range(T)},
           **{"rows%d" % t: mine[t].lookup(np.sort(mine[t].dump()[0]))[0] for t in range(T)})
  dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("grad_bits", [32, 16])
def test_two_ranks_over_gloo_match_the_single_process_reference(tmp_path, grad_bits):
  """grad_bits = 16: the fp16 gradient wire — the block really crosses gloo as 2-byte values, and the
  single-process reference rounds each sender's per-id sums the same way before applying them."""
  import oracle as O
  world = 2
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  mp.spawn(_worker, args=(world, port, str(tmp_path), grad_bits), nprocs=world, join=True)
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
        if grad_bits == 16:
          gu = gu.astype(np.float16).astype(np.float32)
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
