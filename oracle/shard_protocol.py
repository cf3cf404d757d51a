"""TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.

The wire protocol of the id-sharded multi-table step (csrc/mhte_shard_host.h,
mhte_shard_kernels.h: the reference's sync-training exchange, native_training/
distributed_ps_sync.py:95-287 lookup, :289-490 apply_gradients) restated in numpy on oracle tables:
per peer one id block whose header words carry the per-table counts (no size exchange), one exchange
per direction for ALL tables, row slot s <-> id slot s, owners applying the senders' blocks in rank
order (one optimizer application per sender).  `exchange(blocks[world, n]) -> [world, n]` is the
transport (block p goes to peer p, row p of the result came from peer p): torch.distributed over
gloo in tests/test_shard_protocol_oracle_gloo.py.

What pins what: this restatement is checked against a single-process run of the reference semantics
(tests/test_shard_protocol_oracle_gloo.py, world_size 2); the PRODUCT's C++ step is checked against the
same single-process semantics by separate processes on the GPU (tests/test_shard_ipc_gpu.py) and by
N ranks in one process (tests/test_shard_step_gpu.py); block_geometry() is compared with the
product's shard_block_geometry and, on the GPU, with mhte_shard_step_info.
"""
import numpy as np


def block_geometry(dims, batch_per_table, world, ids_per_peer_table=0):
  """ShardStep::init's layout: cap id slots per (peer, table) (default: the whole batch), an id block
  int64[hdr + T * cap] with the counts in words [0, T), a row block float32[sum cap * dim]."""
  T, mb = len(dims), int(batch_per_table)
  c = int(ids_per_peer_table) if ids_per_peer_table > 0 else mb
  cap = (min(c, mb) + 3) & ~3
  hdr = (T + 7) & ~7
  row_off, rw = [], 0
  for d in dims:
    row_off.append(rw)
    rw += cap * int(d)
  idw = hdr + T * cap
  return {"cap": cap, "hdr": hdr, "id_off": [hdr + t * cap for t in range(T)], "row_off": row_off,
          "ids_block": (idw + 1) & ~1, "rows_block": rw}


def unique_sum(O, ids, g, d):
  """distinct ids in first-occurrence order, their gradient sums in occurrence order, occurrence ->
  distinct index (MonolithUniqueKeyWithValueAndOffset + FillWithOffsetMapGradient)."""
  if ids.size == 0:
    return np.zeros(0, np.int64), np.zeros((0, d), np.float32), np.zeros(0, np.int64)
  uk, _, vo, vos, _ = O.unique_key_with_value_and_offset(ids, [0, ids.size], [d])
  gu = O.fill_with_offset_map_gradient(np.arange(uk.size), [0, uk.size], g.ravel(), vo, vos,
                                       [d]).reshape(-1, d)
  index = {int(k): i for i, k in enumerate(uk)}
  inv = np.array([index[int(x)] for x in ids], dtype=np.int64)
  return uk, gu, inv


def rank_step(O, tables, dims, lrs, geo, world, batches, update_time, exchange, grad_bits=32):
  """One training step of one rank.  tables: this rank's oracle tables (the ids it owns); batches:
  [(ids, grads)] per table -> per-table per-occurrence embeddings of this rank's batch.
  grad_bits = 16: the optional fp16 gradient wire (mhte_shard_step_set_grad_bits; the reference's
  distributed_ps_sync.py:424-436): the SENDER rounds its per-id sums to binary16 — after summing in
  fp32 — the block crosses as 2-byte values, the owner widens them and applies."""
  T, cap = len(dims), geo["cap"]
  ids_send = np.zeros((world, geo["ids_block"]), np.int64)
  slot_of, uniq = [], []
  for t, (ids, g) in enumerate(batches):          # sender: dedup, pack into the owners' blocks
    uk, gu, inv = unique_sum(O, ids, g, dims[t])
    owner = np.mod(uk, world)
    slot = np.zeros(uk.size, np.int64)
    for u in range(uk.size):
      p = int(owner[u])
      s = int(ids_send[p, t])
      assert s < cap
      ids_send[p, t] = s + 1
      ids_send[p, geo["id_off"][t] + s] = uk[u]
      slot[u] = p * geo["rows_block"] + geo["row_off"][t] + s * dims[t]
    slot_of.append(slot)
    uniq.append((uk, gu, inv))
  ids_recv = exchange(ids_send)                                      # exchange 1: id blocks
  rows_own = np.zeros((world, geo["rows_block"]), np.float32)        # owner: rows, no insert
  for p in range(world):
    for t in range(T):
      n = int(ids_recv[p, t])
      if n:
        e, _ = tables[t].lookup(ids_recv[p, geo["id_off"][t]:geo["id_off"][t] + n])
        rows_own[p, geo["row_off"][t]:geo["row_off"][t] + n * dims[t]] = e.ravel()
  rows_back = exchange(rows_own).ravel()                             # exchange 2: rows
  embs = []
  for t, (ids, g) in enumerate(batches):                             # sender: rows -> occurrences
    uk, gu, inv = uniq[t]
    d = dims[t]
    ur = np.stack([rows_back[o:o + d] for o in slot_of[t]]) if uk.size else np.zeros((0, d), np.float32)
    embs.append(ur[inv] if ids.size else np.zeros((0, d), np.float32))
  grad_send = np.zeros(world * geo["rows_block"], np.float32)        # sender: sums into the row slots
  for t in range(T):
    uk, gu, inv = uniq[t]
    for u in range(uk.size):
      grad_send[slot_of[t][u]:slot_of[t][u] + dims[t]] = gu[u]
  if grad_bits == 16:
    grad_recv = exchange(grad_send.astype(np.float16).reshape(world, -1)).astype(np.float32)
  else:
    grad_recv = exchange(grad_send.reshape(world, -1))               # exchange 3: gradients
  for p in range(world):                                             # owner: peers in rank order
    for t in range(T):
      n = int(ids_recv[p, t])
      if n:
        ids_p = ids_recv[p, geo["id_off"][t]:geo["id_off"][t] + n]
        g_p = grad_recv[p, geo["row_off"][t]:geo["row_off"][t] + n * dims[t]].reshape(n, dims[t])
        tables[t].optimize(ids_p, g_p, [lrs[t]], update_time)
  return embs
