import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import test_tf_written_gpu as T
from monolith_amd.distributed_ps_sync import ShardedMultiStep
from monolith_amd.fused_step import MultiSparseStep
m = [x for x in T.MANIFEST if x["dim"] == 33][0]
base = os.path.join(T.GOLD, m["basename"])
recs = {}
for r in T.file_records(base):
  e = T.P.EntryDump.FromString(r); recs[e.id] = e
ids_all = np.array(sorted(recs), dtype=np.int64)
rng = np.random.default_rng(33); B = 256
batches = [np.concatenate([rng.choice(ids_all, B - 16), rng.integers(1, 2**40, 16)]).astype(np.int64) for _ in range(3)]
grads = [(rng.standard_normal((B, 33)) * 0.1).astype(np.float32) for _ in range(2)]
res = {}
probe = np.unique(np.concatenate(batches[:2]))
for kind in ("sharded", "multi", "multi_exact", "op"):
  mt = T.table_for(m); mt.restore_table("t", base)
  if kind == "op":
    for s in range(2):
      u, inv = np.unique(batches[s], return_inverse=True)
      gu = np.zeros((u.size, 33), np.float32)
      for p_ in np.argsort(inv, kind="stable"):
        gu[inv[p_]] = gu[inv[p_]] + grads[s][p_]
      mt.apply_gradients({"t": (torch.from_numpy(u).cuda(), torch.from_numpy(gu).cuda())}, req_time=1000 + s)
  else:
    step = ShardedMultiStep(mt, B) if kind == "sharded" else MultiSparseStep(mt, B, exact_order=(kind == "multi_exact"))
    rag = [mt.get_ragged_id({"t": torch.from_numpy(b).cuda()}) for b in batches]
    for s in range(2):
      step.forward(rag[s], rag[s + 1]); step.backward(torch.from_numpy(grads[s].ravel()).cuda(), 1000 + s)
    step.close()
  res[kind] = mt.lookup({"t": torch.from_numpy(probe).cuda()})["t"].cpu().numpy()
for a in res:
  for b in res:
    if a < b:
      d = res[a] != res[b]
      print(a, b, "mismatch elems", int(d.sum()), "cols", sorted(set(np.where(d)[1].tolist()))[:10], "rows", len(set(np.where(d)[0].tolist())))
u, c = np.unique(batches[0], return_counts=True); print("max dup", c.max())
