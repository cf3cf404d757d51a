"""Debug: table size over pipelined steps with an admission filter attached (next_rows_bench reported a
negative size after 70 steps of 65 536 Zipf ids)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from monolith_amd import entry, synthetic as S
from monolith_amd.fused_step import SparseStep
from monolith_amd.multi_hash_table_ops import HashFilter, MultiHashTable
DEV = torch.device("cuda", 0)

def run(B, V, K, thr, exact=False, every=1):
  dim = 64
  flt = HashFilter(capacity=1 << 24, split_num=7)
  occ = entry.SlotOccurrenceThresholdConfig(default_occurrence_threshold=thr)
  slots = 4
  rows_cap = (K + 20) * B
  while slots * 0.5 < rows_cap: slots *= 2
  cfg = entry.make_table_config([entry.CombineAsSegment(dim, entry.ZerosInitializer(), entry.AdagradOptimizer(0.001, 0.1))],
      entry.CuckooHashTableConfig(initial_capacity=slots, reserve_rows=rows_cap), slot_occurrence_threshold_config=occ)
  mt = MultiHashTable.from_configs({"emb": cfg}, name_suffix="dbg%d_%d" % (B, V), hash_filter=flt)
  ids_host = [S.id_batch(s, B, V) for s in range(K + 2)]
  ids_all = [torch.from_numpy(x).to(DEV) for x in ids_host]
  grad = torch.from_numpy(S.grad_batch(0, B, dim)).to(DEV)
  step = SparseStep(mt, "emb", B, exact_order=exact)
  seen, present = {}, set()
  for s in range(K):
    step.forward(ids_all[s], next_ids=ids_all[s + 1])
    step.backward(grad, S.update_time(s))
    u, c = np.unique(ids_host[s], return_counts=True)
    for i, n in zip(u.tolist(), c.tolist()):
      if i in present: continue
      c0 = seen.get(i, 0)
      seen[i] = min(15, c0 + min(15, n))
      if c0 >= thr: present.add(i)
    if (s + 1) % every and s != K - 1:
      continue
    sz = mt.size("emb")
    st = mt.stats("emb")
    print("B %d V %g step %d size %d expected %d  stats.size %s" % (B, V, s, sz, len(present), getattr(st, "size", None)), flush=True)
    if sz != len(present) and s > 6:
      break
  mt.close()

run(65536, 1e9, 70, 2, every=10)
run(65536, 1e9, 70, 2, every=100)
