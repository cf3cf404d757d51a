import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, ctypes as C
from monolith_amd import entry, _lib
from monolith_amd.fused_step import SparseStep
from monolith_amd.multi_hash_table_ops import MultiHashTable
n, dim, steps = 1025, 16, 3
def batch(s_):
  b = np.full(n, 4242, dtype=np.int64); b[-1] = 4243 + s_; return b
cfg = entry.make_table_config([entry.CombineAsSegment(dim, entry.ZerosInitializer(), entry.AdagradOptimizer(0.05, 0.1))])
for exact in (False, True):
  mt = MultiHashTable.from_configs({"emb": cfg}, name_suffix="dbg%d" % exact)
  step = SparseStep(mt, "emb", n, exact_order=exact)
  dev = [torch.from_numpy(batch(s)).cuda() for s in range(steps + 1)]
  for s_ in range(steps):
    g = torch.ones((n, dim), dtype=torch.float32, device="cuda")
    step.forward(dev[s_], next_ids=dev[s_ + 1])
    step.backward(g, 100 + s_)
    torch.cuda.synchronize()
    print("exact", exact, "step", s_, "size", mt.size("emb"), "stats", mt.stats("emb") if hasattr(mt, "stats") else None)
