import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle as O
from monolith_amd import entry
from monolith_amd.multi_hash_table_ops import MultiHashTable
L = O.lib()
def cfg(dim, **kw):
  return entry.make_table_config([entry.CombineAsSegment(dim, entry.ZerosInitializer(), entry.SgdOptimizer(1.0))], entry.CuckooHashTableConfig(**kw))
def analyze(mt, ids, v, tag):
  st = mt.stats("a"); hp = st.hashpower
  d_ids, d_pos, d_ts, d_rows = mt.dump("a")
  d_ids = d_ids.cpu().numpy(); d_pos = d_pos.cpu().numpy()
  print(tag, "size", st.size, "hp", hp, "rows", st.rows_allocated, "dump n", d_ids.size, "unique in dump", np.unique(d_ids).size, "expected", np.unique(ids).size)
  got = mt.lookup({"a": torch.from_numpy(ids).cuda()})["a"].cpu().numpy()
  bad = np.where((got != v).any(1))[0]
  print(" bad lookups:", bad.size, bad[:10])
  posmap = {}
  for k, p in zip(d_ids, d_pos): posmap.setdefault(int(k), []).append(int(p))
  dups = {k: p for k, p in posmap.items() if len(p) > 1}
  print(" dup keys in dump:", len(dups), list(dups.items())[:5])
  nin = 0
  for b in bad[:10]:
    k = int(ids[b]); hv = L.mo_hash(k); i1 = hv & ((1 << hp) - 1); i2 = L.mo_alt_index(hp, L.mo_partial(hv), i1)
    print("  id", k, "idx", b, "i1", i1, "i2", i2, "dump pos", [(p >> 2, p & 3) for p in posmap.get(k, [])], "got", got[b][:3], "exp", v[b][:3])
  invalid = 0
  for k, p in zip(d_ids[:200000], d_pos[:200000]):
    hv = L.mo_hash(int(k)); i1 = hv & ((1 << hp) - 1); i2 = L.mo_alt_index(hp, L.mo_partial(hv), i1)
    if (p >> 2) not in (i1, i2): invalid += 1
  print(" invalid placements:", invalid)

rng = np.random.default_rng(1)
n = 300000
ids = rng.permutation(np.arange(1, 4 * n, 4, dtype=np.int64) * 7919)[:n]
v = rng.standard_normal((n, 8)).astype(np.float32)
mt = MultiHashTable.from_configs({"a": cfg(8)}, name_suffix="dbg1")
half = n // 2
mt.assign({"a": (torch.from_numpy(ids[:half]).cuda(), torch.from_numpy(v[:half]).cuda())})
analyze(mt, ids[:half], v[:half], "after first half")
mt.assign({"a": (torch.from_numpy(ids[half:]).cuda(), torch.from_numpy(v[half:]).cuda())})
analyze(mt, ids, v, "after second half")
# no-growth variant
mt2 = MultiHashTable.from_configs({"a": cfg(8, initial_capacity=1 << 21)}, name_suffix="dbg2")
mt2.assign({"a": (torch.from_numpy(ids[:half]).cuda(), torch.from_numpy(v[:half]).cuda())})
mt2.assign({"a": (torch.from_numpy(ids[half:]).cuda(), torch.from_numpy(v[half:]).cuda())})
analyze(mt2, ids, v, "presized")
