"""Generates tests/golden/*.npz from the REFERENCE's own code (oracle/_ref: the reference's
cuckoohash_map.hpp + avx_utils.h compiled in place from /root/reference with the fixed-hash shim).
Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py

Each fixture is a seeded op sequence plus what the reference produced:
  table_<name>.npz : ops (ids / values per step) -> final dump (ids, bucket*4+slot positions,
                     timestamps, rows incl. optimizer state), per-step lookup outputs, sizes.
The HIP engine (tests/test_parity_gpu.py) and the C restatement (tests/test_oracle.py) are both
checked against these files; the GPU box never sees /root/reference.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402
from monolith_amd import synthetic as S  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def table_fixture(name, dim, opt, steps, batch, universe, dist, lr, init_acc=0.1, wd=0.0,
                  initial_capacity=1):
  """Reference semantics of a training loop on ONE table: per step, worker-side dedup ->
  lookup(unique) -> duplicate-grad sum -> Optimize(unique)  (== RefPs with P=1)."""
  assert O.ref_available(), "needs oracle/_ref (build container)"
  ps = O.RefPs(1, dim, opt, init_acc, wd, 0.0, initial_capacity, avx=False)
  ids_all, emb_all, nuniq = [], [], []
  for s in range(steps):
    ids = S.id_batch(s, batch, universe, dist)
    g = S.grad_batch(s, batch, dim)
    emb, u = ps.step(ids, g, lr, S.update_time(s))
    ids_all.append(ids)
    emb_all.append(emb)
    nuniq.append(u)
  # final state: every id ever seen, looked up through the reference map
  probe = np.unique(np.concatenate(ids_all))
  final, hits = ps.lookup(probe)
  np.savez_compressed(
      os.path.join(OUT, "table_%s.npz" % name), dim=dim, opt=opt, steps=steps, batch=batch,
      universe=universe, dist=dist, lr=np.float32(lr), init_acc=np.float32(init_acc),
      wd=np.float32(wd), n_unique=np.array(nuniq), size=ps.size(), probe_ids=probe,
      final_rows=final, step_emb_first=np.stack([e[:64] for e in emb_all]),
      step_emb_sum=np.stack([e.astype(np.float64).sum(0) for e in emb_all]))
  print(name, "size", ps.size(), "uniq/step", nuniq[:3], "...")


def placement_fixture():
  """Sequential inserts into a PRE-SIZED reference map: exact (bucket, slot) of every key, rows
  and timestamps, including keys that needed BFS displacement."""
  rng = np.random.default_rng(7)
  n = 3000
  ids = rng.integers(-2**62, 2**62, n)
  cap = 4096  # hashpower 10 -> 4096 slots; 3000 keys = load 0.73: exercises displacement
  t = O.RefTable(4, O.OPT_SGD, initial_capacity=cap)
  vals = rng.standard_normal((n, 4)).astype(np.float32)
  for i in range(n):
    t.assign(ids[i:i + 1], vals[i:i + 1], 100 + i)
  d_ids, d_pos, d_ts, d_rows = t.dump()
  assert t.hashpower() == 10
  np.savez_compressed(os.path.join(OUT, "placement_seq.npz"), ids=ids, vals=vals, cap=cap,
                      dump_ids=d_ids, dump_pos=d_pos, dump_ts=d_ts, dump_rows=d_rows)
  print("placement: hp", t.hashpower(), "size", t.size())


def adagrad_fixture():
  """avx_utils.h arithmetic on random inputs, scalar and AVX flavours."""
  rng = np.random.default_rng(3)
  num = rng.standard_normal(224).astype(np.float32)
  norm = (rng.random(224).astype(np.float32) + 0.1)
  grad = rng.standard_normal(224).astype(np.float32)
  out = {}
  for wd in (0.0, 0.1):
    for avx in (False, True):
      n, m = O.ref_adagrad(num, norm, grad, 0.05, wd, avx=avx)
      out["num_wd%g_avx%d" % (wd, avx)] = n
      out["norm_wd%g_avx%d" % (wd, avx)] = m
  np.savez_compressed(os.path.join(OUT, "adagrad_kat.npz"), num=num, norm=norm, grad=grad,
                      lr=np.float32(0.05), **out)
  print("adagrad kat ok")


def adagrad_avx_fixture():
  """The reference AS BUILT (.bazelrc:63-68: -mavx2 -mfma, AdagradOptimizer::Optimize ->
  avx_utils.h:96-119) on ONE id of dim 64 and ONE of dim 27 (three fused blocks + a tail of 3),
  wd = 0.1: 12 Optimize() calls through the reference's own NewAdagradOptimizer object compiled in
  place (oracle/ref_opt_driver.cc).  Weights and accumulators after every call."""
  out = {}
  for dim in (64, 27):
    rng = np.random.default_rng(100 + dim)
    r = O.RefOptimizer(O.OPT_ADAGRAD, dim, (0.1, 0.1), avx=True)
    g = (rng.standard_normal((12, dim)) * 0.5).astype(np.float32)
    nums, norms = [], []
    for s in range(12):
      n, c = r.optimize(g[s], 0.05)
      nums.append(n)
      norms.append(c)
    out["grad_d%d" % dim] = g
    out["num_d%d" % dim] = np.stack(nums)
    out["norm_d%d" % dim] = np.stack(norms)
  np.savez_compressed(os.path.join(OUT, "adagrad_avx_kat.npz"), lr=np.float32(0.05), init_acc=np.float32(0.1),
                      wd=np.float32(0.1), **out)
  print("adagrad avx kat ok")


if __name__ == "__main__":
  if len(sys.argv) > 1 and sys.argv[1] == "adagrad_avx":   # (only this fixture)
    adagrad_avx_fixture()
    sys.exit(0)
  table_fixture("sgd_d8_uniform", 8, O.OPT_SGD, 6, 4096, 20000, "uniform", 0.01)
  table_fixture("adagrad_d16_zipf", 16, O.OPT_ADAGRAD, 8, 8192, 10**6, "zipf", 0.001)
  table_fixture("adagrad_d64_zipf", 64, O.OPT_ADAGRAD, 4, 8192, 10**9, "zipf", 0.001)
  placement_fixture()
  adagrad_fixture()
  adagrad_avx_fixture()
