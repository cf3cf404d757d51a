"""Where the time of the sharded step goes at world 1 (MI355X): host enqueue time of forward /
backward against the stream's wall time, and the kernel-exact time of every tagged launch.
Usage: python scripts/shard_probe.py [--tables 1] [--dim 64] [--batch 65536] [--steps 200]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monolith_amd import _lib, entry, synthetic as S  # noqa: E402
from monolith_amd.distributed_ps_sync import ShardedMultiStep  # noqa: E402
from monolith_amd.fused_step import MultiSparseStep  # noqa: E402
from monolith_amd.multi_hash_table_ops import MultiHashTable, Ragged  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--tables", type=int, default=1)
  ap.add_argument("--dims", default="64")
  ap.add_argument("--batch", type=int, default=65536)
  ap.add_argument("--resident", type=int, default=1 << 21)
  ap.add_argument("--universe", type=int, default=1000000000)
  ap.add_argument("--steps", type=int, default=200)
  ap.add_argument("--which", default="shard,multi")
  args = ap.parse_args()
  T, B = args.tables, args.batch
  dev = torch.device("cuda", 0)
  dl = [int(x) for x in args.dims.split(",")]
  dims = [dl[i % len(dl)] for i in range(T)]
  names = ["f%02d" % (i + 1) for i in range(T)]

  def table(tag):
    cfgs = {}
    rows_cap = args.resident + (3 * args.steps + 64) * B
    slots = 4
    while slots * 0.5 < rows_cap:
      slots *= 2
    for i, n in enumerate(names):
      cfgs[n] = entry.make_table_config(
          [entry.CombineAsSegment(dims[i], entry.ZerosInitializer(), entry.AdagradOptimizer(0.001, 0.1))],
          entry.CuckooHashTableConfig(initial_capacity=slots, reserve_rows=rows_cap))
    return MultiHashTable.from_configs(cfgs, name_suffix=tag)

  nb = 2 * args.steps + 40
  ids_host = np.empty((nb, T * B), dtype=np.int64)
  for s in range(nb):
    for i in range(T):
      ids_host[s, i * B:(i + 1) * B] = S.id_batch(s * 64 + i + 1, B, args.universe // max(1, T), "zipf",
                                                  feature_slot=i + 1)
  ids_all = torch.from_numpy(ids_host).to(dev)
  splits = np.arange(T + 1, dtype=np.int64) * B
  rag = [Ragged(ids_all[s], splits) for s in range(nb)]
  gsz = B * sum(dims)
  grads = [torch.randn(gsz, device=dev) * 0.01 for _ in range(3)]
  out = torch.empty(gsz, dtype=torch.float32, device=dev)
  for which in args.which.split(","):
    mt = table(which)
    step = ShardedMultiStep(mt, B) if which == "shard" else MultiSparseStep(mt, B)
    for s in range(20):
      step.forward(rag[s], rag[s + 1], out=out)
      step.backward(grads[s % 3], S.update_time(s))
    torch.cuda.synchronize()
    tf = tb = 0.0
    t0 = time.perf_counter()
    for s in range(20, 20 + args.steps):
      a = time.perf_counter()
      step.forward(rag[s], rag[s + 1], out=out)
      b = time.perf_counter()
      step.backward(grads[s % 3], S.update_time(s))
      c = time.perf_counter()
      tf += b - a
      tb += c - b
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    res = {"which": which, "tables": T, "host_forward_us": round(tf / args.steps * 1e6, 1),
           "host_backward_us": round(tb / args.steps * 1e6, 1),
           "enqueue_us_per_step": round((t1 - t0) / args.steps * 1e6, 1),
           "wall_us_per_step": round((t2 - t0) / args.steps * 1e6, 1)}
    acc = {}
    n = 10
    for s in range(20 + args.steps, 20 + args.steps + n):
      _lib.profile_arm(64)
      step.forward(rag[s], rag[s + 1], out=out)
      step.backward(grads[s % 3], S.update_time(s))
      torch.cuda.synchronize()
      for name, us in _lib.profile_read():
        a = acc.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += us
    res["kernels"] = {k: [round(v[0] / n, 1), round(v[1] / n, 1)] for k, v in acc.items()}
    res["kernel_us_per_step"] = round(sum(v[1] for v in acc.values()) / n, 1)
    print(json.dumps(res), flush=True)
    step.close()
    del mt


if __name__ == "__main__":
  main()
