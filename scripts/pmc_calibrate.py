#!/usr/bin/env python
"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on THIS engine's access patterns
(MI355X_MICROARCH.md, HBM section: FETCH_SIZE is exactly 1/2 of the bytes of a wide coalesced
streaming read on gfx950; "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known
byte count in your own access pattern").

Run under   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- python scripts/pmc_calibrate.py
(and again with WRITE_SIZE).  Two launches with known byte counts, both far larger than L2 + MALL:

  gather_rows_kernel  n = 2^22 distinct random rows of 256 B out of a 4 GiB source
      reads 4 n (index, streaming) + 256 n (random rows); writes 256 n (streaming)
  lookup_kernel       n = 2^22 distinct resident ids of a 2^24-row table, dim 64 (256-B rows)
      reads 8 n (ids, streaming) + 128 n (two 64-B bucket lines) + 256 n (random rows); writes 256 n

Prints the expected byte counts as JSON; scripts/pmc_traffic.py divides the counters by them."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monolith_amd import entry  # noqa: E402
from monolith_amd.distribution_ops import DedupWorkspace  # noqa: E402
from monolith_amd.multi_hash_table_ops import MultiHashTable  # noqa: E402


def main():
  dev = torch.device("cuda", 0)
  torch.cuda.set_device(0)
  D, n, rows = 64, 1 << 22, 1 << 24
  g = torch.Generator(device="cpu").manual_seed(7)
  src = torch.zeros((rows, D), dtype=torch.float32, device=dev)
  idx = torch.randperm(rows, generator=g)[:n].to(torch.int32).to(dev)
  out = torch.empty((n, D), dtype=torch.float32, device=dev)
  ws = DedupWorkspace(0)
  for _ in range(3):
    ws.gather_rows(src, idx, n, D, out=out)
  torch.cuda.synchronize()
  del src

  cfg = entry.make_table_config(
      [entry.CombineAsSegment(D, entry.ZerosInitializer(), entry.SgdOptimizer(0.01))],
      entry.CuckooHashTableConfig(initial_capacity=rows * 2, reserve_rows=rows + 1024))
  mt = MultiHashTable.from_configs({"emb": cfg}, name_suffix="calib")
  ids = (torch.arange(1, rows + 1, dtype=torch.int64, device=dev) * 0x9E3779B97F4A7C1) & ((1 << 48) - 1)
  ids = torch.unique(ids)
  chunk = 1 << 22
  zeros = torch.zeros((chunk, D), dtype=torch.float32, device=dev)
  for c0 in range(0, ids.numel(), chunk):
    part = ids[c0:c0 + chunk].contiguous()
    mt.assign({"emb": (part, zeros[:part.numel()])}, 1)
  torch.cuda.synchronize()
  pick = ids[torch.randperm(ids.numel(), generator=g)[:n].to(dev)].contiguous()
  for _ in range(3):
    mt.table_lookup_n(0, pick, None, out)
  torch.cuda.synchronize()
  print(json.dumps({
      "n": n,
      "gather_rows_kernel": {"read_stream": 4 * n, "read_rows": 256 * n, "write": 256 * n},
      "lookup_kernel": {"read_stream": 8 * n, "read_buckets": 128 * n, "read_rows": 256 * n,
                        "write": 256 * n},
  }))


if __name__ == "__main__":
  main()
