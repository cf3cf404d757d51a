#!/usr/bin/env python
"""Sweep of the lookup launch shape (ids per group x workgroup size x streaming stores) on the
bench workload: B = 65 536 Zipf(1.2) ids, dim 64, table of --rows resident rows.  Kernel-exact
times (mhte_profile_arm).  Measurement only."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monolith_amd import _lib, entry, synthetic as S  # noqa: E402
from monolith_amd.multi_hash_table_ops import MultiHashTable  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--rows", type=float, default=float(1 << 26))
  ap.add_argument("--reps", type=int, default=40)
  args = ap.parse_args()
  B, D = 65536, 64
  rows = int(args.rows)
  dev = torch.device("cuda", 0)
  slots = 4
  while slots * 0.5 < rows + (1 << 16):
    slots *= 2
  cfg = entry.make_table_config(
      [entry.CombineAsSegment(D, entry.ZerosInitializer(), entry.AdagradOptimizer(0.001, 0.1))],
      entry.CuckooHashTableConfig(initial_capacity=slots, reserve_rows=rows + (1 << 16)))
  mt = MultiHashTable.from_configs({"emb": cfg}, name_suffix="sweep")
  mult = torch.tensor(0x9E3779B97F4A7C15 - (1 << 64), dtype=torch.int64, device=dev)
  chunk = 1 << 22
  ones = torch.ones((chunk, D), dtype=torch.float32, device=dev)
  for r0 in range(1, rows + 1, chunk):
    ranks = torch.arange(r0, min(r0 + chunk, rows + 1), dtype=torch.int64, device=dev)
    fid = ((ranks * mult) & ((1 << 48) - 1)) | (1 << 48)
    rg = mt.get_ragged_id({"emb": fid})
    _lib.check(mt._lib.mhte_assign(mt.handle, _lib.vp(fid),
                                   rg.row_splits.ctypes.data_as(_lib.C.POINTER(_lib.C.c_int64)),
                                   _lib.C.c_int64(2), _lib.vp(ones), _lib.C.c_int64(fid.numel() * D),
                                   _lib.C.c_int64(S.update_time(0)),
                                   _lib.C.c_int32(_lib.MHTE_IDS_UNIQUE), None))
  torch.cuda.synchronize()
  del ones
  ids = [torch.from_numpy(S.id_batch(s, B, 10**9, "zipf")).to(dev) for s in range(args.reps)]
  out = torch.empty((B, D), dtype=torch.float32, device=dev)
  ref = None
  res = []
  for unr in (0, 1, 2, 4, 8, 16):
    for blk in ((256,) if unr == 0 else (256, 512, 1024)):
      for nt in ((0,) if unr == 0 else (0, 1)):
        os.environ["MHTE_LOOKUP_UNR"] = str(unr)
        os.environ["MHTE_LOOKUP_BLOCK"] = str(blk)
        os.environ["MHTE_LOOKUP_NT"] = str(nt)
        for s in range(4):
          mt.table_lookup_n(0, ids[s], None, out)
        torch.cuda.synchronize()
        chk = out.sum().item()
        if ref is None:
          ref = chk
        _lib.profile_arm(args.reps)
        for s in range(args.reps):
          mt.table_lookup_n(0, ids[s], None, out)
        torch.cuda.synchronize()
        us = [u for _, u in _lib.profile_read()]
        res.append({"unr": unr, "block": blk, "nt": nt, "avg_us": round(float(np.mean(us)), 2),
                    "min_us": round(float(np.min(us)), 2), "ok": chk == ref})
        print(json.dumps(res[-1]), flush=True)


if __name__ == "__main__":
  main()
