"""Where the time of the multi-table step goes (MI355X): kernel-exact timings (mhte_profile_arm) of
the roles on their own and fused, for the dlrm26 shape, under the host-side knobs
(MHTE_MSTEP_OVERSUB, MHTE_MSTEP_SCATTER_OVS).  Prints one JSON line per variant.
Usage: python scripts/mstep_probe.py [--tables 26] [--resident 2097152] [--knobs "ovs:lookup,..."]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monolith_amd import _lib, entry, synthetic as S  # noqa: E402
from monolith_amd.fused_step import MultiSparseStep  # noqa: E402
from monolith_amd.multi_hash_table_ops import MultiHashTable, Ragged  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--tables", type=int, default=26)
  ap.add_argument("--batch", type=int, default=65536)
  ap.add_argument("--resident", type=int, default=1 << 21)
  ap.add_argument("--universe", type=int, default=38461538)
  ap.add_argument("--steps", type=int, default=12)
  ap.add_argument("--dims", default="16,32,64")
  ap.add_argument("--knobs", default="4:0")
  ap.add_argument("--trace", default="", help="write a per-wavefront timeline (.npz) of two fused steps")
  args = ap.parse_args()
  T, B = args.tables, args.batch
  dev = torch.device("cuda", 0)
  dl = [int(x) for x in args.dims.split(",")]
  dims = [dl[i % len(dl)] for i in range(T)]
  names = ["f%02d" % (i + 1) for i in range(T)]
  cfgs = {}
  for i, n in enumerate(names):
    rows_cap = args.resident + 200 * 24000
    slots = 4
    while slots * 0.5 < rows_cap:
      slots *= 2
    cfgs[n] = entry.make_table_config(
        [entry.CombineAsSegment(dims[i], entry.ZerosInitializer(), entry.AdagradOptimizer(0.001, 0.1))],
        entry.CuckooHashTableConfig(initial_capacity=slots, reserve_rows=rows_cap))
  mt = MultiHashTable.from_configs(cfgs, name_suffix="probe")
  mult = torch.tensor(0x9E3779B97F4A7C15 - (1 << 64), dtype=torch.int64, device=dev)
  sp0 = np.zeros(T + 1, dtype=np.int64)
  for i in range(T):
    ranks = torch.arange(1, args.resident + 1, dtype=torch.int64, device=dev)
    fid = ((ranks * mult) & ((1 << 48) - 1)) | ((i + 1) << 48)
    zeros = torch.zeros((args.resident, dims[i]), dtype=torch.float32, device=dev)
    sp = sp0.copy()
    sp[i + 1:] = args.resident
    _lib.check(mt._lib.mhte_assign(mt.handle, _lib.vp(fid), sp.ctypes.data_as(_lib.C.POINTER(_lib.C.c_int64)),
                                   _lib.C.c_int64(T + 1), _lib.vp(zeros), _lib.C.c_int64(zeros.numel()),
                                   _lib.C.c_int64(S.update_time(0)), _lib.C.c_int32(_lib.MHTE_IDS_UNIQUE), None))
    del zeros, fid, ranks
  torch.cuda.synchronize()
  torch.cuda.empty_cache()
  nb = len(args.knobs.split(",")) * (3 * args.steps + 3) + 4
  ids_host = np.empty((nb, T * B), dtype=np.int64)
  for s in range(nb):
    for i in range(T):
      ids_host[s, i * B:(i + 1) * B] = S.id_batch(s * 64 + i + 1, B, args.universe, "zipf", feature_slot=i + 1)
  ids_all = torch.from_numpy(ids_host).to(dev)
  splits = np.arange(T + 1, dtype=np.int64) * B
  rag = [Ragged(ids_all[s], splits) for s in range(nb)]
  gsz = B * sum(dims)
  grads = [torch.randn(gsz, device=dev) * 0.01 for _ in range(2)]
  out = torch.empty(gsz, dtype=torch.float32, device=dev)
  cursor = [0]

  def nxt():
    cursor[0] += 1
    return cursor[0] - 1

  for knob in args.knobs.split(","):
    parts_ = knob.split(":")
    ovs, lk = parts_[0], parts_[1]
    os.environ["MHTE_MSTEP_OVERSUB"] = ovs
    os.environ["MHTE_MSTEP_SCATTER_OVS"] = lk
    os.environ["MHTE_MSTEP_ITEM_TARGET"] = parts_[2] if len(parts_) > 2 else "256"
    step = MultiSparseStep(mt, B)
    res = {"knob": knob}
    # unfused: forward without prefetch = [fwd(dedup only), bwd(build only), fwd(lookup only)];
    # backward = [bwd(apply only)]
    parts = {"dedup_only": [], "build_only": [], "lookup_only": [], "apply_only": []}
    for _ in range(args.steps):
      s = nxt()
      _lib.profile_arm(8)
      step.forward(rag[s], None, out=out)
      step.backward(grads[s % 2], S.update_time(s))
      torch.cuda.synchronize()
      pr = _lib.profile_read()
      assert [p[0] for p in pr] == ["dd_kernels", "mstep_bwd_kernel", "mstep_fwd_kernel",
                                    "mstep_bwd_kernel"], pr
      for k, (_, us) in zip(("dedup_only", "build_only", "lookup_only", "apply_only"), pr):
        parts[k].append(us)
    for k, v in parts.items():
      res[k] = round(float(np.median(v)), 1)
    # fused pipeline
    fw, bw = [], []
    s = nxt()
    step.forward(rag[s], rag[s + 1], out=out)
    step.backward(grads[0], S.update_time(s))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    first = None
    for k in range(args.steps):
      s = nxt()
      if k == 2:
        ev0.record()
        first = k
      step.forward(rag[s], rag[s + 1], out=out)
      step.backward(grads[s % 2], S.update_time(s))
    ev1.record()
    torch.cuda.synchronize()
    res["pipelined_step_us_wall"] = round(ev0.elapsed_time(ev1) * 1e3 / (args.steps - first), 1)
    per = {}
    for k in range(args.steps):
      s = nxt()
      _lib.profile_arm(8)
      step.forward(rag[s], rag[s + 1], out=out)
      step.backward(grads[s % 2], S.update_time(s))
      torch.cuda.synchronize()
      for name, us in _lib.profile_read():
        per.setdefault(name, []).append(us)
    cursor[0] += 1
    for name, v in per.items():
      res["fused:" + name] = round(float(np.median(v)), 1)
    print(json.dumps(res), flush=True)
    if args.trace:
      tcap = 1 << 21
      tbuf = torch.zeros(_lib.TRACE_WORDS * tcap, dtype=torch.int64, device=dev)
      s = cursor[0] - 1   # (the batch deduplicated ahead by the last fused step)
      _lib.trace_begin(tbuf, tcap)
      step.forward(rag[s], rag[s + 1], out=out)
      step.backward(grads[0], S.update_time(s))
      torch.cuda.synchronize()
      step.forward(rag[s + 2], None, out=out)   # unfused: dedup, build, lookup
      step.backward(grads[0], S.update_time(s))
      torch.cuda.synchronize()
      tl = _lib.trace_end()
      np.savez_compressed(args.trace, records=tbuf.cpu().numpy().reshape(-1, _lib.TRACE_WORDS),
                          launches=np.array([(t_[0], t_[1], t_[2], t_[3]) for t_ in tl], dtype=object),
                          allow_pickle=True)
      args.trace = ""
    step.close()


if __name__ == "__main__":
  main()
