#!/usr/bin/env python
"""bench.py — BASELINE.json's metric on MI355X: embedding lookups+updates/sec, 1B-id universe x
dim 64, Zipf(1.2), batch 65536, fused Adagrad (configs[2]); HBM-BW% through `roofline`.

One "step" = one pass of the sparse hot path over one batch of B ids already resident in HBM:
  dedup -> lookup(unique) -> scatter to B rows | duplicate-grad sum -> Adagrad apply(unique)
(the op sequence of native_training/distributed_ps.py:282-329 + :489-514).  value =
(B lookups + B updates) * steps * n_gpus / wall time, max over ranks, barrier + synchronize on both
sides.  With --gpus N > 1 (launched by torch.distributed.run) every rank feeds its own B ids and the
table is sharded by fid mod N with four all-to-alls per step over RCCL (weak scaling).

Contract: prints ONE JSON line on rank 0.  Extra objects: `roofline` (dominant kernel, HIP-event
timed on the launch stream), `cpu_baseline` (reference map + AVX Adagrad built from the reference's
own sources, oracle/_ref, timed on this box's host cores; N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable copy
PROBE_BYTES = 72        # keys+tags of both candidate buckets


def parse():
  p = argparse.ArgumentParser()
  p.add_argument("--gpus", type=int, default=1)
  p.add_argument("--steps", type=int, default=200)
  p.add_argument("--warmup", type=int, default=20)
  p.add_argument("--batch", type=int, default=65536)
  p.add_argument("--dim", type=int, default=64)
  p.add_argument("--universe", type=float, default=1e9)
  p.add_argument("--resident-rows", type=float, default=float(1 << 27),
                 help="rows pre-inserted per GPU (the hottest ranks of this GPU's shard)")
  p.add_argument("--opt", default="adagrad", choices=["adagrad", "sgd"])
  p.add_argument("--launch", default="auto", choices=["auto", "eager", "graph"])
  p.add_argument("--no-cpu-baseline", action="store_true")
  p.add_argument("--cpu-steps", type=int, default=150)
  p.add_argument("--exact-order", action="store_true")
  p.add_argument("--force-sharded", action="store_true",
                 help="N=1 through the id-sharded code path (one-rank process group): the floor of "
                      "the multi-GPU step without any link traffic; a measurement, not the bench line")
  p.add_argument("--reserve-ahead", action="store_true",
                 help="forward launch reserves the row handles of the update (SparseStep.reserve_ahead)")
  p.add_argument("--dist-backend", default="nccl",
                 help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo stages through "
                      "host memory and lets several ranks share one GPU: a functional check only)")
  p.add_argument("--trace-out", default="",
                 help="write a per-wavefront timeline (.npz) of three pipelined steps")
  p.add_argument("--no-stage-timing", action="store_true",
                 help="skip the per-kernel HIP-event pass (profiling the overlapped step only)")
  return p.parse_args()


def algorithmic_bytes(B, U, D, S):
  """SURVEY.md §8d / BASELINE.md §3 per step; and per kernel."""
  P = PROBE_BYTES
  lookup = 8 * B + P * U + 4 * D * U + 4 * D * B
  update = 8 * B + 4 * D * B + P * U + (8 * D + 8 * S) * U + 4 * U
  per_kernel = {
      # one probe+gather launch over the B occurrences: distinct buckets and rows are fetched from
      # HBM once (duplicates hit L2) == SURVEY's bytes_lookup
      "lookup_kernel": lookup,
      # fused backward launch (gradient sum + upsert + optimizer) == SURVEY's bytes_update
      "sum_apply_kernel": update,
  }
  return lookup + update, per_kernel


def pmc_traffic(kernel):
  """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_traffic.json,
  written by scripts/pmc_traffic.py from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
  runs of this same command; corrections as MI355X_MICROARCH.md prescribes, see that file's
  "method").  bench.py cannot collect counters itself; null when no summary is committed."""
  path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
  try:
    with open(path) as f:
      d = json.load(f)
    k = d["kernels"][kernel]
    return int(k["hbm_bytes_per_launch"]), "profiles/pmc_traffic.json (%s)" % d.get("source", "")
  except Exception:  # pylint: disable=broad-except
    return None, None


def main():
  args = parse()
  import torch
  import torch.distributed as dist
  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if args.gpus != world:
    if world == 1 and args.gpus > 1:
      raise SystemExit("--gpus %d needs: python -m torch.distributed.run --nproc-per-node %d "
                       "bench.py --gpus %d" % (args.gpus, args.gpus, args.gpus))
  assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
  local_rank %= torch.cuda.device_count()
  torch.cuda.set_device(local_rank)
  dev = torch.device("cuda", local_rank)
  sharded = world > 1 or args.force_sharded
  if sharded:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

  from monolith_amd import _lib, entry, synthetic as S
  from monolith_amd.fused_step import SparseStep
  from monolith_amd.multi_hash_table_ops import MultiHashTable

  B, D = args.batch, args.dim
  V = int(args.universe) * world  # weak scaling: 1B-id universe per GPU
  K, W = args.steps, args.warmup
  S_state = D if args.opt == "adagrad" else 0
  resident = int(args.resident_rows)

  def make_table(res):
    rows_cap = res + (K + W + 8) * B * (2 if sharded else 1) + (1 << 16)
    slots = 4
    while slots * 0.5 < rows_cap:
      slots *= 2
    opt = (entry.AdagradOptimizer(0.001, 0.1) if args.opt == "adagrad" else
           entry.SgdOptimizer(0.01))
    cfg = entry.make_table_config(
        [entry.CombineAsSegment(D, entry.ZerosInitializer(), opt)],
        entry.CuckooHashTableConfig(initial_capacity=slots, reserve_rows=rows_cap))
    return MultiHashTable.from_configs({"emb": cfg}, name_suffix="bench%d_%d" % (rank, res))

  mt = None
  while mt is None:
    try:
      mt = make_table(resident)
    except _lib.MhteError as e:  # out of HBM: halve the resident set and say so
      if e.code != _lib.MHTE_RESOURCE_EXHAUSTED or resident < (1 << 16):
        raise
      resident //= 2
      torch.cuda.empty_cache()

  # ---- prefill: the `resident` hottest ranks of this GPU's shard, rows = initializer (zeros) ----
  t0 = time.time()
  mult = torch.tensor(0x9E3779B97F4A7C15 - (1 << 64), dtype=torch.int64, device=dev)
  chunk = 1 << 22
  zeros = torch.zeros((chunk, D), dtype=torch.float32, device=dev)
  filled, r0 = 0, 1
  while filled < resident:
    ranks = torch.arange(r0, r0 + chunk * world, dtype=torch.int64, device=dev)
    r0 += chunk * world
    fid = ((ranks * mult) & ((1 << 48) - 1)) | (1 << 48)
    if world > 1:
      fid = fid[torch.remainder(fid, world) == rank]
    n = min(fid.numel(), resident - filled)
    fid = fid[:n].contiguous()
    rg = mt.get_ragged_id({"emb": fid})
    _lib.check(mt._lib.mhte_assign(mt.handle, _lib.vp(fid),
                                   rg.row_splits.ctypes.data_as(_lib.C.POINTER(_lib.C.c_int64)),
                                   _lib.C.c_int64(2), _lib.vp(zeros), _lib.C.c_int64(n * D),
                                   _lib.C.c_int64(S.update_time(0)),
                                   _lib.C.c_int32(_lib.MHTE_IDS_UNIQUE), None))
    filled += n
  torch.cuda.synchronize()
  del zeros
  torch.cuda.empty_cache()
  prefill_s = time.time() - t0
  st0 = mt.stats("emb")

  # ---- inputs resident in HBM before the timed region ----
  # Every launch mode trains FRESH batches (a replay of batches the table has already seen would
  # find every id resident and skip the insert path).  The modes run back to back over one
  # contiguous batch sequence: the last step of a mode deduplicates the first batch of the next
  # mode ahead, exactly as it does inside a mode.
  reps = min(K, 100)
  n_batches = (2 * (W + K) + 2 * reps + 8) if not sharded else (K + W + 1)
  ids_host = np.stack([S.id_batch(s * world + rank, B, V, "zipf") for s in range(n_batches)])
  ids_all = torch.from_numpy(ids_host).to(dev)
  grad_pool = [torch.from_numpy(S.grad_batch(s, B, D)).to(dev) for s in range(8)]

  def barrier():
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
      torch.cuda.synchronize()

  results = {}
  steps_of = {}
  graph_err = None
  if not sharded:
    # Steady-state pipeline (fused_step.py): while batch s is looked up and updated, the dedup of
    # batch s+1 — which depends on the ids only — rides in the same launches, as the reference's
    # prefetch queue does.  Every timed step executes exactly one dedup, one lookup and one
    # update: the first timed batch was deduplicated by the last warm-up step, the last timed
    # step deduplicates the batch after it.
    step = SparseStep(mt, "emb", B, exact_order=args.exact_order,
                      reserve_ahead=args.reserve_ahead)

    def run_eager(lo, hi):
      for s in range(lo, hi):
        step.forward(ids_all[s], next_ids=ids_all[s + 1])
        step.backward(grad_pool[s % 8], S.update_time(s))

    def timed(name, c, w, k):
      run_eager(c, c + w)
      barrier()
      t = time.perf_counter()
      run_eager(c + w, c + w + k)
      barrier()
      results[name] = time.perf_counter() - t
      steps_of[name] = k
      return c + w + k

    cur = timed("eager", 0, W, K)          # two launches per step

    # ---- hipGraph replay: the launch-bound loop captured in chunks of `gc` pipelined steps
    # (2 launches per step on one queue).  Each chunk graph reads its batches straight from the
    # resident id array, so nothing is copied or skipped inside the timed region.
    gc = 10
    if args.launch in ("auto", "graph") and K % gc == 0 and W % gc == 0:
      graphs = []
      try:
        # one eager step first: leaves batch cur+1 deduplicated ahead and a displacement pass
        # outstanding, the state every chunk starts from
        run_eager(cur, cur + 1)
        cur += 1
        step.quiesce()
        G0 = cur
        for c0 in range(G0, G0 + W + K, gc):
          g = torch.cuda.CUDAGraph()
          with torch.cuda.graph(g):
            run_eager(c0, c0 + gc)
          graphs.append(g)
        cur = G0 + W + K
        torch.cuda.synchronize()
        # capture executed nothing: replay from batch G0
        for g in graphs[:W // gc]:
          g.replay()
        barrier()
        t = time.perf_counter()
        for g in graphs[W // gc:]:
          g.replay()
        barrier()
        results["graph"] = time.perf_counter() - t
        steps_of["graph"] = K
      except Exception as e:  # pylint: disable=broad-except
        graph_err = repr(e)[:300]
        print("graph path failed: %s" % graph_err, file=sys.stderr)
        torch.cuda.synchronize()
        # the aborted capture advanced the host-side pipeline state without executing anything:
        # start the following passes from a fresh pipeline
        step = SparseStep(mt, "emb", B, exact_order=args.exact_order,
                      reserve_ahead=args.reserve_ahead)
        cur += gc * (len(graphs) + 1)
    P0 = cur
  else:
    from monolith_amd.distributed_ps_sync import HipBackend, ShardedEmbedding
    se = ShardedEmbedding(HipBackend(mt, "emb"))

    def run_sharded(lo, hi):
      for s in range(lo, hi):
        se.lookup(ids_all[s], next_ids=ids_all[s + 1])
        se.apply_gradients(grad_pool[s % 8], S.update_time(s))

    run_sharded(0, W)
    barrier()
    t = time.perf_counter()
    run_sharded(W, W + K)
    barrier()
    results["eager"] = time.perf_counter() - t
    steps_of["eager"] = K
  full = {k: v for k, v in results.items() if steps_of[k] == K}   # modes timed over exactly K steps
  launch = min(full, key=full.get) if args.launch == "auto" else (
      args.launch if args.launch in full else "eager")
  elapsed = results[launch]
  if world > 1:
    tt = torch.tensor([elapsed], dtype=torch.float64,
                      device=dev if args.dist_backend == "nccl" else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())

  # ---- per-kernel timing (N=1): HIP events stamped with each kernel's own begin/end on the launch
  # stream (mhte_profile_arm -> hipExtLaunchKernelGGL), the interval rocprofv3 --kernel-trace
  # reports.  Pass 1: the pipelined step as timed above (2 launches per step).  Pass 2: the
  # same work as separate launches, which attributes time to lookup / backward / dedup.
  roofline, stages, uniq_avg = None, {}, None
  if not sharded and not args.no_stage_timing:
    step.quiesce()
    acc = {}

    def collect():
      torch.cuda.synchronize()
      for name, us in _lib.profile_read():
        a = acc.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += us

    run_eager(P0, P0 + 1)   # (first step after a mode switch may use the plain forward launch)
    step.quiesce()
    _lib.profile_arm(2 * reps)
    run_eager(P0 + 1, P0 + 1 + reps)
    collect()
    pipelined = {k: v[1] / v[0] for k, v in acc.items()}
    pipelined_cnt = {k: v[0] for k, v in acc.items()}
    if args.trace_out:   # per-wavefront timeline of three more pipelined steps
      tcap = 1 << 20
      tbuf = torch.zeros(_lib.TRACE_WORDS * tcap, dtype=torch.int64, device=dev)
      _lib.trace_begin(tbuf, tcap)
      run_eager(P0 + 1 + reps, P0 + 4 + reps)
      torch.cuda.synchronize()
      tl = _lib.trace_end()
      np.savez_compressed(args.trace_out, records=tbuf.cpu().numpy().reshape(-1, _lib.TRACE_WORDS),
                          launches=np.array([(t_[0], t_[1], t_[2], t_[3]) for t_ in tl], dtype=object),
                          allow_pickle=True)
    acc = {}
    us = []
    step.quiesce()
    for s in range(P0 + reps + 5, P0 + 2 * reps + 5):
      ids = ids_all[s]
      _lib.profile_arm(8)
      step._unique(ids)  # pylint: disable=protected-access
      mt.table_lookup_n(step.idx, ids, None, step.emb, n_max=B)
      mt.table_sum_optimize_n(step.idx, step.ws, step.u, grad_pool[s % 8], step.grad_u, step.lrs,
                              S.update_time(s), 0, exact_order=args.exact_order, n_max=B,
                              defer_slowpath=True)
      mt.table_finish_pending(step.idx)
      collect()
      us.append(step.n_unique())
    uniq_avg = float(np.mean(us))
    step_bytes, per_kernel = algorithmic_bytes(B, uniq_avg, D, S_state)
    alg = {"step_fwd_kernel": per_kernel["lookup_kernel"],
           "step_bwd_kernel": per_kernel["sum_apply_kernel"],
           "lookup_kernel": per_kernel["lookup_kernel"], "sum_apply_kernel": per_kernel["sum_apply_kernel"]}
    for name, avg in sorted(pipelined.items()):
      stages[name] = {"avg_us": round(avg, 2),
                      "launches_per_step": round(pipelined_cnt[name] / reps, 2)}
    for name, (cnt, tot) in sorted(acc.items()):
      stages["unzipped:" + name] = {"avg_us": round(tot / cnt, 2), "launches_per_step": cnt // reps}
    for name, st_ in stages.items():
      base = name.split(":")[-1]
      if base in alg:
        st_["alg_bytes"] = int(alg[base])
        st_["GBps"] = round(alg[base] / st_["avg_us"] / 1e3, 1)
    dom = max((k for k in pipelined if k in alg), key=lambda k: pipelined[k])
    a_gbps = alg[dom] / pipelined[dom] / 1e3
    traffic, traffic_src = pmc_traffic(dom)
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(a_gbps, 1), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(a_gbps / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "traffic_source": traffic_src,
                "alg_bytes_per_launch": int(alg[dom]),
                "avg_launch_us": round(pipelined[dom], 2),
                "timing": "hipExtLaunchKernelGGL start/stop events on the launch stream, %d launches" % reps,
                "step_alg_bytes": int(step_bytes),
                "step_GBps": round(step_bytes / (elapsed / K) / 1e9, 1),
                "step_frac": round(step_bytes / (elapsed / K) / 1e9 / HBM_PEAK_GBPS, 4)}

  st1 = mt.stats("emb")

  # ---- CPU baseline: the reference's own map + AVX Adagrad on this box's host cores ----
  cpu = None
  if not sharded and rank == 0 and not args.no_cpu_baseline:
    try:
      import oracle as O
      cores = os.cpu_count() or 1
      opt = O.OPT_ADAGRAD if args.opt == "adagrad" else O.OPT_SGD
      avx = O.ref_available(True)
      ps = O.RefPs(cores, D, opt, 0.1, 0.0, 0.0, 1, avx=avx)
      lr = 0.001 if args.opt == "adagrad" else 0.01
      cw, ck = 10, args.cpu_steps
      grads_h = [S.grad_batch(s, B, D) for s in range(4)]
      times = []
      for s in range(cw + ck):
        ids = S.id_batch(s, B, V, "zipf")
        t = time.perf_counter()
        ps.step(ids, grads_h[s % 4], lr, S.update_time(s), want_emb=True)
        times.append(time.perf_counter() - t)
      med = float(np.median(times[cw:]))
      cpu = {"value": round(2 * B / med, 1), "unit": "lookups+updates/s", "cores": cores,
             "kind": "reference",
             "sample": "%d steps (after %d warm-up) of the same Zipf(1.2) stream, batch %d, dim %d, "
                       "%s; table grown on demand from empty (%d rows at end); PS-style %d "
                       "single-threaded shards of the reference cuckoohash_map + %s Adagrad, "
                       "median step %.2f ms" % (ck, cw, B, D, args.opt, ps.size(), cores,
                                                "AVX2" if avx else "scalar", med * 1e3)}
    except Exception as e:  # pylint: disable=broad-except
      cpu = {"value": None, "unit": "lookups+updates/s", "cores": os.cpu_count(), "kind": "reference",
             "sample": "failed: %r" % (e,)}

  if rank == 0:
    value = 2.0 * B * K * world / elapsed
    out = {
        "metric": "embedding lookups+updates/sec at 1B ids x dim64, Zipf(1.2) batch=65536",
        "value": round(value, 1),
        "unit": "lookups+updates/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": round(elapsed / K * 1e3, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": ("configs[2]: 1 MI355X, 1B-id universe (per GPU), dim=%d, Zipf(1.2), "
                         "batch=%d/GPU, %s; grow-on-demand table" if world == 1 else
                         "configs[3] shape: %d MI355X, %dB ids sharded by fid mod N (1B-id universe "
                         "per GPU), dim=%%d, Zipf(1.2), batch=%%d/GPU, %%s; all-to-all id dispatch + "
                         "row / gradient return" % (world, world)) %
                        (D, B, "fused Adagrad" if args.opt == "adagrad" else "SGD"),
            "batch_per_gpu": B, "dim": D, "universe_ids": V, "optimizer": args.opt,
            "resident_rows_per_gpu_start": int(st0.size), "resident_rows_per_gpu_end": int(st1.size),
            "row_bytes": 4 * (D + S_state), "hashpower": int(st1.hashpower),
            "table_bytes_per_gpu": int(st1.bytes_buckets + st1.bytes_rows),
            "unique_ids_per_batch": uniq_avg, "launch": launch,
            "parallelism": "1 GPU" if not sharded else
                           "fid mod %d sharding, 4 all-to-all/step (%s)" % (world, args.dist_backend),
            "prefill_s": round(prefill_s, 2),
        },
        "timing_ms_per_step": {k: round(v / steps_of[k] * 1e3, 5) for k, v in results.items()},
        "roofline": roofline,
        "stages": stages,
        "cpu_baseline": cpu,
    }
    if cpu and cpu.get("value"):
      out["vs_cpu_baseline"] = round(value / cpu["value"], 2)
    if graph_err:
      out["graph_error"] = graph_err
    print(json.dumps(out))
  if sharded:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
