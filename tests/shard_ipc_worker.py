"""One rank of the multi-PROCESS sharded step (tests/test_shard_ipc_gpu.py spawns world of these, all
on cuda:0 when the box has one GPU): the C++ product path mhte_shard_step_forward / _backward over
the peer-store transport (mhte_shard_step_create_ipc: hipIpc-mapped windows, direct peer stores),
checked against the CPU oracle's single-process replay of the same id streams:

  * every step: this rank's per-occurrence embeddings == the oracle's rows before the step's updates;
  * afterwards: the rows of the ids this rank OWNS (id mod world == rank) == the oracle's rows after
    the ranks' gradient blocks were applied in rank order (one optimizer application per sender,
    native_training/distributed_ps_sync.py:357-479), and the owners' sizes add up.

Bit-exact when every id occurs <= 32 times in a batch (uniform ids); Zipf batches to the multi-step
tolerance.  gloo is only the launcher's side channel (handle gather, barriers).

usage: shard_ipc_worker.py RANK WORLD PORT DIST STEPS OUT_JSON
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
  rank, world, port, dist_kind, steps, out_path = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]),
                                                    sys.argv[4], int(sys.argv[5]), sys.argv[6])
  import torch
  import torch.distributed as dist
  torch.cuda.set_device(rank % torch.cuda.device_count())
  dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
  from monolith_amd import synthetic as S
  from monolith_amd.distributed_ps_sync import ShardedMultiStep
  from test_multi_step_gpu import ATOL, RTOL, dlrm_specs, make, oracle_backward, ragged_of, val_t
  from test_shard_step_gpu import batch_of, grads_of

  result = {"rank": rank, "ok": False}
  if dist_kind == "timeout":
    # rank 1 maps the windows and then never takes part: rank 0's bounded waits must give up with
    # MHTE_UNAVAILABLE instead of hanging its queue
    import ctypes as C
    from monolith_amd import _lib
    try:
      mt = make(dlrm_specs(2, initial_capacity=1 << 10))
      L, h, blob = mt._lib, C.c_void_p(), C.create_string_buffer(128)  # pylint: disable=protected-access
      _lib.check(L.mhte_shard_step_create_ipc(mt.handle, C.c_int64(64), C.c_int32(rank), C.c_int32(world),
                                              C.c_int64(0), C.byref(h)))
      _lib.check(L.mhte_shard_step_ipc_handle(h, blob))
      blobs = [None] * world
      dist.all_gather_object(blobs, blob.raw)
      _lib.check(L.mhte_shard_step_ipc_connect(h, b"".join(blobs), C.c_int32(world)))
      code = None
      if rank == 0:
        try:
          _lib.check(L.mhte_shard_step_ipc_selftest(h, None))
        except _lib.MhteError as e:
          code = e.code
      dist.barrier()
      torch.cuda.synchronize()
      L.mhte_shard_step_destroy(h)
      result.update(ok=(rank != 0 or code == _lib.MHTE_UNAVAILABLE), code=code, pid=os.getpid())
    except BaseException as e:  # pylint: disable=broad-except
      result["error"] = repr(e)[:800]
    with open(out_path, "w") as f:
      json.dump(result, f)
    sys.exit(0 if result["ok"] else 1)
  try:
    if os.environ.get("MHTE_TEST_SPECS") == "bias":
      # the reference's standard row layout: a dim-1 FTRL bias slice in front of the vector (dims 17 / 33)
      from test_multi_step_gpu import bias_slice_specs
      specs = bias_slice_specs()
    else:
      specs = dlrm_specs(8, initial_capacity=1 << 10)   # incl. SGD and bias-FTRL + vector-Adagrad tables
    by_name = sorted(specs, key=lambda s: s.name)
    B = 3000
    universe = 200000 if dist_kind == "uniform" else 7000
    exact = dist_kind == "uniform"
    mt = make(specs)
    want = os.environ.get("MHTE_TEST_TRANSPORT", "ipc")    # "rccl": one DEVICE per rank (a multi-GPU node)
    step = ShardedMultiStep(mt, B, transport=want)
    info = step.info()
    assert info["transport"].startswith(want), info
    if want == "rccl":
      assert step.comm_ranks() == (world, rank), (step.comm_ranks(), world, rank)
    assert info["ids_per_peer_table"] == B, info      # whole-batch blocks: nothing can overflow
    ots = {s.name: s.oracle_table() for s in specs}   # the oracle replays EVERY rank's stream

    def rank_batch(s, r):
      skip = (by_name[min(3, len(by_name) - 1)].name,) if (r == 1 and s % 2 == 0) else ()   # ragged: an empty table on one rank
      n = B if not (r == 0 and s == 2) else 1                  # and a one-id batch
      return batch_of(specs, 100 * s + r, n, universe, dist_kind, skip)

    batches = [[rank_batch(s, r) for r in range(world)] for s in range(steps + 1)]
    rag = [ragged_of(specs, mt, batches[s][rank]) for s in range(steps + 1)]
    worst = 0.0
    for s in range(steps):
      ahead = s % 3 != 2     # (every third step: the next batch is not dispatched ahead)
      emb = step.forward(rag[s], rag[s + 1] if ahead else None)
      views = mt.get_embeddings(rag[s], emb)
      for sp in by_name:
        ids = batches[s][rank].get(sp.name)
        if ids is None:
          continue
        exp = ots[sp.name].lookup(ids)[0]
        got = views[sp.name].cpu().numpy()
        if exact:
          np.testing.assert_array_equal(got, exp, err_msg="rank %d %s step %d" % (rank, sp.name, s))
        else:
          np.testing.assert_allclose(got, exp, rtol=RTOL, atol=ATOL)
          worst = max(worst, float(np.abs(got - exp).max()))
      mine = None
      for r in range(world):     # owners apply the senders' blocks in rank order
        fg = []
        for sp in by_name:
          ids = batches[s][r].get(sp.name)
          if ids is None:
            continue
          g = grads_of(s, r, sp, ids.size)
          fg.append(g.ravel())
          uk, gu = oracle_backward(ots[sp.name], sp, ids, g)
          if os.environ.get("MHTE_SHARD_GRAD_FP16") == "1":   # the sums cross the wire as fp16
            gu = gu.astype(np.float16).astype(np.float32)
          ots[sp.name].optimize(uk, gu, sp.lrs(), S.update_time(s))
        if r == rank:
          mine = val_t(np.concatenate(fg))
      step.backward(mine, S.update_time(s))
    step.check()
    sizes = {}
    for sp in by_name:
      seen = np.unique(np.concatenate([b[sp.name] for st in batches[:steps] for b in st if sp.name in b]))
      own = seen[seen % world == rank]
      got = mt.lookup({sp.name: torch.as_tensor(own).cuda()})[sp.name].cpu().numpy()
      exp = ots[sp.name].lookup(own)[0]
      if exact:
        np.testing.assert_array_equal(got, exp, err_msg="owner %d %s" % (rank, sp.name))
      else:
        np.testing.assert_allclose(got, exp, rtol=RTOL, atol=ATOL)
      assert int(mt.size(sp.name)) == own.size, (sp.name, int(mt.size(sp.name)), own.size)
      sizes[sp.name] = int(own.size)
    step.close()
    result.update(ok=True, transport=info["transport"], sizes=sizes, steps=steps, max_abs=worst,
                  device=torch.cuda.current_device(), pid=os.getpid())
  except BaseException as e:  # pylint: disable=broad-except
    import traceback
    result["error"] = "%s\n%s" % (repr(e)[:500], traceback.format_exc()[-1500:])
  with open(out_path, "w") as f:
    json.dump(result, f)
  try:
    dist.destroy_process_group()
  except Exception:  # pylint: disable=broad-except
    pass
  sys.exit(0 if result["ok"] else 1)


if __name__ == "__main__":
  main()
