"""GPU parity (-m gpu) of the MULTI-PROCESS id-sharded step: `world` processes (sharing cuda:0 on a
1-GPU box) drive the C++ product path mhte_shard_step_forward / _backward over the peer-store
transport — hipIpc-mapped receive windows, device-sized direct stores, credit / arrival flags
(csrc/mhte_shard_host.h, shard_push_kernel / shard_wait_kernel) — and every rank checks itself
against the CPU oracle's single-process replay (tests/shard_ipc_worker.py).  RCCL cannot place two
ranks on one device; this transport runs the same on xGMI peers.
Reference: native_training/distributed_ps_sync.py:95-287 (lookup), :289-490 (apply_gradients).
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def run_world(world, dist_kind, steps, tmp_path, extra_env=None):
  port = _free_port()
  env = dict(os.environ)
  env.setdefault("MHTE_SHARD_TIMEOUT_MS", "20000")
  env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  env.update(extra_env or {})
  outs = [str(tmp_path / ("rank%d.json" % r)) for r in range(world)]
  procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "shard_ipc_worker.py"), str(r), str(world),
                             str(port), dist_kind, str(steps), outs[r]], env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
  logs = []
  try:
    for p in procs:
      try:
        logs.append(p.communicate(timeout=420)[0].decode("utf-8", "replace"))
      except subprocess.TimeoutExpired:
        logs.append("timeout")
  finally:
    for p in procs:   # (exactly the processes started above)
      if p.poll() is None:
        p.kill()
  results = []
  for r in range(world):
    assert os.path.exists(outs[r]), "rank %d wrote no result; log:\n%s" % (r, logs[r][-3000:])
    with open(outs[r]) as f:
      results.append(json.load(f))
  for r, res in enumerate(results):
    assert res["ok"], "rank %d: %s\nlog:\n%s" % (r, res.get("error"), logs[r][-2000:])
  return results


@pytest.mark.parametrize("dist_kind,world", [("uniform", 2), ("zipf", 2), ("uniform", 3)])
def test_processes_against_oracle(dist_kind, world, tmp_path):
  res = run_world(world, dist_kind, 6, tmp_path)
  assert len({r["pid"] for r in res}) == world            # really separate processes
  assert all(r["transport"].startswith("ipc") for r in res)
  # the owners' shards are disjoint and complete (each rank checked its own rows against the oracle)
  for name in res[0]["sizes"]:
    assert sum(r["sizes"][name] for r in res) > 0


def test_processes_bias_slice_rows(tmp_path):
  """FTRL(1) + Adagrad(16 / 32) rows (NT/distributed_ps_test.py:480-505) between PROCESSES over the
  peer-store transport: row slots of 17 / 33 floats on the wire, one float per lane on both sides."""
  res = run_world(2, "zipf", 5, tmp_path, {"MHTE_TEST_SPECS": "bias"})
  assert all(r["transport"].startswith("ipc") for r in res)


def _n_devices():
  try:
    import torch
    return torch.cuda.device_count()
  except Exception:  # pylint: disable=broad-except
    return 0


@pytest.mark.skipif(_n_devices() < 2, reason="needs two GPUs: one DEVICE per rank (runs on the driver's multi-GPU node)")
@pytest.mark.parametrize("transport", ["ipc", "rccl"])
@pytest.mark.parametrize("dist_kind", ["uniform", "zipf"])
def test_two_devices_both_transports_against_oracle(transport, dist_kind, tmp_path):
  """The first execution of the cross-DEVICE paths: two processes, rank r on cuda:r — peer stores
  into the other device's hipIpc-mapped window over xGMI, and RCCL send / recv groups on the library's
  own two-rank communicator (ncclCommCount == 2 asserted in the worker) — every rank's embeddings and
  every owner's rows against the oracle's single-process replay, as the one-GPU process tests do."""
  res = run_world(2, dist_kind, 5, tmp_path, {"MHTE_TEST_TRANSPORT": transport})
  assert len({r["device"] for r in res}) == 2, [r["device"] for r in res]
  assert all(r["transport"].startswith(transport) for r in res)


@pytest.mark.skipif(_n_devices() < 4, reason="needs four GPUs")
def test_four_devices_rccl_bias_rows(tmp_path):
  res = run_world(4, "zipf", 4, tmp_path, {"MHTE_TEST_TRANSPORT": "rccl", "MHTE_TEST_SPECS": "bias"})
  assert len({r["device"] for r in res}) == 4


def test_processes_with_overlap(tmp_path):
  """MHTE_SHARD_OVERLAP=1: the next batch's dedup, numbering, packing and id exchange on the step's own
  stream (pushes and their publication from two streams of every process) — same results."""
  res = run_world(2, "uniform", 6, tmp_path, {"MHTE_SHARD_OVERLAP": "1"})
  assert all(r["ok"] for r in res)
  res = run_world(3, "zipf", 5, tmp_path, {"MHTE_SHARD_OVERLAP": "1"})
  assert all(r["ok"] for r in res)


def test_processes_fp16_gradient_wire(tmp_path):
  """MHTE_SHARD_GRAD_FP16=1 between processes: the push narrows nothing itself — the sender's sums are
  rounded to fp16 by a launch of their own, stored into the peers' windows as 16-bit words and widened by
  the owner per peer — bit-exact against the oracle fed the same rounding."""
  res = run_world(2, "uniform", 5, tmp_path, {"MHTE_SHARD_GRAD_FP16": "1"})
  assert all(r["ok"] for r in res)
  res = run_world(2, "uniform", 5, tmp_path, {"MHTE_SHARD_GRAD_FP16": "1", "MHTE_SHARD_OVERLAP": "1"})
  assert all(r["ok"] for r in res)


def test_coarse_window(tmp_path):
  """MHTE_SHARD_WINDOW=coarse: the windows as plain device memory (the A/B form)."""
  res = run_world(2, "uniform", 4, tmp_path, {"MHTE_SHARD_WINDOW": "coarse"})
  assert all(r["transport"] == "ipc (coarse window)" for r in res)


def test_missing_peer_times_out_instead_of_hanging(tmp_path):
  """A rank whose peer maps the windows and then never takes part gets MHTE_UNAVAILABLE (14) from the
  bounded waits; its queue drains."""
  res = run_world(2, "timeout", 0, tmp_path, {"MHTE_SHARD_TIMEOUT_MS": "1500"})
  assert res[0]["code"] == 14


def test_unconnected_step_is_refused():
  import ctypes as C
  sys.path.insert(0, HERE)
  from monolith_amd import _lib
  from test_multi_step_gpu import dlrm_specs, make
  mt = make(dlrm_specs(2, initial_capacity=1 << 10))
  L, h = mt._lib, C.c_void_p()  # pylint: disable=protected-access
  # world 1 over the transport: the rank's own window, a round trip with itself
  _lib.check(L.mhte_shard_step_create_ipc(mt.handle, C.c_int64(64), C.c_int32(0), C.c_int32(1), C.c_int64(0),
                                          C.byref(h)))
  b = C.create_string_buffer(128)
  _lib.check(L.mhte_shard_step_ipc_handle(h, b))
  with pytest.raises(_lib.MhteError):   # not rank-major handles of this world
    _lib.check(L.mhte_shard_step_ipc_connect(h, b.raw, C.c_int32(2)))
  _lib.check(L.mhte_shard_step_ipc_connect(h, b.raw, C.c_int32(1)))
  _lib.check(L.mhte_shard_step_ipc_selftest(h, None))
  L.mhte_shard_step_destroy(h)
  h0 = C.c_void_p()
  _lib.check(L.mhte_shard_step_create_ipc(mt.handle, C.c_int64(64), C.c_int32(0), C.c_int32(2), C.c_int64(0),
                                          C.byref(h0)))
  with pytest.raises(_lib.MhteError) as ei:
    _lib.check(L.mhte_shard_step_ipc_selftest(h0, None))
  assert ei.value.code == _lib.MHTE_FAILED_PRECONDITION
  L.mhte_shard_step_destroy(h0)


def test_selftest_checks_the_data_not_only_the_flags(monkeypatch):
  """The creation self test moves a pattern through every (sender, receiver) pair of windows, three
  rounds over the same addresses, and checks it on the receiving side behind the sync — the path of a
  real exchange.  With the checker told to expect another round's pattern (MHTE_SHARD_SELFTEST_CORRUPT)
  it reports MHTE_UNAVAILABLE: what a transport whose stores arrive late or stale would get, and what
  makes `transport="auto"` fall back to RCCL."""
  import ctypes as C
  sys.path.insert(0, HERE)
  from monolith_amd import _lib
  from test_multi_step_gpu import dlrm_specs, make
  mt = make(dlrm_specs(2, initial_capacity=1 << 10))
  L = mt._lib  # pylint: disable=protected-access

  def connected():
    h = C.c_void_p()
    _lib.check(L.mhte_shard_step_create_ipc(mt.handle, C.c_int64(64), C.c_int32(0), C.c_int32(1), C.c_int64(0),
                                            C.byref(h)))
    b = C.create_string_buffer(128)
    _lib.check(L.mhte_shard_step_ipc_handle(h, b))
    _lib.check(L.mhte_shard_step_ipc_connect(h, b.raw, C.c_int32(1)))
    return h

  h = connected()
  monkeypatch.setenv("MHTE_SHARD_SELFTEST_CORRUPT", "1")
  with pytest.raises(_lib.MhteError) as ei:
    _lib.check(L.mhte_shard_step_ipc_selftest(h, None))
  assert ei.value.code == _lib.MHTE_UNAVAILABLE and "self test" in str(ei.value)
  monkeypatch.delenv("MHTE_SHARD_SELFTEST_CORRUPT")
  _lib.check(L.mhte_shard_step_ipc_selftest(h, None))   # the same step passes without the hook
  L.mhte_shard_step_destroy(h)
