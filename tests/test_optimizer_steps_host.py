"""CPU suite: the engine's per-element optimizer arithmetic (monolith_amd/csrc/mhte_core.h — one
source for host and device) against oracle/, bit for bit, over 25 Optimize() calls on one row with
normal gradients.  The GPU suite repeats this through the kernels (test_parity_gpu.py); here the
arithmetic itself is gated without a GPU.  Adam / AMSGrad: the reference's unqualified `sqrt`
(adam_optimizer.cc:64,74,76; amsgrad_optimizer.cc:66,77,79) is ::sqrt(double) — both sides form the
effective learning rate and the quotient in double."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [
    # name, engine opt id (mhte_core.h kOpt*), oracle opt, p, lr
    ("sgd", 0, O.OPT_SGD, (), 0.01),
    ("adagrad", 1, O.OPT_ADAGRAD, (0.1, 0.0), 0.05),
    ("adagrad_wd", 1, O.OPT_ADAGRAD, (0.1, 0.01), 0.05),
    # the reference's AVX2 form (avx_utils.h:96-119), opt-in: three fused blocks of 8 + the tail
    ("adagrad_avx_form_wd", 1, O.OPT_ADAGRAD, (0.1, 0.1, 1.0), 0.05),
    ("ftrl", 2, O.OPT_FTRL, (0.1, 1.0, 0.001, 0.01), 0.05),
    ("momentum", 3, O.OPT_MOMENTUM, (0.9, 0.01, 0.0), 0.01),
    ("momentum_nesterov", 3, O.OPT_MOMENTUM, (0.9, 0.0, 1.0), 0.01),
    ("adadelta", 4, O.OPT_ADADELTA, (0.9, 0.01, 0.001), 0.5),
    ("rmsprop", 5, O.OPT_RMSPROP, (0.9, 0.001, 0.02), 0.01),
    ("rmspropv2", 6, O.OPT_RMSPROPV2, (0.9, 0.001, 0.02), 0.01),
    ("adam", 7, O.OPT_ADAM, (0.9, 0.99, 0.01, 0.0, 0.0), 0.01),
    ("adam_wd_nesterov", 7, O.OPT_ADAM, (0.9, 0.999, 1e-8, 0.01, 1.0), 0.003),
    ("amsgrad", 8, O.OPT_AMSGRAD, (0.9, 0.99, 0.01, 0.001, 0.0), 0.01),
    ("moving_average", 9, O.OPT_MOVING_AVERAGE, (0.9,), 0.0),
]


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
  exe = str(tmp_path_factory.mktemp("optim") / "optim_host")
  subprocess.check_call(["g++", "-O2", "-std=c++17", "-DMHTE_HOST_ONLY", "-ffp-contract=off",
                         "-I" + os.path.join(ROOT, "monolith_amd", "csrc"),
                         os.path.join(ROOT, "tests", "optim_host_driver.cc"), "-o", exe])
  return exe


def test_engine_opt_ids_are_the_oracles():
  """the case table pairs the two enumerations by value; keep them the same numbers"""
  for _, eng, orc, _, _ in CASES:
    assert eng == orc


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_optimizer_arithmetic_host_equals_oracle(case, driver, tmp_path):
  _, eng, orc, p, lr = case
  dim, steps = (27 if len(p) > 2 and case[0].startswith("adagrad_avx") else 24), 25
  rng = np.random.default_rng(1234 + eng)
  g = (rng.standard_normal((steps, dim)) * np.float32(0.3)).astype(np.float32)
  pp = np.zeros(8, np.float32)
  pp[:len(p)] = p
  path = str(tmp_path / "case.bin")
  with open(path, "wb") as f:
    f.write(struct.pack("<3i", eng, dim, steps))
    f.write(pp.tobytes())
    f.write(struct.pack("<f", lr))
    f.write(g.tobytes())
  out = subprocess.run([driver, path], capture_output=True, text=True, check=True).stdout
  got = np.array([int(x, 16) for x in out.split()], dtype=np.uint32)

  t = O.Table([O.segment(dim, orc, p=p)], 1)
  one = np.array([42], np.int64)
  for s in range(steps):
    t.optimize(one, g[s:s + 1], [lr], 0)
  want = t.lookup(one)[0][0].astype(np.float32).view(np.uint32)
  np.testing.assert_array_equal(got, want)
  assert np.isfinite(want.view(np.float32)).all() and np.any(want.view(np.float32) != 0)
