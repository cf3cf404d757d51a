"""CPU suite: the host-side helpers of bench.py that decide something about the measured run."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
  spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
  m = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(m)
  return m


def test_peer_unique_max_counts_distinct_ids_per_owner():
  b = _bench()
  ids = np.array([[8, 8, 16, 1, 9, 9, 9, 3],      # owner 0: {8, 16}; owner 1: {1, 9}; owner 3: {3}
                  [5, 5, 5, 5, 13, 21, 29, 2]],   # owner 1: {5, 13, 21, 29}; owner 2: {2}
                 dtype=np.int64)
  assert b.peer_unique_max(ids, 4) == 4
  assert b.peer_unique_max(ids[:1], 4) == 2
  assert b.peer_unique_max(ids, 1) == 5            # one owner: distinct ids of the larger batch


def test_algorithmic_bytes_follow_the_survey_formula():
  """bytes_lookup + bytes_update with P = 72, checked on the closed form at U = B."""
  b = _bench()
  B, D = 65536, 32
  step, per_kernel = b.algorithmic_bytes(B, B, D, 0)   # SGD: no optimizer state
  assert step == B * (16 + 20 * D + 2 * 72 + 4)        # 804 B per id pair at D = 32
  assert per_kernel["sum_apply_kernel"] == B * (8 + 4 * D + 72 + 8 * D + 4)
  assert per_kernel["lookup_kernel"] + per_kernel["sum_apply_kernel"] == step
  step, _ = b.algorithmic_bytes(B, B, 64, 64)          # Adagrad, D = 64: 1 956 B per id pair
  assert step == B * 1956
