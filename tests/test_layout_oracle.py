"""CPU suite: the fused_embedding_to_layout checker (oracle/layout.py layout_model /
layout_grad_model: the op's algorithm over its own offset encoding) pinned to the reference's own
test — its input generation and its TRUTH PROCEDURE, which computes the expectation per feature from
the original fid lists (native_training/fused_embedding_to_layout_test.py:176-530 forward, :553-790
gradient) — at the reference's sizes (batch 256; 199 slots / 5 shards forward, 29 slots / 3 shards
gradient) and the reference's tolerance (np.allclose rtol 1e-4, atol 1e-7, :523)."""
import numpy as np
import pytest

from oracle import layout as L


@pytest.mark.parametrize("seed", [0, 1])
def test_layout_model_reproduces_the_reference_tests_truth(seed):
  c = L.reference_forward_case(seed)
  assert len(c["cfgs"].feature_configs) == 199 and c["batch"] == 256
  got = L.layout_model(c["embs"], c["fid_offset"], c["feature_offset"], c["nfl_offset"], c["batch"], c["cfgs"])
  assert len(got) == len(c["expected"])
  shapes = [g.shape for g in got]
  assert (256, 1) in shapes and (256, 49, 8) in shapes and (256, 3, 20) in shapes     # bias, ffm1, firstN
  for g, e in zip(got, c["expected"]):
    assert g.shape == e.shape
    assert np.allclose(e, g, rtol=1e-4, atol=1e-7)
    assert np.abs(e).max() > 0


def test_layout_grad_model_reproduces_the_reference_tests_truth():
  c = L.reference_grad_case(0)
  got = L.layout_grad_model(c["embs"], c["fid_offset"], c["feature_offset"], c["nfl_offset"], c["batch"],
                            c["cfgs"], c["tensors_grad"])
  touched = 0
  for g, e in zip(got, c["expected_grads"]):
    assert g.shape == e.shape
    np.testing.assert_allclose(g, e, rtol=1e-4, atol=1e-7)
    touched += int((e != 0).sum())
  assert touched > 10000
  # the sequential fp32 form (what the product's atomic-free gradient must equal bit for bit) is the
  # same walk with fp32 adds: it meets the reference test's truth at the test's tolerance too
  seq = L.layout_grad_model(c["embs"], c["fid_offset"], c["feature_offset"], c["nfl_offset"], c["batch"],
                            c["cfgs"], c["tensors_grad"], acc_dtype=np.float32)
  for g, e in zip(seq, c["expected_grads"]):
    assert g.dtype == np.float32
    np.testing.assert_allclose(g, e, rtol=1e-4, atol=1e-6)


def test_small_case_by_hand():
  """two features, one shared: every number written out"""
  cfgs = L.FeatureConfigs(
      {"a": L.FeatureConfig("t", L.SUM, [1, 2], 0), "b": L.FeatureConfig("t", L.MEAN, [1, 2], 0)},
      {"bias": L.infer_shape([L.SliceConfig("a", 0, 1), L.SliceConfig("b", 0, 1)], L.ADDN),
       "vec": L.infer_shape([L.SliceConfig("a", 1, 3), L.SliceConfig("b", 1, 3)], L.CONCAT)})
  embs = [np.array([[1, 10, 20], [2, 30, 40]], np.float32), np.array([[4, 1, 2], [8, 3, 6]], np.float32)]
  # a: batch row 0 -> rows 0 and 1 of matrix 0, batch row 1 -> row 1; b: shared, rows 0 and 1 of matrix 1
  fid_offset = np.array([(0 << 32) | 0, (0 << 32) | 1, (0 << 32) | 1, (1 << 32) | 0, (1 << 32) | 1], np.uint64)
  feature_offset = np.array([0, 2, 3], np.int32)
  nfl_offset = np.array([0, 2 | L.SHARD_BIT], np.uint32)
  bias, vec = L.layout_model(embs, fid_offset, feature_offset, nfl_offset, 2, cfgs)
  np.testing.assert_array_equal(bias, [[1 + 2 + 6], [2 + 6]])
  np.testing.assert_array_equal(vec, [[40, 60, 2, 4], [30, 40, 2, 4]])
  g = L.layout_grad_model(embs, fid_offset, feature_offset, nfl_offset, 2, cfgs,
                          [np.ones((2, 1), np.float32), np.ones((2, 4), np.float32)])
  np.testing.assert_array_equal(g[0], [[1, 1, 1], [2, 2, 2]])
  np.testing.assert_array_equal(g[1], [[1, 1, 1], [1, 1, 1]])      # 2 batch rows x 1/2 each
