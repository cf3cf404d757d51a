"""Seeded synthetic id / gradient streams shared by the parity tests, smoke() and bench.py
(SURVEY.md §8d).  Nothing here touches the GPU: batches are numpy arrays the caller uploads.

  * Zipf(a) over a universe of V ranks: rank r drawn from the truncated zeta pmf p(r) ~ r^-a,
    r in [1, V], by rejection from numpy's Generator.zipf (draws with r > V are discarded);
  * rank -> 48-bit signature by the fixed bijection sig = (r * 0x9E3779B97F4A7C15) mod 2^48, so that
    hot ids are not contiguous;
  * fid = (feature_slot << 48) | sig — FID v2 layout with bit 63 = 0
    (reference data/training_instance/cc/fid.h:61-68, reader_util.h:34-38);
  * per-step seeds: ids 20260921 + step, gradients 20260921 + 10^9 + step.
"""
import numpy as np

SEED0 = 20260921
_MULT = np.uint64(0x9E3779B97F4A7C15)
_MASK48 = np.uint64((1 << 48) - 1)


def rank_to_fid(rank: np.ndarray, feature_slot: int = 1) -> np.ndarray:
  sig = (rank.astype(np.uint64) * _MULT) & _MASK48
  return ((np.uint64(feature_slot) << np.uint64(48)) | sig).astype(np.int64)


def zipf_ranks(rng: np.random.Generator, n: int, universe: int, a: float = 1.2) -> np.ndarray:
  out = np.empty(n, dtype=np.int64)
  filled = 0
  while filled < n:
    draw = rng.zipf(a, size=int((n - filled) * 1.1) + 16)
    draw = draw[(draw >= 1) & (draw <= universe)]
    take = min(draw.size, n - filled)
    out[filled:filled + take] = draw[:take]
    filled += take
  return out


def id_batch(step: int, batch: int, universe: int, dist: str = "zipf", a: float = 1.2,
             feature_slot: int = 1) -> np.ndarray:
  rng = np.random.Generator(np.random.PCG64(SEED0 + step))
  if dist == "zipf":
    ranks = zipf_ranks(rng, batch, universe, a)
  elif dist == "uniform":
    ranks = rng.integers(1, universe + 1, size=batch, dtype=np.int64)
  else:
    raise ValueError(dist)
  return rank_to_fid(ranks, feature_slot)


def grad_batch(step: int, batch: int, dim: int) -> np.ndarray:
  rng = np.random.Generator(np.random.PCG64(SEED0 + 10**9 + step))
  return (rng.standard_normal((batch, dim), dtype=np.float32) * np.float32(0.01))


def update_time(step: int) -> int:
  return 1_700_000_000 + step
