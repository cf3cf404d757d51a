"""monolith_amd — MI355X-native collisionless embedding-table engine for Monolith's sparse path.

Only the hot path named by BASELINE.json's north_star lives here: the MultiHashTable ops
(lookup / optimize / assign / assign_add / reinitialize / fused_lookup / fused_optimize), the
caller-side dedup and packing ops around them, and the id-sharded all-to-all exchange.  Compute
is hand-written HIP for gfx950 in ``csrc/`` behind the C ABI of
``include/monolith_amd_hash_table.h``; PyTorch is used for device memory, streams and
``torch.distributed`` only.  There is no CPU fallback: without the HIP library or a GPU every op
raises.
"""
from monolith_amd import entry  # noqa: F401
from monolith_amd._lib import build_library, library_path, MhteError  # noqa: F401

__all__ = ["entry", "build_library", "library_path", "MhteError"]
