"""One sparse training step of a MonolithModel on a single table, as the reference's worker + PS
execute it (native_training/distributed_ps.py:282-329 lookup, :489-514 apply_gradients), with every
stage on the GPU and no host round trip inside the step:

   dedup (first-occurrence order)                unique_key_with_value_and_offset
   lookup of the unique ids                       MonolithMultiHashTableLookup
   scatter of unique rows to every occurrence     MonolithFillWithOffsetMap
   ... dense forward / backward (the caller's) ...
   duplicate-gradient sum (occurrence order)      MonolithFillWithOffsetMapGradient
   optimizer apply on the unique ids              MonolithMultiHashTableOptimize

The unique-id count stays in device memory; kernels are launched for the batch-size upper bound
and mask themselves."""
from typing import Optional

import numpy as np
import torch

from monolith_amd import _lib
from monolith_amd.distribution_ops import DedupWorkspace, UniqueResult
from monolith_amd.multi_hash_table_ops import MultiHashTable


class SparseStep:

  def __init__(self, table: MultiHashTable, table_name: str, batch: int,
               exact_order: bool = False, direct: bool = True):
    self.direct = direct
    self._joined = True
    self.table = table
    self.name = table_name
    self.idx = table._index(table_name)  # pylint: disable=protected-access
    self.dim = table.get_table_dim_sizes()[self.idx]
    self.batch = batch
    self.exact_order = exact_order
    dev = torch.device("cuda:%d" % table._device)  # pylint: disable=protected-access
    self.ws = DedupWorkspace(dev.index)
    self.side = torch.cuda.Stream(device=dev)
    n = batch
    self.u = UniqueResult(torch.empty(n, dtype=torch.int64, device=dev),
                          torch.empty(n, dtype=torch.int32, device=dev),
                          torch.empty(n + 1, dtype=torch.int32, device=dev),
                          torch.empty(n, dtype=torch.int32, device=dev),
                          torch.zeros(1, dtype=torch.int32, device=dev), None)
    self.emb_u = torch.empty((n, self.dim), dtype=torch.float32, device=dev)
    self.emb = torch.empty((n, self.dim), dtype=torch.float32, device=dev)
    self.grad_u = torch.empty((n, self.dim), dtype=torch.float32, device=dev)
    lr0 = sum(table._slice_sizes[:self.idx])  # pylint: disable=protected-access
    self.lrs = np.ascontiguousarray(
        table.learning_rate[lr0:lr0 + table._slice_sizes[self.idx]])  # pylint: disable=protected-access

  def forward(self, ids: torch.Tensor) -> torch.Tensor:
    """Rows for every occurrence.  ``direct`` (default): ONE probe+gather kernel over the B
    occurrences (duplicates of a Zipf head key are served from L2) — the same values as the
    reference's dedup -> lookup(unique) -> FillWithOffsetMap, without waiting for the dedup.
    ``direct=False`` keeps the reference's three-op shape."""
    assert ids.numel() == self.batch
    if self.direct:
      # the dedup chain (needed by backward only) runs on a side stream beside the lookup
      main = torch.cuda.current_stream()
      self.side.wait_stream(main)
      with torch.cuda.stream(self.side):
        self.ws.unique(ids, want_host_count=False, out=self.u)
      self.table.table_lookup_n(self.idx, ids, None, self.emb, n_max=self.batch)
      self._joined = False
    else:
      self.ws.unique(ids, want_host_count=False, out=self.u)
      self.table.table_lookup_n(self.idx, self.u.unique_ids, self.u.n_unique_dev, self.emb_u,
                                n_max=self.batch)
      self.ws.gather_rows(self.emb_u, self.u.inverse, self.batch, self.dim, out=self.emb)
    return self.emb

  def backward(self, grads: torch.Tensor, update_time: int, global_step: int = 0):
    if self.direct and not self._joined:
      torch.cuda.current_stream().wait_stream(self.side)
      self._joined = True
    self.ws.segment_sum(grads, self.u, self.dim, out=self.grad_u, exact_order=self.exact_order)
    self.table.table_optimize_n(self.idx, self.u.unique_ids, self.u.n_unique_dev, self.grad_u,
                                self.lrs, update_time, global_step, flags=_lib.MHTE_IDS_UNIQUE,
                                n_max=self.batch)

  def n_unique(self) -> int:
    if self.direct and not self._joined:
      torch.cuda.current_stream().wait_stream(self.side)
      self._joined = True
    return int(self.u.n_unique_dev.item())
