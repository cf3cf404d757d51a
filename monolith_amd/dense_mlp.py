"""The dense tower downstream of the embedding path on the MI355X matrix cores (C ABI
mhte_dense_mlp_*, csrc/mhte_gemm_kernels.h): the ranking MLP that consumes
``fused_embedding_to_layout``'s output — a stack of Dense + ReLU layers ending in one logit
(native_training/layers/mlp.py) — as hand-written bf16 MFMA GEMMs with fp32 master weights, fp32
accumulation and SGD inside ``backward``.

  mlp = DenseMlp([1024, 1024, 512, 256, 1], max_batch=65536)
  y = mlp.forward(x)                 # x [B, 1024] fp32 on the GPU -> y [B] fp32
  dx = mlp.backward(dy, lr=1e-3)     # dy [B] = dLoss/dy; SGD step on every layer; dx [B, 1024]

Widths (but the final 1) and batches are multiples of 128 (the GEMM tile).  No fallback: without the
library or a GPU every call raises."""
import ctypes as C

import torch

from . import _lib


class DenseMlp:
  def __init__(self, widths, max_batch, device=None):
    self.widths = [int(w) for w in widths]
    self.max_batch = int(max_batch)
    self._lib = _lib.lib()
    import os
    if os.environ.get("MHTE_DENSE_LIBRARY"):   # development builds of the tower alone (scripts/dbg/gemm_dev.hip)
      self._lib = C.CDLL(os.environ["MHTE_DENSE_LIBRARY"])
      self._lib.mhte_dense_mlp_destroy.restype = None
      self._lib.mhte_dense_mlp_destroy.argtypes = [C.c_void_p]
    self._device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
    arr = (C.c_int32 * len(self.widths))(*self.widths)
    h = C.c_void_p()
    _lib.check(self._lib.mhte_dense_mlp_create(arr, C.c_int32(len(self.widths)), C.c_int64(self.max_batch),
                                               C.c_int32(self._device.index), C.byref(h)))
    self._h = h
    self._batch = 0

  @property
  def n_layers(self):
    return len(self.widths) - 1

  def _stream(self):
    return C.c_void_p(torch.cuda.current_stream(self._device).cuda_stream)

  def set_params(self, layer, weight, bias):
    """weight [out, in] (torch.nn.Linear's layout; [1, in] or [in] for the last layer), bias [out]."""
    w = weight.detach().to(self._device, torch.float32).contiguous()
    b = bias.detach().to(self._device, torch.float32).contiguous()
    assert w.numel() == self.widths[layer] * self.widths[layer + 1] and b.numel() == self.widths[layer + 1]
    _lib.check(self._lib.mhte_dense_mlp_set_params(self._h, C.c_int32(layer), _lib.vp(w), _lib.vp(b),
                                                   self._stream()))
    torch.cuda.current_stream(self._device).synchronize()   # (w, b may be temporaries)

  def get_params(self, layer):
    w = torch.empty(self.widths[layer + 1], self.widths[layer], dtype=torch.float32, device=self._device)
    b = torch.empty(self.widths[layer + 1], dtype=torch.float32, device=self._device)
    _lib.check(self._lib.mhte_dense_mlp_get_params(self._h, C.c_int32(layer), _lib.vp(w), _lib.vp(b),
                                                   self._stream()))
    return w, b

  def forward(self, x, out=None):
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape[1] == self.widths[0]
    B = x.shape[0]
    y = out if out is not None else torch.empty(B, dtype=torch.float32, device=self._device)
    _lib.check(self._lib.mhte_dense_mlp_forward(self._h, _lib.vp(x), C.c_int64(B), _lib.vp(y), self._stream()))
    self._batch = B
    self._keep = x
    return y

  def backward(self, dy, lr, need_dx=True, out=None):
    assert dy.is_cuda and dy.dtype == torch.float32 and dy.is_contiguous() and dy.numel() == self._batch
    dx = None
    if need_dx:
      dx = out if out is not None else torch.empty(self._batch, self.widths[0], dtype=torch.float32,
                                                   device=self._device)
    _lib.check(self._lib.mhte_dense_mlp_backward(self._h, _lib.vp(dy), _lib.vp(dx), C.c_float(float(lr)),
                                                 self._stream()))
    return dx

  def close(self):
    if getattr(self, "_h", None):
      torch.cuda.synchronize(self._device)
      self._lib.mhte_dense_mlp_destroy(self._h)
      self._h = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass
