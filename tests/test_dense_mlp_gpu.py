"""GPU test (-m gpu) of the dense tower on the matrix cores (mhte_dense_mlp_*, csrc/mhte_gemm_kernels.h)
against a plain PyTorch fp32 reference of the same arithmetic: fp32 matmuls over operands rounded to
bf16 at the points where the kernels round them (inputs, weights, stored activations and their
gradients), fp32 accumulation, fp32 master weights.  With the roundings mirrored the comparison is
tight for one hidden layer and looser for deeper towers (see SHALLOW / DEEP below); against an unrounded fp32
MLP the same outputs agree to bf16's 2e-2 (also checked, forward only — a pre-activation near zero
flips its ReLU under rounding, so gradients of single units differ by design)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from monolith_amd.dense_mlp import DenseMlp  # noqa: E402


def _bf16(t):
  return t.to(torch.bfloat16).to(torch.float32)


class RefMlp:
  """fp32 master weights; every GEMM operand rounded to bf16 as the kernels do."""

  def __init__(self, widths, seed):
    torch.manual_seed(seed)
    self.w, self.b = [], []
    for a, b in zip(widths[:-1], widths[1:]):
      lin = torch.nn.Linear(a, b)
      self.w.append(lin.weight.detach().cuda().clone())
      self.b.append(lin.bias.detach().cuda().clone())

  def forward(self, x):
    self.h = [_bf16(x)]
    for w, b in zip(self.w[:-1], self.b[:-1]):
      self.h.append(_bf16(torch.relu(self.h[-1] @ _bf16(w).t() + b)))
    return self.h[-1] @ self.w[-1].view(-1) + self.b[-1]          # (the last layer reads fp32 weights)

  def backward(self, dy, lr):
    gw, gb = [None] * len(self.w), [None] * len(self.w)
    top = self.h[-1]
    gw[-1] = (dy.view(-1, 1) * top).sum(0).view(1, -1)
    gb[-1] = dy.sum().view(1)
    dz = _bf16((top > 0).float() * dy.view(-1, 1) * self.w[-1].view(1, -1))
    dx = None
    for l in range(len(self.w) - 2, -1, -1):
      gw[l] = dz.t() @ self.h[l]
      gb[l] = dz.sum(0)
      d = dz @ _bf16(self.w[l])
      if l > 0:
        dz = _bf16(d * (self.h[l] > 0).float())
      else:
        dx = d
    for l in range(len(self.w)):
      self.w[l] = self.w[l] - lr * gw[l]
      self.b[l] = self.b[l] - lr * gb[l]
    return dx


def _close(got, exp, rel, what):
  """max |got - exp| <= rel * max |exp|."""
  scale = float(exp.abs().max()) + 1e-20
  err = float((got - exp).abs().max())
  assert err <= rel * scale, (what, err, scale)


def _close_norm(got, exp, rel, what):
  """||got - exp|| <= rel * ||exp||: for tensors where a flipped bf16 rounding of ONE stored activation
  moves a few elements by a visible amount (a unit's ReLU gate, its whole column of the gradient)
  while the rest agrees to fp32 noise."""
  err = float((got - exp).norm()) / (float(exp.norm()) + 1e-30)
  assert err <= rel, (what, err)


# one hidden layer: nothing amplifies a flipped rounding — tight (the kernels themselves agree with an
# fp64 product to 1e-7, scripts/dbg/gemm_precision3.py; what remains is a unit whose pre-activation is
# zero to fp32 noise: its gate, and with it one row of the input gradient); deeper towers: a stored activation
# that lands on the other side of a bf16 rounding boundary (fp32 summation order; ~5e-5 of the elements
# of a 1024-wide layer, measured with scripts/dbg/gemm_precision.py — the same rate at which torch's own
# fp32 and fp64 products disagree) shifts every pre-activation of the next layer by ~1e-4 of its scale,
# which flips the ReLU gate of the units that sit that close to zero: ~1e-3 of the logits' scale, and
# a Frobenius 1-2e-2 on the input gradient, growing slowly as the two weight sets drift apart.
SHALLOW = [([128, 128, 1], 128), ([256, 384, 1], 512), ([1024, 1024, 1], 2048)]
DEEP = [([256, 384, 128, 1], 512), ([1024, 1024, 512, 256, 1], 2048)]


@pytest.mark.parametrize("widths,batch,deep", [(w, b, False) for w, b in SHALLOW] + [(w, b, True) for w, b in DEEP])
def test_forward_backward_against_torch_fp32(widths, batch, deep):
  ref = RefMlp(widths, 7)
  mlp = DenseMlp(widths, max_batch=batch)
  for i in range(len(ref.w)):
    mlp.set_params(i, ref.w[i], ref.b[i])
  plain = [(w.clone(), b.clone()) for w, b in zip(ref.w, ref.b)]
  g = torch.Generator(device="cuda").manual_seed(3)
  x = torch.randn(batch, widths[0], device="cuda", generator=g)
  lr = 0.05
  for step in range(3):
    y_ref = ref.forward(x)
    y = mlp.forward(x)
    _close(y, y_ref, 5e-3 if deep else 2e-3, ("logits", step))
    if step == 0:   # against the unrounded fp32 MLP: bf16's tolerance
      hh = x
      for w, b in plain[:-1]:
        hh = torch.relu(hh @ w.t() + b)
      _close(y, hh @ plain[-1][0].view(-1) + plain[-1][1], 2e-2, "logits vs unrounded fp32")
    dy = torch.randn(batch, device="cuda", generator=g) / batch
    dx_ref = ref.backward(dy, lr)
    dx = mlp.backward(dy, lr)
    _close_norm(dx, dx_ref, 5e-2 if deep else 1e-2, ("input gradient", step))   # (one flipped gate = one row)
    for i in range(len(ref.w)):
      w, b = mlp.get_params(i)
      _close(w.view_as(ref.w[i]), ref.w[i], 5e-4, ("weights", step, i))
      _close(b.view_as(ref.b[i]), ref.b[i], 5e-3, ("bias", step, i))
    x = x + 0.1 * torch.randn(batch, widths[0], device="cuda", generator=g)
  mlp.close()


@pytest.mark.parametrize("K,N,B", [(1024, 1024, 2048), (256, 384, 512)])
def test_one_gemm_layer_element_by_element(K, N, B):
  """The last layer as a one-hot row picks single columns of the hidden layer, so the logits ARE the
  stored bf16 activations h[:, j] of ONE MFMA GEMM (256 x 256 tiles for the first shape, 128 x 128 for
  the second): bit-identical to bf16(relu(bf16(x) bf16(W)^T + b)) computed by torch in fp32 except
  where the fp32 summation order puts a value on the other side of a rounding boundary — at most 1e-3
  of the elements (measured 5e-5; torch's own fp32 and fp64 products disagree at 7e-5), and those by
  one unit in the last place."""
  torch.manual_seed(11)
  lin = torch.nn.Linear(K, N).cuda()
  x = torch.randn(B, K, device="cuda")
  mlp = DenseMlp([K, N, 1], max_batch=B)
  mlp.set_params(0, lin.weight, lin.bias)
  h = _bf16(torch.relu(_bf16(x) @ _bf16(lin.weight.detach()).t() + lin.bias.detach()))
  differ = total = 0
  for j in (0, 1, 31, 32, 63, 64, 127, 128, N // 2 + 5, N - 1):
    w = torch.zeros(1, N, device="cuda")
    w[0, j] = 1.0
    mlp.set_params(1, w, torch.zeros(1, device="cuda"))
    y = mlp.forward(x)
    d = (y - h[:, j]).abs()
    differ += int((d > 0).sum())
    total += B
    assert float((d / (h[:, j].abs() + 1e-30))[d > 0].max() if (d > 0).any() else 0.0) <= 2.0 ** -7
  assert differ <= 1e-3 * total, (differ, total)
  mlp.close()


def test_identity_weights_place_every_element():
  """A = I against an ASYMMETRIC B (cdna_hip_programming.md: a symmetric one passes a kernel whose C
  write has rows and columns swapped): one layer with W[n][k] = n * 0.01 - k * 0.003 and one-hot rows."""
  K, N, B = 128, 256, 128
  mlp = DenseMlp([K, N, 1], max_batch=B)
  n_i = torch.arange(N, device="cuda", dtype=torch.float32).view(N, 1)
  k_i = torch.arange(K, device="cuda", dtype=torch.float32).view(1, K)
  W = _bf16(n_i * 0.01 - k_i * 0.003 + 1.0)          # all positive: ReLU is the identity
  mlp.set_params(0, W, torch.zeros(N, device="cuda"))
  w_last = _bf16(torch.linspace(0.5, 1.5, N, device="cuda"))
  mlp.set_params(1, w_last.view(1, N), torch.zeros(1, device="cuda"))
  x = torch.zeros(B, K, device="cuda")
  x[torch.arange(B), torch.arange(B) % K] = 1.0          # row m selects column k = m of W
  y = mlp.forward(x)
  exp = (_bf16(W[:, torch.arange(B) % K].t()) * w_last.view(1, N)).sum(1)   # h = bf16(W[:, m]); y = h . w_last
  np.testing.assert_allclose(y.cpu().numpy(), exp.cpu().numpy(), rtol=1e-5, atol=1e-5)
  mlp.close()


def test_argument_errors():
  from monolith_amd import _lib
  with pytest.raises(_lib.MhteError):
    DenseMlp([100, 128, 1], max_batch=128)      # widths are multiples of 128
  with pytest.raises(_lib.MhteError):
    DenseMlp([128, 128, 2], max_batch=128)      # one output
  mlp = DenseMlp([128, 128, 1], max_batch=256)
  with pytest.raises(_lib.MhteError):
    mlp.forward(torch.zeros(64, 128, device="cuda"))   # batch: a multiple of 128
  mlp.close()
