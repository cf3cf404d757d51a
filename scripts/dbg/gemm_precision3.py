"""Debug: input gradient of one-hidden-layer towers against fp32 and fp64 torch references."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from monolith_amd.dense_mlp import DenseMlp
def bf(t): return t.to(torch.bfloat16).to(torch.float32)
torch.manual_seed(0)
for K, N, B in ((128, 128, 128), (256, 384, 512), (1024, 1024, 2048)):
  l0 = torch.nn.Linear(K, N).cuda(); l1 = torch.nn.Linear(N, 1).cuda()
  x = torch.randn(B, K, device="cuda"); dy = torch.randn(B, device="cuda") / B
  mlp = DenseMlp([K, N, 1], max_batch=B)
  mlp.set_params(0, l0.weight, l0.bias); mlp.set_params(1, l1.weight, l1.bias)
  y = mlp.forward(x); dx = mlp.backward(dy, 0.0)
  W0 = bf(l0.weight.detach()); h = bf(torch.relu(bf(x) @ W0.t() + l0.bias.detach()))
  dz = bf((h > 0).float() * dy.view(-1, 1) * l1.weight.detach().view(1, -1))
  dx32 = dz @ W0
  dx64 = (dz.double() @ W0.double()).float()
  n = lambda a, b: float((a - b).norm() / b.norm())
  # which rows differ?
  rowerr = (dx - dx64).norm(dim=1) / (dx64.norm(dim=1) + 1e-30)
  print("K %d N %d B %d: |mine - fp64| %.3g  |torch32 - fp64| %.3g  rows with error > 1e-3: %d of %d (max %.3g)"
        % (K, N, B, n(dx, dx64), n(dx32, dx64), int((rowerr > 1e-3).sum()), B, float(rowerr.max())))
  mlp.close()
