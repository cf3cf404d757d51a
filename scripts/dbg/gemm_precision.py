"""Debug: element-wise agreement of ONE GEMM layer of the tower with an fp32 torch reference over
bf16-rounded operands: the last layer picks single columns (one-hot weights), so y IS h[:, j]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from monolith_amd.dense_mlp import DenseMlp

def bf(t): return t.to(torch.bfloat16).to(torch.float32)
torch.manual_seed(0)
for K, N, B in ((1024, 1024, 2048), (256, 256, 512), (128, 128, 128)):
  lin = torch.nn.Linear(K, N).cuda()
  x = torch.randn(B, K, device="cuda")
  mlp = DenseMlp([K, N, 1], max_batch=B)
  mlp.set_params(0, lin.weight, lin.bias)
  pre = bf(x) @ bf(lin.weight.detach()).t() + lin.bias.detach()
  pre64 = (bf(x).double() @ bf(lin.weight.detach()).double().t() + lin.bias.detach().double())
  h_ref = bf(torch.relu(pre))
  h_ref64 = bf(torch.relu(pre64).float())
  bad = tot = 0
  worst = 0.0
  bad64 = 0
  for j in (0, 1, 31, 32, 63, 64, 127, N - 1, N // 2 + 5):
    w = torch.zeros(1, N, device="cuda"); w[0, j] = 1.0
    mlp.set_params(1, w, torch.zeros(1, device="cuda"))
    y = mlp.forward(x)
    d = (y - h_ref[:, j]).abs()
    bad += int((d > 0).sum()); tot += B
    bad64 += int(((y - h_ref64[:, j]).abs() > 0).sum())
    worst = max(worst, float((d / (h_ref[:, j].abs() + 1e-30)).max()))
  print("K %d N %d B %d: %d of %d elements differ from the fp32 torch reference (%d from the fp64 one), worst relative %.3g"
        % (K, N, B, bad, tot, bad64, worst))
  print("   torch fp32 vs fp64 reference itself: %d of %d differ" % (int((h_ref != h_ref64).sum()), h_ref.numel()))
  mlp.close()
