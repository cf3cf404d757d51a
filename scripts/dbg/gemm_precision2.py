"""Debug: element-wise agreement of the SECOND and THIRD GEMM layers (their input is the first layer's
stored bf16 output) with the fp32 torch reference, and of the input gradient."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from monolith_amd.dense_mlp import DenseMlp

def bf(t): return t.to(torch.bfloat16).to(torch.float32)
torch.manual_seed(0)
B = 2048
for widths in ([1024, 1024, 512], [1024, 1024, 512, 256]):
  lins = [torch.nn.Linear(a, b).cuda() for a, b in zip(widths[:-1], widths[1:])]
  x = torch.randn(B, widths[0], device="cuda")
  mlp = DenseMlp(widths + [1], max_batch=B)
  h = bf(x)
  for i, l in enumerate(lins):
    mlp.set_params(i, l.weight, l.bias)
    h = bf(torch.relu(h @ bf(l.weight.detach()).t() + l.bias.detach()))
  N = widths[-1]
  bad = tot = 0; worst = 0.0
  for j in (0, 1, 33, 64, 127, N - 1, N // 2 + 5):
    w = torch.zeros(1, N, device="cuda"); w[0, j] = 1.0
    mlp.set_params(len(lins), w, torch.zeros(1, device="cuda"))
    y = mlp.forward(x)
    d = (y - h[:, j]).abs()
    bad += int((d > 0).sum()); tot += B
    worst = max(worst, float((d / (h[:, j].abs() + 1e-30)).max()))
  print("widths %s: %d of %d elements of the last hidden layer differ, worst relative %.3g" % (widths, bad, tot, worst))
  # logits with a dense last layer
  wl = torch.randn(1, N, device="cuda") * 0.05
  mlp.set_params(len(lins), wl, torch.zeros(1, device="cuda"))
  y = mlp.forward(x)
  yr = h @ wl.view(-1)
  print("   logits: max abs diff %.3g, max |y| %.3g" % (float((y - yr).abs().max()), float(yr.abs().max())))
  mlp.close()
