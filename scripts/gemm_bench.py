#!/usr/bin/env python
"""The dense tower on its own: forward + backward + SGD of the bench's MLP (1024-1024-512-256-1 at
batch 65 536) through mhte_dense_mlp_* (hand-written bf16 MFMA GEMMs), per-kernel HIP-event times of
the GEMM launches, and the same MLP through torch (hipBLASLt, the split-K form of bench.py --dense).
One JSON object per line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monolith_amd import _lib  # noqa: E402
from monolith_amd.dense_mlp import DenseMlp  # noqa: E402

PEAK = 2500.0  # dense bf16 TFLOP/s, MI355X_MICROARCH.md


def timed(fn, reps):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / reps


def main():
  widths = [int(w) for w in (sys.argv[1] if len(sys.argv) > 1 else "1024,1024,512,256,1").split(",")]
  B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
  dev = torch.device("cuda", 0)
  torch.manual_seed(0)
  mlp = DenseMlp(widths, max_batch=B)
  lins = [torch.nn.Linear(a, b).to(dev) for a, b in zip(widths[:-1], widths[1:])]
  for i, m in enumerate(lins):
    mlp.set_params(i, m.weight, m.bias)
  x = torch.randn(B, widths[0], device=dev)
  dy = torch.full((B,), 1.0 / B, device=dev)
  y = torch.empty(B, device=dev)
  dx = torch.empty(B, widths[0], device=dev)
  flops = 6.0 * B * sum(a * b for a, b in zip(widths[:-1], widths[1:]))

  def step():
    mlp.forward(x, out=y)
    mlp.backward(dy, 1e-3, out=dx)

  us = timed(step, 10)
  print(json.dumps({"what": "mhte_dense_mlp forward + backward + SGD", "widths": widths, "batch": B,
                    "us": round(us, 1), "TFLOPs": round(flops / us / 1e6, 1), "of_bf16_peak": round(flops / us / 1e6 / PEAK, 4)}))
  # per launch
  _lib.profile_arm(64)
  step()
  torch.cuda.synchronize()
  rec = [(n, u) for n, u in _lib.profile_read() if n == "gemm_nt_bf16_kernel"]
  names = []
  for l in range(len(widths) - 2):
    names.append("fwd %d->%d" % (widths[l], widths[l + 1]))
  for l in range(len(widths) - 3, -1, -1):
    names.append("wgrad %dx%d" % (widths[l + 1], widths[l]))
    names.append("dgrad %d->%d" % (widths[l + 1], widths[l]))
  gsum = 0.0
  for (n, u), what in zip(rec, names):
    a, b_ = [int(v) for v in what.split()[1].replace("->", "x").split("x")]
    fl = 2.0 * B * a * b_
    gsum += u
    print(json.dumps({"kernel": "gemm_nt_bf16_kernel", "what": what, "us": round(u, 1),
                      "TFLOPs": round(fl / u / 1e6, 1), "of_bf16_peak": round(fl / u / 1e6 / PEAK, 4)}))
  print(json.dumps({"what": "GEMM launches of a step, summed", "us": round(gsum, 1),
                    "TFLOPs": round(flops / gsum / 1e6, 1), "of_bf16_peak": round(flops / gsum / 1e6 / PEAK, 4),
                    "other_kernels_us": round(us - gsum, 1)}))

  # torch / hipBLASLt reference of the same arithmetic (bf16 autocast, SGD)
  ref = torch.nn.Sequential(*[m for l in lins for m in (l, torch.nn.ReLU())][:-1])
  opt = torch.optim.SGD(ref.parameters(), lr=1e-3)
  xr = x.clone().requires_grad_(True)

  def tstep():
    xr.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
      yy = ref(xr)
    loss = yy.float().mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()

  tus = timed(tstep, 10)
  print(json.dumps({"what": "torch autocast bf16 (hipBLASLt) forward + backward + SGD", "us": round(tus, 1),
                    "TFLOPs": round(flops / tus / 1e6, 1), "of_bf16_peak": round(flops / tus / 1e6 / PEAK, 4)}))


if __name__ == "__main__":
  main()
