#!/usr/bin/env python3
"""Times the fused training-mode BatchNorm kernels on one layer shape (run under rocprofv3 --kernel-trace --stats
for the per-kernel split).  usage: time_bn.py [images] [side] [channels]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from muzero_amd.learner import bn_act  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
S = int(sys.argv[2]) if len(sys.argv) > 2 else 12
Cc = int(sys.argv[3]) if len(sys.argv) > 3 else 256
bn = nn.BatchNorm2d(Cc).cuda()
bn.native = True
x = torch.randn(N, Cc, S, S, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
res = torch.randn_like(x).contiguous(memory_format=torch.channels_last).requires_grad_()
g = torch.randn_like(x).contiguous(memory_format=torch.channels_last)


def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = bn_act(bn, x, res)
    y.backward(g)


for _ in range(5):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    step()
e1.record()
torch.cuda.synchronize()
print(f"BatchNorm + residual + ReLU forward and backward, {N} x {S}x{S}x{Cc}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us "
      f"(tensor = {x.numel() * 2 / 1e6:.1f} MB)")
