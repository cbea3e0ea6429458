#!/usr/bin/env python3
"""Times mz_conv_wgrad_bf16 (both of its kernels together) at a training batch.  usage: time_wgrad.py [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from muzero_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
g = torch.randn(B, 12, 12, 256, device="cuda").to(torch.bfloat16)
x = torch.randn(B, 12, 12, 256, device="cuda").to(torch.bfloat16)
dw = torch.empty(256, 256, 3, 3, device="cuda")
db = torch.empty(256, device="cuda")
ws = torch.empty(_lib.lib().mz_conv_wgrad_workspace_bytes(B) // 4, device="cuda")
run = lambda: _lib.call("mz_conv_wgrad_bf16", _lib.ptr(g), _lib.ptr(x), B, 12, 12, 256, 256, _lib.ptr(dw), _lib.ptr(db),
                        _lib.ptr(ws), ws.numel() * 4, _lib.current_stream())
for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 200 * 1e3
print(f"wgrad + reduce, B={B}: {us:.1f} us ({2.0 * B * 144 * 256 * 2304 / us / 1e6:.0f} TFLOP/s)")
