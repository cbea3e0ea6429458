#!/usr/bin/env python3
"""One 12x12 256->256 convolution (mz_conv3x3_bf16, the training step's forward / data-gradient kernel) at
batches from 1 to 1024: microseconds per launch by HIP events.  At one image the launch is already ~18 us (the
critical path of a workgroup: image into LDS, 72 k-steps, epilogue); at 128 images - two workgroups per image on
256 CUs - 24 us.

    python tools/trunk_scaling.py        (MUZERO_AMD_LIB=<variant .so> to compare builds)
"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from muzero_amd import _lib, mfma
def timed(fn, iters=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
w = torch.randn(256, 256, 3, 3) / 48
wpk = mfma.pack_weights(w, 256).cuda()
bias = torch.zeros(256, device="cuda")
for B in (1, 16, 64, 128, 129, 256, 512, 1024):
    x = torch.randn(B, 12, 12, 256, device="cuda").to(torch.bfloat16)
    out = torch.empty_like(x)
    t = timed(lambda: _lib.call("mz_conv3x3_bf16", _lib.ptr(x), None, 0, _lib.ptr(wpk), _lib.ptr(bias), None, _lib.ptr(out), None, None, None, B, 12, 12, 256, 256, 1, 0, _lib.current_stream()))
    print(f"B={B:5d}: {t:7.1f} us  {B*0.1699/t:6.3f} PFLOP/s")
