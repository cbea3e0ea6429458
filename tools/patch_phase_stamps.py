#!/usr/bin/env python3
"""Where the cycles of a patch go in the first convolution (on the rotating observation) and in the 48x48 stride-2
layer: a diagnostic build records the shader clock at the phase boundaries of sampled workgroups (second half
of the grid: steady state), and this prints the median per phase.

    make -C muzero_amd/csrc stamps
    MUZERO_AMD_LIB=$PWD/muzero_amd/csrc/libmuzero_amd_stamps.so python3 tools/patch_phase_stamps.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from muzero_amd import _lib, mfma  # noqa: E402

lib = _lib.lib()
lib.mz_debug_read_patch_stamps.restype = C.c_int
buf = (C.c_longlong * (64 * 32))()
B = 1024


class L:
    pass


def read(n_chunks, name, flops, us):
    assert lib.mz_debug_read_patch_stamps(buf) == 0
    a = np.array(buf[:]).reshape(64, 32)
    a = a[a[:, 18] > a[:, 0]]
    a = a[a[:, 0] > 0]
    rows = []
    tot = {"window loads + LDS writes": 0, "barrier": 0, "main loop": 0, "wait for the other wavefronts": 0}
    for c in range(n_chunks):
        s = a[:, 4 * c:4 * c + 4]
        tot["window loads + LDS writes"] += np.median(s[:, 1] - s[:, 0])
        tot["barrier"] += np.median(s[:, 2] - s[:, 1])
        tot["main loop"] += np.median(s[:, 3] - s[:, 2])
        if c + 1 < n_chunks:
            tot["wait for the other wavefronts"] += np.median(a[:, 4 * (c + 1)] - s[:, 3])
    last = 4 * (n_chunks - 1) + 3
    tot["epilogue (bias, ReLU, tile parked in LDS)"] = np.median(a[:, 16] - a[:, last])
    tot["barrier before the store"] = np.median(a[:, 17] - a[:, 16])
    tot["store (issue)"] = np.median(a[:, 18] - a[:, 17])
    whole = np.median(a[:, 18] - a[:, 0])
    print(f"\n== {name}: {us:.0f} us per launch at B = {B} ({flops / us / 1e6:.0f} TFLOP/s); {len(a)} sampled workgroups, "
          f"wavefront 0, median shader-clock cycles per patch ({n_chunks} channel chunk(s))")
    print("| phase | cycles | share |\n|---|---|---|")
    for k, v in tot.items():
        print(f"| {k} | {v:.0f} | {v / whole * 100:.0f} % |")
    print(f"| whole patch | {whole:.0f} | 100 % |")


def timed(run, iters=20):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


# first convolution on the rotating observation
w = mfma.pack_weights(torch.randn(256, 96, 3, 3) / 29.0, 96).cuda()
b = (torch.randn(256) * 0.1).cuda()
x = torch.rand(B, 96, 6, 32, 56, device="cuda").to(torch.bfloat16)
out = torch.empty(B, 48, 48, 256, dtype=torch.bfloat16, device="cuda")
run = lambda: _lib.call("mz_conv3x3_bf16_rot", _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), B, 1, _lib.current_stream())
us = timed(run)
read(1, "first convolution 96x96x96 -> 48x48x256, stride 2, rotating observation (8x8-output patches, matrix work 7.0 k cycles per patch and SIMD)",
     2.0 * B * 2304 * 256 * 864, us)
del x, out

# 48x48x256 -> 24x24x128, stride 2
layer = L()
layer.cin, layer.cout, layer.stride = 256, 128, 2
chunk, _ = mfma.kernel_info(48, 48, 256, 128, 2)
layer.dev = {"w": mfma.pack_weights(torch.randn(128, 256, 3, 3) / 48.0, chunk).cuda(), "bias": torch.randn(128).cuda()}
x = torch.randn(B, 48, 48, 256, device="cuda").to(torch.bfloat16)
out = torch.empty(B, 24, 24, 128, dtype=torch.bfloat16, device="cuda")
run = lambda: mfma.conv(layer, x, out, B, 48, 48, True, None, None, 0, None, None, None)
us = timed(run)
read(256 // chunk, "48x48x256 -> 24x24x128, stride 2 (8x8-output patches, 64-channel chunks, matrix work 9.2 k cycles per patch and SIMD)",
     2.0 * B * 576 * 128 * 2304, us)
