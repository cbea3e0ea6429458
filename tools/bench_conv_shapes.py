#!/usr/bin/env python3
"""The representation network's five convolution shapes other than the 12x12 trunk (model.py:165, :182), one
training batch each: torch's bf16 NHWC convolution (MIOpen: forward, data gradient, weight gradient, solver
search on) next to this library's kernels for the same work.  Microseconds per call, HIP events.

    python tools/bench_conv_shapes.py --batch 128
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from muzero_amd import _lib, mfma  # noqa: E402

SHAPES = [("conv1", 96, 96, 256, 2), ("down0", 48, 256, 128, 2), ("res24", 24, 128, 128, 1), ("down1", 24, 128, 256, 2),
          ("trunk", 12, 256, 256, 1)]


def timed(fn, iters=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--wgrad-only", action="store_true", help="time only this library's weight-gradient kernels")
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    B = args.batch
    print(f"batch {B}; microseconds per call (PFLOP/s)")
    print(f"{'layer':8} {'GFLOP':>7} | {'torch fwd':>16} {'torch dgrad':>16} {'torch wgrad':>16} | {'native fwd':>16} "
          f"{'native dgrad':>16} {'native wgrad':>16}")
    for name, H, cin, cout, s in SHAPES:
        Ho = H // s
        gflop = 2.0 * B * Ho * Ho * cout * cin * 9 / 1e9
        x = torch.randn(B, cin, H, H, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, 3, 3, device="cuda") / (3 * cin ** 0.5))
        wb = w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        g = torch.randn(B, cout, Ho, Ho, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        conv = torch.ops.aten.convolution_backward
        bw = lambda mask: conv(g, x, wb, None, [s, s], [1, 1], [1, 1], False, [0, 0], 1, mask)
        xh = x.permute(0, 2, 3, 1)
        gh = g.permute(0, 2, 3, 1)
        if args.wgrad_only:
            need = _lib.lib().mz_conv_wgrad_workspace_bytes_for(B, H, H, cin, cout)
            ws = torch.empty(need // 4, device="cuda")
            dw = torch.empty(cout, cin, 3, 3, device="cuda")
            db = torch.empty(cout, device="cuda")
            n_w = timed(lambda: _lib.call("mz_conv_wgrad_bf16", _lib.ptr(gh), _lib.ptr(xh), B, H, H, cin, cout, _lib.ptr(dw),
                                          _lib.ptr(db), _lib.ptr(ws), ws.numel() * 4, _lib.current_stream()))
            print(f"{name:8} {gflop:7.1f} GFLOP  native wgrad + reduce {n_w:8.1f} us  ({gflop / n_w:6.3f} PFLOP/s)")
            continue
        t_f = timed(lambda: torch.nn.functional.conv2d(x, wb, None, s, 1))
        t_d = timed(lambda: bw([True, False, False]))
        t_w = timed(lambda: bw([False, True, False]))
        chunk, _ = mfma.kernel_info(H, H, cin, cout, s)
        wpk = mfma.pack_weights(w.cpu(), chunk).cuda()
        bias = torch.zeros(cout, device="cuda")
        out = torch.empty(B, Ho, Ho, cout, device="cuda", dtype=torch.bfloat16)
        n_f = timed(lambda: _lib.call("mz_conv3x3_bf16", _lib.ptr(xh), None, 0, _lib.ptr(wpk), _lib.ptr(bias), None,
                                      _lib.ptr(out), None, None, None, B, H, H, cin, cout, s, 0, _lib.current_stream()))
        n_d = n_w = float("nan")
        try:
            dx = torch.empty(B, H, H, cin, device="cuda", dtype=torch.bfloat16)
            if s == 1:
                wt = mfma.pack_weights(w.permute(1, 0, 2, 3).flip(2, 3).contiguous().cpu(), chunk).cuda()
                zb = torch.zeros(cin, device="cuda")
                n_d = timed(lambda: _lib.call("mz_conv3x3_bf16", _lib.ptr(gh), None, 0, _lib.ptr(wt), _lib.ptr(zb), None,
                                              _lib.ptr(dx), None, None, None, B, H, H, cout, cin, 1, 0,
                                              _lib.current_stream()))
            elif hasattr(_lib.lib(), "mz_conv_dgrad_s2_bf16"):
                from muzero_amd.learner import pack_dgrad_s2
                wt = pack_dgrad_s2(w)
                n_d = timed(lambda: _lib.call("mz_conv_dgrad_s2_bf16", _lib.ptr(gh), _lib.ptr(wt), _lib.ptr(dx), B, H, H,
                                              cin, cout, _lib.current_stream()))
        except Exception as e:          # a shape without a kernel
            print("   (no native data gradient:", str(e)[:80], ")")
        try:
            need = _lib.lib().mz_conv_wgrad_workspace_bytes_for(B, H, H, cin, cout)
            ws = torch.empty(need // 4, device="cuda")
            dw = torch.empty(cout, cin, 3, 3, device="cuda")
            db = torch.empty(cout, device="cuda")
            n_w = timed(lambda: _lib.call("mz_conv_wgrad_bf16", _lib.ptr(gh), _lib.ptr(xh), B, H, H, cin, cout, _lib.ptr(dw),
                                          _lib.ptr(db), _lib.ptr(ws), ws.numel() * 4, _lib.current_stream()))
        except Exception as e:
            print("   (no native weight gradient:", str(e)[:80], ")")
        cell = lambda t: f"{t:8.1f} ({gflop / t:5.3f})" if t == t else f"{'-':>16}"
        print(f"{name:8} {gflop:7.1f} | {cell(t_f)} {cell(t_d)} {cell(t_w)} | {cell(n_f)} {cell(n_d)} {cell(n_w)}")


if __name__ == "__main__":
    main()
