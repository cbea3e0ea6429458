#!/usr/bin/env python3
"""Time one bf16 MFMA convolution shape in isolation (HIP events on the launch stream)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from muzero_amd import mfma

SHAPES = {"stem": (96, 96, 256, 2), "down0": (48, 256, 128, 2), "mid": (24, 128, 128, 1),
          "down1": (24, 128, 256, 2), "trunk": (12, 256, 256, 1)}


class L:
    pass


def bench_chain(n_layers, B, iters):
    """The fused LDS-resident chain of 12x12x256->256 convolutions (mz_conv_chain_bf16)."""
    import os
    from muzero_amd import _lib
    keep = []
    arr = (_lib.ChainLayer * n_layers)()
    for l in range(n_layers):
        w = mfma.pack_weights(torch.randn(256, 256, 3, 3) / 48.0, 256).cuda()
        b = (torch.randn(256) * 0.1).cuda()
        keep += [w, b]
        arr[l] = _lib.ChainLayer(w.data_ptr(), b.data_ptr(), 1, 1 if (l % 2 == 0 and l > 0) else 0)
    x = torch.randn(B, 12, 12, 256, device="cuda").to(torch.bfloat16)
    out = torch.empty_like(x)
    mean = torch.empty(B, 256, device="cuda")
    run = lambda: _lib.call("mz_conv_chain_bf16", _lib.ptr(x), None, 0, arr, n_layers, _lib.ptr(out), None, None,
                            _lib.ptr(mean), B, _lib.current_stream())
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * B * 144 * 256 * 2304 * n_layers
    print(f"chain{n_layers} variant={os.environ.get('MZ_CHAIN_VARIANT', '0')} B={B}: {ms * 1e3:.1f} us/launch "
          f"({ms * 1e3 / n_layers:.1f} us/layer), {flops / ms / 1e9:.1f} TFLOP/s")


def bench_stem_rot(B, iters):
    """The first convolution on the rotating, tiled observation (mz_conv3x3_bf16_rot)."""
    from muzero_amd import _lib
    w = mfma.pack_weights(torch.randn(256, 96, 3, 3) / 29.0, 96).cuda()
    b = (torch.randn(256) * 0.1).cuda()
    x = torch.rand(B, 96, 6, 32, 56, device="cuda").to(torch.bfloat16)
    out = torch.empty(B, 48, 48, 256, dtype=torch.bfloat16, device="cuda")
    run = lambda: _lib.call("mz_conv3x3_bf16_rot", _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), B, 1,
                            _lib.current_stream())
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * B * 2304 * 256 * 864
    print(f"first convolution, rotating observation, B={B}: {ms * 1e3:.1f} us, {flops / ms / 1e9:.1f} TFLOP/s, "
          f"{(x.numel() + out.numel()) * 2 / ms / 1e9:.2f} TB/s of input + output")


def bench_mid_chain(B, iters):
    """The two residual blocks at 24x24x128 as one kernel (mz_conv_chain_mid_bf16) and as four."""
    from muzero_amd import _lib
    layers = []
    for _ in range(4):
        l = L()
        l.cin, l.cout, l.stride = 128, 128, 1
        l.dev = {"w": mfma.pack_weights(torch.randn(128, 128, 3, 3) / 34.0, 128).cuda(), "bias": (torch.randn(128) * 0.1).cuda()}
        layers.append(l)
    arr = (_lib.ChainLayer * 4)()
    for i, l in enumerate(layers):
        arr[i] = _lib.ChainLayer(l.dev["w"].data_ptr(), l.dev["bias"].data_ptr(), 1, i % 2)
    x = torch.randn(B, 24, 24, 128, device="cuda").to(torch.bfloat16)
    t, y, out = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)

    def fused():
        _lib.call("mz_conv_chain_mid_bf16", _lib.ptr(x), arr, 4, _lib.ptr(out), B, _lib.current_stream())

    def layered():
        mfma.conv(layers[0], x, t, B, 24, 24, True, None, None, 0, None, None, None)
        mfma.conv(layers[1], t, y, B, 24, 24, True, x, None, 0, None, None, None)
        mfma.conv(layers[2], y, t, B, 24, 24, True, None, None, 0, None, None, None)
        mfma.conv(layers[3], t, out, B, 24, 24, True, y, None, 0, None, None, None)

    flops = 2.0 * B * 576 * 128 * 1152 * 4
    for name, run in (("one kernel", fused), ("four kernels", layered), ("one kernel", fused), ("four kernels", layered)):
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print(f"24x24x128 x4 layers, {name}, B={B}: {ms * 1e3:.1f} us, {flops / ms / 1e9:.1f} TFLOP/s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="trunk")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    if args.shape == "stemrot":
        return bench_stem_rot(args.batch, args.iters)
    if args.shape == "midchain":
        return bench_mid_chain(args.batch, args.iters)
    if args.shape.startswith("chain"):
        return bench_chain(int(args.shape[5:] or 5), args.batch, args.iters)
    H, cin, cout, stride = SHAPES[args.shape]
    B = args.batch
    layer = L()
    layer.cin, layer.cout, layer.stride = cin, cout, stride
    chunk, _ = mfma.kernel_info(H, H, cin, cout, stride)
    w = torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5)
    layer.dev = {"w": mfma.pack_weights(w, chunk).cuda(), "bias": torch.randn(cout).cuda()}
    HO = (H - 1) // stride + 1
    x = torch.randn(B, H, H, cin, device="cuda").to(torch.bfloat16)
    res = torch.randn(B, HO, HO, cout, device="cuda").to(torch.bfloat16)
    out = torch.empty(B, HO, HO, cout, dtype=torch.bfloat16, device="cuda")
    for _ in range(5):
        mfma.conv(layer, x, out, B, H, H, True, res, None, 0, None, None, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        mfma.conv(layer, x, out, B, H, H, True, res, None, 0, None, None, None)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    flops = 2.0 * B * HO * HO * cout * cin * 9
    print(f"{args.shape} B={B}: {ms * 1e3:.1f} us/launch, {flops / ms / 1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
