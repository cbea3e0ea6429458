#!/usr/bin/env python3
"""The shader clock each launch of the two-image chain kernel ran at, in place in the search and in isolation:
a diagnostic build of the library records, per launch, the shader-clock cycles and the constant-rate 100 MHz
ticks between the first and the last instruction of workgroup 0 (their ratio x 100 MHz = the clock).

    make -C muzero_amd/csrc stamps
    MUZERO_AMD_LIB=$PWD/muzero_amd/csrc/libmuzero_amd_stamps.so python3 tools/launch_clocks.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from muzero_amd import _lib  # noqa: E402
from muzero_amd.config import MuZeroConfig  # noqa: E402
from muzero_amd.network import MuZeroNetwork, init_state_dict  # noqa: E402
from muzero_amd.selfplay import SelfPlayWorker, SyntheticEnv  # noqa: E402

lib = _lib.lib()
lib.mz_debug_read_launches.restype = C.c_int
buf = (C.c_longlong * (3 * 4096))()


def records(since):
    n = lib.mz_debug_read_launches(buf, 4096)
    assert 0 <= n - since <= 4096
    a = np.array(buf[:]).reshape(4096, 3)
    return n, np.stack([a[i % 4096] for i in range(since, n)]) if n > since else np.zeros((0, 3))


def report(name, r):
    for layers in sorted(set(r[:, 0].tolist())):
        x = r[r[:, 0] == layers]
        mhz = x[:, 1] / x[:, 2] * 100.0
        us = x[:, 2] / 100.0
        print(f"{name}: {int(layers):2d}-layer launches: {len(x):4d}, {us.mean():8.1f} us ({us.mean() / layers:6.1f} us/layer), "
              f"{x[:, 1].mean() / layers / 1e3:7.1f} k cycles/layer, clock {mhz.mean():6.0f} MHz (min {mhz.min():.0f}, max {mhz.max():.0f})")


B, A, sims = 1024, 18, 50
cfg = MuZeroConfig()
cfg.action_space_size, cfg.num_simulations = A, sims
net = MuZeroNetwork(cfg, precision="bf16", state_dict=init_state_dict(cfg, seed=0, randomize_bn=True))
worker = SelfPlayWorker(cfg, net, B, sims)
env = SyntheticEnv(B, device="cuda")
worker.reset(env.reset())
for _ in range(31):
    worker.ring.push(env.step())
worker.draw_randomness()
for _ in range(3):
    worker.step(env.step())
torch.cuda.synchronize()
n, _ = records(0)
n0 = n
for _ in range(10):
    worker.step(env.step())
torch.cuda.synchronize()
n, r = records(n0)
report("in the search (10 moves)", r)


def spans(name):
    """Start and end of every workgroup of the most recent 5- and 18-layer launch, relative to the first start."""
    sb = (C.c_longlong * (2 * 512 * 2))()
    assert lib.mz_debug_read_wg_spans(sb) == 0
    a = np.array(sb[:]).reshape(2, 512, 2)[:, :256] / 100.0       # us
    for k, layers in enumerate((5, 18)):
        s, e = a[k, :, 0] - a[k, :, 0].min(), a[k, :, 1] - a[k, :, 0].min()
        d = e - s
        print(f"{name}: last {layers:2d}-layer launch, 256 workgroups: start {s.min():.1f}..{s.max():.1f} us (median {np.median(s):.1f}), "
              f"end {e.min():.1f}..{e.max():.1f} (median {np.median(e):.1f}), duration {d.min():.1f}..{d.max():.1f} (median {np.median(d):.1f})")
        late = np.argsort(-e)[:5]
        print("   the five last to finish: " + ", ".join(f"wg {i}: {s[i]:.0f}->{e[i]:.0f}" for i in late))


spans("in the search")

# the same kernel alone, back to back: random data, then all-zero data (same instruction stream, no operand toggling)
from muzero_amd import mfma  # noqa: E402
for data in ("random", "zeros"):
    for layers in (5, 18):
        keep = []
        arr = (_lib.ChainLayer * layers)()
        for l in range(layers):
            wt = torch.randn(256, 256, 3, 3) / 48.0 if data == "random" else torch.zeros(256, 256, 3, 3)
            w = mfma.pack_weights(wt, 256).cuda()
            b = (torch.randn(256) * 0.1).cuda() if data == "random" else torch.zeros(256, device="cuda")
            keep += [w, b]
            arr[l] = _lib.ChainLayer(w.data_ptr(), b.data_ptr(), 1, l % 2)
        x = (torch.randn(B, 12, 12, 256, device="cuda") if data == "random" else torch.zeros(B, 12, 12, 256, device="cuda")).to(torch.bfloat16)
        out = torch.empty_like(x)
        mean = torch.empty(B, 256, device="cuda")
        for _ in range(60):
            _lib.call("mz_conv_chain_bf16", _lib.ptr(x), None, 0, arr, layers, _lib.ptr(out), None, None, _lib.ptr(mean), B,
                      _lib.current_stream())
        torch.cuda.synchronize()
        n0 = n
        n, r = records(n0)
        report(f"alone, back to back, {data} data", r[10:])
