#!/usr/bin/env python3
"""Timing of the training-side rows on one GPU (not the headline benchmark):
the two HIP kernels against their HBM bound, and a whole update_weights step at the
full network size.

    python tools/bench_train.py --batch 128 --steps 5
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from muzero_amd import _lib  # noqa: E402
from muzero_amd.config import MuZeroConfig  # noqa: E402
from muzero_amd.learner import EpisodeStore, GameHistory, Learner  # noqa: E402
from oracle.loss import support_ce as _support_ce_torch  # noqa: E402  (the op sequence the kernel is timed against)
from muzero_amd.network import init_state_dict  # noqa: E402

HBM_PEAK = 8.0e12


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--autotune", action="store_true", help="let MIOpen search its convolution solvers")
    ap.add_argument("--rows", type=int, default=1024 * 6, help="loss rows: batch 1024 x (K+1) unroll steps")
    ap.add_argument("--only", default="", help="comma-separated variants to run (default: all five)")
    args = ap.parse_args()
    dev = torch.device("cuda")
    torch.backends.cudnn.benchmark = args.autotune
    out = {}

    # support cross-entropy, forward + gradient in one launch: reads n*601*4 B, writes the same
    S, n = 300, args.rows
    x = torch.randn(n, 2 * S + 1, device=dev)
    t = torch.randn(n, device=dev) * 30
    loss = torch.empty(n, device=dev)
    dlog = torch.empty_like(x)
    s = timed(lambda: _lib.call("mz_support_ce", _lib.ptr(x), _lib.ptr(t), n, S, _lib.ptr(loss), _lib.ptr(dlog),
                                _lib.current_stream()))
    nbytes = 2 * x.numel() * 4 + 8 * n
    xr = x.clone().requires_grad_(True)

    def torch_version():
        xr.grad = None
        _support_ce_torch(xr, t, S).sum().backward()
    s_torch = timed(torch_version, 20)
    out["support_ce"] = {"rows": n, "us": s * 1e6, "GB/s": nbytes / s / 1e9, "frac_of_hbm_peak": nbytes / s / HBM_PEAK,
                         "torch_op_sequence_us": s_torch * 1e6}

    # n-step targets: 4096 queries over 512 episodes of up to 3000 steps
    rng = np.random.default_rng(0)
    games = []
    for _ in range(512):
        T = int(rng.integers(100, 3000))
        g = GameHistory(18, 0.997)
        g.rewards, g.values = list(rng.normal(0, 1, T)), list(rng.normal(0, 3, T))
        g.policies = list(rng.dirichlet(np.ones(18), T))
        games.append(g)
    queries = [(int(gi), int(rng.integers(0, len(games[gi].rewards)))) for gi in rng.integers(0, 512, 4096)]
    store = EpisodeStore(18, 0.997, capacity=1 << 21)
    for g in games:
        store.add_game(g)
    q = np.asarray(queries)
    store.targets(q[:, 0], q[:, 1], 5, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        store.targets(q[:, 0], q[:, 1], 5, 10)
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for gi, p in queries:
        games[gi].make_target(p, 5, 10)
    out["nstep_targets"] = {"queries": 4096, "device_store_query_ms": t_dev * 1e3,
                            "host_python_ms": (time.perf_counter() - t0) * 1e3}

    # a whole update at the full network size (fp32 torch autograd convolutions + the HIP loss)
    cfg = MuZeroConfig()
    cfg.observation_shape, cfg.action_space_size = (96, 96, 96), 18
    B, K, A = args.batch, cfg.num_unroll_steps, cfg.action_space_size
    obs = torch.rand(B, 96, 96, 96, device=dev)
    acts = rng.integers(0, A, (B, K))
    tv, tr = rng.normal(0, 5, (B, K + 1)), rng.normal(0, 1, (B, K + 1))
    tp = rng.dirichlet(np.ones(A), (B, K + 1))
    w = np.ones(B)
    fwd = 5_181_014_016 + K * 850_364_416      # forward FLOPs per position; backward ~2x forward
    for tag, precision, native, graphed in (("fp32", "fp32", False, False), ("bf16_torch_convs", "bf16", False, False),
                                            ("bf16_native_convs", "bf16", True, False),
                                            ("bf16_torch_convs_graphed", "bf16", False, True),
                                            ("bf16_native_convs_graphed", "bf16", True, True)):
        if args.only and tag not in args.only.split(","):
            continue
        lrn = Learner(cfg, device=dev, state_dict=init_state_dict(cfg, seed=0), precision=precision,
                      native_convs=native, autotune=args.autotune)
        if graphed:
            lrn.update_from_tensors = lrn.update_graphed
        if args.channels_last:
            lrn.network.to(memory_format=torch.channels_last)
        for _ in range(4):
            lrn.update_from_tensors(obs, acts, tv, tr, tp, w)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            m = lrn.update_from_tensors(obs, acts, tv, tr, tp, w)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        out["update_weights_" + tag] = {"batch": B, "unroll": K, "ms_per_step": dt * 1e3,
                                              "positions_per_s": B / dt, "model_TFLOP/s": 3 * fwd * B / dt / 1e12,
                                              "loss": m["total_loss"]}
        print(f"{tag}: {dt * 1e3:.2f} ms per update", file=sys.stderr, flush=True)     # (a long run stays audible)
        del lrn
    print(json.dumps(out))


if __name__ == "__main__":
    main()
