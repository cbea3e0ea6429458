#!/usr/bin/env python3
"""Same-box A/B of search variants: alternates the variants move by move so that clock and
thermal drift hit both equally.  usage: ab_bench.py [B] [A]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from muzero_amd.config import MuZeroConfig  # noqa: E402
from muzero_amd.network import MuZeroNetwork, init_state_dict  # noqa: E402
from muzero_amd.selfplay import SelfPlayWorker, SyntheticEnv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
A = int(sys.argv[2]) if len(sys.argv) > 2 else 18
cfg = MuZeroConfig()
cfg.action_space_size, cfg.num_simulations = A, 50
dev = torch.device("cuda")
net = MuZeroNetwork(cfg, precision="bf16", device=dev, state_dict=init_state_dict(cfg, seed=0, randomize_bn=True))
w = SelfPlayWorker(cfg, net, B, 50, 1)
env = SyntheticEnv(B, device=dev)
w.reset(env.reset())
for _ in range(31):
    w.ring.push(env.step())
w.draw_randomness()

_rot = w.ring.rot


def _rotating(on):
    w.rotating = on
    w.ring.rot = _rot if on else None   # off: the frame stack is rebuilt from the u8 ring every move


VARIANTS = {"rotating observation": lambda: _rotating(True), "rebuilt stack": lambda: _rotating(False)}
total = {k: 0.0 for k in VARIANTS}
for k, setup in VARIANTS.items():
    setup()
    w.step(env.step())
torch.cuda.synchronize()
ROUNDS = 8
for _ in range(ROUNDS):
    for k, setup in VARIANTS.items():
        setup()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        w.step(env.step())
        w.step(env.step())
        torch.cuda.synchronize()
        total[k] += time.perf_counter() - t0
for k, t in total.items():
    print(f"{k:20s} {t / ROUNDS / 2 * 1e3:8.3f} ms/move  {B * 50 * 2 * ROUNDS / t:10.0f} sims/s")
