#!/usr/bin/env python3
"""Self-play with environments that cost host time (an emulator stand-in that sleeps, i.e. releases
the GIL as ALE does): all games in one group (search, then step, serially) against two groups that
take turns on the GPU (loop.play_pipelined).   usage: bench_pipelined.py [games] [env_step_us] [moves]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from muzero_amd import loop  # noqa: E402
from muzero_amd.config import MuZeroConfig  # noqa: E402
from muzero_amd.envpool import HostEnvPool  # noqa: E402
from muzero_amd.network import MuZeroNetwork, init_state_dict  # noqa: E402
from muzero_amd.selfplay import SelfPlayWorker  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
env_us = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
moves = int(sys.argv[3]) if len(sys.argv) > 3 else 12
SIMS, A = 50, 18


class SleepyEnv:
    def __init__(self, idx):
        self.rng = np.random.default_rng(idx)
        self.frame = self.rng.integers(0, 256, (96, 96, 3), dtype=np.uint8)
        self.t = 0

    def reset(self):
        self.t = 0
        return self.frame, {}

    def step(self, a):
        time.sleep(env_us * 1e-6)
        self.t += 1
        self.frame[self.t % 96] += 1
        return self.frame, 0.0, self.t >= 400, False, {}


cfg = MuZeroConfig()
cfg.action_space_size, cfg.num_simulations = A, SIMS
net = MuZeroNetwork(cfg, precision="bf16", state_dict=init_state_dict(cfg, seed=0, randomize_bn=True))


def run(groups):
    b = B // groups
    workers = [SelfPlayWorker(cfg, net, b, SIMS, seed=g) for g in range(groups)]
    pools = [HostEnvPool(SleepyEnv, b, workers=16 // groups, auto_reset=True) for _ in range(groups)]
    if groups == 1:
        play = lambda n: loop.play_continuous(workers[0], pools[0], n, lambda g: None)
    else:
        play = lambda n: loop.play_pipelined(workers, pools, n, lambda g: None)
    play(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    play(moves)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for p in pools:
        p.close()
    return B * SIMS * moves / dt, dt / moves * 1e3


one = run(1)
two = run(2)
print(f"{B} games, {env_us:.0f} us per environment step on 16 host threads, {moves} moves")
print(f"one group  (search then step): {one[0]:10.0f} sims/s  {one[1]:7.2f} ms per move of all games")
print(f"two groups (taking turns)    : {two[0]:10.0f} sims/s  {two[1]:7.2f} ms per move of all games")
