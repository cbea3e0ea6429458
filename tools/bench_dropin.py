#!/usr/bin/env python3
"""Latency of the reference's own per-game call sequence (train.py:230-277) on the drop-in surface:
prepare_observation -> initial_inference -> Node.expand -> add_exploration_noise -> MCTS.run ->
select_action, one game at a time as the reference does it."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from muzero_amd.config import MuZeroConfig  # noqa: E402
from muzero_amd.network import MuZeroNetwork, init_state_dict  # noqa: E402
from muzero_amd.prep import prepare_observation  # noqa: E402
from muzero_amd.search import MCTS, MinMaxStats, Node, select_action  # noqa: E402

sims = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cfg = MuZeroConfig()
cfg.observation_shape, cfg.action_space_size, cfg.num_simulations = (96, 96, 96), 6, sims
net = MuZeroNetwork(cfg, precision="bf16", state_dict=init_state_dict(cfg, seed=0))
rng = np.random.default_rng(0)
obs = rng.integers(0, 256, (96, 96, 96), dtype=np.uint8)


def move():
    x = prepare_observation(torch.from_numpy(obs)).unsqueeze(0)
    out = net.initial_inference(x)
    root = Node(0, None)
    root.expand(list(range(6)), out)
    root.add_exploration_noise(0.3, 0.25)
    MCTS(cfg).run(root, 6, net, MinMaxStats())
    return select_action(cfg, root, 1.0)


for _ in range(3):
    move()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 30
for _ in range(n):
    move()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / n * 1e3:.2f} ms per move (1 game, run() with {sims} simulations in one wave)")
