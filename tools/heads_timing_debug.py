#!/usr/bin/env python3
"""Per-launch HIP-event timings of the heads and tree launches of two moves (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from muzero_amd.config import MuZeroConfig
from muzero_amd.network import init_state_dict
import bench
cfg = MuZeroConfig(); cfg.action_space_size = 18; cfg.num_simulations = 50
sd = init_state_dict(cfg, seed=0, randomize_bn=True)
wl = bench.Workload(cfg, sd, torch.device("cuda", 0), 1024, 50, 1, "bf16", 0)
wl.timed(3, 2)
net = wl.worker.net
net.dynamics_events, net.heads_events, wl.worker.mcts.tree_events = [], [], []
for _ in range(2):
    wl.worker.step(wl.env.step())
torch.cuda.synchronize()
for name, ev in (("dyn", net.dynamics_events), ("heads", net.heads_events), ("tree", wl.worker.mcts.tree_events)):
    t = np.array([a.elapsed_time(b) * 1e3 for a, b in ev])
    print(name, len(t), "mean %.1f median %.1f min %.1f max %.1f" % (t.mean(), np.median(t), t.min(), t.max()), np.round(t[:12], 1))
