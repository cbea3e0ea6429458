#!/usr/bin/env python3
"""How far the throughput precision moves the search result: the same observations, noise and
weights searched with the fp32 validation path and with the bf16 path; reports agreement of the
most-visited action, the mean total-variation distance between visit distributions and the
root-value difference.   usage: bf16_vs_fp32_search.py [B] [A] [sims]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from muzero_amd.config import MuZeroConfig  # noqa: E402
from muzero_amd.network import MuZeroNetwork, init_state_dict  # noqa: E402
from muzero_amd.prep import FrameRing  # noqa: E402
from muzero_amd.search import BatchedMCTS  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
A = int(sys.argv[2]) if len(sys.argv) > 2 else 18
sims = int(sys.argv[3]) if len(sys.argv) > 3 else 50
cfg = MuZeroConfig()
cfg.action_space_size, cfg.num_simulations = A, sims
sd = init_state_dict(cfg, seed=0, randomize_bn=True)
rng = np.random.default_rng(0)
ring = FrameRing(B)
ring.reset(torch.from_numpy(rng.integers(0, 256, (B, 96, 96, 3), dtype=np.uint8)))
for _ in range(31):
    ring.push(torch.from_numpy(rng.integers(0, 256, (B, 96, 96, 3), dtype=np.uint8)))
noise = torch.from_numpy(rng.dirichlet(np.full(A, 0.3), size=B)).cuda()
res = {}
for precision in ("fp32", "bf16"):
    net = MuZeroNetwork(cfg, precision=precision, state_dict=sd)
    m = BatchedMCTS(cfg, net, B, sims)
    r = m.run(ring.observation(net.obs_dtype), noise)
    res[precision] = (r.visit_counts.cpu().numpy().copy(), r.root_values.cpu().numpy().copy())
    del m, net
v32, q32 = res["fp32"]
v16, q16 = res["bf16"]
tv = 0.5 * np.abs(v32 - v16).sum(1) / sims
print(json.dumps({"games": B, "actions": A, "sims": sims,
                  "same_most_visited_action": float((v32.argmax(1) == v16.argmax(1)).mean()),
                  "identical_visit_counts": float((v32 == v16).all(1).mean()),
                  "mean_total_variation": float(tv.mean()), "max_total_variation": float(tv.max()),
                  "root_value_abs_diff_mean": float(np.abs(q32 - q16).mean()),
                  "root_value_abs_mean": float(np.abs(q32).mean())}))
