#!/usr/bin/env python3
"""How far the throughput precision moves the search result: the same observations, noise and
weights searched with the fp32 validation path, with the bf16 path (stacked observation), with the bf16
path on the rotating observation (what self-play runs: the first convolution sums in another order), and
with IEEE-half operands (f16) and with the float32-accurate bf16x3 path; for each pair: agreement of the most-visited action, fraction of games
with identical visit counts, mean total-variation distance between visit distributions, root-value difference.
usage: bf16_vs_fp32_search.py [B] [A] [sims] [seed]"""
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
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 0
cfg = MuZeroConfig()
cfg.action_space_size, cfg.num_simulations = A, sims
sd = init_state_dict(cfg, seed=0, randomize_bn=True)
rng = np.random.default_rng(seed)
ring = FrameRing(B)
ring.enable_rotating(torch.bfloat16)
ring.reset(torch.from_numpy(rng.integers(0, 256, (B, 96, 96, 3), dtype=np.uint8)))
for _ in range(31):
    ring.push(torch.from_numpy(rng.integers(0, 256, (B, 96, 96, 3), dtype=np.uint8)))
noise = torch.from_numpy(rng.dirichlet(np.full(A, 0.3), size=B)).cuda()
res = {}
for name, precision, rotating in (("fp32", "fp32", False), ("bf16", "bf16", False), ("bf16_rotating", "bf16", True),
                                  ("f16", "f16", False), ("bf16x3", "bf16x3", False), ("f16x2", "f16x2", False)):
    net = MuZeroNetwork(cfg, precision=precision, state_dict=sd)
    m = BatchedMCTS(cfg, net, B, sims)
    if rotating:
        obs, newest = ring.rotating_observation()
        r = m.run(obs, noise, rotation=newest)
    else:
        r = m.run(ring.observation(net.obs_dtype), noise)
    res[name] = (r.visit_counts.cpu().numpy().copy(), r.root_values.cpu().numpy().copy())
    del m, net
    torch.cuda.empty_cache()


def compare(a, b):
    (va, qa), (vb, qb) = res[a], res[b]
    tv = 0.5 * np.abs(va - vb).sum(1) / sims
    return {"pair": f"{a} vs {b}", "same_most_visited_action": float((va.argmax(1) == vb.argmax(1)).mean()),
            "identical_visit_counts": float((va == vb).all(1).mean()),
            "mean_total_variation": float(tv.mean()), "max_total_variation": float(tv.max()),
            "root_value_rel_diff": float(np.abs(qa - qb).mean() / np.abs(qa).mean())}


print(json.dumps({"games": B, "actions": A, "sims": sims, "seed": seed,
                  "pairs": [compare("fp32", "bf16"), compare("fp32", "bf16_rotating"), compare("bf16", "bf16_rotating"),
                            compare("fp32", "f16"), compare("fp32", "bf16x3"), compare("fp32", "f16x2")]}))
