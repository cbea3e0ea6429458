#!/usr/bin/env python3
"""Per-parameter gradient agreement of the learner's precisions on one batch (diagnostic):
native bf16 and torch bf16 convolutions, each against the float32 learner."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from muzero_amd.config import MuZeroConfig
from muzero_amd.learner import Learner
from muzero_amd.network import init_state_dict
cfg = MuZeroConfig(); cfg.observation_shape = (96, 96, 96); cfg.action_space_size = 6
sd = init_state_dict(cfg, seed=0)
rng = np.random.default_rng(0)
B, K, A = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 5, 6
batch = (torch.from_numpy(rng.random((B, 96, 96, 96), dtype=np.float32)), rng.integers(0, A, (B, K)),
         rng.normal(0, 3, (B, K + 1)), rng.normal(0, 1, (B, K + 1)), rng.dirichlet(np.ones(A), (B, K + 1)), np.ones(B))
g = {}
for name, kw in (("fp32", dict(precision="fp32")), ("torch16", dict(precision="bf16", native_convs=False)),
                 ("native16", dict(precision="bf16", native_convs=True))):
    lrn = Learner(cfg, state_dict=sd, **kw)
    v, r, p, per = lrn.losses_from_tensors(*batch)
    per.mean().backward()
    g[name] = {k: q.grad.detach().float().cpu() for k, q in lrn.network.named_parameters()}
    print(name, v.item(), r.item(), p.item())
for k in g["fp32"]:
    if not k.endswith("weight") or "bn" in k:
        continue
    n = float(g["fp32"][k].norm())
    print(f"{k:50s} |g|={n:9.3e}  torch16-fp32 {float((g['torch16'][k]-g['fp32'][k]).norm())/n:6.3f}  "
          f"native16-fp32 {float((g['native16'][k]-g['fp32'][k]).norm())/n:6.3f}  native16-torch16 {float((g['native16'][k]-g['torch16'][k]).norm())/n:6.3f}")
