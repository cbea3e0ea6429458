#!/usr/bin/env python3
"""A few training updates at the full network size, for rocprofv3 --kernel-trace --stats.
usage: profile_train.py [batch] [fp32|bf16] [native|torch] [eager|graphed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from muzero_amd.config import MuZeroConfig  # noqa: E402
from muzero_amd.learner import Learner  # noqa: E402
from muzero_amd.network import init_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
precision = sys.argv[2] if len(sys.argv) > 2 else "bf16"
native = (sys.argv[3] if len(sys.argv) > 3 else "native") == "native"
cfg = MuZeroConfig()
cfg.observation_shape, cfg.action_space_size = (96, 96, 96), 18
rng = np.random.default_rng(0)
lrn = Learner(cfg, state_dict=init_state_dict(cfg, seed=0), precision=precision, native_convs=native)
K, A = cfg.num_unroll_steps, cfg.action_space_size
obs = torch.rand(B, 96, 96, 96, device="cuda")
args = (obs, rng.integers(0, A, (B, K)), rng.normal(0, 5, (B, K + 1)), rng.normal(0, 1, (B, K + 1)),
        rng.dirichlet(np.ones(A), (B, K + 1)), np.ones(B))
step = lrn.update_graphed if len(sys.argv) > 4 and sys.argv[4] == "graphed" else lrn.update_from_tensors
for _ in range(6):
    step(*args)
torch.cuda.synchronize()
if os.environ.get("MZ_TIME_UPDATES"):      # same-box comparisons: milliseconds per update after the warm-up above
    import time
    t0 = time.perf_counter()
    for _ in range(20):
        step(*args)
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per update")
