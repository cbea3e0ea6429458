"""Multi-rank host logic on CPU (gloo, world_size 2): game sharding and the weight
broadcast that is the path's only collective."""
import os
import socket
import subprocess
import sys

from muzero_amd import dist as mzdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["MZ_ROOT"])
import torch
from muzero_amd import dist as mzdist
from muzero_amd.config import MuZeroConfig
from muzero_amd.network import init_state_dict

rank, world, local = mzdist.init_from_env("gloo")
cfg = MuZeroConfig(); cfg.num_channels = 16; cfg.downsample_steps = [(8, 1), (16, 1)]
cfg.fc_reward_layers = cfg.fc_value_layers = cfg.fc_policy_layers = [16]; cfg.support_size = 5
cfg.observation_shape = (6, 16, 16); cfg.action_space_size = 4
mine = init_state_dict(cfg, seed=100 + rank, randomize_bn=True)   # different on every rank
assert not mzdist.verify_same_everywhere(mine)
got = mzdist.broadcast_state_dict(mine, src=0)
want = init_state_dict(cfg, seed=100, randomize_bn=True)
assert list(got) == list(want)
for k in want:
    assert got[k].dtype == want[k].dtype and torch.equal(got[k], want[k]), k
assert mzdist.verify_same_everywhere(got)
lo, hi = mzdist.shard_range(4097, rank, world)
sizes = [None] * world
torch.distributed.all_gather_object(sizes, (lo, hi))
assert sizes[0][0] == 0 and sizes[-1][1] == 4097 and all(sizes[i][1] == sizes[i + 1][0] for i in range(world - 1))
torch.distributed.destroy_process_group()
print("rank", rank, "ok")
'''


DP_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["MZ_ROOT"])
import numpy as np, torch
from muzero_amd import dist as mzdist
from muzero_amd.config import MuZeroConfig
from muzero_amd.learner import Learner
from muzero_amd.network import init_state_dict
from oracle.loss import support_ce

rank, world, local = mzdist.init_from_env("gloo")
cfg = MuZeroConfig(); cfg.num_channels = 16; cfg.downsample_steps = [(8, 1), (16, 1)]
cfg.fc_reward_layers = cfg.fc_value_layers = cfg.fc_policy_layers = [16]; cfg.support_size = 5
cfg.observation_shape = (6, 16, 16); cfg.action_space_size = 4; cfg.num_unroll_steps = 2
sd = init_state_dict(cfg, seed=3)


def batch(r):
    rng = np.random.default_rng(50 + r)
    B, K, A = 3, 2, 4
    return (torch.from_numpy(rng.random((B, 6, 16, 16), dtype=np.float32)), rng.integers(0, A, (B, K)),
            rng.normal(0, 2, (B, K + 1)), rng.normal(0, 1, (B, K + 1)), rng.dirichlet(np.ones(A), (B, K + 1)), np.ones(B))


# what the averaged gradient must be: the mean of the two ranks' own gradients
grads = []
for r in range(world):
    solo = Learner(cfg, device="cpu", state_dict=sd, loss_impl=support_ce)
    solo.losses_from_tensors(*batch(r))[3].mean().backward()
    grads.append([p.grad.clone() for p in solo.network.parameters()])
want = [sum(g) / world for g in zip(*grads)]

dp = Learner(cfg, device="cpu", state_dict=sd, loss_impl=support_ce, data_parallel=True)
dp.losses_from_tensors(*batch(rank))[3].mean().backward()
mzdist.average_gradients(dp.network.parameters())
for p, w in zip(dp.network.parameters(), want):
    assert torch.allclose(p.grad, w, rtol=1e-5, atol=1e-8)
# a full update keeps the replicas identical
dp.optimizer.zero_grad()
dp.update_from_tensors(*batch(rank))
assert mzdist.verify_same_everywhere({k: v for k, v in dp.state_dict().items() if "running" not in k and "num_batches" not in k})
torch.distributed.destroy_process_group()
print("rank", rank, "ok")
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_weight_broadcast_and_sharding_world_size_2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MZ_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o


def test_data_parallel_update_averages_gradients_world_size_2(tmp_path):
    script = tmp_path / "dp_worker.py"
    script.write_text(DP_WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MZ_ROOT=ROOT, OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o


def test_shard_range_is_a_partition():
    for total in (1, 7, 256, 4096, 4097):
        for world in (1, 2, 3, 8):
            edges = [mzdist.shard_range(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
