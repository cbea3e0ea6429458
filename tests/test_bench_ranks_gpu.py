"""bench.py's multi-rank path on real kernels: two ranks started by bench.py itself from a plain command line,
both on the one GPU of the test box (MZ_FORCE_DEVICE=0), rendezvous and collectives on gloo (RCCL wants one
GPU per rank).  Everything but the collective backend is the path an 8-GPU run takes: per-rank environment,
weight broadcast from rank 0 and its digest on every rank, game sharding, barriers around the timed region,
the maximum over ranks, rank 0 printing the one JSON line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MZ_FORCE_DEVICE="0", MZ_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "320", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak"
    assert out["config"]["games_per_gpu"] == 320 and out["config"]["parallelism"] == "games sharded x2"
    assert out["visit_sum_check"] is True
    assert len(out["weights_sha256_per_rank"]) == 2 and len(set(out["weights_sha256_per_rank"])) == 1
    assert out["collective_backend"] == "gloo"
    # whole-job value: both ranks' games over the slower rank's time
    assert abs(out["value"] - 2 * 320 * 50 * 2 / (out["ms_per_step"] * 2e-3)) <= 1e-6 * out["value"]
    assert out["roofline"]["frac"] > 0 and "sustained" not in out      # the 1-GPU extras are not run with ranks


_ONE_RANK_RCCL = r"""
import os, sys
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
flat = torch.arange(1 << 20, dtype=torch.float32, device="cuda")
want = flat.clone()
dist.broadcast(flat, src=0)                        # the weight hand-over (muzero_amd.dist._flat_broadcast)
t = torch.tensor([3.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)           # max_over_ranks
g = torch.ones(1 << 16, device="cuda")
dist.all_reduce(g, op=dist.ReduceOp.SUM)           # average_gradients
dist.barrier()
box = [None]
dist.all_gather_object(box, "digest")              # gather_checksums
torch.cuda.synchronize()
assert torch.equal(flat, want) and float(t) == 3.25 and float(g.sum()) == 1 << 16 and box == ["digest"]
print("backend", dist.get_backend())
dist.destroy_process_group()
"""


def test_rccl_backend_of_torch_distributed_runs_the_collectives_used():
    """torch.distributed's "nccl" backend (RCCL on this image) with the one rank a one-GPU box allows: the
    collectives bench.py and muzero_amd.dist issue on device tensors - broadcast, MAX / SUM all-reduce, barrier,
    object all-gather - initialise and complete.  (Two or more ranks need as many GPUs; their plumbing is the
    gloo rehearsal above.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", _ONE_RANK_RCCL, "29541"], capture_output=True, text=True, timeout=300,
                       cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "backend nccl" in r.stdout
