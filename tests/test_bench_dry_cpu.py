"""bench.py's rank plumbing without a GPU: `--dry` under torch.distributed.run with two gloo ranks does the
rendezvous, the weight broadcast and its checksum, the barriers and the max over ranks, and rank 0 alone
prints the JSON line."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_dry_two_ranks():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--batch", "8", "--dry"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry"] is True and out["value"] is None
    assert out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["games_per_gpu"] == 8 and out["config"]["parallelism"] == "games sharded x2"
    assert out["elapsed_max_over_ranks_s"] >= 0.02        # the slower rank (sleeps 20 ms) sets the time


def test_bench_dry_single_process():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry", "--batch", "4"], capture_output=True,
                       text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["dry"] is True


def test_bench_refuses_world_size_mismatch():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry", "--gpus", "2"], capture_output=True,
                       text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
