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


def test_bench_starts_its_own_ranks():
    """`bench.py --gpus 2` from a plain shell (no launcher, WORLD_SIZE unset): the script starts its two ranks
    itself, relays rank 0's single JSON line and reports every rank's weight digest."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry", "--gpus", "2", "--batch", "8"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry"] is True
    assert len(out["weights_sha256_per_rank"]) == 2 and len(set(out["weights_sha256_per_rank"])) == 1
    assert out["elapsed_max_over_ranks_s"] >= 0.02


def test_bench_refuses_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry", "--gpus", "2"], capture_output=True,
                       text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_bench_self_launch_propagates_a_failing_rank():
    """A rank that dies makes the whole command fail instead of leaving the others waiting in a collective."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MZ_BENCH_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry", "--gpus", "2", "--batch", "8"],
                       capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert "rank 1 exited" in r.stderr
