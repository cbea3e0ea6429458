#!/usr/bin/env python3
"""Headline benchmark: MCTS simulations per second of the batched self-play search.

One "step" = one move for every game on every GPU: push one new 96x96x3 frame per
game into the device frame ring and the stacked observation, run the search
(initial inference + `sims` simulations at wave 1, each a distinct recurrent
inference) and choose an action per game.  Synthetic frames and weights.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N --steps K --warmup W          (starts its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rehearsal of the N-rank path on ONE GPU (every rank on device 0, rendezvous and collectives on gloo because
RCCL wants one GPU per rank):  MZ_FORCE_DEVICE=0 MZ_DIST_BACKEND=gloo python bench.py --gpus 2 --batch 256

Games are sharded over ranks with no data-path collective (weak scaling: the
per-GPU batch is fixed); RCCL only broadcasts the weights before the timed region.
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {
    # name: (actions, games per GPU, label)
    "synthetic18": (18, 1024, "synthetic"),
    "pong6": (6, 256, "Pong-shaped"),
    "mspacman9": (9, 512, "MsPacman-shaped"),
}
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak of MI355X
# convolution FLOPs (2*MACs) per sample: representation 5 181 014 016, one recurrent inference
# 849 346 560 (+ heads, negligible): 12x12 trunk conv = 2*144*256*2304
TRUNK_FLOPS = 2 * 144 * 256 * 2304


def cpu_baseline(cfg, sd, sims, budget_s=20.0):
    """The reference's search restated on the CPU (oracle: per-node Python tree + torch fp32
    networks, eval-mode BatchNorm), timed on this host: 1 game at a time, wave 1."""
    from oracle import nets as onets
    from oracle import prep as oprep
    from oracle import tree as otree
    cores = os.cpu_count() or 1
    threads = min(cores, 16) if os.environ.get("OMP_NUM_THREADS") is None else int(os.environ["OMP_NUM_THREADS"])
    torch.set_num_threads(threads)
    net = onets.Net(cfg).eval()
    net.load_state_dict({k: v.cpu() for k, v in sd.items()}, strict=True)
    infer = onets.make_infer(net)
    rng = np.random.default_rng(1234)
    done, t_search = 0, 0.0
    t_end = time.time() + budget_s
    games = 0
    while time.time() < t_end or games < 1:
        frames = rng.integers(0, 256, size=(96, 96, 96), dtype=np.uint8)
        t0 = time.time()
        obs = torch.tensor(oprep.prepare(frames)).unsqueeze(0)
        with torch.no_grad():
            h, p, v = net.initial(obs)
        pri = torch.softmax(p, dim=1).numpy()[0]
        noise = rng.dirichlet(np.full(cfg.action_space_size, 0.3))
        otree.search(pri, h, infer, sims, 1, noise=noise)
        t_search += time.time() - t0
        done += sims
        games += 1
    return {"value": done / t_search, "unit": "sims/s", "cores": threads, "kind": "port",
            "sample": f"{games} game-moves x {sims} sims, 1 game at a time (as train.py:257), wave 1, "
                      f"A={cfg.action_space_size}, torch fp32 CPU nets with {threads} threads"}


def measure_dynamics_kernel(worker, one_step, moves=2):
    """Average duration of the dominant kernel - the fused dynamics chain (5 convolutions
    12x12x256->256 with the image resident in LDS) - measured in place: HIP events on the launch
    stream around every dynamics launch of `moves` further moves of the same workload.  The tree
    and heads launches of the same moves are timed the same way (secondary kernels)."""
    net = worker.net
    net.dynamics_events, net.heads_events, worker.mcts.tree_events = [], [], []
    for _ in range(moves):
        one_step()
    torch.cuda.synchronize()
    pairs, net.dynamics_events = net.dynamics_events, None
    heads, net.heads_events = net.heads_events, None
    tree, worker.mcts.tree_events = worker.mcts.tree_events, None
    mean_s = lambda ps: sum(e0.elapsed_time(e1) for e0, e1 in ps) / max(len(ps), 1) * 1e-3
    # the short launches are summarised by their median: an event pair that straddles a moment when the
    # stream ran dry (the host preparing the next move) reads milliseconds for a 17 us kernel
    median_s = lambda ps: float(np.median([e0.elapsed_time(e1) for e0, e1 in ps])) * 1e-3 if ps else 0.0
    n_convs = 1 + 2 * len(net.dyn_blocks)
    # s per launch, convs per launch, launches, (tree s per launch, heads s per launch)
    return mean_s(pairs), n_convs, len(pairs), (median_s(tree), median_s(heads))


def load_profiled_traffic():
    """HBM bytes per launch of the dominant kernel from the last rocprofv3 PMC passes
    (FETCH_SIZE / WRITE_SIZE, corrected as documented in profiles/); None if never profiled."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


class Workload:
    """One self-play configuration on this rank's GPU: network, worker, synthetic frames."""

    def __init__(self, cfg, sd, device, B, sims, wave, precision, rank, graph=False, from_host=False):
        from muzero_amd.network import MuZeroNetwork
        from muzero_amd.selfplay import SelfPlayWorker, SyntheticEnv
        from muzero_amd.tree import select_actions
        self.B, self.sims, self.from_host = B, sims, from_host
        self.net = MuZeroNetwork(cfg, precision=precision, device=device, state_dict=sd)
        self.worker = worker = SelfPlayWorker(cfg, self.net, B, sims, wave, seed=5678 + rank)
        self.env = env = SyntheticEnv(B, seed=1234 + rank, device=device)
        worker.reset(env.reset())
        for _ in range(31):
            worker.ring.push(env.step())
        worker.draw_randomness()
        self.replay = None
        if graph:   # the whole search as one HIP graph; frame push and action choice stay outside it
            if worker.rotating:
                obs, _ = worker.ring.rotating_observation()
                self.replay = worker.mcts.capture(obs, worker.noise, rotation=worker.ring.newest_dev)
            else:
                worker.ring.observation(self.net.obs_dtype, out=worker.obs)
                self.replay = worker.mcts.capture(worker.obs, worker.noise)
        self._select = select_actions

    def one_step(self):
        w = self.worker
        w.draw_randomness()  # fresh Dirichlet rows and uniforms every move (host RNG, async copy)
        frames = self.env.step(on_device=not self.from_host)
        if self.replay is None:
            return w.step(frames)[0]
        w.ring.push(frames)
        if not w.rotating:
            w.ring.observation(self.net.obs_dtype, out=w.obs)
        res = self.replay()
        return self._select(res.visit_counts, 1.0, w.u)

    def timed(self, steps, warmup, barrier=lambda: None, min_seconds=0.0):
        """(seconds, steps done): `steps` moves, or as many as fill `min_seconds` (checked every 10 moves)."""
        for _ in range(warmup):
            self.one_step()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        done = 0
        while True:
            for _ in range(steps if not min_seconds else 10):
                actions = self.one_step()
            done += steps if not min_seconds else 10
            if not min_seconds:
                break
            torch.cuda.synchronize()
            if time.perf_counter() - t0 >= min_seconds:
                break
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        assert actions.shape == (self.B,)
        assert int(self.worker.mcts.visits.sum().item()) == self.B * self.sims
        self.worker.mcts.check_status()
        return elapsed, done

    def free(self):
        self.replay = self.worker = self.net = self.env = None
        torch.cuda.empty_cache()


def chain_roofline(wl, precision, traffic=None):
    """Roofline entry of the dominant kernel (the fused dynamics chain) of a workload, timed live."""
    dur, n_convs, n_launches, (tree_s, heads_s) = measure_dynamics_kernel(
        wl.worker, lambda: wl.worker.step(wl.env.step())[0])
    flops = TRUNK_FLOPS * n_convs * wl.B
    achieved = flops / dur / 1e12
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    if precision in ("bf16x3", "f16x2"):
        # three bf16 (two half) MFMAs per product: the matrix pipe executes 3x (2x) the algorithmic FLOPs, so the
        # bound on ALGORITHMIC FLOP/s is a third (half) of the dense 16-bit peak
        per = 3 if precision == "bf16x3" else 2
        kernel, peak, executed = "conv_chain_x3_kernel", PEAK_BF16_TFLOPS / per, per * achieved
    else:
        kernel, peak, executed = ("conv_chain_pair_kernel" if wl.B > n_cu else "conv_chain_kernel"), PEAK_BF16_TFLOPS, achieved
    out = {"bound": "mfma", "kernel": f"{kernel}: dynamics network, {n_convs} fused 3x3 convs 12x12x256->256",
           "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
           "launch_us": dur * 1e6, "launches_timed": n_launches, "flops_per_launch": flops}
    if precision in ("bf16x3", "f16x2"):
        out["executed_bf16_tflops"] = executed
        out["peak_note"] = ("2500 TFLOP/s dense bf16 / 3 MFMAs per product" if precision == "bf16x3" else
                            "2500 TFLOP/s dense f16 / 2 MFMAs per product")
    return out, tree_s, heads_s


def launch_ranks(n: int) -> int:
    """Start `n` copies of this script as ranks 0..n-1 of one job (one per GPU) and wait for them: what
    `torch.distributed.run --nproc-per-node n` would do, for a caller that just runs `bench.py --gpus n`.
    Called before anything in this process touches a GPU.  Rank 0 inherits stdout, so its JSON line is this
    process's output; the first rank to fail ends the others.  Returns the exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    code = 0
    live = dict(enumerate(procs))
    while live and code == 0:
        for r, p in list(live.items()):
            rc = p.poll()
            if rc is not None:
                del live[r]
                if rc != 0:
                    print(f"bench.py: rank {r} exited with code {rc}", file=sys.stderr)
                    code = rc if rc > 0 else 1
        time.sleep(0.05)
    for p in live.values():            # a rank failed: the others would wait for it in a collective for ever
        p.terminate()
    for p in live.values():
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
    return code


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="synthetic18", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="games per GPU (default: the workload's)")
    ap.add_argument("--sims", type=int, default=50)
    ap.add_argument("--wave", type=int, default=1)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f16", "bf16x3", "f16x2"],
                    help="bf16: throughput precision (the headline); f16: IEEE-half operands at the same matrix rate "
                         "(outputs within 1e-3 of the reference's float32 path on its fixtures); bf16x3 / f16x2: float32-accurate "
                         "(three bf16 / two half MFMAs per product)")
    ap.add_argument("--graph", action="store_true", help="replay the search from a HIP graph")
    ap.add_argument("--from-host", action="store_true", help="include the pinned H2D copy of the new frames")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the additional blocks of a 1-GPU run (secondary precision, sustained, from-host, small batch)")
    ap.add_argument("--sustained-seconds", type=float, default=10.0)
    ap.add_argument("--dry", action="store_true",
                    help="rank plumbing only (no GPU, no kernels): gloo rendezvous, weight broadcast, barriers, "
                         "max over ranks, rank 0 prints the line with value null")
    args = ap.parse_args()

    from muzero_amd import dist as mzdist
    from muzero_amd.config import MuZeroConfig
    from muzero_amd.network import init_state_dict

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    rank, world, local = mzdist.init_from_env("gloo" if args.dry else None)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("MZ_BENCH_FAIL_RANK") == str(rank):     # test hook: this rank dies after the rendezvous
        os._exit(3)
    A, B, label = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    desc = f"{label} 96x96 obs (32 RGB frames), {A} actions, {B} games x {args.sims} sims per GPU"
    cfg = MuZeroConfig()
    cfg.action_space_size = A
    cfg.num_simulations = args.sims
    if args.dry:
        cfg.num_channels, cfg.downsample_steps = 16, [(8, 1), (16, 1)]       # a small parameter set to broadcast
        cfg.fc_reward_layers = cfg.fc_value_layers = cfg.fc_policy_layers = [16]

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # weights: created on rank 0, handed to the other ranks with one broadcast (RCCL on GPUs)
    sd = init_state_dict(cfg, seed=0 if rank == 0 else 1 + rank, randomize_bn=True)
    sd = mzdist.broadcast_state_dict(sd, src=0)
    if not mzdist.verify_same_everywhere(sd):
        raise SystemExit("weight broadcast left ranks with different parameters")
    digests = mzdist.gather_checksums(sd) if world > 1 else None     # every rank's digest, reported by rank 0

    if args.dry:
        barrier()
        t0 = time.perf_counter()
        time.sleep(0.01 * (1 + rank))          # ranks finish at different times: the maximum must win
        barrier()
        elapsed = mzdist.max_over_ranks(time.perf_counter() - t0)
        lo, hi = mzdist.shard_range(world * B, rank, world)
        assert hi - lo == B
        if rank == 0:
            print(json.dumps({"metric": "mcts_sims_per_sec", "value": None, "unit": "sims/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "dry": True,
                              "elapsed_max_over_ranks_s": elapsed, "scaling": "weak",
                              "weights_sha256_per_rank": digests,
                              "config": {"workload": desc, "games_per_gpu": B, "parallelism": f"games sharded x{world}"}}))
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    wl = Workload(cfg, sd, device, B, args.sims, args.wave, args.precision, rank, args.graph, args.from_host)
    elapsed, _ = wl.timed(args.steps, args.warmup, barrier)
    elapsed = mzdist.max_over_ranks(elapsed)

    if rank == 0:
        total_sims = world * B * args.sims * args.steps
        sims_per_s = total_sims / elapsed
        out = {
            "metric": "mcts_sims_per_sec", "value": sims_per_s, "unit": "sims/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "f16": "f16", "f16x2": "f16x2 (float32-accurate: 2 half MFMAs per product)"}.get(
                args.precision, "bf16x3 (float32-accurate: 3 bf16 MFMAs per product)"),
            "data": "synthetic",
            "config": {"workload": desc + f", wave {args.wave}", "games_per_gpu": B, "actions": A,
                       "sims_per_move": args.sims, "wave": args.wave, "parallelism": f"games sharded x{world}",
                       "precision": args.precision, "hip_graph": bool(args.graph),
                       "frames_from_host": bool(args.from_host)},
            "env_steps_per_sec": world * B * args.steps / elapsed,
            # ALE "-v5" environments repeat each action over 4 emulator frames (built-in frameskip)
            "env_frames_per_sec": 4 * world * B * args.steps / elapsed,
            "visit_sum_check": True,
        }
        if digests is not None:
            out["weights_sha256_per_rank"] = digests
            out["collective_backend"] = torch.distributed.get_backend()
        prof = load_profiled_traffic()
        traffic = None
        if prof and prof.get("games_per_gpu") == B and args.precision == "bf16":
            traffic = prof.get("hbm_bytes_per_launch")
        out["roofline"], tree_s, heads_s = chain_roofline(wl, args.precision, traffic)
        out["roofline"]["end_to_end_frac"] = sims_per_s / world * 953.98e6 / (out["roofline"]["peak"] * 1e12)
        # the other launches of a simulation step, timed the same way (HIP events include ~2 us of
        # launch latency).  Tree: algorithmic bytes per simulation = D*(A*20+20) read by the descent
        # + A*20+12 written by the expansion + (D+1)*32 for the backup at mean depth D ~ 4.
        tree_bytes = B * (4 * (A * 20 + 20) + A * 20 + 12 + 5 * 32)
        out["secondary_kernels"] = [
            {"kernel": "expand_backup_select_staged_kernel (tree step, tree staged in LDS)", "bound": "hbm (latency-bound: two load round trips, then ALU latency)",
             "launch_us": tree_s * 1e6, "achieved": tree_bytes / tree_s / 1e9, "peak": 8000.0, "unit": "GB/s",
             "frac": tree_bytes / tree_s / 8e12},
            {"kernel": "heads_kernel (3 MLPs + softmax + support decode)", "bound": "latency (4 dependent phases)",
             "launch_us": heads_s * 1e6},
        ]
        if world == 1 and not args.no_extras:
            # (1) the same workload long enough for clocks and temperature to settle
            t, n = wl.timed(0, 0, min_seconds=args.sustained_seconds)
            out["sustained"] = {"value": B * args.sims * n / t, "unit": "sims/s", "seconds": t, "steps": n,
                                "ms_per_step": t / n * 1e3}
            # (2) new frames arriving from pinned host memory every move (PCIe-inclusive; never `value`)
            if not args.from_host:
                wl.from_host = True
                t, n = wl.timed(args.steps, 2)
                wl.from_host = False
                out["from_host"] = {"value": B * args.sims * n / t, "unit": "sims/s", "ms_per_step": t / n * 1e3,
                                    "h2d_bytes_per_step": B * 96 * 96 * 3}
            wl.free()
            # (3) the other precisions on the same workload: the float32-accurate bf16x3 and f16x2 (meet the reference's
            # fp32 outputs to 1e-3 everywhere) and IEEE-half operands (same matrix rate as bf16, 8x smaller rounding steps)
            for key, other in (("secondary", "bf16x3" if args.precision != "bf16x3" else "bf16"),
                               ("f16x2", "f16x2" if args.precision != "f16x2" else "bf16"),
                               ("f16", "f16" if args.precision != "f16" else "bf16")):
                w2 = Workload(cfg, sd, device, B, args.sims, args.wave, other, rank)
                t, n = w2.timed(max(2, args.steps // 2 if other in ("bf16", "f16") else args.steps // 4), 1)
                sec = {"precision": other, "value": B * args.sims * n / t, "unit": "sims/s", "ms_per_step": t / n * 1e3,
                       "steps": n, "config": out["config"]["workload"]}
                sec["roofline"], _, _ = chain_roofline(w2, other)
                sec["roofline"]["end_to_end_frac"] = sec["value"] * 953.98e6 / (sec["roofline"]["peak"] * 1e12)
                out[key] = sec
                w2.free()
            # (4) BASELINE configs[1]: Pong-shaped, 6 actions, 256 games, eager launches and one HIP graph per search
            cfg6 = MuZeroConfig()
            cfg6.action_space_size, cfg6.num_simulations = 6, args.sims
            sd6 = init_state_dict(cfg6, seed=0, randomize_bn=True)
            small = {"workload": f"Pong-shaped 96x96 obs, 6 actions, 256 games x {args.sims} sims, wave 1"}
            for key, graph in (("eager", False), ("hip_graph", True)):
                w3 = Workload(cfg6, sd6, device, 256, args.sims, 1, "bf16", rank, graph=graph)
                t, n = w3.timed(args.steps, 2)
                small[key] = {"value": 256 * args.sims * n / t, "unit": "sims/s", "ms_per_step": t / n * 1e3}
                w3.free()
            out["small_batch"] = small
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, sd, args.sims, args.cpu_seconds)
            out["speedup_vs_cpu"] = sims_per_s / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
