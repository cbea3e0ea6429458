#!/usr/bin/env python3
"""Headline benchmark: MCTS simulations per second of the batched self-play search.

One "step" = one move for every game on every GPU: push one new 96x96x3 frame per
game into the device frame ring and the stacked observation, run the search
(initial inference + `sims` simulations at wave 1, each a distinct recurrent
inference) and choose an action per game.  Synthetic frames and weights.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Games are sharded over ranks with no data-path collective (weak scaling: the
per-GPU batch is fixed); RCCL only broadcasts the weights before the timed region.
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
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
    net.load_state_dict(sd, strict=True)
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
    n_convs = 1 + 2 * len(net.dyn_blocks)
    # s per launch, convs per launch, launches, (tree s per launch, heads s per launch)
    return mean_s(pairs), n_convs, len(pairs), (mean_s(tree), mean_s(heads))


def load_profiled_traffic():
    """HBM bytes per launch of the dominant kernel from the last rocprofv3 PMC passes
    (FETCH_SIZE / WRITE_SIZE, corrected as documented in profiles/); None if never profiled."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="synthetic18", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="games per GPU (default: the workload's)")
    ap.add_argument("--sims", type=int, default=50)
    ap.add_argument("--wave", type=int, default=1)
    ap.add_argument("--graph", action="store_true", help="replay the search from a HIP graph")
    ap.add_argument("--from-host", action="store_true", help="include the pinned H2D copy of the new frames")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    args = ap.parse_args()

    from muzero_amd import dist as mzdist
    from muzero_amd.config import MuZeroConfig
    from muzero_amd.network import MuZeroNetwork, init_state_dict
    from muzero_amd.selfplay import SelfPlayWorker, SyntheticEnv
    from muzero_amd.tree import select_actions

    rank, world, local = mzdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    A, B, label = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    desc = f"{label} 96x96 obs (32 RGB frames), {A} actions, {B} games x {args.sims} sims per GPU"
    cfg = MuZeroConfig()
    cfg.action_space_size = A
    cfg.num_simulations = args.sims

    # weights: created on rank 0, handed to the other ranks with one RCCL broadcast
    sd = init_state_dict(cfg, seed=0 if rank == 0 else 1 + rank, randomize_bn=True)
    sd = mzdist.broadcast_state_dict(sd, src=0)
    if not mzdist.verify_same_everywhere(sd):
        raise SystemExit("weight broadcast left ranks with different parameters")
    net = MuZeroNetwork(cfg, precision="bf16", device=device, state_dict=sd)
    worker = SelfPlayWorker(cfg, net, B, args.sims, args.wave, seed=5678 + rank)
    env = SyntheticEnv(B, seed=1234 + rank, device=device)
    worker.reset(env.reset())
    for _ in range(31):
        worker.ring.push(env.step())
    worker.draw_randomness()

    replay = None
    if args.graph:
        worker.ring.observation(net.obs_dtype, out=worker.obs)
        replay = worker.mcts.capture(worker.obs, worker.noise)

    def one_step():
        worker.draw_randomness()  # fresh Dirichlet rows and uniforms every move (host RNG, async copy)
        frames = env.step(on_device=not args.from_host)
        if replay is None:
            return worker.step(frames)[0]
        worker.ring.push(frames)
        worker.ring.observation(net.obs_dtype, out=worker.obs)
        res = replay()
        return select_actions(res.visit_counts, 1.0, worker.u)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        actions = one_step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    assert actions.shape == (B,)
    elapsed = mzdist.max_over_ranks(elapsed)
    visits_ok = int(worker.mcts.visits.sum().item()) == B * args.sims
    worker.mcts.check_status()

    if rank == 0:
        total_sims = world * B * args.sims * args.steps
        sims_per_s = total_sims / elapsed
        out = {
            "metric": "mcts_sims_per_sec", "value": sims_per_s, "unit": "sims/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": desc + f", wave {args.wave}", "games_per_gpu": B, "actions": A,
                       "sims_per_move": args.sims, "wave": args.wave, "parallelism": f"games sharded x{world}",
                       "hip_graph": bool(args.graph), "frames_from_host": bool(args.from_host)},
            "env_steps_per_sec": world * B * args.steps / elapsed,
            # ALE "-v5" environments repeat each action over 4 emulator frames (built-in frameskip)
            "env_frames_per_sec": 4 * world * B * args.steps / elapsed,
            "visit_sum_check": visits_ok,
        }
        dur, n_convs, n_launches, (tree_s, heads_s) = measure_dynamics_kernel(worker, lambda: worker.step(env.step())[0])
        flops = TRUNK_FLOPS * n_convs * B
        achieved = flops / dur / 1e12
        prof = load_profiled_traffic()
        traffic = None
        if prof and prof.get("games_per_gpu") == B:
            traffic = prof.get("hbm_bytes_per_launch")
        out["roofline"] = {"bound": "mfma",
                           "kernel": ("conv_chain_pair_kernel" if B > torch.cuda.get_device_properties(0).multi_processor_count
                                      else "conv_chain_kernel") + f": dynamics network, {n_convs} fused 3x3 convs 12x12x256->256",
                           "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                           "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic,
                           "launch_us": dur * 1e6, "launches_timed": n_launches, "flops_per_launch": flops,
                           "end_to_end_frac": sims_per_s / world * 953.98e6 / (PEAK_BF16_TFLOPS * 1e12)}
        # the other launches of a simulation step, timed the same way (HIP events include ~2 us of
        # launch latency).  Tree: algorithmic bytes per simulation = D*(A*20+20) read by the descent
        # + A*20+12 written by the expansion + (D+1)*32 for the backup at mean depth D ~ 4.
        tree_bytes = B * (4 * (A * 20 + 20) + A * 20 + 12 + 5 * 32)
        out["secondary_kernels"] = [
            {"kernel": "expand_backup_select_kernel (tree step)", "bound": "hbm (latency-bound dependent loads)",
             "launch_us": tree_s * 1e6, "achieved": tree_bytes / tree_s / 1e9, "peak": 8000.0, "unit": "GB/s",
             "frac": tree_bytes / tree_s / 8e12},
            {"kernel": "heads_kernel (3 MLPs + softmax + support decode)", "bound": "latency (4 dependent phases)",
             "launch_us": heads_s * 1e6},
        ]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, sd, args.sims, args.cpu_seconds)
            out["speedup_vs_cpu"] = sims_per_s / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
