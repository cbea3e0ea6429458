#!/usr/bin/env python3
"""Self-play -> replay -> K-step unrolled training -> fresh weights back into the search, on
synthetic frames.  One GPU:  python examples/full_loop.py --games 64 --moves 8 --iters 2
N GPUs (all ranks search; rank 0 trains and the weights travel by one RCCL broadcast, or with
--data-parallel every rank trains on its own replay buffer and the gradients are averaged by one
RCCL all-reduce per update):
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 examples/full_loop.py
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from muzero_amd import dist as mzdist, loop  # noqa: E402
from muzero_amd.config import MuZeroConfig  # noqa: E402
from muzero_amd.learner import EpisodeStore, Learner, ReplayBuffer  # noqa: E402
from muzero_amd.network import MuZeroNetwork, init_state_dict  # noqa: E402
from muzero_amd.selfplay import SelfPlayWorker, SyntheticEnv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=64, help="concurrent games per GPU")
    ap.add_argument("--moves", type=int, default=8, help="moves per self-play phase")
    ap.add_argument("--sims", type=int, default=50)
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--train-steps", type=int, default=4)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--actions", type=int, default=4)
    ap.add_argument("--data-parallel", action="store_true", help="every rank trains; gradients all-reduced")
    ap.add_argument("--learner-precision", default="bf16", choices=["fp32", "bf16"])
    ap.add_argument("--graphed", action="store_true", help="replay each update from a HIP graph")
    args = ap.parse_args()

    rank, world, local = mzdist.init_from_env()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = MuZeroConfig()
    cfg.observation_shape, cfg.action_space_size = (96, 96, 96), args.actions
    cfg.num_simulations, cfg.batch_size = args.sims, args.batch
    sd = mzdist.broadcast_state_dict(init_state_dict(cfg, seed=rank), src=0)
    net = MuZeroNetwork(cfg, precision="bf16", device=dev, state_dict=sd)
    trains = rank == 0 or args.data_parallel
    learner = Learner(cfg, device=dev, state_dict=sd, data_parallel=args.data_parallel,
                      precision=args.learner_precision) if trains else None
    replay = ReplayBuffer(cfg, EpisodeStore(args.actions, cfg.discount, dev))
    worker = SelfPlayWorker(cfg, net, args.games, args.sims, seed=100 + rank)
    env = SyntheticEnv(args.games, seed=7 + rank, device=dev)
    rng = np.random.default_rng(rank)

    def env_step(actions):
        return env.step(actions), (rng.random(args.games) < 0.1) * np.where(actions == 0, 1.0, -1.0)

    for it in range(args.iters):
        t0 = time.time()
        games = loop.new_games(worker, env.reset())
        loop.play_moves(worker, games, env_step, args.moves, cfg.visit_softmax_temperature_fn(it))
        torch.cuda.synchronize()
        t1 = time.time()
        if trains:
            loop.save_games(replay, games, cfg.td_steps)
            metrics = loop.train_steps(learner, replay, args.train_steps, graphed=args.graphed)
        torch.cuda.synchronize()
        t2 = time.time()
        loop.publish(learner, net)   # rank 0's weights (and BatchNorm statistics) reach every search engine
        torch.cuda.synchronize()
        t3 = time.time()
        if rank == 0:
            m = metrics[-1]
            print(f"iter {it}: self-play {args.games * args.moves * args.sims / (t1 - t0):.0f} sims/s/GPU, "
                  f"{args.train_steps} updates in {t2 - t1:.2f}s, publish {1e3 * (t3 - t2):.1f} ms, loss {m['total_loss']:.4f} "
                  f"(value {m['value_loss']:.3f} reward {m['reward_loss']:.3f} policy {m['policy_loss']:.3f})",
                  flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
