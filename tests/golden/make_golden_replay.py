#!/usr/bin/env python3
"""Golden fixture for the replay buffer, generated from the reference in the build container:

    python tests/golden/make_golden_replay.py

``model.ReplayBuffer`` (model.py:549-618) is imported as is and driven with the reference's own
``Game`` class (compiled from train.py's text as in make_golden_train.py).  Every position's
"observation" is the pair (game id, position), so the (game, position) a sampled index was
mapped to can be read back from the batch.  Only inputs and outputs are stored.

  g10_replay.npz
    lens                 positions per game, in save order
    prio_in_<g>          priorities passed to save_game for game g (absent: default = current maximum)
    s<k>_seed            np.random.seed before sample k
    s<k>_prio            the flat priority list the sample saw
    s<k>_u               the uniforms np.random.choice consumed (same seed, random_sample(batch))
    s<k>_flat            the flat indices np.random.choice drew
    s<k>_game, s<k>_pos  (game id, position) of every batch entry
    s<k>_batch_indices   what sample_batch returns as indices: the IN-GAME offset (model.py:595, :607)
    s<k>_weights         importance weights
    s<k>_tv/_tr          targets of every entry (K = 5, td = 10)
    upd_values           priorities handed to update_priorities(s0_batch_indices, .)
    prio_after_update    the flat list afterwards: the reference writes slot [in-game offset], not the
                         sampled slot - recorded as it is, the build departs from it on purpose
    prio_after_evict     after saving one more game into a full window (oldest game evicted)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, HERE)
import model as M  # noqa: E402

from make_golden_train import reference_definitions  # noqa: E402

assert int(np.__version__.split(".")[0]) >= 2, "fixtures are generated under NumPy >= 2 scalar promotion"


def main():
    Game = reference_definitions()["Game"]
    rng = np.random.default_rng(77)
    A, gamma, K, td = 4, 0.997, 5, 10
    cfg = M.MuZeroConfig()
    cfg.replay_buffer_size, cfg.batch_size = 3, 64
    lens = [5, 12, 7, 20]
    games = []
    for gi, T in enumerate(lens):
        g = Game(A, gamma)
        for t in range(T):
            g.store_transition(int(rng.integers(0, A)) if t else None, np.array([gi, t]), float(rng.normal()) if t else 0)
            root = M.Node(0, None)
            root.visit_count, root.value_sum = 1, float(rng.normal(0, 3))
            g.store_search_stats(root, {a: float(p) for a, p in enumerate(rng.dirichlet(np.ones(A)))})
        games.append(g)
    out = dict(lens=np.array(lens), A=A, gamma=gamma, K=K, td=td, batch_size=cfg.batch_size,
               window=cfg.replay_buffer_size)
    for gi, g in enumerate(games):
        out[f"g{gi}_rewards"] = np.array(g.rewards, np.float64)
        out[f"g{gi}_values"] = np.array(g.values, np.float64)
        out[f"g{gi}_policies"] = np.array(g.policies, np.float64)
        out[f"g{gi}_actions"] = np.array(g.actions, np.int64)

    rb = M.ReplayBuffer(cfg)
    rb.save_game(games[0])                                  # default priorities: 1
    out["prio_in_1"] = rng.random(lens[1]) * 3 + 0.01
    rb.save_game(games[1], out["prio_in_1"])
    rb.save_game(games[2])                                  # default: current maximum

    def sample(k, seed):
        out[f"s{k}_seed"] = seed
        out[f"s{k}_prio"] = np.array(rb.priorities, np.float64)
        np.random.seed(seed)
        out[f"s{k}_u"] = np.random.random_sample(cfg.batch_size)
        total = sum(rb.priorities)
        probs = [p / total for p in rb.priorities] if total != 0 else [1 / len(rb.priorities)] * len(rb.priorities)
        np.random.seed(seed)
        out[f"s{k}_flat"] = np.random.choice(len(rb.priorities), size=cfg.batch_size, p=probs)
        np.random.seed(seed)
        batch, idx, w = rb.sample_batch(K, td)
        assert len(batch) == cfg.batch_size
        out[f"s{k}_game"] = np.array([b[0][0] for b in batch])
        out[f"s{k}_pos"] = np.array([b[0][1] for b in batch])
        out[f"s{k}_batch_indices"] = np.array(idx)
        out[f"s{k}_weights"] = np.asarray(w, np.float64)
        out[f"s{k}_tv"] = np.array([[t[0] for t in b[2]] for b in batch], np.float64)
        out[f"s{k}_tr"] = np.array([[t[1] for t in b[2]] for b in batch], np.float64)
        return idx

    idx0 = sample(0, 11)
    out["upd_values"] = rng.random(len(idx0)) * 2
    rb.update_priorities(idx0, out["upd_values"])
    out["prio_after_update"] = np.array(rb.priorities, np.float64)
    rb.save_game(games[3])                                  # window of 3: evicts game 0
    out["prio_after_evict"] = np.array(rb.priorities, np.float64)
    sample(1, 12)
    rb.priorities = [0.0] * len(rb.priorities)              # all-zero priorities: uniform sampling (model.py:581-583)
    sample(2, 13)
    np.savez_compressed(os.path.join(HERE, "g10_replay.npz"), **out)
    print("g10 ok", {k: np.asarray(v).shape for k, v in out.items() if k.startswith("s0_")})


if __name__ == "__main__":
    main()
