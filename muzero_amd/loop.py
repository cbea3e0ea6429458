"""Self-play feeding training feeding self-play: the pieces of the reference's outer loop
(train.py:181-206) that touch the search path, for B concurrent games on one GPU.

    play_moves      the loop body train.py:227-282 for all games at once: search, record the
                    search statistics (Game.store_search_stats), choose, step, record the
                    transition (Game.store_transition)
    train_steps     train.py:295-319 on positions sampled from the replay buffer
    publish         hand the fresh weights to every rank's search engine (one broadcast)
"""
from typing import Callable, List, Optional

import numpy as np
import torch

from . import dist as mzdist
from .learner import GameHistory, Learner, ReplayBuffer, compute_priorities, unroll_actions
from .selfplay import SelfPlayWorker


def new_games(worker: SelfPlayWorker, first_frames: torch.Tensor) -> List[GameHistory]:
    """Reset the frame rings and open one episode record per game (train.py:214-222)."""
    worker.reset(first_frames)
    host = first_frames.cpu().numpy() if first_frames.is_cuda else first_frames.numpy().copy()   # a pool reuses its buffers
    games = [GameHistory(worker.A, worker.config.discount, worker.ring.depth) for _ in range(worker.B)]
    for g, f in zip(games, host):
        g.store_transition(None, f, 0)
    return games


def play_moves(worker: SelfPlayWorker, games: List[GameHistory], env_step: Callable, moves: int,
               temperature: float = 1.0) -> None:
    """``env_step(actions int array [B]) -> (frames uint8 [B,H,W,3] tensor, rewards [B][, done [B]])``
    (for a ``HostEnvPool`` p: ``lambda a: (p.step(a), p.rewards, p.terminated | p.truncated)``).
    As in the reference, the step that ends an episode is not recorded (train.py:280-282) and a
    finished game records nothing further; after ``moves`` moves every game still running is
    truncated the same way, so every recorded position has search statistics."""
    B = len(games)
    frames = None
    V, P, Aq, R, F, E = [], [], [], [], [], []
    for move in range(moves):
        worker.draw_randomness()
        actions, res = worker.step(frames, temperature)
        counts = res.visit_counts.cpu().numpy()
        V.append(res.root_values.cpu().numpy().copy())
        acts = actions.cpu().numpy()
        P.append(counts / np.maximum(counts.sum(1, keepdims=True), 1))
        stepped = env_step(acts)
        frames = stepped[0]
        Aq.append(acts)
        R.append(np.array(stepped[1], np.float64))           # copies: an environment pool reuses its arrays
        E.append(np.array(stepped[2], bool) if len(stepped) > 2 else np.zeros(B, bool))
        F.append(frames.cpu().numpy() if frames.is_cuda else frames.numpy().copy())
    # hand every game its moves in one go (the per-move work above stays free of per-game Python)
    V, R, E, Aq = np.stack(V), np.stack(R), np.stack(E), np.stack(Aq)
    E[moves - 1] = True                                      # the phase ends by truncation
    for i, g in enumerate(games):
        if g.done:
            continue
        last = int(np.argmax(E[:, i]))                       # first move that ends this game
        g.values.extend(V[:last + 1, i].tolist())            # store_search_stats, train.py:83-87
        g.policies.extend(P[m][i].astype(np.float64) for m in range(last + 1))
        for m in range(last):                                # the ending move itself is not recorded
            g.store_transition(int(Aq[m, i]), F[m][i], float(R[m, i]))
        g.done = True


def play_continuous(worker: SelfPlayWorker, pool, moves: int, on_game_end: Callable,
                    temperature: float = 1.0) -> int:
    """B games that never wait for each other: ``pool`` is a ``HostEnvPool(auto_reset=True)``; a
    game whose episode ends is handed to ``on_game_end(game)`` (e.g. ``replay.save_game``), its
    environment has already restarted, its frame stack is refilled from the new episode's first
    frame (per-game reset on the device) and a fresh record is opened - the reference plays
    every batch of games to the end of the longest one (train.py:227).  Recording follows
    train.py:268 / :280-282.  Returns the number of finished games."""
    games = new_games(worker, pool.reset())
    frames, restarted, finished = None, None, 0
    for _ in range(moves):
        worker.draw_randomness()
        actions, res = worker.step(frames, temperature, restarted=restarted)
        counts = res.visit_counts.cpu().numpy()
        values = res.root_values.cpu().numpy()
        acts = actions.cpu().numpy()
        probs = counts / np.maximum(counts.sum(1, keepdims=True), 1)
        frames = pool.step(acts)
        host = frames.numpy()
        ended = pool.terminated | pool.truncated
        restarted = pool.restarted.copy() if pool.restarted.any() else None
        for i, g in enumerate(games):
            g.values.append(float(values[i]))                 # store_search_stats, train.py:83-87
            g.policies.append(probs[i].astype(np.float64))
            if ended[i]:                                      # the ending move is not recorded (train.py:281)
                g.done = True
                on_game_end(g)
                finished += 1
                games[i] = GameHistory(worker.A, worker.config.discount, worker.ring.depth)
                games[i].store_transition(None, host[i].copy(), 0)
            else:
                g.store_transition(int(acts[i]), host[i].copy(), float(pool.rewards[i]))
    return finished


def evaluate(worker: SelfPlayWorker, first_frames: torch.Tensor, env_step: Callable, max_moves: int) -> np.ndarray:
    """The reference's evaluation episodes (train.py:433-478) for B games at once: no root noise
    (train.py:456-460), most-visited action (select_action with temperature 0, train.py:467),
    rewards summed until each game ends or ``max_moves`` is reached.  ``env_step`` as in
    ``play_moves`` (it must report ``done``).  Returns the episode returns [B]."""
    worker.reset(first_frames)
    total = np.zeros(worker.B)
    alive = np.ones(worker.B, bool)
    frames = None
    for _ in range(max_moves):
        actions, _ = worker.step(frames, temperature=0, add_noise=False)
        stepped = env_step(actions.cpu().numpy())
        frames, rewards = stepped[0], np.asarray(stepped[1], np.float64)
        total += np.where(alive, rewards, 0.0)
        if len(stepped) > 2:
            alive &= ~np.asarray(stepped[2], bool)
        if not alive.any():
            break
    return total


def save_games(replay: ReplayBuffer, games: List[GameHistory], td_steps: int) -> None:
    """train.py:286-289.  A game holds one position more than it holds search values (the last
    frame was never searched); the buffer keeps one priority per position, so that position
    takes the game's largest priority."""
    for g in games:
        pr = compute_priorities(g, td_steps)
        full = np.full(len(g.rewards), pr.max() if len(pr) else 1.0)
        n = min(len(pr), len(full))
        full[:n] = pr[:n]
        replay.save_game(g, full)


def train_steps(learner: Learner, replay: ReplayBuffer, steps: int) -> List[dict]:
    """train.py:295-319.  With a device episode store behind the replay buffer the targets of
    the whole batch come from one kernel launch; otherwise from ``make_target`` per position."""
    cfg = learner.config
    K = cfg.num_unroll_steps
    out = []
    for _ in range(steps):
        gi, pos, flat, weights = replay.sample_positions()
        games = [replay.buffer[int(g)] for g in gi]
        if replay.store is not None:
            tv, tr, tp = replay.store.targets(gi, pos, K, cfg.td_steps)
            obs = torch.from_numpy(np.stack([g.get_observation(int(p)) for g, p in zip(games, pos)]))
            obs = obs.to(learner.device).permute(0, 3, 1, 2).float() / 255.0
            acts = unroll_actions([g.actions[int(p):int(p) + K] for g, p in zip(games, pos)], K)    # model.py:604
            m = learner.update_from_tensors(obs, acts, tv, tr, tp, weights)
        else:
            batch = [(g.get_observation(int(p)), g.actions[int(p):int(p) + K], g.make_target(int(p), K, cfg.td_steps))
                     for g, p in zip(games, pos)]                                                  # model.py:602-606
            m = learner.update_weights(batch, weights)
        replay.update_priorities(flat, m["priorities"])
        out.append(m)
    return out


def publish(learner: Optional[Learner], network, src: int = 0) -> None:
    """Rank ``src`` holds the learner; every rank's search engine gets its weights."""
    sd = learner.state_dict() if learner is not None else network.state_dict()
    network.load_state_dict(mzdist.broadcast_state_dict(sd, src=src))
