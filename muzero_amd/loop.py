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
        worker.check_status()             # e.g. no selectable child under non-finite priors: the reference crashes there
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


class _ContinuousGroup:
    """One worker + one auto-resetting environment pool and the episode records of their games.
    A move is: ``launch`` (enqueue the search, no wait) -> ``record`` the previous move on the host
    while the GPU works -> ``fetch`` the search result -> step the environments -> ``stepped``."""

    def __init__(self, worker: SelfPlayWorker, pool, on_game_end: Callable):
        self.worker, self.pool, self.on_game_end = worker, pool, on_game_end
        self.games = new_games(worker, pool.reset())
        self.frames = self.restarted = self._search = self._last = None
        self.finished = 0

    def launch(self, temperature: float) -> None:
        self.worker.draw_randomness()
        self._search = self.worker.step(self.frames, temperature, restarted=self.restarted)

    def fetch(self):
        """Wait for the search; returns host arrays (actions, root values, visit distributions)."""
        actions, res = self._search
        counts = res.visit_counts.cpu().numpy()
        self.worker.check_status()        # the host waits here anyway; trouble in a tree must not reach the replay buffer
        probs = counts / np.maximum(counts.sum(1, keepdims=True), 1)
        self._last = (actions.cpu().numpy(), res.root_values.cpu().numpy().copy(), probs)
        return self._last[0]

    def stepped(self, frames) -> None:
        """The environments were stepped with the last actions: next move's input."""
        self.frames = frames
        self.restarted = self.pool.restarted.copy() if self.pool.restarted.any() else None
        self._flags = ((self.pool.terminated | self.pool.truncated).copy(), self.pool.rewards.copy())

    def record(self) -> None:
        """Book-keeping of the last move (train.py:268, :280-282); pure host work."""
        if self._last is None:
            return
        (acts, values, probs), self._last = self._last, None
        (ended, rewards), host, w = self._flags, self.frames.numpy(), self.worker
        for i, g in enumerate(self.games):
            g.values.append(float(values[i]))                 # store_search_stats, train.py:83-87
            g.policies.append(probs[i].astype(np.float64))
            if ended[i]:                                      # the ending move is not recorded (train.py:281)
                g.done = True
                self.on_game_end(g)
                self.finished += 1
                self.games[i] = GameHistory(w.A, w.config.discount, w.ring.depth)
                self.games[i].store_transition(None, host[i].copy(), 0)
            else:
                g.store_transition(int(acts[i]), host[i].copy(), float(rewards[i]))


def play_continuous(worker: SelfPlayWorker, pool, moves: int, on_game_end: Callable,
                    temperature: float = 1.0) -> int:
    """B games that never wait for each other: ``pool`` is a ``HostEnvPool(auto_reset=True)``; a
    game whose episode ends is handed to ``on_game_end(game)`` (e.g. ``replay.save_game``), its
    environment has already restarted, its frame stack is refilled from the new episode's first
    frame (per-game reset on the device) and a fresh record is opened - the reference plays
    every batch of games to the end of the longest one (train.py:227).  Recording follows
    train.py:268 / :280-282 and runs on the host while the GPU searches the next position.
    Returns the number of finished games."""
    grp = _ContinuousGroup(worker, pool, on_game_end)
    for _ in range(moves):
        grp.launch(temperature)
        grp.record()
        grp.stepped(pool.step(grp.fetch()))
    grp.record()
    return grp.finished


def play_pipelined(workers, pools, moves: int, on_game_end: Callable, temperature: float = 1.0) -> int:
    """``play_continuous`` for two groups of games that take turns on the GPU: while one group's
    positions are searched, the other group's environments are stepped on the host (a background
    thread; emulators release the GIL), so the emulation time disappears behind the search
    instead of adding to it (the reference alternates the two serially, train.py:262 / :279).
    ``workers`` / ``pools``: two of each, one per group.  ``moves`` counts moves per group."""
    from concurrent.futures import ThreadPoolExecutor
    if len(workers) != 2 or len(pools) != 2:
        raise ValueError("play_pipelined takes exactly two workers and two pools")
    groups = [_ContinuousGroup(w, p, on_game_end) for w, p in zip(workers, pools)]
    stepping = [None, None]                    # future of each group's environment step
    with ThreadPoolExecutor(max_workers=1) as bg:
        for _ in range(moves):
            for k, grp in enumerate(groups):
                if stepping[k] is not None:    # stepped behind the other group's search
                    grp.stepped(stepping[k].result())
                grp.launch(temperature)
                grp.record()                   # the previous move's book-keeping, behind this search
                stepping[k] = bg.submit(grp.pool.step, grp.fetch())
        for k, grp in enumerate(groups):
            grp.stepped(stepping[k].result())
            grp.record()
    return sum(g.finished for g in groups)


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
        acts = actions.cpu().numpy()
        worker.check_status()
        stepped = env_step(acts)
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


def train_steps(learner: Learner, replay: ReplayBuffer, steps: int, graphed: bool = False) -> List[dict]:
    """train.py:295-319.  With a device episode store behind the replay buffer the targets of
    the whole batch come from one kernel launch; otherwise from ``make_target`` per position.
    graphed: replay each update from a HIP graph (``Learner.update_graphed``; needs the episode store)."""
    cfg = learner.config
    K = cfg.num_unroll_steps
    out = []
    for _ in range(steps):
        gi, pos, flat, weights = replay.sample_positions()
        games = [replay.buffer[int(g)] for g in gi]
        if replay.store is not None:
            tv, tr, tp = replay.store.targets(gi, pos, K, cfg.td_steps)
            if replay.store.frames is not None:      # frames live on the device: one gather kernel
                obs = replay.store.observations(gi, pos)
            else:
                obs = torch.from_numpy(np.stack([g.get_observation(int(p)) for g, p in zip(games, pos)]))
                obs = obs.to(learner.device).permute(0, 3, 1, 2).float() / 255.0
            acts = unroll_actions([g.actions[int(p):int(p) + K] for g, p in zip(games, pos)], K)    # model.py:604
            m = (learner.update_graphed if graphed else learner.update_from_tensors)(obs, acts, tv, tr, tp, weights)
        else:
            batch = [(g.get_observation(int(p)), g.actions[int(p):int(p) + K], g.make_target(int(p), K, cfg.td_steps))
                     for g, p in zip(games, pos)]                                                  # model.py:602-606
            m = learner.update_weights(batch, weights)
        replay.update_priorities(flat, m["priorities"])
        out.append(m)
    return out


def publish(learner: Optional[Learner], network, src: int = 0) -> None:
    """Rank ``src`` holds the learner; every rank's search engine gets its weights: one flat broadcast
    (device to device under RCCL), then BatchNorm folding and fragment packing by kernels into the
    engine's existing weight buffers - the parameters never visit the host."""
    sd = learner.device_state_dict() if learner is not None else network.state_dict()
    network.load_state_dict(mzdist.broadcast_state_dict(sd, src=src))
