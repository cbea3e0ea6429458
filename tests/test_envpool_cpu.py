"""Host environment pool: threaded stepping into double-buffered frame buffers."""
import numpy as np
import pytest

from muzero_amd.envpool import HostEnvPool


class FakeEnv:
    """Frames encode (env id, step); episode i ends after 3 + i steps; reward = action."""

    def __init__(self, idx, stacked):
        self.idx, self.t, self.stacked = idx, 0, stacked

    def _obs(self):
        f = np.full((8, 8, 3), (self.idx * 16 + self.t) % 256, np.uint8)
        return np.concatenate([np.zeros((8, 8, 9), np.uint8), f], axis=2) if self.stacked else f

    def reset(self):
        self.t = 0
        return self._obs(), {}

    def step(self, a):
        self.t += 1
        return self._obs(), float(a), self.t >= 3 + self.idx, False, {}


@pytest.mark.parametrize("stacked", [False, True])
def test_pool_steps_all_envs_and_double_buffers(stacked):
    pool = HostEnvPool(lambda i: FakeEnv(i, stacked), 5, workers=3, frame_shape=(8, 8, 3), pin=False)
    f0 = pool.reset()
    assert f0.shape == (5, 8, 8, 3) and [int(f0[i, 0, 0, 0]) for i in range(5)] == [0, 16, 32, 48, 64]
    f1 = pool.step(np.arange(5))
    assert [int(f1[i, 0, 0, 0]) for i in range(5)] == [1, 17, 33, 49, 65]
    assert int(f0[1, 0, 0, 0]) == 16                      # the previous buffer is still intact
    assert pool.rewards.tolist() == [0, 1, 2, 3, 4] and not pool.finished.any()
    pool.step(np.zeros(5, int))
    f3 = pool.step(np.ones(5, int))                       # env 0 ends here (3 steps)
    assert pool.terminated.tolist() == [True, False, False, False, False]
    f4 = pool.step(np.ones(5, int))                       # env 0 is frozen: last frame, reward 0
    assert int(f4[0, 0, 0, 0]) == 3 and pool.rewards[0] == 0 and pool.rewards[1] == 1
    assert pool.envs[0].t == 3 and not pool.all_done()
    for _ in range(4):
        pool.step(np.zeros(5, int))
    assert pool.all_done()
    pool.close()


def test_pool_rejects_wrong_shapes():
    pool = HostEnvPool(lambda i: FakeEnv(i, False), 2, workers=1, frame_shape=(8, 8, 3), pin=False)
    pool.reset()
    with pytest.raises(ValueError):
        pool.step([0])
    bad = HostEnvPool(lambda i: FakeEnv(i, False), 2, workers=2, frame_shape=(4, 4, 3), pin=False)
    with pytest.raises(ValueError):
        bad.reset()
    pool.close(); bad.close()


def test_play_moves_records_like_the_reference_loop():
    """loop.play_moves against a stub worker: every game gets search statistics for each position it
    reached, the move that ends a game is not recorded as a transition, and frames coming from an
    environment pool's reused buffers are copied."""
    import torch
    from muzero_amd import loop
    from muzero_amd.learner import GameHistory

    B, A = 5, 3

    class Res:
        def __init__(self, t):
            self.visit_counts = torch.tensor([[t + 1, 1, 0]] * B, dtype=torch.int32)
            self.root_values = torch.full((B,), float(t), dtype=torch.float64)

    class Worker:
        def __init__(self):
            self.t = 0

        def draw_randomness(self):
            pass

        def check_status(self):
            pass

        def step(self, frames, temperature):
            self.t += 1
            return torch.full((B,), self.t % A, dtype=torch.int32), Res(self.t)

    pool = HostEnvPool(lambda i: FakeEnv(i, False), B, workers=2, frame_shape=(8, 8, 3), pin=False)
    first = pool.reset()
    games = [GameHistory(A, 0.997, frame_stack=4) for _ in range(B)]
    for g, f in zip(games, first.numpy().copy()):
        g.store_transition(None, f, 0)
    loop.play_moves(Worker(), games, lambda a: (pool.step(a), pool.rewards, pool.terminated | pool.truncated), 6)
    for i, g in enumerate(games):
        n = min(3 + i, 6)                       # FakeEnv i ends at its move 3 + i; the phase truncates at 6
        assert g.done and len(g.values) == n and len(g.policies) == n
        assert len(g.rewards) == n and len(g.actions) == n      # reset entry + n - 1 transitions
        assert g.values == [float(t) for t in range(1, n + 1)]
        assert [int(f[0, 0, 0]) for f in g.frames] == [(i * 16 + t) % 256 for t in range(n)]
        assert g.actions[1:] == [t % A for t in range(1, n)]
        assert abs(g.policies[0].sum() - 1) < 1e-12
    pool.close()


def test_pool_auto_reset_restarts_finished_games():
    pool = HostEnvPool(lambda i: FakeEnv(i, False), 3, workers=2, frame_shape=(8, 8, 3), pin=False, auto_reset=True)
    pool.reset()
    for _ in range(2):
        pool.step(np.zeros(3, int))
    f = pool.step(np.ones(3, int))                          # env 0 ends (3 steps) and restarts in the same step
    assert pool.terminated.tolist() == [True, False, False] and pool.restarted.tolist() == [True, False, False]
    assert int(f[0, 0, 0, 0]) == 0 and pool.rewards[0] == 1 and pool.envs[0].t == 0
    f = pool.step(np.ones(3, int))                          # env 1 ends now; env 0 is one step into its new episode
    assert pool.restarted.tolist() == [False, True, False] and int(f[0, 0, 0, 0]) == 1 and int(f[1, 0, 0, 0]) == 16
    assert not pool.finished.any() and not pool.all_done()
    pool.close()


def test_play_continuous_hands_over_finished_games_and_restarts_them():
    import torch
    from muzero_amd import loop

    B, A = 3, 3

    class Res:
        def __init__(self, t):
            self.visit_counts = torch.tensor([[t + 1, 1, 0]] * B, dtype=torch.int32)
            self.root_values = torch.full((B,), float(t), dtype=torch.float64)

    class Ring:
        depth = 4

    class Cfg:
        discount = 0.997

    class Worker:
        A, B, config, ring = 3, 3, Cfg, Ring

        def __init__(self):
            self.t, self.masks = 0, []

        def reset(self, frames):
            pass

        def draw_randomness(self):
            pass

        def check_status(self):
            pass

        def step(self, frames, temperature, restarted=None):
            self.t += 1
            self.masks.append(None if restarted is None else restarted.tolist())
            return torch.full((B,), self.t % A, dtype=torch.int32), Res(self.t)

    pool = HostEnvPool(lambda i: FakeEnv(i, False), B, workers=1, frame_shape=(8, 8, 3), pin=False, auto_reset=True)
    done, w = [], Worker()
    n = loop.play_continuous(w, pool, 8, done.append)
    # episode lengths 3, 4, 5: env 0 ends at moves 3 and 6, env 1 at 4 and 8, env 2 at 5
    assert n == 5 and [len(g.values) for g in done] == [3, 4, 5, 3, 4]
    for g in done:
        assert g.done and len(g.rewards) == len(g.values) == len(g.policies)
        assert [int(f[0, 0, 0]) % 16 for f in g.frames] == list(range(len(g.frames)))   # frames 0..n-1 of ONE episode
    assert w.masks[3] == [True, False, False] and w.masks[4] == [False, True, False] and w.masks[1] is None
    pool.close()


def test_play_pipelined_matches_two_independent_continuous_runs():
    """Two groups taking turns (search of one behind the environment steps of the other) record
    exactly what each group records when played on its own."""
    import torch
    from muzero_amd import loop

    B, A = 3, 3

    class Res:
        def __init__(self, t):
            self.visit_counts = torch.tensor([[t + 1, 1, 0]] * B, dtype=torch.int32)
            self.root_values = torch.full((B,), float(t), dtype=torch.float64)

    class Ring:
        depth = 4

    class Cfg:
        discount = 0.997

    class Worker:
        A, B, config, ring = 3, 3, Cfg, Ring

        def __init__(self, base):
            self.t = base

        def reset(self, frames):
            pass

        def draw_randomness(self):
            pass

        def check_status(self):
            pass

        def step(self, frames, temperature, restarted=None):
            self.t += 1
            return torch.full((B,), self.t % A, dtype=torch.int32), Res(self.t)

    def mk_pool(off):
        return HostEnvPool(lambda i: FakeEnv(i + off, False), B, workers=1, frame_shape=(8, 8, 3), pin=False,
                           auto_reset=True)

    def summary(games):
        return sorted((len(g.values), tuple(g.values), tuple(g.actions), tuple(int(f[0, 0, 0]) for f in g.frames))
                      for g in games)

    solo = []
    for base, off in ((0, 0), (100, 2)):
        pool = mk_pool(off)
        loop.play_continuous(Worker(base), pool, 9, solo.append)
        pool.close()
    both, pools = [], [mk_pool(0), mk_pool(2)]
    n = loop.play_pipelined([Worker(0), Worker(100)], pools, 9, both.append)
    assert n == len(both) == len(solo) and summary(both) == summary(solo)
    for p in pools:
        p.close()
