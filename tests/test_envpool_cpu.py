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
