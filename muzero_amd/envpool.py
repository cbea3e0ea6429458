"""Host-side environment stepping for B games: the counterpart of the reference's
``SyncVectorEnv`` of wrapped ALE environments (train.py:510-527) feeding the device frame
ring.  Environments are any objects with the gymnasium calls the reference uses
(``reset() -> (obs, info)``, ``step(a) -> (obs, reward, terminated, truncated, info)``,
train.py:45-60); they are stepped by a pool of threads (ALE releases the GIL while it
emulates) and write their newest frame straight into one of two pinned host buffers, so the
copy of step t's frames to the GPU can overlap the emulation of step t+1.

An observation may be a single ``H x W x 3`` frame or the reference's stacked
``H x W x 3k`` observation (train.py:62-63), of which only the newest frame is taken: the
stack itself lives on the device (``prep.FrameRing``).
"""
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, List, Sequence

import numpy as np
import torch


class HostEnvPool:
    def __init__(self, make_env: Callable[[int], object], num_envs: int, workers: int = 8,
                 frame_shape=(96, 96, 3), pin: bool = None, auto_reset: bool = False):
        """auto_reset: an environment whose episode ends in a step is reset within that step, which
        then returns the NEW episode's first frame with ``restarted[i]`` set (reward and end flags
        are the ended episode's; its terminal frame is dropped - the reference never records it,
        train.py:280-282), so B games keep playing without waiting for the longest one."""
        self.envs: List[object] = [make_env(i) for i in range(num_envs)]
        self.B = num_envs
        self.frame_shape = tuple(frame_shape)
        pin = torch.cuda.is_available() if pin is None else pin
        self._buffers = [torch.empty((num_envs,) + self.frame_shape, dtype=torch.uint8) for _ in range(2)]
        if pin:
            self._buffers = [b.pin_memory() for b in self._buffers]
        self._views = [b.numpy() for b in self._buffers]
        self._cur = 0
        workers = max(1, min(workers, num_envs))
        self._pool = ThreadPoolExecutor(max_workers=workers)
        bounds = np.linspace(0, num_envs, workers + 1).astype(int)
        self._slices = [range(bounds[i], bounds[i + 1]) for i in range(workers) if bounds[i] < bounds[i + 1]]
        self.rewards = np.zeros(num_envs, np.float64)
        self.terminated = np.zeros(num_envs, bool)
        self.truncated = np.zeros(num_envs, bool)
        self.finished = np.zeros(num_envs, bool)      # episodes that ended earlier and were not reset
        self.auto_reset = auto_reset
        self.restarted = np.zeros(num_envs, bool)     # this step returned the first frame of a new episode

    def _newest(self, obs) -> np.ndarray:
        obs = np.asarray(obs)
        c = self.frame_shape[2]
        if obs.shape[:2] != self.frame_shape[:2] or obs.shape[2] % c:
            raise ValueError(f"observation {obs.shape} does not end in frames of shape {self.frame_shape}")
        return obs[:, :, -c:]

    def _run(self, fn) -> None:
        for f in [self._pool.submit(fn, sl) for sl in self._slices]:
            f.result()                                  # re-raises a worker's exception

    def reset(self) -> torch.Tensor:
        """Reset every environment; returns the pinned uint8 [B, H, W, 3] first frames."""
        out = self._views[self._cur]

        def work(sl):
            for i in sl:
                obs, _ = self.envs[i].reset()
                out[i] = self._newest(obs)
        self._run(work)
        self.finished[:] = False
        return self._buffers[self._cur]

    def step(self, actions: Sequence[int]) -> torch.Tensor:
        """Step every live environment; an environment whose episode has ended keeps its last
        frame and reports reward 0 (the reference stops recording it, train.py:280-282).
        Returns this step's pinned frame buffer; it stays valid until the step after next.
        ``rewards``, ``terminated``, ``truncated`` hold this step's results."""
        actions = np.asarray(actions)
        if actions.shape != (self.B,):
            raise ValueError(f"expected {self.B} actions, got shape {actions.shape}")
        prev = self._views[self._cur]
        self._cur ^= 1
        out = self._views[self._cur]

        def work(sl):
            for i in sl:
                self.restarted[i] = False
                if self.finished[i]:
                    out[i] = prev[i]
                    self.rewards[i], self.terminated[i], self.truncated[i] = 0.0, False, False
                    continue
                obs, r, term, trunc, _ = self.envs[i].step(int(actions[i]))
                self.rewards[i], self.terminated[i], self.truncated[i] = float(r), bool(term), bool(trunc)
                if self.auto_reset and (term or trunc):
                    obs, _ = self.envs[i].reset()
                    self.restarted[i] = True
                out[i] = self._newest(obs)
        self._run(work)
        self.finished |= (self.terminated | self.truncated) & ~self.restarted
        return self._buffers[self._cur]

    def all_done(self) -> bool:
        return bool(self.finished.all())                # the loop condition of train.py:227

    def close(self) -> None:
        self._pool.shutdown(wait=True)
        for e in self.envs:
            if hasattr(e, "close"):
                e.close()
