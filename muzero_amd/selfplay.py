"""Batched self-play step: the device counterpart of one iteration of the
reference's self-play loop body (train.py:227-282) for B games at once.

    frame -> FrameRing.push -> stacked observation -> BatchedMCTS.run
          -> select_action for every game

The environment itself stays on the host (the reference steps gymnasium / ALE
there, train.py:279); ``SyntheticEnv`` stands in for it with random frames of the
same shape so that the search path can be driven and timed without ALE.
"""
from typing import Optional

import numpy as np
import torch

from .prep import FrameRing
from .search import BatchedMCTS, SearchResult
from .tree import select_actions


class SyntheticEnv:
    """B environments producing uniform random 96x96x3 uint8 frames."""

    def __init__(self, B: int, H: int = 96, W: int = 96, seed: int = 1234, device=None, pool: int = 4):
        self.B, self.H, self.W = B, H, W
        rng = np.random.default_rng(seed)
        frames = rng.integers(0, 256, size=(pool, B, H, W, 3), dtype=np.uint8)
        self.host = torch.from_numpy(frames)
        if torch.cuda.is_available():
            self.host = self.host.pin_memory()
        self.dev = self.host.to(device) if device is not None else None
        self.t = 0

    def reset(self, on_device: bool = True) -> torch.Tensor:
        self.t = 0
        return (self.dev if on_device and self.dev is not None else self.host)[0]

    def step(self, actions=None, on_device: bool = True) -> torch.Tensor:
        self.t += 1
        src = self.dev if on_device and self.dev is not None else self.host
        return src[self.t % src.shape[0]]


class SelfPlayWorker:
    """Owns the frame ring, the search and the action choice for B games on one GPU."""

    def __init__(self, config, network, batch_size: int, num_simulations: int = 50, wave: int = 1,
                 frame_stack: int = 32, obs_hw=(96, 96), seed: int = 5678, rotating: bool = True):
        self.config = config
        self.net = network
        self.B = batch_size
        self.A = config.action_space_size
        self.ring = FrameRing(batch_size, frame_stack, obs_hw[0], obs_hw[1], network.device)
        self.rotating = rotating and not getattr(network, "split", False)   # bf16x3 reads the stacked observation
        if self.rotating:
            self.ring.enable_rotating(network.obs_dtype)   # a move then rewrites 3 of the 96 channels per pixel
        self.mcts = BatchedMCTS(config, network, batch_size, num_simulations, wave, obs_hw)
        self._obs = None
        self.pairs = bool(getattr(network, "split", False))    # bf16x3 takes the observation as (hi, lo) pairs
        self._obs_shape = ((batch_size, 2) if self.pairs else (batch_size,)) + (obs_hw[0], obs_hw[1], 3 * frame_stack)
        self.rng = np.random.default_rng(seed)
        self.noise = torch.empty(batch_size, self.A, dtype=torch.float64, device=network.device)
        self.u = torch.empty(batch_size, dtype=torch.float64, device=network.device)
        self.sims = num_simulations

    @property
    def obs(self) -> torch.Tensor:
        """Buffer for the stacked observation in the reference's channel order (allocated on first
        use: the rotating path never needs it)."""
        if self._obs is None:
            self._obs = torch.empty(*self._obs_shape, dtype=self.net.act_dtype if self.pairs else self.net.obs_dtype,
                                    device=self.net.device)
        return self._obs

    def reset(self, frames: torch.Tensor) -> None:
        self.ring.reset(frames)

    def check_status(self) -> None:
        """Raise if a tree reported trouble in the last search (one small device read-back)."""
        self.mcts.check_status()

    def draw_randomness(self) -> None:
        """Dirichlet rows for the roots and uniforms for the action choice (host RNG)."""
        alpha = getattr(self.config, "root_dirichlet_alpha", 0.3)
        self.noise.copy_(torch.from_numpy(self.rng.dirichlet(np.full(self.A, alpha), size=self.B)), non_blocking=True)
        self.u.copy_(torch.from_numpy(self.rng.random(self.B)), non_blocking=True)

    def step(self, new_frames: Optional[torch.Tensor], temperature: float = 1.0, add_noise: bool = True,
             restarted: Optional[torch.Tensor] = None):
        """One move for all B games.  Returns (actions int32 [B] on the device, SearchResult).
        ``restarted`` marks games for which ``new_frames`` holds the first frame of a new episode."""
        if new_frames is not None:
            self.ring.push(new_frames, restarted)
        if self.rotating:
            obs, newest = self.ring.rotating_observation()
        else:
            dtype = self.net.act_dtype if self.pairs else self.net.obs_dtype
            obs, newest = self.ring.observation(dtype, out=self.obs, pairs=self.pairs), None
        res: SearchResult = self.mcts.run(obs, self.noise if add_noise else None, rotation=newest)
        actions = select_actions(res.visit_counts, temperature, self.u if temperature != 0 else None)
        return actions, res
