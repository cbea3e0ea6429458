"""CPU restatement of the frame stack and observation preparation
(test infrastructure).

Follows /root/reference/train.py:
  CustomFrameStack.reset/step/_get_observation   train.py:45-63
  MuZeroTrainer.prepare_observation              train.py:529-540

``train.py`` cannot be imported in the build container (gymnasium, ale_py and
tensorboard are absent), so these two functions are pinned by reading the
source: the stack keeps the last ``depth`` frames, a reset fills every slot
with the first frame, and the stacked observation concatenates the frames
oldest-first along the channel axis (channel = slot * 3 + rgb).  The
normalisation divides by 255.0 in float64 and rounds to float32; that this
equals a float32 division for every byte value is checked exhaustively in the
tests.
"""
from collections import deque

import numpy as np


class FrameStack:
    def __init__(self, depth: int):
        self.depth = depth
        self.q = deque(maxlen=depth)

    def reset(self, frame: np.ndarray) -> np.ndarray:
        for _ in range(self.depth):
            self.q.append(frame)
        return self.observation()

    def push(self, frame: np.ndarray) -> np.ndarray:
        self.q.append(frame)
        return self.observation()

    def observation(self) -> np.ndarray:
        """(H, W, 3*depth) uint8, oldest frame first."""
        return np.concatenate(list(self.q), axis=2)


def prepare(obs: np.ndarray) -> np.ndarray:
    """HWC (or NHWC) uint8 -> CHW (or NCHW) float32 in [0, 1]."""
    t = np.transpose(obs, (2, 0, 1)) if obs.ndim == 3 else np.transpose(obs, (0, 3, 1, 2))
    return (t / 255.0).astype(np.float32)
