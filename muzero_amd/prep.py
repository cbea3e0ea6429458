"""Device frame stack and observation preparation (host wrappers of mz_ring_* /
mz_prepare_observation).  Counterpart of CustomFrameStack (train.py:22-63) and
prepare_observation (train.py:529-540) for B games at once."""
import torch

from . import _lib


class FrameRing:
    """Ring of the last ``depth`` uint8 frames of B games, resident on the device.

    ``reset(frames)`` fills every slot with the given frame (train.py:49-50);
    ``push(frames)`` replaces the oldest (train.py:59).  ``frames`` is a uint8
    tensor [B, H, W, 3]; a pinned host tensor is copied asynchronously on the
    current stream, so only 27 648 bytes per game cross PCIe per step."""

    def __init__(self, B: int, depth: int = 32, H: int = 96, W: int = 96, device="cuda"):
        self.B, self.depth, self.H, self.W = B, depth, H, W
        self.device = torch.device(device)
        self.frame_bytes = H * W * 3
        if self.frame_bytes % 16:
            raise ValueError("frame size must be a multiple of 16 bytes")
        self.ring = torch.zeros(B, depth, H, W, 3, dtype=torch.uint8, device=self.device)
        self.staging = torch.empty(B, H, W, 3, dtype=torch.uint8, device=self.device)
        self.newest = depth - 1
        self.rot = None  # rotating observation [B, H, W, 3*depth], see enable_rotating

    def _stage(self, frames: torch.Tensor) -> torch.Tensor:
        _lib.expect(frames, "frames", torch.uint8, (self.B, self.H, self.W, 3), cuda=False, contiguous=False)
        if frames.is_cuda:
            return frames.contiguous()
        self.staging.copy_(frames, non_blocking=True)
        return self.staging

    def enable_rotating(self, dtype=torch.bfloat16) -> None:
        """Also keep the stacked observation in rotating, tiled form [B, H, W/16, depth, 56] (per row, 16-pixel
        tile and slot: the last pixel of the tile to the left, the tile's 16 pixels, 5 values of padding; see
        mz_obs_slot_write): a push rewrites only the newest slot (3 values per pixel instead of the whole
        stack).  Call before ``reset``."""
        if dtype not in _OBS_FORMAT:
            raise ValueError("observation dtype must be bfloat16, float16 or float32")
        if self.W % 16:
            raise ValueError("the rotating observation needs an image width that is a multiple of 16")
        self.rot = torch.zeros(self.B, self.H, self.W // 16, self.depth, ROT_RUN, dtype=dtype, device=self.device)
        self.newest_dev = torch.full((1,), self.newest, dtype=torch.int32, device=self.device)   # for graph replay

    def _write_rot(self, src: torch.Tensor, fill_all: int, mask: torch.Tensor = None) -> None:
        if self.rot is not None:
            _lib.call("mz_obs_slot_write", _lib.ptr(src), _lib.ptr(self.rot), _OBS_FORMAT[self.rot.dtype],
                      self.B, self.depth, self.H, self.W, self.newest, fill_all, _lib.ptr(mask), _lib.current_stream())
            self.newest_dev.fill_(self.newest)

    def reset(self, frames: torch.Tensor) -> None:
        src = self._stage(frames)
        self.newest = self.depth - 1
        _lib.call("mz_ring_push", _lib.ptr(self.ring), _lib.ptr(src), self.B, self.depth, self.frame_bytes,
                  0, 1, None, _lib.current_stream())
        self._write_rot(src, 1)

    def push(self, frames: torch.Tensor, restarted: torch.Tensor = None) -> None:
        """Append the newest frame of every game.  ``restarted`` (bool / int [B], optional) marks
        games whose frame is the first of a NEW episode: their whole stack is refilled with it
        (the reset rule) while the other games just push - games end at different moves."""
        src = self._stage(frames)
        mask = None
        if restarted is not None:
            mask = torch.as_tensor(restarted).to(self.device, torch.int32, non_blocking=True).contiguous()
            _lib.expect(mask, "restarted", torch.int32, (self.B,))
        self.newest = (self.newest + 1) % self.depth
        _lib.call("mz_ring_push", _lib.ptr(self.ring), _lib.ptr(src), self.B, self.depth, self.frame_bytes,
                  self.newest, 0, _lib.ptr(mask), _lib.current_stream())
        self._write_rot(src, 0, mask)

    def rotating_observation(self):
        """(tensor [B, H, W/16, depth, 56], newest slot): slot s holds channels
        3*((s - newest - 1) mod depth) + {0, 1, 2} of ``observation()``."""
        if self.rot is None:
            raise ValueError("enable_rotating() was not called")
        return self.rot, self.newest

    def rotation_permutation(self, newest: int = None) -> torch.Tensor:
        """perm with observation()[..., perm[j]] == untile(rotating tensor)[..., j]."""
        newest = self.newest if newest is None else newest
        return rotation_permutation(self.depth, newest)

    def observation(self, dtype=torch.bfloat16, out: torch.Tensor = None, pairs: bool = False) -> torch.Tensor:
        """Stacked observation NHWC [B, H, W, 3*depth], oldest frame first, x / 255.  pairs: the float32
        value as (hi, lo) 16-bit pairs [B, 2, H, W, 3*depth] in ``dtype`` - bfloat16 (the bf16x3 precision's
        input) or float16 (f16x2)."""
        if pairs:
            if dtype not in (torch.bfloat16, torch.float16):
                raise ValueError("pairs are bfloat16 or float16")
            shape, mode = (self.B, 2, self.H, self.W, 3 * self.depth), 2 if dtype == torch.bfloat16 else 4
        elif dtype in _OBS_FORMAT:
            shape, mode = (self.B, self.H, self.W, 3 * self.depth), _OBS_FORMAT[dtype]
        else:
            raise ValueError("observation dtype must be bfloat16, float16 or float32")
        if out is None:
            out = torch.empty(*shape, dtype=dtype, device=self.device)
        _lib.expect(out, "out", dtype, shape)
        _lib.call("mz_ring_to_obs", _lib.ptr(self.ring), _lib.ptr(out), mode,
                  self.B, self.depth, self.H * self.W, self.newest, _lib.current_stream())
        return out


_OBS_FORMAT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 3}   # mz_ring_to_obs / mz_obs_slot_write codes
ROT_RUN = 56     # values per (row, tile, slot) run of the rotating observation: 17 pixels x 3 + 5 of padding


def untile(rot: torch.Tensor) -> torch.Tensor:
    """[B, H, W/16, D, 56] -> NHWC [B, H, W, 3*D] with the channels still in slot order (the halo pixel and
    the padding of every run dropped)."""
    B, H, T, D, _ = rot.shape
    own = rot[..., 3:51].reshape(B, H, T, D, 16, 3)
    return own.permute(0, 1, 2, 4, 3, 5).reshape(B, H, T * 16, 3 * D)


def rotation_permutation(depth: int, newest: int) -> torch.Tensor:
    j = torch.arange(3 * depth)
    return ((j // 3 - newest - 1) % depth) * 3 + j % 3


def prepare_observation(observation: torch.Tensor) -> torch.Tensor:
    """train.py:529-540 on the device: uint8 HWC or NHWC -> float32 CHW or NCHW, x / 255."""
    single = observation.dim() == 3
    x = observation.unsqueeze(0) if single else observation
    _lib.expect(x, "observation", torch.uint8, cuda=False, contiguous=False)
    x = x.cuda().contiguous()
    B, H, W, Cc = x.shape
    out = torch.empty(B, Cc, H, W, dtype=torch.float32, device=x.device)
    _lib.call("mz_prepare_observation", _lib.ptr(x), _lib.ptr(out), B, Cc, H * W, _lib.current_stream())
    return out[0] if single else out
