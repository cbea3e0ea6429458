"""Checkpoint files in the reference's format (train.py:481-495): one ``torch.save`` of
``{'training_step', 'model_state_dict', 'optimizer_state_dict', 'scheduler_state_dict',
'config'}`` where ``config`` is the configuration object's ``__dict__``.

Files are read with ``torch.load(..., weights_only=True)`` only: nothing in a checkpoint is
executed, and a file that needs more than tensors and plain containers is refused.
"""
from typing import Any, Dict

import torch

from .config import MuZeroConfig

KEYS = ("training_step", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "config")


def read(path: str) -> Dict[str, Any]:
    ck = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(ck, dict) or "model_state_dict" not in ck:
        raise ValueError(f"{path}: not a MuZero checkpoint (no 'model_state_dict')")
    return ck


def config_from(ck: Dict[str, Any]) -> MuZeroConfig:
    """A configuration object carrying every field the checkpoint recorded."""
    cfg = MuZeroConfig()
    for k, v in (ck.get("config") or {}).items():
        setattr(cfg, k, v)
    cfg.downsample_steps = [tuple(s) for s in cfg.downsample_steps]
    cfg.observation_shape = tuple(cfg.observation_shape)
    return cfg


def load_network(path: str, precision: str = "bf16", device="cuda", input_hw=(96, 96)):
    """The inference engine for the search, from a checkpoint the reference (or ``Learner``)
    wrote.  BatchNorm is folded with the stored running statistics."""
    from .network import MuZeroNetwork
    ck = read(path)
    return MuZeroNetwork(config_from(ck), precision=precision, device=device, state_dict=ck["model_state_dict"],
                         input_hw=input_hw)


def write(path: str, training_step: int, network, optimizer, scheduler, config) -> None:
    cfg = {k: v for k, v in vars(config).items() if not callable(v)}
    torch.save({"training_step": int(training_step),
                "model_state_dict": {k: v.detach().cpu() for k, v in network.state_dict().items()},
                "optimizer_state_dict": optimizer.state_dict(),
                "scheduler_state_dict": scheduler.state_dict(),
                "config": cfg}, path)
