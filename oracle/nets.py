"""torch fp32 CPU mirror of the reference networks (test infrastructure).

Follows /root/reference/model.py:
  inverse_transform_value / support_to_scalar   model.py:89-109
  ResidualBlock / DownsampleBlock               model.py:143-175
  RepresentationNetwork                         model.py:178-204
  DynamicsNetwork                               model.py:207-245
  PredictionNetwork                             model.py:248-281
  MuZeroNetwork.initial/recurrent_inference     model.py:332-348

Sub-module names and construction order equal the reference's, so
(a) ``state_dict`` keys are interchangeable with a reference checkpoint and
(b) building both under the same ``torch.manual_seed`` yields bit-identical
parameters (checked against the golden state-dict digest).

Inference semantics: BatchNorm in eval mode (running statistics); the caller
is expected to call ``.eval()``.
"""
from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def decode_support(logits: torch.Tensor, support_size: int,
                   eps: float = 0.001) -> torch.Tensor:
    """softmax -> expectation over [-S..S] -> inverse of the invertible value
    transform, all in the dtype of ``logits`` (model.py:89-109)."""
    p = torch.softmax(logits, dim=1)
    bins = torch.arange(-support_size, support_size + 1,
                        device=logits.device, dtype=logits.dtype)
    x = torch.sum(bins * p, dim=1)
    return torch.sign(x) * (
        ((torch.sqrt(1 + 4 * eps * (torch.abs(x) + 1 + eps)) - 1) / (2 * eps)) ** 2 - 1)


class Res(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv1 = nn.Conv2d(c, c, 3, 1, 1)
        self.bn1 = nn.BatchNorm2d(c)
        self.conv2 = nn.Conv2d(c, c, 3, 1, 1)
        self.bn2 = nn.BatchNorm2d(c)

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        return F.relu(self.bn2(self.conv2(y)) + x)


class Down(nn.Module):
    def __init__(self, cin: int, cout: int, n: int):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3, 2, 1)
        self.bn = nn.BatchNorm2d(cout)
        self.blocks = nn.ModuleList([Res(cout) for _ in range(n)])

    def forward(self, x):
        x = F.relu(self.bn(self.conv(x)))
        for b in self.blocks:
            x = b(x)
        return x


class Representation(nn.Module):
    TAIL_BLOCKS = 6  # model.py:191 hard-codes six trailing blocks

    def __init__(self, cin: int, width: int,
                 steps: Sequence[Tuple[int, int]]):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, width, 3, 2, 1)
        self.bn1 = nn.BatchNorm2d(width)
        stages, c = [], width
        for cout, n in steps:
            stages.append(Down(c, cout, n))
            c = cout
        self.downsample = nn.Sequential(*stages)
        self.resblocks = nn.ModuleList([Res(c) for _ in range(self.TAIL_BLOCKS)])

    def forward(self, x):
        x = self.downsample(F.relu(self.bn1(self.conv1(x))))
        for b in self.resblocks:
            x = b(x)
        return x


def _mlp(cin: int, hidden: Sequence[int], cout: int) -> nn.Sequential:
    layers: List[nn.Module] = []
    for h in hidden:
        layers += [nn.Linear(cin, h), nn.ReLU()]
        cin = h
    layers.append(nn.Linear(cin, cout))
    return nn.Sequential(*layers)


class Dynamics(nn.Module):
    def __init__(self, width: int, n_blocks: int, n_actions: int,
                 fc_reward: Sequence[int], support_size: int):
        super().__init__()
        self.n_actions = n_actions
        self.conv_action = nn.Conv2d(n_actions, width, 1)
        self.conv_state = nn.Conv2d(width, width, 3, 1, 1)
        self.bn = nn.BatchNorm2d(width)
        self.resblocks = nn.ModuleList([Res(width) for _ in range(n_blocks)])
        self.reward_head = _mlp(width, fc_reward, 2 * support_size + 1)

    def forward(self, h, action):
        # one-hot action planes through a 1x1 conv (model.py:231-236)
        b, _, hh, ww = h.shape
        planes = F.one_hot(action.view(b), self.n_actions).to(h.dtype)
        planes = planes.view(b, self.n_actions, 1, 1).expand(-1, -1, hh, ww)
        x = h + self.conv_action(planes)
        x = F.relu(self.bn(self.conv_state(x)))
        for blk in self.resblocks:
            x = blk(x)
        return x, self.reward_head(x.mean(dim=(2, 3)))


class Prediction(nn.Module):
    def __init__(self, width: int, n_actions: int, fc_policy: Sequence[int],
                 fc_value: Sequence[int], support_size: int):
        super().__init__()
        self.policy_head = _mlp(width, fc_policy, n_actions)
        self.value_head = _mlp(width, fc_value, 2 * support_size + 1)

    def forward(self, h):
        m = h.mean(dim=(2, 3))
        return self.policy_head(m), self.value_head(m)


class Net(nn.Module):
    """Mirror of MuZeroNetwork (model.py:307-348).  ``cfg`` needs the
    reference's attribute names; input channels are ``observation_shape[0]``
    as in model.py:181."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.representation = Representation(
            cfg.observation_shape[0], cfg.num_channels, cfg.downsample_steps)
        self.dynamics = Dynamics(cfg.num_channels, cfg.num_blocks,
                                 cfg.action_space_size, cfg.fc_reward_layers,
                                 cfg.support_size)
        self.prediction = Prediction(cfg.num_channels, cfg.action_space_size,
                                     cfg.fc_policy_layers, cfg.fc_value_layers,
                                     cfg.support_size)

    def initial(self, obs):
        """-> (hidden, policy_logits, value_logits)"""
        h = self.representation(obs)
        p, v = self.prediction(h)
        return h, p, v

    def recurrent(self, hidden, action):
        """-> (hidden', reward_logits, policy_logits, value_logits)"""
        h, r = self.dynamics(hidden, action)
        p, v = self.prediction(h)
        return h, r, p, v


def make_infer(net: Net):
    """Adapter from ``Net`` to the ``infer`` callable of ``oracle.tree``: one
    batched recurrent inference, then per-row softmax / support decode on
    ``[i:i+1]`` slices exactly as model.py:461-467 does."""
    s = net.cfg.support_size

    def infer(hiddens, actions):
        with torch.no_grad():
            h = torch.cat(hiddens, dim=0)
            a = torch.tensor(actions, dtype=torch.long).unsqueeze(1)
            h2, r, p, v = net.recurrent(h, a)
            rows = []
            for i in range(h2.shape[0]):
                pri = torch.softmax(p[i:i + 1], dim=1).numpy()[0]
                rew = decode_support(r[i:i + 1], r.shape[-1] // 2).item()
                val = decode_support(v[i:i + 1], s).item()
                rows.append((pri, val, rew, h2[i:i + 1]))
        return rows

    return infer


def randomize_bn(net: nn.Module, seed: int = 1) -> None:
    """Give every BatchNorm non-trivial running statistics and affine
    parameters so that folding them into the convolutions is exercised."""
    g = torch.Generator().manual_seed(seed)
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            n = m.num_features
            m.running_mean.copy_(torch.randn(n, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(n, generator=g) + 0.5)
            with torch.no_grad():
                m.weight.copy_(torch.rand(n, generator=g) * 0.5 + 0.75)
                m.bias.copy_(torch.randn(n, generator=g) * 0.1)


def state_digest(net: nn.Module) -> str:
    """SHA-256 over the raw bytes of every state_dict tensor in key order."""
    import hashlib
    hsh = hashlib.sha256()
    for k, v in net.state_dict().items():
        hsh.update(k.encode())
        hsh.update(np.ascontiguousarray(v.detach().cpu().numpy()).tobytes())
    return hsh.hexdigest()
