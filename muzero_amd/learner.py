"""The training side of the loop: episode records and n-step targets, the replay buffer, and
the K-step unrolled training step that produces the weights the search path consumes.

Counterparts in the reference:
  GameHistory          Game                        train.py:66-129
  ReplayBuffer         ReplayBuffer                model.py:549-618, compute_priorities train.py:542-550
  Learner              compute_losses              train.py:330-431
                       update_weights              train.py:295-328 (SGD + momentum + weight decay,
                                                   StepLR, gradient clipping; train.py:156-166, :309-313)
  support_ce_loss      scalar_to_support_loss      model.py:132-140

Division of labour: the learner is the one place where gradients are needed.  It keeps the
network as torch modules (same parameter names as the reference, BatchNorm in training mode
as train.py:155 leaves it) and lets torch autograd differentiate the convolutions; the
support cross-entropy (forward and gradient) and the n-step targets are HIP kernels
(``mz_support_ce``, ``mz_nstep_targets``).  After an update ``Learner.state_dict()`` is
broadcast (``muzero_amd.dist.broadcast_state_dict``) and loaded into the inference engine
(``MuZeroNetwork.load_state_dict``), which folds BatchNorm for the search.
"""
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, checkpoint


# --------------------------------------------------------------------------- modules
class _Res(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv1 = nn.Conv2d(c, c, 3, 1, 1)
        self.bn1 = nn.BatchNorm2d(c)
        self.conv2 = nn.Conv2d(c, c, 3, 1, 1)
        self.bn2 = nn.BatchNorm2d(c)

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        return F.relu(self.bn2(self.conv2(y)) + x)


class _Down(nn.Module):
    def __init__(self, cin, cout, n):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3, 2, 1)
        self.bn = nn.BatchNorm2d(cout)
        self.blocks = nn.ModuleList([_Res(cout) for _ in range(n)])

    def forward(self, x):
        x = F.relu(self.bn(self.conv(x)))
        for b in self.blocks:
            x = b(x)
        return x


def _mlp(cin, hidden, cout):
    layers = []
    for h in hidden:
        layers += [nn.Linear(cin, h), nn.ReLU()]
        cin = h
    layers.append(nn.Linear(cin, cout))
    return nn.Sequential(*layers)


class _Representation(nn.Module):
    def __init__(self, cin, width, steps):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, width, 3, 2, 1)
        self.bn1 = nn.BatchNorm2d(width)
        stages, c = [], width
        for cout, n in steps:
            stages.append(_Down(c, cout, n))
            c = cout
        self.downsample = nn.Sequential(*stages)
        self.resblocks = nn.ModuleList([_Res(c) for _ in range(6)])

    def forward(self, x):
        x = self.downsample(F.relu(self.bn1(self.conv1(x))))
        for b in self.resblocks:
            x = b(x)
        return x


class _Dynamics(nn.Module):
    def __init__(self, width, n_blocks, n_actions, fc_reward, support):
        super().__init__()
        self.n_actions = n_actions
        self.conv_action = nn.Conv2d(n_actions, width, 1)
        self.conv_state = nn.Conv2d(width, width, 3, 1, 1)
        self.bn = nn.BatchNorm2d(width)
        self.resblocks = nn.ModuleList([_Res(width) for _ in range(n_blocks)])
        self.reward_head = _mlp(width, fc_reward, 2 * support + 1)

    def forward(self, h, action):
        # hidden + conv1x1(one-hot planes) == hidden + W[:, action] + bias on every pixel
        w = self.conv_action.weight[:, :, 0, 0]                      # [width][A]
        x = h + (w.t()[action.view(-1)] + self.conv_action.bias)[:, :, None, None]
        x = F.relu(self.bn(self.conv_state(x)))
        for b in self.resblocks:
            x = b(x)
        return x, self.reward_head(x.mean(dim=(2, 3)))


class _Prediction(nn.Module):
    def __init__(self, width, n_actions, fc_policy, fc_value, support):
        super().__init__()
        self.policy_head = _mlp(width, fc_policy, n_actions)
        self.value_head = _mlp(width, fc_value, 2 * support + 1)

    def forward(self, h):
        m = h.mean(dim=(2, 3))
        return self.policy_head(m), self.value_head(m)


class TrainableNetwork(nn.Module):
    """The network with gradients; ``state_dict`` keys equal the reference's."""

    def __init__(self, cfg):
        super().__init__()
        self.representation = _Representation(cfg.observation_shape[0], cfg.num_channels, cfg.downsample_steps)
        self.dynamics = _Dynamics(cfg.num_channels, cfg.num_blocks, cfg.action_space_size, cfg.fc_reward_layers,
                                  cfg.support_size)
        self.prediction = _Prediction(cfg.num_channels, cfg.action_space_size, cfg.fc_policy_layers,
                                      cfg.fc_value_layers, cfg.support_size)

    def initial(self, obs):
        h = self.representation(obs)
        p, v = self.prediction(h)
        return h, p, v

    def recurrent(self, h, action):
        h2, r = self.dynamics(h, action)
        p, v = self.prediction(h2)
        return h2, r, p, v


# --------------------------------------------------------------------------- loss
def _support_ce_torch(logits, targets, support_size: int):
    """The reference's own sequence of torch ops (model.py:84-86, :112-140)."""
    t = torch.sign(targets) * (torch.sqrt(torch.abs(targets) + 1) - 1 + 0.001 * targets)
    t = torch.clamp(t, -support_size, support_size)
    fl = t.floor()
    up_p = t - fl
    lo_p = 1 - up_p
    sup = torch.zeros(t.size(0), 2 * support_size + 1, device=t.device)
    lo = (fl + support_size).long()
    up = (lo + 1).clamp_(0, 2 * support_size)
    lo = lo.clamp_(0, 2 * support_size)
    sup.scatter_(1, lo.unsqueeze(1), lo_p.unsqueeze(1))
    sup.scatter_(1, up.unsqueeze(1), up_p.unsqueeze(1))
    return -(sup * F.log_softmax(logits, dim=1)).sum(1)


class _SupportCEHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, support_size):
        x = logits.contiguous()
        t = targets.contiguous().float()
        loss = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        dlog = torch.empty_like(x)
        _lib.call("mz_support_ce", _lib.ptr(x), _lib.ptr(t), x.shape[0], support_size, _lib.ptr(loss),
                  _lib.ptr(dlog), _lib.current_stream())
        ctx.save_for_backward(dlog)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        (dlog,) = ctx.saved_tensors
        return dlog * grad_out.unsqueeze(1), None, None


def support_ce_loss(logits, targets, support_size: int, impl: str = "hip"):
    """-(two_hot(transform(target)) * log_softmax(logits)).sum(1).  impl "hip": fused kernel
    (device tensors only); "torch": the reference's op sequence."""
    if impl == "hip":
        if not logits.is_cuda:
            raise _lib.MuZeroAmdError("support_ce_loss(impl='hip') needs device tensors; pass impl='torch' on the CPU")
        return _SupportCEHip.apply(logits, targets, support_size)
    return _support_ce_torch(logits, targets, support_size)


# --------------------------------------------------------------------------- episodes and targets
class GameHistory:
    """One episode (train.py:66-129).  Observations are kept as the single new frame per
    step; the stacked observation of a position is rebuilt from the 32 frames before it."""

    def __init__(self, action_space_size: int, discount: float, frame_stack: int = 32, frame_channels: int = 3):
        self.action_space_size = action_space_size
        self.discount = discount
        self.frame_stack = frame_stack
        self.frame_channels = frame_channels
        self.done = False
        self.actions: List[int] = []
        self.frames: List[np.ndarray] = []
        self.rewards: List[float] = []
        self.values: List[float] = []
        self.policies: List[np.ndarray] = []
        self.environment_steps = 0

    def store_search_stats(self, root, action_probs: Dict[int, float]) -> None:
        self.values.append(root.value())
        policy = np.zeros(self.action_space_size)
        for a, p in action_probs.items():
            policy[a] = p
        self.policies.append(policy)

    def store_transition(self, action: Optional[int], observation: np.ndarray, reward: float) -> None:
        """``observation`` is either the new frame or the whole stacked observation the
        reference passes (train.py:246, :281); of a stack only the newest frame is kept."""
        self.actions.append(action if action is not None else 0)   # dummy action 0 at index 0 (train.py:123)
        if observation.ndim == 3 and observation.shape[2] == self.frame_channels * self.frame_stack:
            observation = np.ascontiguousarray(observation[:, :, -self.frame_channels:])
        self.frames.append(observation)
        self.rewards.append(reward)
        self.environment_steps += 1

    def get_observation(self, index: int) -> np.ndarray:
        """(H, W, 3*frame_stack) uint8, oldest first; the first frame pads the start (train.py:49-50)."""
        idx = [max(0, index - self.frame_stack + 1 + k) for k in range(self.frame_stack)]
        return np.concatenate([self.frames[i] for i in idx], axis=2)

    def compute_target_value(self, state_index: int, td_steps: int) -> float:
        value, discount = 0.0, 1.0
        for i in range(td_steps):
            idx = state_index + i
            if idx >= len(self.rewards):
                break
            value += self.rewards[idx] * discount
            discount *= self.discount
        if state_index + td_steps < len(self.values):
            value += self.values[state_index + td_steps] * discount
        return value

    def make_target(self, state_index: int, num_unroll_steps: int, td_steps: int):
        out = []
        for cur in range(state_index, state_index + num_unroll_steps + 1):
            v = self.compute_target_value(cur, td_steps)
            if cur < len(self.rewards):
                out.append((v, self.rewards[cur], self.policies[cur]))
            else:
                out.append((v, 0, np.zeros(self.action_space_size)))
        return out


def batched_targets(games: Sequence[GameHistory], queries: Sequence[Tuple[int, int]], K: int, td_steps: int,
                    device="cuda"):
    """make_target for many (game, position) pairs in one HIP launch.  Returns float64 tensors
    target_values [n, K+1], target_rewards [n, K+1] and float64 target_policies [n, K+1, A]."""
    G = len(games)
    t_max = max(max(len(g.rewards), len(g.values)) for g in games)
    rew = np.zeros((G, t_max)); val = np.zeros((G, t_max))
    nr = np.zeros(G, np.int32); nv = np.zeros(G, np.int32)
    for i, g in enumerate(games):
        nr[i], nv[i] = len(g.rewards), len(g.values)
        rew[i, :nr[i]] = g.rewards
        val[i, :nv[i]] = g.values
    q = np.asarray(queries, np.int32).reshape(-1, 2)
    n = q.shape[0]
    dev = torch.device(device)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_rew, d_val, d_nr, d_nv, d_g, d_p = t(rew), t(val), t(nr), t(nv), t(q[:, 0].copy()), t(q[:, 1].copy())
    tv = torch.empty(n, K + 1, dtype=torch.float64, device=dev)
    tr = torch.empty(n, K + 1, dtype=torch.float64, device=dev)
    pi = torch.empty(n, K + 1, dtype=torch.int32, device=dev)
    _lib.call("mz_nstep_targets", _lib.ptr(d_rew), _lib.ptr(d_val), _lib.ptr(d_nr), _lib.ptr(d_nv), t_max,
              _lib.ptr(d_g), _lib.ptr(d_p), n, K, td_steps, float(games[0].discount), _lib.ptr(tv), _lib.ptr(tr),
              _lib.ptr(pi), _lib.current_stream())
    A = games[0].action_space_size
    idx = pi.cpu().numpy()
    pol = np.zeros((n, K + 1, A))
    for i in range(n):
        g = games[q[i, 0]]
        for k in range(K + 1):
            if idx[i, k] >= 0:
                pol[i, k] = g.policies[idx[i, k]]
    return tv, tr, torch.from_numpy(pol).to(dev)


class ReplayBuffer:
    """Prioritised replay over positions (model.py:549-618), host side.  One departure from
    the reference, on purpose: ``sample_batch`` returns the FLAT position index of every
    sample, so ``update_priorities`` writes the slots that were sampled (the reference keeps
    the index after it was reduced to an in-game offset, model.py:595 / :607, and so updates
    the wrong slots)."""

    def __init__(self, config):
        self.window_size = config.replay_buffer_size
        self.batch_size = config.batch_size
        self.alpha = config.priority_alpha
        self.beta = config.priority_beta
        self.buffer: List[GameHistory] = []
        self.priorities: List[float] = []

    def __len__(self):
        return len(self.buffer)

    def save_game(self, game: GameHistory, priorities: Optional[np.ndarray] = None) -> None:
        if priorities is None:
            priorities = np.full(len(game.rewards), max(self.priorities) if self.priorities else 1)
        if len(priorities) != len(game.rewards):
            raise ValueError(f"{len(priorities)} priorities for a game of {len(game.rewards)} positions: the buffer "
                             "keeps exactly one per position (sampling maps flat indices through game lengths)")
        self.buffer.append(game)
        self.priorities.extend(priorities)
        while len(self.buffer) > self.window_size:
            old = self.buffer.pop(0)
            self.priorities = self.priorities[len(old.rewards):]

    def sample_positions(self, uniforms: Optional[np.ndarray] = None):
        """-> (game index, in-game position, flat index) arrays and importance weights.
        ``uniforms`` are the [0,1) draws np.random.choice would consume (global generator if None)."""
        total = sum(self.priorities)
        n = len(self.priorities)
        probs = np.array([1 / n] * n if total == 0 else [p / total for p in self.priorities])
        u = np.random.random_sample(self.batch_size) if uniforms is None else np.asarray(uniforms)
        cdf = np.cumsum(probs)
        cdf /= cdf[-1]
        flat = np.searchsorted(cdf, u, side="right")
        weights = (n * probs[flat]) ** -self.beta
        weights = weights / weights.max()
        starts = np.cumsum([0] + [len(g.rewards) for g in self.buffer])
        gi = np.searchsorted(starts, flat, side="right") - 1
        return gi, flat - starts[gi], flat, weights

    def update_priorities(self, flat_indices, priorities) -> None:
        for idx, p in zip(flat_indices, priorities):
            self.priorities[int(idx)] = (float(p) + 1e-6) ** self.alpha


def compute_priorities(game: GameHistory, td_steps: int) -> np.ndarray:
    """|search value - n-step target| + 1e-6 per position (train.py:542-550)."""
    return np.array([abs(game.values[i] - game.compute_target_value(i, td_steps)) + 1e-6
                     for i in range(len(game.values))])


# --------------------------------------------------------------------------- the training step
def unroll_actions(action_lists, K: int) -> np.ndarray:
    """[n, K] actions in the order the unroll consumes them.  The reference feeds the dynamics
    network at unroll step t the action list's entry t-1, and action 0 at step 0 and past the
    end of the list (train.py:370-373 builds the tensor, :410-413 uses it in the same
    iteration)."""
    out = np.zeros((len(action_lists), max(K, 1)), np.int64)
    for i, acts in enumerate(action_lists):
        for t in range(1, K):
            if t - 1 < len(acts):
                out[i, t] = acts[t - 1]
    return out[:, :K]


class Learner:
    def __init__(self, config, device="cuda", state_dict=None, loss_impl: Optional[str] = None):
        self.config = config
        self.device = torch.device(device)
        self.loss_impl = loss_impl or ("hip" if self.device.type == "cuda" else "torch")
        self.network = TrainableNetwork(config).to(self.device)      # stays in training mode (train.py:155)
        if state_dict is not None:
            self.network.load_state_dict(state_dict)
        self.optimizer = torch.optim.SGD(self.network.parameters(), lr=config.lr_init, momentum=config.momentum,
                                         weight_decay=config.weight_decay)
        self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=config.lr_decay_steps,
                                                         gamma=config.lr_decay_rate)
        self.training_step = 0

    def state_dict(self):
        return {k: v.detach().cpu() for k, v in self.network.state_dict().items()}

    def save_checkpoint(self, path: str) -> None:
        """The reference's checkpoint dictionary (train.py:481-495)."""
        checkpoint.write(path, self.training_step, self.network, self.optimizer, self.scheduler, self.config)

    @classmethod
    def from_checkpoint(cls, path: str, device="cuda", loss_impl: Optional[str] = None) -> "Learner":
        """Resume from a checkpoint written by the reference or by ``save_checkpoint``:
        weights, BatchNorm statistics, momentum buffers, learning-rate schedule and step."""
        ck = checkpoint.read(path)
        lrn = cls(checkpoint.config_from(ck), device=device, state_dict=ck["model_state_dict"], loss_impl=loss_impl)
        if ck.get("optimizer_state_dict") is not None:
            lrn.optimizer.load_state_dict(ck["optimizer_state_dict"])
        if ck.get("scheduler_state_dict") is not None:
            lrn.scheduler.load_state_dict(ck["scheduler_state_dict"])
        lrn.training_step = int(ck.get("training_step", 0))
        return lrn

    def compute_losses(self, batch, weights):
        """The reference's signature (train.py:330-333): ``batch`` is a list of (stacked uint8
        observation HWC, action list, [(value, reward, policy)] * (K+1)) as ``sample_batch``
        returns it."""
        K, A = self.config.num_unroll_steps, self.config.action_space_size
        n = len(batch)
        obs = torch.stack([torch.from_numpy(np.ascontiguousarray(o)) for o, _, _ in batch]).to(self.device)
        obs = obs.permute(0, 3, 1, 2).float() / 255.0                       # prepare_observation, train.py:529-540
        tv = np.zeros((n, K + 1)); tr = np.zeros((n, K + 1)); tp = np.zeros((n, K + 1, A))
        for i, (_, _, targets) in enumerate(batch):
            for t, (v, r, p) in enumerate(targets[:K + 1]):                 # missing entries stay 0 (train.py:363-368)
                tv[i, t], tr[i, t], tp[i, t] = v, r, p
        unroll = unroll_actions([a for _, a, _ in batch], K)
        return self.losses_from_tensors(obs, unroll, tv, tr, tp, weights)

    def losses_from_tensors(self, observations, actions, target_values, target_rewards, target_policies, weights):
        """train.py:343-431.  observations: float32 [B, C, H, W] (already prepared); actions
        int64 [B, K] in unroll order (see ``unroll_actions``); target_* [B, K+1(, A)];
        weights [B].  Returns (value_loss, reward_loss, policy_loss, per-sample loss) with the
        reference's averaging conventions."""
        cfg = self.config
        K, S = cfg.num_unroll_steps, cfg.support_size
        dev = self.device
        f32 = lambda x: torch.as_tensor(x, device=dev).float()
        tv, tr, tp, w = f32(target_values), f32(target_rewards), f32(target_policies), f32(weights)
        act = torch.as_tensor(actions, device=dev).long()
        hidden, policy_logits, value_logits = self.network.initial(observations.to(dev))
        reward_logits = None
        total_v = total_r = total_p = 0.0
        per_sample = 0.0
        for t in range(K + 1):
            v_loss = support_ce_loss(value_logits, tv[:, t], S, self.loss_impl)
            p_loss = -(tp[:, t] * torch.log_softmax(policy_logits, dim=1)).sum(1)
            r_loss = support_ce_loss(reward_logits, tr[:, t], S, self.loss_impl) if t > 0 \
                else torch.zeros(tv.shape[0], device=dev)
            total = cfg.value_loss_weight * v_loss + cfg.reward_loss_weight * r_loss + cfg.policy_loss_weight * p_loss
            per_sample = per_sample + total * w
            if t < K:
                hidden, reward_logits, policy_logits, value_logits = self.network.recurrent(hidden, act[:, t])
            total_v = total_v + v_loss.mean()
            total_r = total_r + r_loss.mean()
            total_p = total_p + p_loss.mean()
        total_v = total_v / (K + 1)
        total_p = total_p / (K + 1)
        total_r = total_r / K if K > 0 else torch.zeros(1, device=dev)
        return total_v, total_r, total_p, per_sample / (K + 1)

    def update_weights(self, batch, weights):
        """One optimisation step (train.py:303-313).  Returns the metrics and the per-sample
        losses that become the new replay priorities (train.py:318-319)."""
        v, r, p, per = self.compute_losses(batch, weights)
        loss = per.mean()
        self.optimizer.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.network.parameters(), self.config.max_grad_norm)
        self.optimizer.step()
        self.scheduler.step()
        self.training_step += 1
        return {"value_loss": v.item(), "reward_loss": r.item(), "policy_loss": p.item(), "total_loss": loss.item(),
                "priorities": per.detach().cpu().numpy()}
