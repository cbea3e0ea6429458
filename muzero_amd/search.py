"""Batched MCTS over B concurrent games on the device, plus the reference's
single-root call surface (Node / MinMaxStats / MCTS.run / select_action,
model.py:351-546) implemented on the same kernels.

``BatchedMCTS.run`` is the batched counterpart of the serial per-game loop of
train.py:238-277: one initial inference for all games, then ``num_simulations /
wave`` steps of {select -> recurrent inference -> expand + backup} with no host
synchronisation in between, then a read-out of visit counts, root values and
action probabilities.

``wave`` is the number of simulations per tree update.  The reference's
``MCTS.run`` performs all its ``num_simulations`` descents before any backup
(model.py:433-447), i.e. it is one wave of that width; ``wave=1`` is the
canonical one-simulation-at-a-time search (what calling the reference's ``run``
with ``num_simulations = 1`` repeatedly on one root does).
"""
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from .network import MuZeroNetwork, NetworkOutput
from .tree import DeviceTrees, select_actions


class SearchResult:
    def __init__(self, visit_counts, root_values, action_probs):
        self.visit_counts = visit_counts    # int32  [B, A]
        self.root_values = root_values      # float64 [B]
        self.action_probs = action_probs    # float64 [B, A]

    def roots(self) -> List["RootView"]:
        v = self.visit_counts.cpu().numpy()
        r = self.root_values.cpu().numpy()
        return [RootView(v[i], float(r[i])) for i in range(v.shape[0])]

    def action_prob_dicts(self) -> List[Dict[int, float]]:
        p = self.action_probs.cpu().numpy()
        return [{a: float(x) for a, x in enumerate(row)} for row in p]


class _Child:
    __slots__ = ("visit_count",)

    def __init__(self, n):
        self.visit_count = int(n)


class RootView:
    """What the consumers of a searched root read: ``children[a].visit_count``
    (select_action, model.py:535) and ``value()`` (Game.store_search_stats,
    train.py:83)."""

    def __init__(self, visits, value: float):
        self.children = {a: _Child(n) for a, n in enumerate(visits)}
        self._value = value

    def value(self) -> float:
        return self._value


class BatchedMCTS:
    def __init__(self, config, network: MuZeroNetwork, batch_size: int, num_simulations: Optional[int] = None,
                 wave: int = 1, obs_hw=(96, 96)):
        self.config = config
        self.net = network
        self.B = batch_size
        self.sims = num_simulations if num_simulations is not None else config.num_simulations
        self.wave = wave
        if self.sims % wave:
            raise ValueError("num_simulations must be a multiple of wave")
        self.steps = self.sims // wave
        self.A = config.action_space_size
        dev = network.device
        self.trees = DeviceTrees(batch_size, self.A, self.steps + 1, dev, config.pb_c_base, config.pb_c_init,
                                 max_visits=self.sims)
        h, w = network.hidden_hw(*obs_hw)
        self.pool = torch.empty(self.steps + 1, batch_size, *network.hidden_shape(h, w), dtype=network.act_dtype,
                                device=dev)
        self.visits = torch.empty(batch_size, self.A, dtype=torch.int32, device=dev)
        self.root_values = torch.empty(batch_size, dtype=torch.float64, device=dev)
        self.probs = torch.empty(batch_size, self.A, dtype=torch.float64, device=dev)
        self.leaf_parent = self.trees.array("leaf_parent")
        self.leaf_action = self.trees.array("leaf_action")
        self.fuse_tree_step = True  # False: separate expand_backup and select launches (same results)
        self.tree_events = None     # set to a list to time every tree launch of the search loop with HIP events

    def _search(self, observations: torch.Tensor, noise: Optional[torch.Tensor], rotation: Optional[int] = None) -> None:
        cfg = self.config
        priors, value = self.net.initial_to_pool(observations, self.pool[0], rotation)
        self.root_priors, self.root_net_value = priors, value
        self.trees.root_init(priors, noise, cfg.root_exploration_fraction)
        fuse = self.fuse_tree_step
        self.trees.select()
        for s in range(self.steps):
            priors, value, reward = self.net.recurrent_in_pool(self.pool, self.leaf_parent, self.B,
                                                                self.leaf_action, self.pool[s + 1])
            # the descent of the next simulation rides in the same launch as this one's backup
            step = self.trees.expand_backup_select if fuse and s + 1 < self.steps else self.trees.expand_backup
            timing = self.tree_events
            if timing is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            step(priors, value, reward, self.wave, cfg.discount)
            if timing is not None:
                e1.record()
                timing.append((e0, e1))
            if not fuse and s + 1 < self.steps:
                self.trees.select()
        self.trees.readout(self.visits, self.root_values, self.probs)

    def run(self, observations: torch.Tensor, noise: Optional[torch.Tensor] = None,
            rotation: Optional[int] = None) -> SearchResult:
        """observations: NHWC [B, H, W, C] in the network's activation dtype (e.g. from
        FrameRing.observation).  noise: float64 [B, A] Dirichlet rows, or None for the
        noise-free search of evaluate() (train.py:456-460).  rotation: newest slot when
        ``observations`` is FrameRing.rotating_observation().  The returned tensors are
        reused by the next call."""
        self._search(observations, noise, rotation)
        return SearchResult(self.visits, self.root_values, self.probs)

    def check_status(self) -> None:
        """Raise if any tree reported trouble during the last search (one device read-back;
        call it outside timed or captured regions)."""
        st = self.trees.array("status")
        if bool((st != 0).any()):
            codes = {1: "no selectable child", 2: "tree full", 3: "score tables shorter than a visit count"}
            bad = int((st != 0).nonzero()[0].item())
            code = int(st[bad].item())
            raise _lib.MuZeroAmdError(f"search: game {bad} reported status {code} ({codes.get(code, '?')})")

    def capture(self, observations: torch.Tensor, noise: Optional[torch.Tensor] = None,
                rotation: Optional[torch.Tensor] = None):
        """Record one whole search (initial inference, every simulation step, read-out)
        into a HIP graph reading ``observations`` / ``noise`` in place.  Returns a
        ``replay()`` callable: refill the two tensors, call it, read the result
        tensors.  Worth it when B is small and the per-kernel launch cost shows.
        rotation: for an observation in rotating form, the device int32 [1] that holds the newest
        slot (``FrameRing.newest_dev``): the first convolution picks its weight copy from it when the
        graph runs.  The graph holds the addresses of the network's device weight buffers.
        ``network.load_state_dict`` refills those buffers in place, so a later ``replay()`` searches with the
        new weights; if a load had to allocate new buffers (other shapes), ``replay()`` raises instead of
        reading freed memory - capture again."""
        if rotation is not None and not torch.is_tensor(rotation):
            raise ValueError("capture(): pass the rotation as the device tensor FrameRing.newest_dev")
        self._search(observations, noise, rotation)  # warm-up: sizes every scratch buffer
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._search(observations, noise, rotation)
        result = SearchResult(self.visits, self.root_values, self.probs)
        generation = self.net.generation

        def replay() -> SearchResult:
            if self.net.generation != generation:
                raise _lib.MuZeroAmdError("the network allocated new weight buffers after this search was captured: "
                                          "the graph points at the old ones; call capture() again")
            graph.replay()
            return result
        return replay


def policy_softmax(logits: torch.Tensor) -> torch.Tensor:
    """softmax over actions with the heads' own device code (model.py:397)."""
    x = logits.detach().to("cuda", torch.float32).contiguous()
    out = torch.empty_like(x)
    _lib.call("mz_policy_softmax", _lib.ptr(x), _lib.ptr(out), x.shape[0], x.shape[1], _lib.current_stream())
    return out


# --------------------------------------------------------------------------
# The reference's single-root surface
# --------------------------------------------------------------------------
class MinMaxStats:
    """model.py:351-370.  MCTS.run mirrors the device statistics into this object."""

    def __init__(self):
        self.maximum = -float("inf")
        self.minimum = float("inf")

    def update(self, value: float) -> None:
        self.maximum = max(self.maximum, value)
        self.minimum = min(self.minimum, value)

    def normalize(self, value: float) -> float:
        if self.maximum > self.minimum:
            return (value - self.minimum) / (self.maximum - self.minimum)
        return 0.5


class Node:
    """Host handle of a search node with the reference's attributes
    (model.py:373-416).  The tree below a root lives on the device; after
    ``MCTS.run`` the root and its children carry the visit counts and value sums
    the reference's callers read."""

    def __init__(self, prior, hidden_state=None):
        self.visit_count = 0
        self.prior = prior
        self.value_sum = 0.0
        self.children: Dict[int, "Node"] = {}
        self.hidden_state = hidden_state
        self.reward = 0.0
        self.expanded_flag = False
        self._device = None  # (DeviceTrees, pool, steps_done) once searched

    def is_expanded(self) -> bool:
        return self.expanded_flag

    def value(self) -> float:
        return self.value_sum / self.visit_count if self.visit_count else 0.0

    def expand(self, actions: List[int], network_output: NetworkOutput) -> None:
        self.expanded_flag = True
        policy = policy_softmax(network_output.policy_logits).cpu().numpy()[0]
        self.hidden_state = network_output.hidden_state
        self.reward = network_output.scalar_reward() if network_output.reward is not None else 0
        self.children = {a: Node(policy[a], None) for a in actions}
        self._device = None

    def add_exploration_noise(self, dirichlet_alpha: float, exploration_fraction: float) -> None:
        actions = list(self.children.keys())
        noise = np.random.dirichlet([dirichlet_alpha] * len(actions))  # the reference's global generator
        keep = np.float32(1.0 - exploration_fraction)
        for a, n in zip(actions, noise):
            p = self.children[a].prior
            kept = float(np.float32(p * keep)) if isinstance(p, np.float32) else p * (1.0 - exploration_fraction)
            self.children[a].prior = np.float64(kept + float(n) * exploration_fraction)


class MCTS:
    """model.py:419-480 for one root.  Each ``run`` call is one wave of
    ``config.num_simulations`` simulations on the device tree attached to
    ``root`` (created on the first call), exactly as the reference's ``run``
    performs all its descents before any backup."""

    CAPACITY = 512  # expanded nodes per root across repeated run() calls

    def __init__(self, config):
        self.config = config

    def run(self, root: Node, action_space_size: int, network: MuZeroNetwork,
            min_max_stats: MinMaxStats) -> Dict[int, float]:
        cfg = self.config
        A = action_space_size
        if not root.is_expanded():
            raise IndexError("root must be expanded before MCTS.run (model.py:444)")
        if root._device is None:
            trees = DeviceTrees(1, A, self.CAPACITY, network.device, cfg.pb_c_base, cfg.pb_c_init,
                                max_visits=self.CAPACITY * 64)
            hid = network._to_nhwc(root.hidden_state.to(network.device))
            pool = torch.empty(self.CAPACITY, 1, *hid.shape[1:], dtype=network.act_dtype, device=network.device)
            pool[0].copy_(hid)
            pri = [root.children[a].prior for a in range(A)]
            p32 = torch.tensor([float(np.float32(p)) for p in pri], dtype=torch.float32,
                               device=network.device).view(1, A)
            if any(isinstance(p, np.float64) for p in pri):
                # priors already mixed on the host: hand the doubles over unchanged
                p64 = torch.tensor([float(p) for p in pri], dtype=torch.float64, device=network.device).view(1, A)
                trees.root_init(p32, p64, frac=1.0)
            else:
                trees.root_init(p32, None)
            root._device = [trees, pool, 0]
        trees, pool, done = root._device
        if cfg.num_simulations > 0:
            if done + 1 >= self.CAPACITY:
                raise _lib.MuZeroAmdError("MCTS.run: device tree capacity exhausted")
            trees.select()
            priors, value, reward = network.recurrent_in_pool(pool, trees.array("leaf_parent"), 1,
                                                              trees.array("leaf_action"), pool[done + 1])
            trees.expand_backup(priors, value, reward, cfg.num_simulations, cfg.discount)
            root._device[2] = done + 1
        visits, _, probs = trees.readout()
        visits = visits.cpu().numpy()[0]
        w = trees.array("value_sum")[0, 0].cpu().numpy()
        for a in range(A):
            root.children[a].visit_count = int(visits[a])
            root.children[a].value_sum = float(w[a])
        root.visit_count = int(trees.array("root_visit").item())
        root.value_sum = float(trees.array("root_value_sum").item())
        mm = trees.array("minmax").cpu().numpy()[0]
        min_max_stats.minimum, min_max_stats.maximum = float(mm[0]), float(mm[1])
        return {a: float(p) for a, p in enumerate(probs.cpu().numpy()[0])}


def select_action(config, root, temperature: float = 1.0) -> int:
    """model.py:530-546.  Consumes one uniform draw from NumPy's global generator
    when temperature > 0, as np.random.choice does in the reference."""
    counts = torch.tensor([[c.visit_count for c in root.children.values()]], dtype=torch.int32, device="cuda")
    actions = list(root.children.keys())
    u = None
    if temperature != 0:
        u = torch.tensor([np.random.random_sample()], dtype=torch.float64, device="cuda")
    return actions[int(select_actions(counts, float(temperature), u).item())]
