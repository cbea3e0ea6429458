"""Per-node CPU restatement of the reference search tree (test infrastructure).

Follows /root/reference/model.py:
  MinMaxStats            model.py:351-370
  Node (+ value)         model.py:373-389
  Node.expand            model.py:391-403
  add_exploration_noise  model.py:405-416
  MCTS.run               model.py:424-480
  select_child           model.py:482-497
  ucb_score              model.py:499-513
  backpropagate          model.py:515-527
  select_action          model.py:530-546

The reference leans on NumPy >= 2 scalar promotion: a child prior is an
``np.float32`` unless Dirichlet noise has turned it into an ``np.float64``, and
a Python float multiplied with an ``np.float32`` is first rounded to float32.
Here every rounding step is written out, so the result does not depend on the
NumPy version in use.

The network is abstracted as ``infer(parent_hiddens, actions) -> rows`` where
each row is ``(priors: np.float32[A], value: float, reward: float, hidden)``;
``value`` and ``reward`` are float32 numbers widened to Python floats, which is
what ``support_to_scalar(...).item()`` yields in the reference
(model.py:399, model.py:467).
"""
import math
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

F32 = np.float32


class Bounds:
    """Running minimum / maximum of the node values seen in backups
    (model.py:351-370)."""

    def __init__(self):
        self.lo = math.inf
        self.hi = -math.inf

    def see(self, v: float) -> None:
        if v > self.hi:
            self.hi = v
        if v < self.lo:
            self.lo = v

    def scale(self, v: float) -> float:
        if self.hi > self.lo:
            return (v - self.lo) / (self.hi - self.lo)
        return 0.5


class TreeNode:
    """One search node.  ``prior`` is an ``np.float32`` for every node whose
    parent was expanded from network output and a Python float (double) for
    the children of a root that had noise mixed in."""

    __slots__ = ("n", "w", "prior", "kids", "hidden", "reward", "expanded")

    def __init__(self, prior):
        self.n = 0
        self.w = 0.0
        self.prior = prior
        self.kids: List["TreeNode"] = []
        self.hidden = None
        self.reward = 0.0
        self.expanded = False

    def q(self) -> float:
        return self.w / self.n if self.n else 0.0

    def grow(self, priors: np.ndarray, reward: float, hidden) -> None:
        """Expansion (model.py:391-403): every call replaces all children."""
        assert priors.dtype == np.float32
        self.expanded = True
        self.hidden = hidden
        self.reward = reward
        self.kids = [TreeNode(F32(p)) for p in priors]

    def mix_noise(self, noise: Sequence[float], frac: float) -> None:
        """Dirichlet mixing (model.py:412-415).  ``prior * (1 - frac)`` is a
        float32 product (the Python float is rounded to float32 first),
        ``noise * frac`` is a double product, the sum is a double."""
        keep = F32(1.0 - frac)
        for kid, eta in zip(self.kids, noise):
            if isinstance(kid.prior, np.float32):
                kept = float(F32(kid.prior * keep))
            else:  # already a double: a second mixing stays in double
                kept = kid.prior * (1.0 - frac)
            kid.prior = kept + float(eta) * frac


def puct(parent_n: int, kid: TreeNode, bounds: Bounds,
         c_base: float, c_init: float):
    """model.py:505-513 with the roundings spelled out."""
    c = math.log((parent_n + c_base + 1) / c_base) + c_init
    c = c * (math.sqrt(parent_n) / (kid.n + 1))
    vs = bounds.scale(kid.w / kid.n) if kid.n > 0 else 0.0
    if isinstance(kid.prior, np.float32):
        return F32(F32(F32(c) * kid.prior) + F32(vs))
    return c * kid.prior + vs


def pick_child(node: TreeNode, bounds: Bounds, c_base: float,
               c_init: float) -> Tuple[int, Optional[TreeNode]]:
    """model.py:487-497: strict '>' from -inf, children in action order, so
    ties go to the lowest action and a NaN score never wins."""
    best, best_a, best_kid = -math.inf, -1, None
    for a, kid in enumerate(node.kids):
        s = puct(node.n, kid, bounds, c_base, c_init)
        if s > best:
            best, best_a, best_kid = s, a, kid
    return best_a, best_kid


def backup(path: List[TreeNode], value: float, gamma: float,
           bounds: Bounds) -> None:
    """model.py:522-526, all in double."""
    for node in reversed(path):
        node.w += value
        node.n += 1
        bounds.see(node.w / node.n)
        value = node.reward + gamma * value


InferFn = Callable[[list, List[int]], list]


def run_wave(root: TreeNode, wave: int, infer: InferFn, bounds: Bounds,
             c_base: float = 19652, c_init: float = 1.25,
             gamma: float = 0.997, trace: Optional[list] = None) -> None:
    """One call of the reference's ``MCTS.run`` with ``num_simulations = wave``
    (model.py:433-469): ``wave`` descents on frozen statistics, one batched
    inference over the ``wave`` leaves, then expand + backup for each."""
    pending = []
    for _ in range(wave):
        node, path, acts = root, [root], []
        while node.expanded:
            a, node = pick_child(node, bounds, c_base, c_init)
            path.append(node)
            acts.append(a)
        pending.append((path[-2].hidden, acts[-1], path))
        if trace is not None:
            trace.append(list(acts))
    if not pending:
        return
    rows = infer([p[0] for p in pending], [p[1] for p in pending])
    for (_, _, path), (priors, value, reward, hidden) in zip(pending, rows):
        path[-1].grow(priors, reward, hidden)
        backup(path, value, gamma, bounds)


def visit_counts(root: TreeNode) -> np.ndarray:
    return np.array([k.n for k in root.kids], dtype=np.int32)


def action_probs(root: TreeNode) -> np.ndarray:
    """model.py:472-480."""
    n = [k.n for k in root.kids]
    tot = sum(n)
    if tot == 0:
        return np.full(len(n), 1.0 / len(n))
    return np.array([c / tot for c in n], dtype=np.float64)


def choose_action(counts: np.ndarray, temperature: float,
                  u: Optional[float] = None) -> int:
    """model.py:535-546.  ``u`` is the uniform [0,1) draw that
    ``np.random.choice`` would take from the global generator: choice(p)
    computes ``cdf = cumsum(p); cdf /= cdf[-1]`` and returns
    ``searchsorted(cdf, u, side='right')``."""
    counts = np.asarray(counts, dtype=np.int64)
    if temperature == 0:
        return int(np.argmax(counts))
    p = counts ** (1 / temperature)
    p = p / p.sum()
    cdf = np.cumsum(p)
    cdf = cdf / cdf[-1]
    return int(np.searchsorted(cdf, u, side="right"))


def search(priors_root: np.ndarray, hidden_root, infer: InferFn, sims: int,
           wave: int = 1, noise: Optional[Sequence[float]] = None,
           frac: float = 0.25, c_base: float = 19652, c_init: float = 1.25,
           gamma: float = 0.997, trace: Optional[list] = None):
    """Build a root the way train.py:240-253 does (expand with reward 0,
    optional noise), then call run_wave sims // wave times on one Bounds."""
    assert sims % wave == 0
    root = TreeNode(0)
    root.grow(np.asarray(priors_root, dtype=np.float32), 0, hidden_root)
    if noise is not None:
        root.mix_noise(noise, frac)
    bounds = Bounds()
    for _ in range(sims // wave):
        run_wave(root, wave, infer, bounds, c_base, c_init, gamma, trace)
    return root, bounds
