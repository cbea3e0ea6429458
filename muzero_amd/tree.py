"""Device-resident search trees for B games (host wrapper of the mz_tree_* entry points).

Replaces the per-node ``Node`` / ``MinMaxStats`` objects of model.py:351-416 by
struct-of-arrays trees in one torch-owned workspace tensor.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib

_FIELD_DTYPES = {
    "child": torch.int32, "visit": torch.int32, "value_sum": torch.float64, "prior": torch.float32,
    "root_prior64": torch.float64, "reward": torch.float64, "num_nodes": torch.int32,
    "root_visit": torch.int32, "root_value_sum": torch.float64, "minmax": torch.float64,
    "root_is_f64": torch.int32, "path_node": torch.int32, "path_action": torch.int32,
    "path_len": torch.int32, "leaf_parent": torch.int32, "leaf_action": torch.int32, "status": torch.int32,
}


def score_tables(c_base: float, c_init: float, length: int):
    """pb_c and sqrt tables indexed by the parent visit count, computed with the
    same libm calls as model.py:505-507 (math.log / math.sqrt)."""
    pbc = np.array([math.log((n + c_base + 1) / c_base) + c_init for n in range(length)], dtype=np.float64)
    sq = np.array([math.sqrt(n) for n in range(length)], dtype=np.float64)
    return pbc, sq


class DeviceTrees:
    def __init__(self, B: int, A: int, max_nodes: int, device="cuda", c_base: float = 19652,
                 c_init: float = 1.25, max_visits: int = None):
        self.B, self.A, self.max_nodes = B, A, max_nodes
        self.device = torch.device(device)
        self.layout = _lib.TreeLayout()
        _lib.call("mz_tree_layout_query", B, A, max_nodes, C.byref(self.layout))
        self.ws = torch.zeros(self.layout.total_bytes, dtype=torch.uint8, device=self.device)
        n = (max_visits if max_visits is not None else max_nodes * 64) + 2
        pbc, sq = score_tables(c_base, c_init, n)
        self.pbc = torch.from_numpy(pbc).to(self.device)
        self.sq = torch.from_numpy(sq).to(self.device)
        self.table_len = n

    def _shape(self, name):
        B, A, N = self.B, self.A, self.max_nodes
        return {"child": (B, N, A), "visit": (B, N, A), "value_sum": (B, N, A), "prior": (B, N, A),
                "root_prior64": (B, A), "reward": (B, N), "minmax": (B, 2), "path_node": (B, N),
                "path_action": (B, N)}.get(name, (B,))

    def array(self, name: str) -> torch.Tensor:
        """A typed view of one tree array inside the workspace (for inspection)."""
        dt = _FIELD_DTYPES[name]
        shape = self._shape(name)
        nbytes = int(np.prod(shape)) * torch.empty((), dtype=dt).element_size()
        off = getattr(self.layout, name)
        return self.ws[off:off + nbytes].view(dt).view(*shape)

    def root_init(self, priors: torch.Tensor, noise: torch.Tensor = None, frac: float = 0.25):
        _lib.expect(priors, "priors", torch.float32, (self.B, self.A))
        if noise is not None:
            _lib.expect(noise, "noise", torch.float64, (self.B, self.A))
        keep = float(np.float32(1.0 - frac))  # the float32 factor of prior * (1 - frac)
        _lib.call("mz_tree_root_init", _lib.ptr(self.ws), C.byref(self.layout), _lib.ptr(priors),
                  _lib.ptr(noise), keep, float(frac), _lib.current_stream())

    def select(self):
        _lib.call("mz_tree_select", _lib.ptr(self.ws), C.byref(self.layout), _lib.ptr(self.pbc),
                  _lib.ptr(self.sq), self.table_len, _lib.current_stream())

    def expand_backup(self, priors, value, reward, wave: int = 1, gamma: float = 0.997):
        _lib.expect(priors, "priors", torch.float32, (self.B, self.A))
        _lib.expect(value, "value", torch.float32, (self.B,))
        _lib.expect(reward, "reward", torch.float32, (self.B,))
        _lib.call("mz_tree_expand_backup", _lib.ptr(self.ws), C.byref(self.layout), _lib.ptr(priors),
                  _lib.ptr(value), _lib.ptr(reward), int(wave), float(gamma), _lib.current_stream())

    def expand_backup_select(self, priors, value, reward, wave: int = 1, gamma: float = 0.997):
        """``expand_backup`` followed by the next ``select`` in one launch."""
        _lib.expect(priors, "priors", torch.float32, (self.B, self.A))
        _lib.expect(value, "value", torch.float32, (self.B,))
        _lib.expect(reward, "reward", torch.float32, (self.B,))
        _lib.call("mz_tree_expand_backup_select", _lib.ptr(self.ws), C.byref(self.layout), _lib.ptr(priors),
                  _lib.ptr(value), _lib.ptr(reward), int(wave), float(gamma), _lib.ptr(self.pbc), _lib.ptr(self.sq),
                  self.table_len, _lib.current_stream())

    def readout(self, visits=None, root_value=None, probs=None):
        if visits is None:
            visits = torch.empty(self.B, self.A, dtype=torch.int32, device=self.device)
        if root_value is None:
            root_value = torch.empty(self.B, dtype=torch.float64, device=self.device)
        if probs is None:
            probs = torch.empty(self.B, self.A, dtype=torch.float64, device=self.device)
        _lib.expect(visits, "visits", torch.int32, (self.B, self.A))
        _lib.expect(root_value, "root_value", torch.float64, (self.B,))
        _lib.expect(probs, "probs", torch.float64, (self.B, self.A))
        _lib.call("mz_tree_readout", _lib.ptr(self.ws), C.byref(self.layout), _lib.ptr(visits),
                  _lib.ptr(root_value), _lib.ptr(probs), _lib.current_stream())
        return visits, root_value, probs


def select_actions(visits: torch.Tensor, temperature: float, u: torch.Tensor = None) -> torch.Tensor:
    """Device select_action (model.py:530-546) for every row of ``visits``."""
    B, A = visits.shape
    _lib.expect(visits, "visits", torch.int32)
    if temperature != 0:
        _lib.expect(u, "u", torch.float64, (B,))
    out = torch.empty(B, dtype=torch.int32, device=visits.device)
    _lib.call("mz_select_action", _lib.ptr(visits), B, A, float(temperature), _lib.ptr(u), _lib.ptr(out),
              _lib.current_stream())
    return out
