"""Shared test helpers: golden loading and the table-driven fake network that
the tree fixtures were generated with (tests/golden/make_golden.py)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HASH_MOD = 2147483647
TREE_CASES = ["a6_w1_noise", "a18_w1_noise", "a9_w1_plain", "a6_w50_noise",
              "a18_w5_noise", "a4_w1_deep", "a9_w2_noise", "a64_w1_noise"]


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def child_hash(h, a):
    """Vectorised over numpy int64 arrays or plain ints."""
    return (h * 1000003 + (a + 1) * 7919 + 12345) % HASH_MOD


class TableNet:
    """Oracle-side fake network: hidden state == node hash (python int)."""

    def __init__(self, g):
        self.pri = g["table_priors"]
        self.val = g["table_value"]
        self.rew = g["table_reward"]
        self.rows = self.pri.shape[0]

    def root_priors(self, h0):
        return self.pri[int(h0) % self.rows]

    def infer(self, hiddens, actions):
        out = []
        for h, a in zip(hiddens, actions):
            k = child_hash(int(h), int(a))
            r = k % self.rows
            out.append((self.pri[r], float(self.val[r]), float(self.rew[r]), k))
        return out
