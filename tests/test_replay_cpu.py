"""The replay buffer against g10_replay.npz, which tests/golden/make_golden_replay.py produced by
running the reference's own ReplayBuffer (model.py:549-618): first the CPU restatement in oracle/,
then the host logic of learner.ReplayBuffer."""
import numpy as np

from oracle import replay as oreplay
from tests.helpers import load

from muzero_amd.learner import GameHistory, ReplayBuffer


def test_oracle_replay_matches_reference_fixture():
    g = load("g10_replay")
    lens = g["lens"].tolist()
    windows = {0: lens[:3], 1: lens[1:4], 2: lens[1:4]}
    for k in range(3):
        prio, u = g[f"s{k}_prio"], g[f"s{k}_u"]
        flat, w = oreplay.draw(list(prio), u, 1.0)
        assert np.array_equal(flat, g[f"s{k}_flat"])
        assert np.array_equal(w, g[f"s{k}_weights"])                 # same operations in the same order
        game, pos = oreplay.locate(flat, windows[k])
        first = 0 if k == 0 else 1                                   # the window of samples 1, 2 starts at game 1
        assert np.array_equal(game + first, g[f"s{k}_game"]) and np.array_equal(pos, g[f"s{k}_pos"])
        # what the reference hands back as "indices" is the in-game offset, not the flat index
        assert np.array_equal(pos, g[f"s{k}_batch_indices"])
    after = oreplay.update(g["s0_prio"], g["s0_batch_indices"], g["upd_values"], 1.0)
    assert np.array_equal(after, g["prio_after_update"])
    assert (g["s2_weights"] == 1.0).all()                            # zero priorities: uniform draw, unit weights


def _games(g):
    out = []
    for gi in range(len(g["lens"])):
        h = GameHistory(int(g["A"]), float(g["gamma"]))
        h.rewards, h.values = list(g[f"g{gi}_rewards"]), list(g[f"g{gi}_values"])
        h.policies, h.actions = list(g[f"g{gi}_policies"]), list(g[f"g{gi}_actions"])
        out.append(h)
    return out


class _Cfg:
    replay_buffer_size, batch_size, priority_alpha, priority_beta = 3, 64, 1.0, 1.0


def filled_buffer(g, **kw):
    games = _games(g)
    rb = ReplayBuffer(_Cfg, **kw)
    rb.save_game(games[0])
    rb.save_game(games[1], g["prio_in_1"])
    rb.save_game(games[2])
    return rb, games


def test_replay_buffer_matches_reference_fixture():
    g = load("g10_replay")
    K, td = int(g["K"]), int(g["td"])
    rb, games = filled_buffer(g)
    assert np.array_equal(rb.priorities, g["s0_prio"])               # default priorities = current maximum
    gi, pos, flat, w = rb.sample_positions(g["s0_u"])
    assert np.array_equal(flat, g["s0_flat"]) and np.array_equal(w, g["s0_weights"])
    assert np.array_equal(gi, g["s0_game"]) and np.array_equal(pos, g["s0_pos"])
    for j in range(len(flat)):                                        # the targets the reference put in its batch
        t = games[gi[j]].make_target(int(pos[j]), K, td)
        assert [x[0] for x in t] == g["s0_tv"][j].tolist() and [x[1] for x in t] == g["s0_tr"][j].tolist()
    # the deliberate departure: update_priorities addresses the sampled (flat) slots.  The reference
    # writes slot [in-game offset] instead (model.py:595, :607); the fixture records that outcome.
    rb.update_priorities(flat, g["upd_values"])
    assert np.array_equal(rb.priorities, oreplay.update(g["s0_prio"], flat, g["upd_values"], 1.0))
    assert not np.array_equal(rb.priorities, g["prio_after_update"])
    # window eviction, from the reference's state so that the fixture applies
    rb.priorities[:] = g["prio_after_update"]
    rb.save_game(games[3])
    assert len(rb) == 3 and np.array_equal(rb.priorities, g["prio_after_evict"])
    gi, pos, flat, w = rb.sample_positions(g["s1_u"])
    assert np.array_equal(flat, g["s1_flat"]) and np.array_equal(w, g["s1_weights"])
    assert np.array_equal(gi + 1, g["s1_game"]) and np.array_equal(pos, g["s1_pos"])
    rb.priorities[:] = 0.0
    gi, pos, flat, w = rb.sample_positions(g["s2_u"])
    assert np.array_equal(flat, g["s2_flat"]) and np.array_equal(w, g["s2_weights"])
