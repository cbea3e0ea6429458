"""Device-side prioritised sampling (mz_replay_sample / mz_replay_update) against the reference
fixture g10_replay.npz and, at sizes the fixture does not reach, against the CPU restatement."""
import numpy as np
import pytest
import torch

from oracle import replay as oreplay
from tests.helpers import load
from tests.test_replay_cpu import filled_buffer

pytestmark = pytest.mark.gpu


def test_device_sampler_matches_reference_fixture():
    from muzero_amd.learner import DevicePriorities
    g = load("g10_replay")
    rb, games = filled_buffer(g, device_priorities=DevicePriorities())
    game, pos, flat, w = rb.sample_positions_device(g["s0_u"])
    assert np.array_equal(flat.cpu().numpy(), g["s0_flat"])
    assert np.array_equal(game.cpu().numpy(), g["s0_game"]) and np.array_equal(pos.cpu().numpy(), g["s0_pos"])
    np.testing.assert_allclose(w.cpu().numpy(), g["s0_weights"], rtol=1e-14, atol=0)
    # priorities of the sampled slots, last entry for a slot wins; host and device copies agree
    rb.update_priorities(flat.cpu().numpy(), g["upd_values"])
    want = oreplay.update(g["s0_prio"], g["s0_flat"], g["upd_values"], 1.0)
    assert np.array_equal(rb.device_priorities.values.cpu().numpy(), want)
    assert np.array_equal(rb.priorities, want)
    # eviction + the second and third draws of the fixture, from the reference's own state
    rb.priorities[:] = g["prio_after_update"]
    rb.device_priorities.values.copy_(torch.from_numpy(g["prio_after_update"]))
    rb.save_game(games[3])
    assert np.array_equal(rb.device_priorities.values.cpu().numpy(), g["prio_after_evict"])
    game, pos, flat, w = rb.sample_positions_device(g["s1_u"])
    assert np.array_equal(flat.cpu().numpy(), g["s1_flat"])
    assert np.array_equal(game.cpu().numpy() + 1, g["s1_game"]) and np.array_equal(pos.cpu().numpy(), g["s1_pos"])
    np.testing.assert_allclose(w.cpu().numpy(), g["s1_weights"], rtol=1e-14, atol=0)
    rb.device_priorities.values.zero_()
    game, pos, flat, w = rb.sample_positions_device(g["s2_u"])
    assert np.array_equal(flat.cpu().numpy(), g["s2_flat"]) and (w.cpu().numpy() == 1.0).all()


@pytest.mark.parametrize("n,batch,beta", [(1, 5, 1.0), (2047, 64, 1.0), (2049, 257, 0.5), (1_000_003, 4096, 1.0)])
def test_device_sampler_against_oracle_at_size(n, batch, beta):
    """Sizes around the chunk width and a million positions.  The device sums in a fixed two-level order,
    the reference left to right: an index may differ only where the draw sits within rounding of a
    boundary of the cumulative distribution, which the bound below expresses."""
    from muzero_amd.learner import DevicePriorities
    rng = np.random.default_rng(n)
    prio = rng.random(n) ** 3 * 5
    prio[rng.random(n) < 0.1] = 0.0                       # never-drawn positions
    if not prio.any():
        prio[0] = 1.0
    lens = []
    left = n
    while left:
        lens.append(int(min(left, rng.integers(1, 4000))))
        left -= lens[-1]
    dp = DevicePriorities(capacity=16)
    off = 0
    for L in lens:
        dp.append_game(prio[off:off + L]); off += L
    u = rng.random(batch)
    flat, game, pos, w = (t.cpu().numpy() for t in dp.sample(u, beta))
    f_ref, w_ref = oreplay.draw(list(prio), u, beta)
    cdf = np.cumsum(prio / prio.sum())
    cdf /= cdf[-1]
    same = flat == f_ref
    assert same.mean() >= 0.999
    eps = 1e-11
    lo = np.where(flat > 0, cdf[np.maximum(flat - 1, 0)], 0.0)
    assert (lo <= u + eps).all() and (u < cdf[flat] + eps).all() and (prio[flat] > 0).all()
    starts = np.concatenate([[0], np.cumsum(lens)])
    assert np.array_equal(game, np.searchsorted(starts, flat, side="right") - 1)
    assert np.array_equal(pos, flat - starts[game])
    if same.all():
        np.testing.assert_allclose(w, w_ref, rtol=1e-12)
    # the update kernel: duplicates resolve to the last entry, out-of-range entries are ignored
    idx = np.concatenate([flat[:8], flat[:3], [-1, n]]).astype(np.int32)
    vals = rng.random(len(idx))
    dp.update(idx, vals, 1.0)
    want = prio.copy()
    for i, v in zip(idx, vals):
        if 0 <= i < n:
            want[i] = v + 1e-6
    assert np.array_equal(dp.values.cpu().numpy(), want)


def test_device_sampler_rejects_bad_arguments():
    from muzero_amd import _lib
    p = torch.ones(8, dtype=torch.float64, device="cuda")
    u = torch.zeros(4, dtype=torch.float64, device="cuda")
    flat = torch.empty(4, dtype=torch.int32, device="cuda")
    w = torch.empty(4, dtype=torch.float64, device="cuda")
    ws = torch.empty(1, dtype=torch.float64, device="cuda")      # too small
    with pytest.raises(_lib.MuZeroAmdError):
        _lib.call("mz_replay_sample", _lib.ptr(p), 8, _lib.ptr(u), 4, 1.0, None, 0, _lib.ptr(flat), None, None,
                  _lib.ptr(w), _lib.ptr(ws), 8, _lib.current_stream())


def test_draws_on_boundaries_next_to_zero_priority_runs_never_pick_a_zero():
    """Uniforms placed exactly on (and an ulp either side of) chunk boundaries, lane boundaries and the total,
    with runs of zero priority on both sides of every boundary: every drawn position has a positive priority
    (a zero would make its importance weight infinite and the normalised batch NaN), lies where the cumulative
    distribution puts the draw to rounding, and all weights are finite."""
    from muzero_amd.learner import DevicePriorities
    rng = np.random.default_rng(17)
    n = 5 * 2048 + 777
    prio = rng.random(n) * 3
    for b in list(range(0, n, 32)):                        # zero runs straddling every lane and chunk boundary
        if rng.random() < 0.7:
            prio[max(b - int(rng.integers(1, 40)), 0):b + int(rng.integers(1, 40))] = 0.0
    prio[-300:] = 0.0                                       # the tail of the last chunk
    prio[2048 * 3:2048 * 4] = 0.0                           # a whole chunk
    assert (prio > 0).sum() > 1000
    cdf = np.cumsum(prio) / prio.sum()
    edges = cdf[np.arange(31, n, 32)]
    edges = edges[edges < 1.0]
    u = np.concatenate([edges, np.nextafter(edges, 0), np.nextafter(edges, 1), [0.0, np.nextafter(1.0, 0)]])
    u = np.clip(u, 0.0, np.nextafter(1.0, 0))
    dp = DevicePriorities(capacity=4)
    dp.append_game(prio)
    flat, game, pos, w = (t.cpu().numpy() for t in dp.sample(u, 1.0))
    assert (prio[flat] > 0).all()
    assert np.isfinite(w).all() and w.max() == 1.0 and (w > 0).all()
    eps = 1e-11
    lo = np.where(flat > 0, cdf[np.maximum(flat - 1, 0)], 0.0)
    assert (lo <= u + eps).all() and (u < cdf[flat] + eps).all()


def test_priority_update_is_linear_and_last_entry_wins_at_size():
    """64 Ki entries with heavy duplication over 5 000 slots (the quadratic scan this replaces would take
    4 G comparisons): the slot values equal a sequential last-entry-wins loop, untouched slots keep theirs."""
    from muzero_amd.learner import DevicePriorities
    rng = np.random.default_rng(23)
    n, m = 200_000, 65_536
    prio = rng.random(n) + 0.5
    prio[::7] = 0.0                                         # zero priorities are valid old values
    dp = DevicePriorities(capacity=4)
    dp.append_game(prio)
    idx = rng.choice(rng.choice(n, 5000, replace=False), m).astype(np.int32)
    idx[::1000] = -1
    idx[1::1000] = n + 3
    vals = rng.random(m)
    dp.update(idx, vals, 1.0)
    want = prio.copy()
    for i, v in zip(idx, vals):
        if 0 <= i < n:
            want[i] = v + 1e-6
    assert np.array_equal(dp.values.cpu().numpy(), want)
