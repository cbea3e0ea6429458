"""The CPU oracle against fixtures generated from the reference's model.py."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import nets as onets
from oracle import prep as oprep
from oracle import tree as otree
from tests.helpers import TREE_CASES, TableNet, load


@pytest.mark.parametrize("case", TREE_CASES)
def test_tree_search_bit_exact(case):
    g = load("g1_tree_" + case)
    net = TableNet(g)
    A, sims, wave = int(g["A"]), int(g["sims"]), int(g["wave"])
    for gi in range(g["root_hash"].shape[0]):
        h0 = int(g["root_hash"][gi])
        trace = []
        root, bounds = otree.search(
            net.root_priors(h0), h0, net.infer, sims, wave,
            noise=g["noise"][gi] if int(g["noise_on"]) else None,
            frac=float(g["frac"]), c_base=float(g["c_base"]),
            c_init=float(g["c_init"]), gamma=float(g["gamma"]), trace=trace)
        assert otree.visit_counts(root).tolist() == g["visits"][gi].tolist()
        assert root.n == g["root_n"][gi]
        assert root.w == g["root_w"][gi]
        assert root.q() == g["root_value"][gi]
        assert [k.w for k in root.kids] == g["child_w"][gi].tolist()
        assert bounds.lo == g["lo"][gi] and bounds.hi == g["hi"][gi]
        assert otree.action_probs(root).tolist() == g["probs"][gi].tolist()
        assert len(trace) == sims
        for si, p in enumerate(trace):
            n = int(g["path_len"][gi, si])
            assert p == g["path_act"][gi, si, :n].tolist()


def test_puct_scores_bit_exact():
    g = load("g5_ucb")
    for i in range(g["pn"].shape[0]):
        b = otree.Bounds()
        if not g["unset"][i]:
            b.lo, b.hi = float(g["lo"][i]), float(g["hi"][i])
        for prior, want in ((np.float32(g["p32"][i]), g["s32"][i]),
                            (float(g["p64"][i]), g["s64"][i])):
            kid = otree.TreeNode(prior)
            kid.n, kid.w = int(g["cn"][i]), float(g["cw"][i])
            got = otree.puct(int(g["pn"][i]), kid, b, float(g["c_base"]), float(g["c_init"]))
            assert type(got) is (np.float32 if isinstance(prior, np.float32) else float)
            assert got == want, (i, got, want)


def test_choose_action_matches():
    g = load("g6_action")
    for i in range(g["A"].shape[0]):
        A = int(g["A"][i])
        got = otree.choose_action(g["counts"][i, :A], float(g["temps"][i]), float(g["u"][i]))
        assert got == int(g["action"][i]), i


def test_decode_support_bit_exact():
    """The restatement is the reference's own sequence of torch operations, so on one machine the two agree
    bit for bit (checked live when the reference is importable). The committed fixture was produced on a CPU
    whose torch build sums the 601-term softmax in a different vector width: the expectation then differs in
    its last bit, and the inverse transform's `sqrt(1 + 4 eps (|x| + 1 + eps)) - 1` cancels two to three
    digits, so across machines the fixture holds to ~1e-4 relative, not to the bit."""
    g = load("g4_decode")
    y = onets.decode_support(torch.tensor(g["logits"]), 300)
    ys = onets.decode_support(torch.tensor(g["logits_small"]), 10)
    np.testing.assert_allclose(y.numpy(), g["scalar"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(ys.numpy(), g["scalar_small"], rtol=2e-4, atol=2e-4)
    same = np.mean(y.numpy() == g["scalar"])
    assert same >= 0.5, same   # most inputs are not in the cancelling range and agree exactly
    if os.path.exists("/root/reference/model.py"):
        import logging
        sys.path.insert(0, "/root/reference")
        try:
            import model as ref_model   # the reference itself: build container only
        finally:
            sys.path.pop(0)
        logging.getLogger().setLevel(logging.WARNING)
        assert np.array_equal(ref_model.support_to_scalar(torch.tensor(g["logits"]), 300).reshape(-1).numpy(),
                              y.reshape(-1).numpy())
        assert np.array_equal(ref_model.support_to_scalar(torch.tensor(g["logits_small"]), 10).reshape(-1).numpy(),
                              ys.reshape(-1).numpy())


class _Cfg:
    pass


def _small_cfg():
    c = _Cfg()
    c.observation_shape = (12, 32, 32)
    c.action_space_size = 5
    c.num_channels = 16
    c.num_blocks = 2
    c.downsample_steps = [(8, 2), (16, 3)]
    c.fc_reward_layers = [16]
    c.fc_value_layers = [16]
    c.fc_policy_layers = [16]
    c.support_size = 10
    return c


def test_nets_small_config_matches_reference_outputs():
    g = load("g2_nets_small")
    net = onets.Net(_small_cfg()).eval()
    sd = {k[3:]: torch.tensor(g[k]) for k in g.files if k.startswith("w::")}
    net.load_state_dict(sd, strict=True)  # same key set as the reference
    with torch.no_grad():
        h0, p0, v0 = net.initial(torch.tensor(g["obs"]))
        h1, r1, p1, v1 = net.recurrent(h0, torch.tensor(g["act"]))
    for got, key in ((h0, "h0"), (p0, "p0"), (v0, "v0"), (h1, "h1"),
                     (r1, "r1"), (p1, "p1"), (v1, "v1")):
        np.testing.assert_allclose(got.numpy(), g[key], rtol=1e-6, atol=1e-6)


def full_cfg(A):
    c = _Cfg()
    c.observation_shape = (96, 96, 12)
    c.action_space_size = A
    c.num_channels = 256
    c.num_blocks = 2
    c.downsample_steps = [(128, 2), (256, 3)]
    c.fc_reward_layers = [256]
    c.fc_value_layers = [256]
    c.fc_policy_layers = [256]
    c.support_size = 300
    return c


@pytest.mark.parametrize("A", [6, 18])
def test_nets_full_size_same_init_stream_and_outputs(A):
    g = load("g3_nets_full")
    torch.manual_seed(0)
    net = onets.Net(full_cfg(A)).eval()
    assert onets.state_digest(net) == str(g[f"digest_init_A{A}"])
    onets.randomize_bn(net, seed=1)
    assert onets.state_digest(net) == str(g[f"digest_bn_A{A}"])
    frames = np.random.default_rng(1234).integers(0, 256, size=(2, 96, 96, 96), dtype=np.uint8)
    obs = torch.tensor(oprep.prepare(frames))
    act = torch.tensor([[1], [A - 1]])
    with torch.no_grad():
        h0, p0, v0 = net.initial(obs)
        h1, r1, p1, v1 = net.recurrent(h0, act)
    tol = dict(rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(h0.numpy()[:1], g[f"A{A}_h0"], **tol)
    np.testing.assert_allclose(p0.numpy(), g[f"A{A}_p0"], **tol)
    np.testing.assert_allclose(v0.numpy(), g[f"A{A}_v0"], **tol)
    np.testing.assert_allclose(h1.numpy()[:1], g[f"A{A}_h1"], **tol)
    np.testing.assert_allclose(r1.numpy(), g[f"A{A}_r1"], **tol)
    np.testing.assert_allclose(p1.numpy(), g[f"A{A}_p1"], **tol)
    np.testing.assert_allclose(v1.numpy(), g[f"A{A}_v1"], **tol)
    # decoded scalars: the inverse transform amplifies last-bit differences of the logits (see the decode test)
    np.testing.assert_allclose(onets.decode_support(v0, 300).numpy(), g[f"A{A}_sv0"], rtol=2e-4, atol=2e-4)


def test_prepare_divide_is_float32_divide_for_every_byte():
    u = np.arange(256, dtype=np.uint8)
    via_f64 = (u / 255.0).astype(np.float32)
    via_f32 = u.astype(np.float32) / np.float32(255.0)
    assert np.array_equal(via_f64, via_f32)
    # a float32 multiply by the reciprocal is NOT the same
    assert not np.array_equal(via_f64, u.astype(np.float32) * np.float32(1.0 / 255.0))


def test_frame_stack_order_and_reset():
    rng = np.random.default_rng(0)
    fs = oprep.FrameStack(4)
    f0 = rng.integers(0, 256, (5, 6, 3), dtype=np.uint8)
    o = fs.reset(f0)
    assert o.shape == (5, 6, 12)
    for s in range(4):
        assert np.array_equal(o[:, :, 3 * s:3 * s + 3], f0)
    frames = [rng.integers(0, 256, (5, 6, 3), dtype=np.uint8) for _ in range(6)]
    for f in frames:
        o = fs.push(f)
    for s in range(4):  # oldest first
        assert np.array_equal(o[:, :, 3 * s:3 * s + 3], frames[2 + s])
    p = oprep.prepare(o)
    assert p.shape == (12, 5, 6) and p.dtype == np.float32
    assert p[4, 2, 3] == np.float32(o[2, 3, 4] / 255.0)
