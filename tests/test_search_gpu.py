"""End-to-end search on the device (float32-accurate networks: fp32 and bf16x3) against the CPU oracle."""
import numpy as np
import pytest
import torch

from tests.test_oracle_golden import full_cfg

pytestmark = pytest.mark.gpu


def _setup(A, seed=0, precision="fp32"):
    from muzero_amd.network import MuZeroNetwork
    from oracle import nets as onets
    cfg = full_cfg(A)
    cfg.pb_c_base, cfg.pb_c_init = 19652, 1.25
    cfg.root_exploration_fraction, cfg.discount = 0.25, 0.997
    cfg.num_simulations = 50
    torch.manual_seed(seed)
    ref = onets.Net(cfg).eval()
    onets.randomize_bn(ref, seed=1)
    net = MuZeroNetwork(cfg, precision=precision, state_dict=ref.state_dict())
    return cfg, ref, net


def _replay_with_device_outputs(trees_log, priors0, noise, sims, wave, cfg):
    """Oracle tree fed with the network outputs the device search consumed."""
    from oracle import tree as otree
    step = [0]

    def infer(hiddens, actions):
        pri, val, rew = trees_log[step[0]]
        step[0] += 1
        return [(pri, float(val), float(rew), None)] * len(actions)

    root, bounds = otree.search(priors0, None, infer, sims, wave, noise=noise,
                                frac=cfg.root_exploration_fraction, c_base=cfg.pb_c_base,
                                c_init=cfg.pb_c_init, gamma=cfg.discount)
    return root, bounds


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "f16x2"])
@pytest.mark.parametrize("A,wave,noisy", [(6, 1, True), (18, 1, False), (6, 5, True)])
def test_batched_search_matches_oracle(A, wave, noisy, precision):
    """fp32: exact float32 kernels on the vector units; bf16x3: the float32-accurate matrix-core path.
    Both are held to the same 1e-3 bound against the CPU oracle."""
    from muzero_amd.prep import FrameRing
    from muzero_amd.search import BatchedMCTS
    from oracle import nets as onets
    from oracle import prep as oprep
    from oracle import tree as otree
    cfg, ref, net = _setup(A, precision=precision)
    B, sims = 4, 50
    rng = np.random.default_rng(1234)
    frames = rng.integers(0, 256, size=(40, B, 96, 96, 3), dtype=np.uint8)
    ring = FrameRing(B)
    ring.reset(torch.from_numpy(frames[0]))
    stacks = [oprep.FrameStack(32) for _ in range(B)]
    obs_ref = [s.reset(frames[0, i]) for i, s in enumerate(stacks)]
    for t in range(1, 40):
        ring.push(torch.from_numpy(frames[t]))
        obs_ref = [s.push(frames[t, i]) for i, s in enumerate(stacks)]
    noise = rng.dirichlet(np.full(A, 0.3), size=B) if noisy else None

    mcts = BatchedMCTS(cfg, net, B, sims, wave)
    # run step by step to log what the device networks fed the trees
    obs = ring.observation(torch.float32)
    priors0, value0 = net.initial_to_pool(obs, mcts.pool[0])
    mcts.trees.root_init(priors0, None if noise is None else torch.from_numpy(noise).cuda(),
                         cfg.root_exploration_fraction)
    log = []
    for s in range(sims // wave):
        mcts.trees.select()
        pri, val, rew = net.recurrent_in_pool(mcts.pool, mcts.leaf_parent, B, mcts.leaf_action, mcts.pool[s + 1])
        log.append((pri.cpu().numpy(), val.cpu().numpy(), rew.cpu().numpy()))
        mcts.trees.expand_backup(pri, val, rew, wave, cfg.discount)
    visits, root_values, probs = (t.cpu().numpy() for t in mcts.trees.readout())
    p0 = priors0.cpu().numpy()

    # (1) whole BatchedMCTS.run gives the same result as the stepwise drive
    res = mcts.run(obs, None if noise is None else torch.from_numpy(noise).cuda())
    assert np.array_equal(res.visit_counts.cpu().numpy(), visits)
    assert np.array_equal(res.root_values.cpu().numpy(), root_values)

    # (2) fed identical network outputs, the tree is bit-exact
    for b in range(B):
        per_game = [(l[0][b], l[1][b], l[2][b]) for l in log]
        root, _ = _replay_with_device_outputs(per_game, p0[b], None if noise is None else noise[b], sims, wave, cfg)
        assert otree.visit_counts(root).tolist() == visits[b].tolist()
        assert root.q() == root_values[b]

    # (3) against the full CPU oracle (torch fp32 networks): network outputs and root values
    # within 1e-3; visit counts equal unless float32 rounding flips a near-tie
    infer = onets.make_infer(ref)
    same = 0
    for b in range(B):
        o = torch.tensor(oprep.prepare(obs_ref[b])).unsqueeze(0)
        with torch.no_grad():
            h, p, v = ref.initial(o)
        pri_ref = torch.softmax(p, dim=1).numpy()[0]
        np.testing.assert_allclose(p0[b], pri_ref, rtol=1e-3, atol=1e-6)
        v_ref = onets.decode_support(v, 300).item()
        assert abs(value0[b].item() - v_ref) <= 1e-3 * max(1.0, abs(v_ref))
        root, _ = otree.search(pri_ref, h, infer, sims, wave, noise=None if noise is None else noise[b])
        same += otree.visit_counts(root).tolist() == visits[b].tolist()
        assert abs(root.q() - root_values[b]) <= 1e-3 * max(1.0, abs(root.q()))
    assert same >= B - 1, f"visit counts differ from the CPU oracle in {B - same} of {B} games"


def test_reference_call_surface_single_root():
    """Node / MCTS.run / select_action used the way train.py:447-467 uses them."""
    from muzero_amd.prep import prepare_observation
    from muzero_amd.search import MCTS, MinMaxStats, Node, select_action
    from oracle import nets as onets
    from oracle import tree as otree
    A = 6
    cfg, ref, net = _setup(A)
    frames = np.random.default_rng(5).integers(0, 256, size=(96, 96, 96), dtype=np.uint8)
    obs = prepare_observation(torch.from_numpy(frames)).unsqueeze(0)
    out = net.initial_inference(obs)

    # as written: one run() with num_simulations = 50 -> every simulation takes the same path
    cfg.num_simulations = 50
    root = Node(0, None)
    root.expand(list(range(A)), out)
    mm = MinMaxStats()
    probs = MCTS(cfg).run(root, A, net, mm)
    counts = [root.children[a].visit_count for a in range(A)]
    assert counts == [50, 0, 0, 0, 0, 0] and probs[0] == 1.0 and root.visit_count == 50
    assert select_action(cfg, root, temperature=0) == 0

    # canonical: num_simulations = 1, fifty calls on the same root and statistics
    cfg.num_simulations = 1
    np.random.seed(3)
    root = Node(0, None)
    root.expand(list(range(A)), out)
    root.add_exploration_noise(cfg.root_dirichlet_alpha if hasattr(cfg, "root_dirichlet_alpha") else 0.3, 0.25)
    mm = MinMaxStats()
    for _ in range(50):
        probs = MCTS(cfg).run(root, A, net, mm)
    counts = [root.children[a].visit_count for a in range(A)]
    assert sum(counts) == 50 and root.visit_count == 50 and mm.maximum >= mm.minimum

    np.random.seed(3)
    noise = np.random.dirichlet([0.3] * A)
    with torch.no_grad():
        h, p, v = ref.initial(obs.cpu())
    oroot, ob = otree.search(torch.softmax(p, dim=1).numpy()[0], h, onets.make_infer(ref), 50, 1, noise=noise)
    assert abs(oroot.q() - root.value()) <= 1e-3 * max(1.0, abs(oroot.q()))
    assert otree.visit_counts(oroot).tolist() == counts
    np.random.seed(11)
    a = select_action(cfg, root, temperature=1.0)
    np.random.seed(11)
    assert a == otree.choose_action(np.array(counts), 1.0, np.random.random_sample())
