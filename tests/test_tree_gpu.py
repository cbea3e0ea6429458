"""HIP search-tree kernels against the golden fixtures (bit-exact) and against
the CPU oracle at full batch size, driven by the table-based fake network."""
import numpy as np
import pytest
import torch

from tests.helpers import TREE_CASES, TableNet, child_hash, load

pytestmark = pytest.mark.gpu


def run_device_search(g, root_hash, noise, sims, wave, record_paths=False, fused=False, slack=0):
    """fused: every descent after the first runs inside the expand+backup launch of the step before.
    slack: node records allocated beyond the sims // wave + 1 the search fills."""
    from muzero_amd.tree import DeviceTrees
    A = int(g["A"])
    B = len(root_hash)
    rows = g["table_priors"].shape[0]
    pri_t, val_t, rew_t = g["table_priors"], g["table_value"], g["table_reward"]
    max_nodes = sims // wave + 1 + slack
    trees = DeviceTrees(B, A, max_nodes, c_base=float(g["c_base"]), c_init=float(g["c_init"]),
                        max_visits=sims)
    dev = trees.device
    node_hash = np.zeros((B, max_nodes), np.int64)
    node_hash[:, 0] = root_hash
    trees.root_init(torch.from_numpy(pri_t[root_hash % rows]).to(dev),
                    None if noise is None else torch.from_numpy(noise).to(dev), float(g["frac"]))
    paths = []
    ar = np.arange(B)
    steps = sims // wave
    for s in range(steps):
        if s == 0 or not fused:
            trees.select()
        parent = trees.array("leaf_parent").cpu().numpy()
        act = trees.array("leaf_action").cpu().numpy()
        if record_paths:
            paths.append((trees.array("path_len").cpu().numpy().copy(),
                          trees.array("path_action").cpu().numpy().copy()))
        kid = child_hash(node_hash[ar, parent], act.astype(np.int64))
        node_hash[:, s + 1] = kid
        r = kid % rows
        step = trees.expand_backup_select if fused and s + 1 < steps else trees.expand_backup
        step(torch.from_numpy(pri_t[r]).to(dev), torch.from_numpy(val_t[r]).to(dev),
             torch.from_numpy(rew_t[r]).to(dev), wave, float(g["gamma"]))
    torch.cuda.synchronize()
    assert int(trees.array("status").abs().sum()) == 0
    return trees, paths


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("case", TREE_CASES)
def test_tree_kernels_match_reference_fixtures(case, fused):
    g = load("g1_tree_" + case)
    sims, wave = int(g["sims"]), int(g["wave"])
    noise = g["noise"] if int(g["noise_on"]) else None
    trees, paths = run_device_search(g, g["root_hash"].astype(np.int64), noise, sims, wave, True, fused)
    visits, root_value, probs = trees.readout()
    assert np.array_equal(visits.cpu().numpy(), g["visits"])
    assert np.array_equal(root_value.cpu().numpy(), g["root_value"])
    assert np.array_equal(probs.cpu().numpy(), g["probs"])
    assert np.array_equal(trees.array("value_sum")[:, 0, :].cpu().numpy(), g["child_w"])
    assert np.array_equal(trees.array("root_visit").cpu().numpy(), g["root_n"])
    assert np.array_equal(trees.array("root_value_sum").cpu().numpy(), g["root_w"])
    mm = trees.array("minmax").cpu().numpy()
    assert np.array_equal(mm[:, 0], g["lo"]) and np.array_equal(mm[:, 1], g["hi"])
    # per-simulation descent paths (all `wave` descents of one run() call are identical)
    for s, (plen, pact) in enumerate(paths):
        for w in range(wave):
            want_len = g["path_len"][:, s * wave + w]
            assert np.array_equal(plen, want_len)
            for b in range(len(plen)):
                assert pact[b, :plen[b]].tolist() == g["path_act"][b, s * wave + w, :plen[b]].tolist()


@pytest.mark.parametrize("slack", [5, 9, 30])
def test_fused_step_with_and_without_the_lds_copy_of_the_tree(slack):
    """The fused step keeps the game's tree in LDS when it has at most 64 nodes and 1024 edges (56 x 18 here)
    and walks global memory otherwise (60 x 18 edges; 81 nodes): both must reproduce the fixture."""
    g = load("g1_tree_a18_w1_noise")
    sims, wave = int(g["sims"]), int(g["wave"])
    trees, _ = run_device_search(g, g["root_hash"].astype(np.int64), g["noise"], sims, wave, fused=True, slack=slack)
    visits, root_value, probs = trees.readout()
    assert np.array_equal(visits.cpu().numpy(), g["visits"])
    assert np.array_equal(root_value.cpu().numpy(), g["root_value"])
    assert np.array_equal(probs.cpu().numpy(), g["probs"])
    assert np.array_equal(trees.array("value_sum")[:, 0, :].cpu().numpy(), g["child_w"])
    mm = trees.array("minmax").cpu().numpy()
    assert np.array_equal(mm[:, 0], g["lo"]) and np.array_equal(mm[:, 1], g["hi"])


@pytest.mark.parametrize("A,B,wave,fused", [(18, 1024, 1, True), (6, 256, 1, False), (9, 333, 5, True)])
def test_tree_kernels_match_oracle_at_batch_size(A, B, wave, fused):
    from oracle import tree as otree
    case = {18: "a18_w1_noise", 6: "a6_w1_noise", 9: "a9_w1_plain"}[A]
    g = load("g1_tree_" + case)
    sims = 50
    rng = np.random.default_rng(A * 1000 + B)
    root_hash = rng.integers(1, 2147483647, size=B).astype(np.int64)
    noise = rng.dirichlet(np.full(A, 0.3), size=B) if A != 9 else None
    trees, _ = run_device_search(g, root_hash, noise, sims, wave, fused=fused)
    visits, root_value, probs = (t.cpu().numpy() for t in trees.readout())
    assert (visits.sum(1) == sims).all()
    assert (trees.array("root_visit").cpu().numpy() == sims).all()
    assert (trees.array("num_nodes").cpu().numpy() == sims // wave + 1).all()
    net = TableNet(g)
    mm = trees.array("minmax").cpu().numpy()
    for b in rng.choice(B, size=48, replace=False):
        root, bounds = otree.search(net.root_priors(root_hash[b]), int(root_hash[b]), net.infer, sims, wave,
                                    noise=None if noise is None else noise[b], frac=float(g["frac"]),
                                    c_base=float(g["c_base"]), c_init=float(g["c_init"]),
                                    gamma=float(g["gamma"]))
        assert otree.visit_counts(root).tolist() == visits[b].tolist()
        assert root.q() == root_value[b]
        assert otree.action_probs(root).tolist() == probs[b].tolist()
        assert (bounds.lo, bounds.hi) == (mm[b, 0], mm[b, 1])


def test_select_action_kernel_matches_reference_fixture():
    from muzero_amd.tree import select_actions
    g = load("g6_action")
    for A in (4, 6, 9, 18):
        for T in (0.0, 1.0, 0.5, 0.25):
            m = (g["A"] == A) & (g["temps"] == T)
            if not m.any():
                continue
            counts = torch.from_numpy(np.ascontiguousarray(g["counts"][m][:, :A])).cuda()
            u = torch.from_numpy(g["u"][m]).cuda()
            got = select_actions(counts, T, u).cpu().numpy()
            assert np.array_equal(got, g["action"][m]), (A, T)


def test_select_action_general_temperature_agrees_with_numpy():
    from muzero_amd.tree import select_actions
    from oracle.tree import choose_action
    rng = np.random.default_rng(3)
    counts = rng.multinomial(50, np.full(18, 1 / 18), size=512).astype(np.int32)
    u = rng.random(512)
    got = select_actions(torch.from_numpy(counts).cuda(), 0.7, torch.from_numpy(u).cuda()).cpu().numpy()
    want = np.array([choose_action(counts[i], 0.7, u[i]) for i in range(512)])
    # pow() on the device may differ from NumPy's in the last ulp; a draw within one ulp of a
    # cdf edge could then pick the neighbour - none of 512 random draws is that close
    assert np.array_equal(got, want)
