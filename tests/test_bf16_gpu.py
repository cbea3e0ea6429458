"""The bf16 MFMA network path end to end: accuracy against the fp32 oracle (stated
separately from the fp32 path's 1e-3 bound) and bit-exact trees on its outputs."""
import numpy as np
import pytest
import torch

from tests.helpers import load
from tests.test_oracle_golden import full_cfg

pytestmark = pytest.mark.gpu


def _nets(A, precision="bf16"):
    from muzero_amd.network import MuZeroNetwork
    from oracle import nets as onets
    cfg = full_cfg(A)
    cfg.pb_c_base, cfg.pb_c_init, cfg.root_exploration_fraction, cfg.discount = 19652, 1.25, 0.25, 0.997
    torch.manual_seed(0)
    ref = onets.Net(cfg).eval()
    onets.randomize_bn(ref, seed=1)
    return cfg, ref, MuZeroNetwork(cfg, precision=precision, state_dict=ref.state_dict())


@pytest.mark.parametrize("A", [6, 18])
def test_bf16_network_close_to_reference_fixture(A):
    """bf16 activations and weights in the convolutions (IEEE-half heads), fp32 accumulation, against the
    reference's float32 outputs (g3) at both fixture action counts - 18 is the benchmark's.  The bound stated
    for this precision: the largest error of a logit tensor at most 0.8 % of that tensor's RMS (measured
    <= 0.5 %), decoded scalars 2 %, hidden states 6 % (measured 3-4 %); the measured ratios are printed."""
    from muzero_amd.network import support_to_scalar
    from muzero_amd.prep import prepare_observation
    g = load("g3_nets_full")
    cfg, ref, net = _nets(A)
    frames = np.random.default_rng(1234).integers(0, 256, size=(2, 96, 96, 96), dtype=np.uint8)
    obs = prepare_observation(torch.from_numpy(frames))
    act = torch.tensor([[1], [A - 1]])
    o0 = net.initial_inference(obs)
    o1 = net.recurrent_inference(o0.hidden_state, act)
    o2 = net.recurrent_inference(o1.hidden_state, act.flip(0))
    torch.cuda.synchronize()
    report, multiples = {}, {}

    def close(got, key, frac):
        want = g[f"A{A}_{key}"]
        rms = float(np.sqrt((want.astype(np.float64) ** 2).mean()))
        diff = np.abs(got.cpu().numpy() - want)
        err = float(diff.max())
        report[key] = round(err / max(rms, 1e-3), 4)
        # the same error as a multiple of the float32-accurate precisions' bound |got - want| <= 2e-4 + 1e-3 |want|
        multiples[key] = round(float((diff / (2e-4 + 1e-3 * np.abs(want))).max()), 2)
        assert err <= frac * max(rms, 1e-3), (key, err, rms)
    close(o0.hidden_state[:1], "h0", 0.06); close(o0.policy_logits, "p0", 0.008); close(o0.value, "v0", 0.008)
    close(o1.hidden_state[:1], "h1", 0.06); close(o1.policy_logits, "p1", 0.008)
    close(o1.value, "v1", 0.008); close(o1.reward, "r1", 0.008)
    close(o2.policy_logits, "p2", 0.008); close(o2.value, "v2", 0.008); close(o2.reward, "r2", 0.008)
    close(support_to_scalar(o0.value, 300), "sv0", 0.02); close(support_to_scalar(o1.value, 300), "sv1", 0.02)
    close(support_to_scalar(o1.reward, 300), "sr1", 0.02)
    print(f"bf16 vs reference fixture, A={A}: max error / RMS per output", report)
    print(f"bf16 vs reference fixture, A={A}: the same as multiples of the 1e-3 bound", multiples)
    assert max(v for k, v in multiples.items() if k[0] in "pvr") <= 4.0, multiples      # logits: within 4x of it


@pytest.mark.parametrize("A,B,wave", [(18, 16, 1), (6, 9, 5)])
def test_bf16_search_trees_bit_exact_on_their_own_outputs(A, B, wave):
    from muzero_amd.selfplay import SelfPlayWorker, SyntheticEnv
    from oracle import tree as otree
    cfg, ref, net = _nets(A)
    sims = 50
    worker = SelfPlayWorker(cfg, net, B, sims, wave)
    env = SyntheticEnv(B, device="cuda")
    worker.reset(env.reset())
    for _ in range(5):
        worker.ring.push(env.step())
    worker.draw_randomness()
    m = worker.mcts
    obs = worker.ring.observation(net.obs_dtype)
    pri0, _ = net.initial_to_pool(obs, m.pool[0])
    pri0 = pri0.clone()
    m.trees.root_init(pri0, worker.noise, cfg.root_exploration_fraction)
    log = []
    for s in range(sims // wave):
        m.trees.select()
        p, v, r = net.recurrent_in_pool(m.pool, m.leaf_parent, B, m.leaf_action, m.pool[s + 1])
        log.append((p.cpu().numpy(), v.cpu().numpy(), r.cpu().numpy()))
        m.trees.expand_backup(p, v, r, wave, cfg.discount)
    visits, values, probs = (t.cpu().numpy().copy() for t in m.trees.readout())
    noise, p0 = worker.noise.cpu().numpy(), pri0.cpu().numpy()
    for b in range(B):
        it = iter([(l[0][b], float(l[1][b]), float(l[2][b]), None) for l in log])
        root, _ = otree.search(p0[b], None, lambda h, a, it=it: [next(it)] * len(a), sims, wave, noise=noise[b])
        assert otree.visit_counts(root).tolist() == visits[b].tolist()
        assert root.q() == values[b]
    # the one-call run() and the HIP-graph replay reproduce the stepwise result exactly
    res = m.run(obs, worker.noise)
    assert np.array_equal(res.visit_counts.cpu().numpy(), visits)
    assert np.array_equal(res.root_values.cpu().numpy(), values)
    replay = m.capture(obs, worker.noise)
    m.visits.zero_()
    res = replay()
    torch.cuda.synchronize()
    assert np.array_equal(res.visit_counts.cpu().numpy(), visits)
    assert np.array_equal(res.action_probs.cpu().numpy(), probs)


def test_selfplay_worker_evaluate_mode_and_temperature_zero():
    """The evaluate() flavour of the step (train.py:456-467): no Dirichlet noise, first-maximum
    action; and the host RNG hooks of the self-play flavour."""
    from muzero_amd.selfplay import SelfPlayWorker, SyntheticEnv
    cfg, ref, net = _nets(6)
    B, sims = 5, 20
    worker = SelfPlayWorker(cfg, net, B, sims)
    env = SyntheticEnv(B, device="cuda")
    worker.reset(env.reset())
    actions, res = worker.step(env.step(), temperature=0, add_noise=False)
    visits = res.visit_counts.cpu().numpy()
    assert (visits.sum(1) == sims).all()
    assert actions.cpu().tolist() == visits.argmax(1).tolist()
    assert int(worker.mcts.trees.array("root_is_f64").sum()) == 0          # float32 root scores
    roots = res.roots()
    assert [roots[0].children[a].visit_count for a in range(6)] == visits[0].tolist()
    assert roots[0].value() == float(res.root_values[0])
    assert abs(sum(res.action_prob_dicts()[0].values()) - 1.0) < 1e-12
    worker.draw_randomness()
    actions2, res2 = worker.step(env.step(), temperature=1.0, add_noise=True)
    assert int(worker.mcts.trees.array("root_is_f64").sum()) == B
    v2 = res2.visit_counts.cpu().numpy()
    assert all(v2[i, a] > 0 for i, a in enumerate(actions2.cpu().tolist()))   # only visited actions are drawn


@pytest.mark.parametrize("A,B,precision", [(6, 256, "bf16"), (18, 1024, "bf16"), (9, 4096, "bf16"), (18, 1024, "bf16x3"),
                                           (18, 1024, "f16"), (18, 1024, "f16x2")])
def test_full_size_search_properties_and_sampled_replay(A, B, precision):
    """The benchmark configurations at their full sizes, through the calls the benchmark makes (for bf16 the
    ROTATING observation and the first convolution's permuted weight copies, after enough moves to wrap the ring):
    size-independent properties for every game (visit counts sum to the simulation count, one node per simulation,
    no status flags), the oracle tree replaying the recorded network outputs bit-exactly for a sample of games,
    the root priors of a few games against the CPU oracle networks, and SelfPlayWorker.step() giving the same
    result as the stepwise search."""
    from muzero_amd.selfplay import SelfPlayWorker, SyntheticEnv
    from oracle import prep as oprep
    from oracle import tree as otree
    cfg, ref, net = _nets(A, precision)
    sims = 50
    worker = SelfPlayWorker(cfg, net, B, sims, 1)
    assert worker.rotating == (precision in ("bf16", "f16"))
    env = SyntheticEnv(B, device="cuda", pool=3)
    worker.reset(env.reset())
    for _ in range(34):                       # past the ring's wrap: newest slot 1
        worker.ring.push(env.step())
    worker.draw_randomness()
    m = worker.mcts
    rng = np.random.default_rng(B)
    sample = rng.choice(B, size=24, replace=False)
    idx = torch.from_numpy(sample).cuda()
    if worker.rotating:
        obs, newest = worker.ring.rotating_observation()
        assert newest == 1
    else:
        obs, newest = worker.ring.observation(net.act_dtype if worker.pairs else net.obs_dtype, out=worker.obs,
                                              pairs=worker.pairs), None
    pri0, val0 = net.initial_to_pool(obs, m.pool[0], newest)
    p0 = pri0[idx].cpu().numpy()
    # root priors / values of four sampled games against the float32 CPU networks on the stacked observation
    # rebuilt from the synthetic frames on the host (reset frame 0, then frames t % 3 for t = 1..34)
    few = sample[:4]
    host = env.host.numpy()
    stack = [host[t % 3][few] for t in range(3, 35)]                      # the last 32 pushes, oldest first
    stacked = np.concatenate(stack, axis=-1)
    with torch.no_grad():
        _, p_ref, v_ref = ref.initial(torch.tensor(oprep.prepare(stacked)))
    want = torch.softmax(p_ref, dim=1).numpy()
    tol = {"bf16": 0.02, "f16": 0.003}.get(precision, 1e-3)       # bf16x3, f16x2: float32-accurate
    assert np.abs(p0[:4] - want).max() <= tol * want.max(), np.abs(p0[:4] - want).max()
    m.trees.root_init(pri0, worker.noise, cfg.root_exploration_fraction)
    m.trees.select()
    log = []
    for s in range(sims):
        p, v, r = net.recurrent_in_pool(m.pool, m.leaf_parent, B, m.leaf_action, m.pool[s + 1])
        log.append((p[idx].cpu().numpy(), v[idx].cpu().numpy(), r[idx].cpu().numpy()))
        (m.trees.expand_backup_select if s + 1 < sims else m.trees.expand_backup)(p, v, r, 1, cfg.discount)
    visits, values, probs = (t.cpu().numpy().copy() for t in m.trees.readout())
    assert (visits.sum(1) == sims).all() and (visits >= 0).all()
    assert (m.trees.array("root_visit").cpu().numpy() == sims).all()
    assert (m.trees.array("num_nodes").cpu().numpy() == sims + 1).all()
    assert int(m.trees.array("status").abs().sum()) == 0
    assert np.allclose(probs.sum(1), 1.0, atol=1e-12) and np.isfinite(values).all()
    noise = worker.noise.cpu().numpy()
    for j, b in enumerate(sample):
        it = iter([(l[0][j], float(l[1][j]), float(l[2][j]), None) for l in log])
        root, _ = otree.search(p0[j], None, lambda h, a, it=it: [next(it)] * len(a), sims, 1, noise=noise[b])
        assert otree.visit_counts(root).tolist() == visits[b].tolist()
        assert root.q() == values[b]
    res = m.run(obs, worker.noise, rotation=newest)
    m.check_status()
    assert np.array_equal(res.visit_counts.cpu().numpy(), visits)
    assert np.array_equal(res.root_values.cpu().numpy(), values)
    # the benchmark's own call: no new frame, same noise -> the same search
    actions, res = worker.step(None, temperature=1.0, add_noise=True)
    assert np.array_equal(res.visit_counts.cpu().numpy(), visits)
    assert np.array_equal(res.root_values.cpu().numpy(), values)
    a = actions.cpu().numpy()
    assert (visits[np.arange(B), a] > 0).all()


@pytest.mark.parametrize("B", [1, 15, 16, 333, 1024])
def test_bf16_heads_match_their_arithmetic_in_torch(B):
    """The heads of the 16-bit precisions - IEEE-half weights and half-rounded inputs with float32 accumulation -
    restated in torch (the spatial means and the hidden layer rounded to half exactly where the kernel rounds
    them).  Only the order of the float32 partial sums differs - and with it, now and then, the rounding of a
    hidden unit that sits on a rounding boundary: logits to 5e-4, priors to 1e-5 of the softmax of the
    kernel's own logits, decoded scalars to 1e-3 relative / 2e-3 absolute of the decode of the kernel's own logits
    (the decode itself is pinned bit for bit by the fixture test in test_nets_gpu.py).
    Batch sizes that leave the last workgroup's 16 samples partly or wholly empty included."""
    from oracle import nets as onets
    cfg, ref, net = _nets(18)
    torch.manual_seed(B)
    mean = (torch.randn(B, 256, device="cuda") * 0.5).contiguous()
    priors, value, reward, pl, vl, rl = net._heads(mean, B, True, True)
    torch.cuda.synchronize()
    rb = lambda t: t.to(torch.float16).to(torch.float32)

    def mlp(seq, x):
        l1, l2 = seq[0], seq[2]
        h = torch.relu(rb(x) @ rb(l1.weight.detach().cuda()).T + l1.bias.detach().cuda())
        return rb(h) @ rb(l2.weight.detach().cuda()).T + l2.bias.detach().cuda()

    for got, seq in ((pl, ref.prediction.policy_head), (vl, ref.prediction.value_head), (rl, ref.dynamics.reward_head)):
        want = mlp(seq, mean)
        rms = float(want.pow(2).mean().sqrt())
        assert float((got - want).abs().max()) <= 5e-4 * max(rms, 1.0), (float((got - want).abs().max()), rms)
    assert float((priors - torch.softmax(pl, 1)).abs().max()) <= 1e-5
    for got, lg in ((value, vl), (reward, rl)):
        want = onets.decode_support(lg.cpu(), cfg.support_size).view(-1)
        assert torch.allclose(got.cpu(), want, rtol=1e-3, atol=2e-3)  # float32 cancellation near zero
