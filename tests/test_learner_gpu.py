"""Training-side HIP kernels and the device training step against the reference fixtures
(tests/golden/make_golden_train.py)."""
import numpy as np
import pytest
import torch

from oracle.loss import support_ce
from tests.helpers import load
from tests.test_learner_cpu import build_games, check_against_g7, g7_batch, g7_learner

pytestmark = pytest.mark.gpu


def test_nstep_targets_bit_exact():
    from muzero_amd.learner import batched_targets
    g8 = load("g8_targets")
    K, td = int(g8["K"]), int(g8["td"])
    games = build_games(g8)
    queries = [(gi, s) for gi, g in enumerate(games) for s in range(len(g.rewards))]
    tv, tr, tp = batched_targets(games, queries, K, td)
    tv, tr, tp = tv.cpu().numpy(), tr.cpu().numpy(), tp.cpu().numpy()
    row = 0
    for gi, g in enumerate(games):
        n = len(g.rewards)
        assert np.array_equal(tv[row:row + n], g8[f"g{gi}_tv"])       # float64, same operation order: bit-exact
        assert np.array_equal(tr[row:row + n], g8[f"g{gi}_tr"])
        assert np.array_equal(tp[row:row + n], g8[f"g{gi}_tp"])
        row += n
    assert row == len(queries)


def test_nstep_targets_large_random_against_host():
    """Size-independent check at a realistic replay size: 512 episodes up to 3000 steps, 4096 queries."""
    from muzero_amd.learner import GameHistory, batched_targets
    rng = np.random.default_rng(4)
    games = []
    for _ in range(512):
        T = int(rng.integers(1, 3000))
        g = GameHistory(6, 0.997)
        g.rewards = list(rng.normal(0, 1, T))
        g.values = list(rng.normal(0, 3, T))
        g.policies = list(rng.dirichlet(np.ones(6), T))
        games.append(g)
    queries = [(int(gi), int(rng.integers(0, len(games[gi].rewards)))) for gi in rng.integers(0, 512, 4096)]
    tv, tr, tp = batched_targets(games, queries, 5, 10)
    tv, tr, tp = tv.cpu().numpy(), tr.cpu().numpy(), tp.cpu().numpy()
    for i in range(0, 4096, 7):
        gi, s = queries[i]
        for k, (v, r, p) in enumerate(games[gi].make_target(s, 5, 10)):
            assert tv[i, k] == v and tr[i, k] == r and np.array_equal(tp[i, k], p)


def test_episode_store_follows_the_replay_window():
    """The device episode store behind a replay buffer: eviction shifts game indices, targets stay
    bit-exact against Game.make_target, and a training batch built from it equals the per-position one."""
    from muzero_amd.learner import EpisodeStore, ReplayBuffer
    g8 = load("g8_targets")
    games = build_games(g8)

    class Cfg:
        replay_buffer_size, batch_size, priority_alpha, priority_beta = 3, 16, 1.0, 1.0
    rb = ReplayBuffer(Cfg, EpisodeStore(6, 0.997, capacity=4))       # tiny capacity: exercises growth
    for g in games:
        rb.save_game(g)
    assert len(rb) == 3 and len(rb.store) == 3
    gi, pos, flat, w = rb.sample_positions((np.arange(16) + 0.5) / 16)
    tv, tr, tp = (t.cpu().numpy() for t in rb.store.targets(gi, pos, 5, 10))
    for i, (g_idx, p) in enumerate(zip(gi, pos)):
        for k, (v, r, pol) in enumerate(rb.buffer[g_idx].make_target(int(p), 5, 10)):
            assert tv[i, k] == v and tr[i, k] == r and np.array_equal(tp[i, k], pol)
    with pytest.raises(IndexError):
        rb.store.targets([3], [0], 5, 10)


def _edge_targets(S):
    # zeros, signs, exact bin centres, the clamp on both sides (the +S clamp makes both scatter indices coincide)
    inv = lambda t: np.sign(t) * (((np.sqrt(1 + 4 * 0.001 * (abs(t) + 1 + 0.001)) - 1) / 0.002) ** 2 - 1)
    vals = [0.0, -0.0, 1e-8, -1e-8, 0.5, -0.5, 3.0, -3.0, inv(S - 0.25), inv(-(S - 0.25)), inv(S + 5.0), inv(-S - 5.0),
            1e9, -1e9]
    return np.array(vals, np.float32)


@pytest.mark.parametrize("S", [10, 300])
def test_support_ce_matches_reference_ops(S):
    from muzero_amd.learner import support_ce_loss
    from oracle.loss import support_ce as _support_ce_torch
    rng = np.random.default_rng(S)
    t = np.concatenate([_edge_targets(S), rng.normal(0, 30, 1000).astype(np.float32)])
    x = rng.normal(0, 2, (t.size, 2 * S + 1)).astype(np.float32)
    x[5] += 40 * (rng.random(2 * S + 1) > 0.9)                 # a peaked row
    xd = torch.tensor(x, device="cuda", requires_grad=True)
    td = torch.tensor(t, device="cuda")
    loss = support_ce_loss(xd, td, S, impl="hip")
    w = torch.tensor(rng.random(t.size).astype(np.float32), device="cuda")
    (loss * w).sum().backward()
    # the reference's own op sequence on the CPU, float32
    xr = torch.tensor(x, requires_grad=True)
    ref = _support_ce_torch(xr, torch.tensor(t), S)
    (ref * w.cpu()).sum().backward()
    np.testing.assert_allclose(loss.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-6)  # softmax - target cancels


def test_support_ce_golden_rows():
    """The g7 fixture's value targets through the kernel against a float64 evaluation of
    model.py:112-140 on the same rows."""
    from muzero_amd.learner import support_ce_loss
    g7 = load("g7_losses")
    S = 10
    t = g7["target_rewards"].reshape(-1).astype(np.float32)      # holds the 400.0 that clamps at +S
    rng = np.random.default_rng(0)
    x = rng.normal(0, 1, (t.size, 2 * S + 1)).astype(np.float32)
    got = support_ce_loss(torch.tensor(x, device="cuda"), torch.tensor(t, device="cuda"), S, impl="hip").cpu().numpy()
    tt = np.clip(np.sign(t) * (np.sqrt(np.abs(t) + 1) - 1 + np.float32(0.001) * t), -S, S).astype(np.float32)
    fl = np.floor(tt); up_p = tt - fl; lo_p = 1 - up_p
    lo = np.clip((fl + S).astype(int), 0, 2 * S); up = np.clip((fl + S).astype(int) + 1, 0, 2 * S)
    sup = np.zeros((t.size, 2 * S + 1))
    sup[np.arange(t.size), lo] = lo_p
    sup[np.arange(t.size), up] = up_p
    x64 = x.astype(np.float64)
    logp = x64 - x64.max(1, keepdims=True)
    logp = logp - np.log(np.exp(logp).sum(1, keepdims=True))
    np.testing.assert_allclose(got, -(sup * logp).sum(1), rtol=1e-5, atol=2e-6)


def test_device_training_step_matches_reference():
    """Whole unrolled loss + backward on the device with the HIP loss: fp32 within 1e-3 of the
    reference's CPU numbers (losses, per-sample loss, every parameter gradient, BN statistics)."""
    torch.backends.cudnn.allow_tf32 = False
    g7 = load("g7_losses")
    lrn = g7_learner(g7, "cuda", "hip")
    assert lrn.loss_impl == "hip"
    check_against_g7(lrn, g7, rtol=1e-3)


def test_hip_and_torch_loss_agree_on_device():
    g7 = load("g7_losses")
    a, b = g7_learner(g7, "cuda", "hip"), g7_learner(g7, "cuda", support_ce)
    ra = a.update_weights(g7_batch(g7), g7["weights"])
    rb = b.update_weights(g7_batch(g7), g7["weights"])
    np.testing.assert_allclose(ra["priorities"], rb["priorities"], rtol=1e-5)
    for (k, pa), (_, pb) in zip(a.network.state_dict().items(), b.network.state_dict().items()):
        np.testing.assert_allclose(pa.cpu().numpy(), pb.cpu().numpy(), rtol=1e-4, atol=1e-6, err_msg=k)


def test_trained_weights_reach_the_search_engine():
    """After an update the learner's state_dict loads into the inference engine, whose fp32
    outputs equal the learner network's own eval-mode outputs."""
    from muzero_amd.network import MuZeroNetwork
    g7 = load("g7_losses")
    lrn = g7_learner(g7, "cuda", "hip")
    for _ in range(3):
        lrn.update_weights(g7_batch(g7), g7["weights"])
    sd = lrn.state_dict()
    eng = MuZeroNetwork(lrn.config, precision="fp32", device="cuda", state_dict=sd)
    obs = torch.tensor(g7["obs"], device="cuda").permute(0, 3, 1, 2).float() / 255.0
    out = eng.initial_inference(obs)
    lrn.network.eval()
    with torch.no_grad():
        h, p, v = lrn.network.initial(obs)
    lrn.network.train()
    np.testing.assert_allclose(out.policy_logits.cpu().numpy(), p.cpu().numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(out.hidden_state.cpu().numpy(), h.cpu().numpy(), rtol=1e-3, atol=1e-4)


def test_reference_checkpoint_feeds_the_search_engine():
    """A checkpoint written by the reference loads into the inference engine (BatchNorm folded at
    load); fp32 outputs within 1e-3 of the reference's eval-mode outputs stored beside it."""
    import os
    from tests.helpers import GOLDEN
    from muzero_amd import checkpoint
    g9 = load("g9_outputs")
    net = checkpoint.load_network(os.path.join(GOLDEN, "g9_checkpoint.pt"), precision="fp32", input_hw=(32, 32))
    o0 = net.initial_inference(torch.tensor(g9["obs"]).cuda())
    o1 = net.recurrent_inference(o0.hidden_state, torch.tensor([[1], [4], [0], [2]]).cuda())
    for got, key in ((o0.hidden_state, "h0"), (o0.policy_logits, "p0"), (o0.value, "v0"), (o1.hidden_state, "h1"),
                     (o1.reward, "r1"), (o1.policy_logits, "p1"), (o1.value, "v1")):
        np.testing.assert_allclose(got.cpu().numpy(), g9[key], rtol=1e-3, atol=2e-4, err_msg=key)


def test_selfplay_training_roundtrip():
    """Search -> episode records -> replay -> update -> weights back into the search engine, at the full
    network size on a few small games: the loss falls over repeated updates on one replay buffer and the
    engine's outputs change with the published weights."""
    from muzero_amd import loop
    from muzero_amd.config import MuZeroConfig
    from muzero_amd.learner import Learner, ReplayBuffer
    from muzero_amd.network import MuZeroNetwork, init_state_dict
    from muzero_amd.selfplay import SelfPlayWorker, SyntheticEnv
    cfg = MuZeroConfig()
    cfg.observation_shape, cfg.action_space_size, cfg.batch_size = (96, 96, 96), 4, 8
    cfg.lr_init = 0.01
    sd = init_state_dict(cfg, seed=0)
    net = MuZeroNetwork(cfg, precision="bf16", state_dict=sd)
    learner = Learner(cfg, state_dict=sd)
    B, moves = 8, 7
    worker = SelfPlayWorker(cfg, net, B, num_simulations=10)
    env = SyntheticEnv(B, device="cuda")
    rng = np.random.default_rng(0)
    games = loop.new_games(worker, env.reset())
    loop.play_moves(worker, games, lambda a: (env.step(a), rng.normal(0, 1, B)), moves)
    for g in games:
        assert len(g.rewards) == moves and len(g.values) == moves and len(g.policies) == moves
        assert abs(g.policies[0].sum() - 1) < 1e-12 and g.get_observation(3).shape == (96, 96, 96)
    from muzero_amd.learner import EpisodeStore
    replay = ReplayBuffer(cfg, EpisodeStore(cfg.action_space_size, cfg.discount))
    loop.save_games(replay, games, cfg.td_steps)
    assert len(replay.priorities) == B * moves
    # the batched-target route and the per-position route give the same update
    plain = ReplayBuffer(cfg)
    loop.save_games(plain, games, cfg.td_steps)
    la, lb = Learner(cfg, state_dict=sd), Learner(cfg, state_dict=sd)
    np.random.seed(3)
    ma = loop.train_steps(la, replay, 1)[0]
    np.random.seed(3)
    mb = loop.train_steps(lb, plain, 1)[0]
    np.testing.assert_allclose(ma["priorities"], mb["priorities"], rtol=1e-4)
    loop.save_games(replay, [], cfg.td_steps)
    replay.priorities = list(plain.priorities[:0]) + [1.0] * len(replay.priorities)     # undo the priority update
    np.random.seed(0)
    obs = worker.ring.observation(net.obs_dtype).permute(0, 3, 1, 2).float()
    before = net.initial_inference(obs).policy_logits.clone()
    metrics = loop.train_steps(learner, replay, 6)
    assert all(np.isfinite(m["total_loss"]) for m in metrics)
    assert min(m["total_loss"] for m in metrics[3:]) < metrics[0]["total_loss"]
    loop.publish(learner, net)
    after = net.initial_inference(obs).policy_logits
    assert not torch.equal(before, after)
    actions, res = worker.step(None)                       # and the search runs on the new weights
    assert int(res.visit_counts.sum()) == B * 10


def test_episode_store_compaction_keeps_targets_exact():
    """More than a thousand evictions trigger the store's compaction of its evicted prefix; game
    indices and targets stay exact across it."""
    from muzero_amd.learner import EpisodeStore, GameHistory
    rng = np.random.default_rng(11)
    store = EpisodeStore(4, 0.997, capacity=8)
    games = []
    for _ in range(1200):
        T = int(rng.integers(1, 6))
        g = GameHistory(4, 0.997)
        g.rewards, g.values = list(rng.normal(0, 1, T)), list(rng.normal(0, 1, T))
        g.policies = list(rng.dirichlet(np.ones(4), T))
        games.append(g)
        store.add_game(g)
    for _ in range(1100):
        store.pop_front()
    assert len(store) == 100 and store.base < 1100          # compacted at least once
    live = games[1100:]
    q = [(i, int(rng.integers(0, len(g.rewards)))) for i, g in enumerate(live)]
    tv, tr, tp = (t.cpu().numpy() for t in store.targets([a for a, _ in q], [b for _, b in q], 3, 2))
    for i, (gi, p) in enumerate(q):
        for k, (v, r, pol) in enumerate(live[gi].make_target(p, 3, 2)):
            assert tv[i, k] == v and tr[i, k] == r and np.array_equal(tp[i, k], pol)
    store.add_game(games[0])                                 # and it keeps growing after compaction
    tv2, _, _ = store.targets([100], [0], 3, 2)
    assert tv2.cpu().numpy()[0, 0] == games[0].make_target(0, 3, 2)[0][0]


def test_evaluate_runs_noise_free_greedy_episodes():
    from muzero_amd import loop
    from muzero_amd.config import MuZeroConfig
    from muzero_amd.network import MuZeroNetwork, init_state_dict
    from muzero_amd.selfplay import SelfPlayWorker, SyntheticEnv
    cfg = MuZeroConfig()
    cfg.observation_shape, cfg.action_space_size = (96, 96, 96), 4
    net = MuZeroNetwork(cfg, precision="bf16", state_dict=init_state_dict(cfg, seed=0))
    B = 4
    worker = SelfPlayWorker(cfg, net, B, num_simulations=8)
    env = SyntheticEnv(B, device="cuda")
    lens = np.array([2, 3, 5, 9])
    t = [0]

    def env_step(actions):
        t[0] += 1
        return env.step(actions), np.ones(B), t[0] >= lens
    ret = loop.evaluate(worker, env.reset(), env_step, max_moves=20)
    assert ret.tolist() == lens.tolist() and t[0] == 9          # reward 1 per move until each game's end
    assert int(worker.mcts.trees.array("root_is_f64").sum()) == 0   # no exploration noise at the roots


def test_continuous_selfplay_on_the_device():
    """play_continuous with a real worker: episodes of different lengths end, are handed over with
    search statistics for every position, and a restarted game's frame stack on the device holds
    only frames of its new episode."""
    from muzero_amd import loop
    from muzero_amd.config import MuZeroConfig
    from muzero_amd.envpool import HostEnvPool
    from muzero_amd.network import MuZeroNetwork, init_state_dict
    from muzero_amd.selfplay import SelfPlayWorker

    class Env:
        """Frames carry (episode, step) in two channels; episode i ends after 3 + i steps."""

        def __init__(self, idx):
            self.idx, self.ep, self.t = idx, -1, 0

        def _obs(self):
            f = np.zeros((96, 96, 3), np.uint8)
            f[..., 0], f[..., 1], f[..., 2] = self.ep, self.t, self.idx
            return f

        def reset(self):
            self.ep, self.t = self.ep + 1, 0
            return self._obs(), {}

        def step(self, a):
            self.t += 1
            return self._obs(), 1.0, self.t >= 3 + self.idx, False, {}

    cfg = MuZeroConfig()
    cfg.observation_shape, cfg.action_space_size = (96, 96, 96), 4
    net = MuZeroNetwork(cfg, precision="bf16", state_dict=init_state_dict(cfg, seed=0))
    B = 4
    worker = SelfPlayWorker(cfg, net, B, num_simulations=6)
    pool = HostEnvPool(Env, B, workers=2, auto_reset=True)
    done = []
    n = loop.play_continuous(worker, pool, 9, done.append)
    assert n == len(done) == 3 + 2 + 1 + 1                    # lengths 3, 4, 5, 6 within 9 moves
    for g in done:
        idx = int(g.frames[0][0, 0, 2])
        assert len(g.values) == len(g.rewards) == 3 + idx and abs(g.policies[0].sum() - 1) < 1e-12
        assert len({int(f[0, 0, 0]) for f in g.frames}) == 1                       # one episode's frames only
        assert [int(f[0, 0, 1]) for f in g.frames] == list(range(3 + idx))
    obs = worker.ring.observation(torch.float32).cpu().numpy()                    # [B, 96, 96, 96]
    for i in range(B):
        episodes = np.round(obs[i, 0, 0, 0::3] * 255).astype(int)                 # channel 0 of the 32 frames
        # the frames of the final environment step are not in the stack yet (the next move pushes them)
        assert len(set(episodes.tolist())) == 1 and episodes[0] == pool.envs[i].ep - int(pool.restarted[i])
    pool.close()


def test_device_resident_frames_give_the_reference_observations():
    """EpisodeStore.observations (one gather kernel over device-resident frames) against Game.get_observation
    restated on the host (the stack of the 32 frames before a position, first frame repeated) followed by the
    oracle's prepare_observation: bit for bit, including positions near the start of an episode, the window's
    eviction and the store's compaction offsets."""
    from muzero_amd.learner import EpisodeStore, GameHistory
    from oracle import prep as oprep
    rng = np.random.default_rng(4)
    A, D = 4, 32
    store = EpisodeStore(A, 0.997)
    games = []
    for T in (1, 5, 40, 33):
        g = GameHistory(A, 0.997, frame_stack=D)
        for t in range(T):
            g.store_transition(None if t == 0 else int(rng.integers(0, A)), rng.integers(0, 256, (96, 96, 3), dtype=np.uint8),
                               float(rng.normal()))
            g.values.append(float(rng.normal()))
            g.policies.append(rng.dirichlet(np.ones(A)))
        games.append(g)
        store.add_game(g)
    q = [(0, 0), (1, 0), (1, 4), (2, 0), (2, 30), (2, 31), (2, 39), (3, 32), (3, 7)]
    got = store.observations([a for a, _ in q], [b for _, b in q]).cpu().numpy()
    for i, (gi, p) in enumerate(q):
        want = oprep.prepare(games[gi].get_observation(p))               # HWC uint8 -> CHW float32 / 255
        assert np.array_equal(got[i], want), (gi, p)
    store.pop_front()                                                     # game 0 leaves the window
    got = store.observations([0, 2], [3, 10]).cpu().numpy()
    assert np.array_equal(got[0], oprep.prepare(games[1].get_observation(3)))
    assert np.array_equal(got[1], oprep.prepare(games[3].get_observation(10)))
    with pytest.raises(IndexError):
        store.observations([0], [5])                                      # game 1 has 5 positions: 0..4
