"""Training-side host logic against fixtures generated from the reference
(tests/golden/make_golden_train.py): episode targets, the unrolled loss and its gradients,
the replay buffer.  CPU only: the loss runs through its torch implementation here; the HIP
implementation is checked against the same fixtures in test_learner_gpu.py."""
import numpy as np
import pytest
import torch

from tests.helpers import load
from oracle.loss import support_ce

from muzero_amd.config import MuZeroConfig
from muzero_amd.learner import (GameHistory, Learner, ReplayBuffer, compute_priorities, support_ce_loss,
                                unroll_actions)


def build_games(g8):
    games = []
    A, gamma = int(g8["A"]), float(g8["gamma"])
    for gi in range(int(g8["n_games"])):
        g = GameHistory(A, gamma)
        g.rewards = list(g8[f"g{gi}_rewards"])
        g.values = list(g8[f"g{gi}_values"])
        g.policies = list(g8[f"g{gi}_policies"])
        g.actions = list(g8[f"g{gi}_actions"])
        games.append(g)
    return games


def test_make_target_matches_reference():
    g8 = load("g8_targets")
    K, td = int(g8["K"]), int(g8["td"])
    for gi, g in enumerate(build_games(g8)):
        for s in range(len(g.rewards)):
            for k, (v, r, p) in enumerate(g.make_target(s, K, td)):
                assert v == g8[f"g{gi}_tv"][s, k]          # same operations in the same order: bit-exact
                assert r == g8[f"g{gi}_tr"][s, k]
                assert np.array_equal(p, g8[f"g{gi}_tp"][s, k])


def test_store_calls_and_frame_stack():
    g = GameHistory(4, 0.997, frame_stack=4)
    rng = np.random.default_rng(0)
    frames = [rng.integers(0, 256, (5, 5, 3), dtype=np.uint8) for _ in range(7)]
    stack = [frames[0]] * 4
    g.store_transition(None, np.concatenate(stack, axis=2), 0)
    expect = [np.concatenate(stack, axis=2)]
    for f in frames[1:]:
        stack = stack[1:] + [f]
        g.store_transition(1, np.concatenate(stack, axis=2), 1.0)     # the reference hands over the whole stack
        expect.append(np.concatenate(stack, axis=2))
    assert g.actions[0] == 0 and g.environment_steps == 7
    for i, e in enumerate(expect):
        assert np.array_equal(g.get_observation(i), e)

    class Root:
        def value(self):
            return 1.5
    g.store_search_stats(Root(), {0: 0.25, 3: 0.75})
    assert g.values == [1.5] and np.array_equal(g.policies[0], [0.25, 0, 0, 0.75])


def g7_batch(g7):
    K = g7["target_values"].shape[1] - 1
    batch = []
    for i in range(g7["obs"].shape[0]):
        acts = [int(a) for a in g7["actions"][i, :int(g7["n_actions"][i])]]
        targets = [(float(g7["target_values"][i, t]), float(g7["target_rewards"][i, t]), g7["target_policies"][i, t])
                   for t in range(K + 1)]
        batch.append((g7["obs"][i], acts, targets))
    return batch


def g7_learner(g7, device, loss_impl):
    cfg = MuZeroConfig()
    cfg.observation_shape, cfg.action_space_size, cfg.support_size = (12, 32, 32), 5, 10
    cfg.num_channels, cfg.num_blocks, cfg.downsample_steps = 16, 2, [(8, 2), (16, 3)]
    cfg.fc_reward_layers = cfg.fc_value_layers = cfg.fc_policy_layers = [16]
    cfg.num_unroll_steps = 5
    sd = {k[3:]: torch.from_numpy(np.array(v)) for k, v in g7.items() if k.startswith("w::")}
    return Learner(cfg, device=device, state_dict=sd, loss_impl=loss_impl)


def check_against_g7(lrn, g7, rtol):
    v, r, p, per = lrn.compute_losses(g7_batch(g7), g7["weights"])
    per.mean().backward()
    np.testing.assert_allclose(v.item(), g7["value_loss"], rtol=rtol)
    np.testing.assert_allclose(r.item(), g7["reward_loss"], rtol=rtol)
    np.testing.assert_allclose(p.item(), g7["policy_loss"], rtol=rtol)
    np.testing.assert_allclose(per.detach().cpu().numpy(), g7["per_sample"], rtol=rtol)
    total = np.sqrt(sum(float(g7[k]) ** 2 for k in g7.files if k.startswith("gnorm::")))
    floor = 1e-6 * total      # a convolution bias in front of a training-mode BatchNorm has an exactly-zero
    #                           gradient in real arithmetic; what both sides hold there is rounding noise
    for name, prm in lrn.network.named_parameters():
        gn = float(g7[f"gnorm::{name}"])
        head = g7[f"ghead::{name}"]
        got = prm.grad.detach().cpu().reshape(-1)[:64].numpy()
        assert abs(prm.grad.norm().item() - gn) <= rtol * gn + floor, name
        np.testing.assert_allclose(got, head, rtol=0, atol=rtol * max(np.abs(head).max(), gn / 30) + floor, err_msg=name)
    for k, val in lrn.network.state_dict().items():          # BatchNorm running statistics moved the same way
        if "running" in k:
            np.testing.assert_allclose(val.cpu().numpy(), g7[f"bn_after::{k}"], rtol=rtol, atol=1e-6, err_msg=k)


def test_unrolled_loss_and_gradients_match_reference():
    g7 = load("g7_losses")
    check_against_g7(g7_learner(g7, "cpu", support_ce), g7, rtol=2e-5)


def test_unroll_action_order():
    got = unroll_actions([[3, 4, 1, 2, 2], [1, 2], []], 5)
    assert got.tolist() == [[0, 3, 4, 1, 2], [0, 1, 2, 0, 0], [0, 0, 0, 0, 0]]


def test_update_weights_steps_and_reports_priorities():
    g7 = load("g7_losses")
    lrn = g7_learner(g7, "cpu", support_ce)
    before = {k: v.clone() for k, v in lrn.network.state_dict().items()}
    m = lrn.update_weights(g7_batch(g7), g7["weights"])
    np.testing.assert_allclose(m["priorities"], g7["per_sample"], rtol=2e-5)
    np.testing.assert_allclose(m["total_loss"], g7["per_sample"].mean(), rtol=2e-5)
    assert lrn.training_step == 1
    # SGD with momentum 0.9, weight decay 1e-4, lr 0.05, clip 10 (train.py:156-166, :309): first step is
    # w - lr * clip(g + wd * w)
    name = "prediction.value_head.2.bias"
    g = torch.from_numpy(np.array(g7[f"ghead::{name}"]))
    total = np.sqrt(sum(float(g7[k]) ** 2 for k in g7.files if k.startswith("gnorm::")))
    scale = min(1.0, 10.0 / (total + 1e-6))
    expect = before[name][:64] - 0.05 * (g * scale + 1e-4 * before[name][:64])
    np.testing.assert_allclose(lrn.network.state_dict()[name][:64].numpy(), expect.numpy(), rtol=1e-4, atol=1e-7)
    sd = lrn.state_dict()
    assert set(sd) == set(before) and all(not v.is_cuda for v in sd.values())


def test_hip_loss_refuses_cpu_tensors():
    from muzero_amd._lib import MuZeroAmdError
    with pytest.raises(MuZeroAmdError):
        support_ce_loss(torch.zeros(2, 21), torch.zeros(2), 10, impl="hip")


def test_replay_buffer_sampling_and_window():
    class Cfg:
        replay_buffer_size, batch_size, priority_alpha, priority_beta = 2, 8, 1.0, 1.0
    g8 = load("g8_targets")
    games = build_games(g8)
    rb = ReplayBuffer(Cfg)
    rb.save_game(games[1])                                    # 5 positions, priority 1
    rb.save_game(games[2], np.arange(1, 13, dtype=float))     # 12 positions
    assert len(rb) == 2 and len(rb.priorities) == 17
    u = (np.arange(8) + 0.5) / 8
    gi, pos, flat, w = rb.sample_positions(u)
    p = np.array(rb.priorities) / sum(rb.priorities)
    cdf = np.cumsum(p); cdf /= cdf[-1]
    assert np.array_equal(flat, np.searchsorted(cdf, u, side="right"))     # np.random.choice's inversion
    assert np.array_equal(np.where(flat < 5, 0, 1), gi) and np.array_equal(np.where(flat < 5, flat, flat - 5), pos)
    wexp = (17 * p[flat]) ** -1.0
    np.testing.assert_allclose(w, wexp / wexp.max())
    rb.update_priorities(flat[:2], [0.5, 2.0])
    assert rb.priorities[flat[0]] == 0.5 + 1e-6 and rb.priorities[flat[1]] == 2.0 + 1e-6
    rb.save_game(games[3])                                    # evicts the oldest game and its priorities
    assert len(rb) == 2 and len(rb.priorities) == 12 + len(games[3].rewards)
    pr = compute_priorities(games[2], 10)
    assert pr.shape == (len(games[2].values),) and pr[0] == abs(games[2].values[0] - games[2].compute_target_value(0, 10)) + 1e-6


def test_reference_checkpoint_resumes_training(tmp_path):
    """g9_checkpoint.pt was written by the reference's save_checkpoint after two steps; resuming from it
    (weights, BN statistics, momentum) and taking the third step lands on the reference's weights."""
    import os
    from tests.helpers import GOLDEN
    from muzero_amd import checkpoint
    g7, g9 = load("g7_losses"), load("g9_outputs")
    path = os.path.join(GOLDEN, "g9_checkpoint.pt")
    ck = checkpoint.read(path)
    assert set(ck) == set(checkpoint.KEYS) and ck["training_step"] == 2
    cfg = checkpoint.config_from(ck)
    assert cfg.downsample_steps == [(8, 2), (16, 3)] and cfg.support_size == 10 and cfg.lr_init == 0.05
    lrn = Learner.from_checkpoint(path, device="cpu", loss_impl=support_ce)
    assert lrn.training_step == 2 and lrn.scheduler.last_epoch == 2
    m = lrn.update_weights(g7_batch(g7), g7["weights"])
    np.testing.assert_allclose(m["priorities"], g9["per_sample_step3"], rtol=2e-5)
    for k, v in lrn.network.state_dict().items():
        np.testing.assert_allclose(v.numpy(), g9[f"w3::{k}"], rtol=1e-4, atol=2e-6, err_msg=k)
    # and a checkpoint written here has the reference's structure and reads back the same way
    out = str(tmp_path / "ck.pt")
    lrn.save_checkpoint(out)
    ck2 = checkpoint.read(out)
    assert set(ck2) == set(ck) and ck2["training_step"] == 3
    assert set(ck2["model_state_dict"]) == set(ck["model_state_dict"])
    assert set(ck2["config"]) >= set(ck["config"]) - {"eval_interval", "eval_episodes", "num_workers"}
    again = Learner.from_checkpoint(out, device="cpu", loss_impl=support_ce)
    for (k, a), (_, b) in zip(again.network.state_dict().items(), lrn.network.state_dict().items()):
        assert torch.equal(a, b), k
    mom = lambda l: [st["momentum_buffer"] for st in l.optimizer.state_dict()["state"].values()]
    assert all(torch.equal(a, b) for a, b in zip(mom(again), mom(lrn)))


def test_checkpoint_reader_refuses_other_files(tmp_path):
    from muzero_amd import checkpoint
    p = str(tmp_path / "x.pt")
    torch.save({"something": 1}, p)
    with pytest.raises(ValueError):
        checkpoint.read(p)
