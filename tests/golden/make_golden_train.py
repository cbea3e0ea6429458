#!/usr/bin/env python3
"""Golden fixtures for the training-side rows (n-step targets, unrolled loss), generated from
the reference in the build container:

    python tests/golden/make_golden_train.py

``model.py`` is imported as is.  ``train.py`` cannot be imported (gymnasium / ale_py /
tensorboard are absent), so the few pure-Python definitions needed from it - the ``Game``
class, ``MuZeroTrainer.compute_losses`` and ``MuZeroTrainer.prepare_observation`` - are
compiled from their own source text, located with ``ast``, and executed here against
``model.py``'s functions.  Only inputs and outputs are stored.

  g7_losses.npz    small-config network in training mode, a batch of positions ->
                   value / reward / policy losses, per-sample loss, parameter gradients
  g8_targets.npz   Game.make_target / compute_target_value on random episodes
  g9_checkpoint.pt a checkpoint written by the reference's save_checkpoint after two SGD steps
  g9_outputs.npz   eval-mode outputs of that network, and the weights one further step gives
"""
import ast
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import model as M  # noqa: E402

from make_golden import small_cfg  # noqa: E402


def reference_definitions():
    """Game, compute_losses and prepare_observation compiled from train.py's own text."""
    path = "/root/reference/train.py"
    src = open(path).read()
    tree = ast.parse(src)
    ns = {"np": np, "torch": torch, "device": torch.device("cpu"), "logging": __import__("logging"),
          "scalar_to_support_loss": M.scalar_to_support_loss, "Node": M.Node, "os": os,
          "List": __import__("typing").List, "Tuple": __import__("typing").Tuple,
          "Dict": __import__("typing").Dict, "Optional": __import__("typing").Optional}
    out = {}
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == "Game":
            exec(compile(ast.Module([node], []), path, "exec"), ns)
            out["Game"] = ns["Game"]
        if isinstance(node, ast.ClassDef) and node.name == "MuZeroTrainer":
            keep = [f for f in node.body if isinstance(f, ast.FunctionDef)
                    and f.name in ("compute_losses", "prepare_observation", "save_checkpoint")]
            stub = ast.ClassDef(name="TrainerStub", bases=[], keywords=[], body=keep, decorator_list=[])
            mod = ast.fix_missing_locations(ast.Module([stub], []))
            exec(compile(mod, path, "exec"), ns)
            out["TrainerStub"] = ns["TrainerStub"]
    return out


def losses_fixture(defs):
    cfg = small_cfg(A=5)
    cfg.num_unroll_steps = 5
    torch.manual_seed(21)
    net = M.MuZeroNetwork(cfg)          # training mode, as train.py:155 leaves it
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    rng = np.random.default_rng(3)
    B, K, A = 4, cfg.num_unroll_steps, cfg.action_space_size
    obs = rng.integers(0, 256, size=(B, 32, 32, 12), dtype=np.uint8)
    n_act = [5, 5, 3, 0]                                     # positions near the end of a game have fewer actions
    actions = [list(rng.integers(0, A, n)) for n in n_act]
    t_val = rng.normal(0, 6, (B, K + 1))
    t_rew = rng.normal(0, 1, (B, K + 1)); t_rew[:, 0] = 0
    t_rew[1, 2] = 400.0                                      # clamps at the top of the support
    t_pol = rng.dirichlet(np.ones(A), (B, K + 1)); t_pol[3, 3:] = 0
    weights = rng.random(B) + 0.2
    batch = [(obs[i], actions[i], [(float(t_val[i, t]), float(t_rew[i, t]), t_pol[i, t]) for t in range(K + 1)])
             for i in range(B)]
    stub = defs["TrainerStub"]()
    stub.config, stub.network = cfg, net
    vl, rl, pl, per = stub.compute_losses(batch, weights)
    per.mean().backward()
    out = {f"w::{k}": v.numpy() for k, v in sd0.items()}
    out.update({f"bn_after::{k}": v.numpy() for k, v in net.state_dict().items() if "running" in k})
    grads = {k: p.grad for k, p in net.named_parameters()}
    out.update({f"gnorm::{k}": np.array(g.norm().item()) for k, g in grads.items()})
    out.update({f"ghead::{k}": g.reshape(-1)[:64].numpy() for k, g in grads.items()})
    act_pad = np.zeros((B, K), np.int64)
    for i, a in enumerate(actions):
        act_pad[i, :len(a)] = a
    out.update(obs=obs, actions=act_pad, n_actions=np.array(n_act), target_values=t_val, target_rewards=t_rew,
               target_policies=t_pol, weights=weights, value_loss=vl.detach().numpy(), reward_loss=rl.detach().numpy(),
               policy_loss=pl.detach().numpy(), per_sample=per.detach().numpy())
    np.savez_compressed(os.path.join(HERE, "g7_losses.npz"), **out)
    print("g7", float(vl), float(rl), float(pl), per.detach().numpy())
    return stub, batch, weights


def checkpoint_fixture(stub, batch, weights):
    """Two optimiser steps as train.py:303-313 does them, then the reference's save_checkpoint."""
    import shutil
    import tempfile
    cfg, net = stub.config, stub.network
    net.zero_grad()
    opt = torch.optim.SGD(net.parameters(), lr=cfg.lr_init, momentum=cfg.momentum, weight_decay=cfg.weight_decay)
    sch = torch.optim.lr_scheduler.StepLR(opt, step_size=cfg.lr_decay_steps, gamma=cfg.lr_decay_rate)

    def step():
        _, _, _, per = stub.compute_losses(batch, weights)
        opt.zero_grad()
        per.mean().backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), cfg.max_grad_norm)
        opt.step()
        sch.step()
        return per.detach().numpy()

    step(); step()
    stub.optimizer, stub.scheduler, stub.training_step = opt, sch, 2
    stub.checkpoint_dir = tempfile.mkdtemp()
    stub.save_checkpoint()
    shutil.copy(os.path.join(stub.checkpoint_dir, "checkpoint_2.pt"), os.path.join(HERE, "g9_checkpoint.pt"))
    shutil.rmtree(stub.checkpoint_dir)
    out = {}
    net.eval()
    obs = torch.stack([stub.prepare_observation(o) for o, _, _ in batch])
    with torch.no_grad():
        o0 = net.initial_inference(obs)
        o1 = net.recurrent_inference(o0.hidden_state, torch.tensor([[1], [4], [0], [2]]))
    out.update(obs=obs.numpy(), h0=o0.hidden_state.numpy(), p0=o0.policy_logits.numpy(), v0=o0.value.numpy(),
               h1=o1.hidden_state.numpy(), r1=o1.reward.numpy(), p1=o1.policy_logits.numpy(), v1=o1.value.numpy())
    net.train()
    out["per_sample_step3"] = step()                      # resumes with the momentum in the checkpoint
    out.update({f"w3::{k}": v.numpy() for k, v in net.state_dict().items()})
    np.savez_compressed(os.path.join(HERE, "g9_outputs.npz"), **out)
    print("g9 ok", out["per_sample_step3"])


def targets_fixture(defs):
    Game = defs["Game"]
    rng = np.random.default_rng(8)
    A, gamma, td, K = 6, 0.997, 10, 5
    out = dict(A=A, gamma=gamma, td=td, K=K)
    lens = [1, 4, 11, 12, 37, 60]
    for gi, T in enumerate(lens):
        g = Game(A, gamma)
        g.store_transition(None, np.zeros((1,), np.uint8), 0)
        rewards = rng.normal(0, 1, T)
        acts = rng.integers(0, A, T)
        # self-play order (train.py:227-282): search stats for the current position, then the
        # transition unless the episode ended - so values / policies are as long as rewards
        vals = rng.normal(0, 3, T + 1)
        for t in range(T + 1):
            root = M.Node(0, None)
            root.visit_count, root.value_sum = 1, float(vals[t])
            g.store_search_stats(root, {a: float(p) for a, p in enumerate(rng.dirichlet(np.ones(A)))})
            if t < T:
                g.store_transition(int(acts[t]), np.zeros((1,), np.uint8), float(rewards[t]))
        n_pos = len(g.rewards)
        tv = np.zeros((n_pos, K + 1)); tr = np.zeros((n_pos, K + 1)); tp = np.zeros((n_pos, K + 1, A))
        for s in range(n_pos):
            for k, (v, r, p) in enumerate(g.make_target(s, K, td)):
                tv[s, k], tr[s, k], tp[s, k] = v, r, p
        out[f"g{gi}_rewards"] = np.array(g.rewards, np.float64)
        out[f"g{gi}_values"] = np.array(g.values, np.float64)
        out[f"g{gi}_policies"] = np.array(g.policies, np.float64)
        out[f"g{gi}_actions"] = np.array(g.actions, np.int64)
        out[f"g{gi}_tv"], out[f"g{gi}_tr"], out[f"g{gi}_tp"] = tv, tr, tp
    out["n_games"] = len(lens)
    np.savez_compressed(os.path.join(HERE, "g8_targets.npz"), **out)
    print("g8 ok", lens)


if __name__ == "__main__":
    defs = reference_definitions()
    checkpoint_fixture(*losses_fixture(defs))
    targets_fixture(defs)
