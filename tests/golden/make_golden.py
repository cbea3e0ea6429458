#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference itself.

Run once in the build container (the reference is mounted read-only at
/root/reference and is NOT available on the GPU box):

    python tests/golden/make_golden.py

It imports the reference's ``model.py`` and records inputs / expected outputs
as small ``.npz`` files; no reference source text is stored.  Fixtures:

  g1_tree_*.npz   search with a table-driven fake network: visit counts,
                  root value, per-simulation paths, min/max, action probs
  g2_nets_small   small-config network: weights, inputs, outputs
  g3_nets_full    full-size network: state-dict digest + outputs (weights are
                  re-created from the seed by the mirror modules)
  g4_decode       support_to_scalar on random and extreme logits
  g5_ucb          ucb_score on random node statistics, f32 and f64 priors
  g6_action       select_action: counts, temperature, uniform draw -> action
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import model as M  # noqa: E402  (the reference)

from oracle import nets as onets  # noqa: E402  (only for randomize_bn / digest helpers)

# ucb_score mixes np.float32 priors with Python floats (model.py:505-513): under NumPy >= 2 (NEP 50) the
# product stays float32, under NumPy 1.x it is promoted to float64 and near-tie argmaxes can differ.  The
# fixtures (and the device kernels checked against them) are the NumPy >= 2 behaviour.
assert int(np.__version__.split(".")[0]) >= 2, "generate the fixtures under NumPy >= 2"

HASH_MOD = 2147483647
T_ROWS = 1024
SMALL_S = 10


def child_hash(h: int, a: int) -> int:
    return (h * 1000003 + (a + 1) * 7919 + 12345) % HASH_MOD


class FakeNet:
    """Stand-in for MuZeroNetwork in MCTS.run: the 'hidden state' is a
    [n,1,1,1] int64 tensor holding a node hash; outputs are table rows."""

    def __init__(self, pol, val, rew):
        self.pol, self.val, self.rew = pol, val, rew

    def recurrent_inference(self, hidden, action):
        hs = hidden.view(-1).tolist()
        acts = action.view(-1).tolist()
        kids = [child_hash(h, a) for h, a in zip(hs, acts)]
        idx = torch.tensor([k % T_ROWS for k in kids], dtype=torch.long)
        return M.NetworkOutput(self.val[idx], self.rew[idx], self.pol[idx],
                               torch.tensor(kids, dtype=torch.long).view(-1, 1, 1, 1))


def make_tables(A, seed):
    g = torch.Generator().manual_seed(seed)
    pol = torch.randn(T_ROWS, A, generator=g) * 1.5
    nb = 2 * SMALL_S + 1
    # peaked logits so decoded values / rewards spread over the support
    val = torch.randn(T_ROWS, nb, generator=g)
    rew = torch.randn(T_ROWS, nb, generator=g)
    peak_v = torch.randint(0, nb, (T_ROWS,), generator=g)
    peak_r = torch.randint(SMALL_S - 3, SMALL_S + 4, (T_ROWS,), generator=g)
    val[torch.arange(T_ROWS), peak_v] += 4.0
    rew[torch.arange(T_ROWS), peak_r] += 3.0
    return pol, val, rew


def consumed_scalars(pol, val, rew):
    """What Node.expand / MCTS.run read per row, computed with the reference's
    own functions on [i:i+1] slices (model.py:397-399, :467)."""
    T, A = pol.shape
    pri = np.zeros((T, A), np.float32)
    v = np.zeros(T, np.float32)
    r = np.zeros(T, np.float32)
    for i in range(T):
        out = M.NetworkOutput(val[i:i + 1], rew[i:i + 1], pol[i:i + 1], None)
        pri[i] = torch.softmax(out.policy_logits, dim=1).detach().cpu().numpy()[0]
        r[i] = out.scalar_reward()
        v[i] = M.support_to_scalar(out.value, SMALL_S).item()
    return pri, v, r


def tree_case(name, A, sims, wave, noise_on, games, seed):
    cfg = M.MuZeroConfig()
    cfg.action_space_size = A
    cfg.support_size = SMALL_S
    cfg.num_simulations = wave
    pol, val, rew = make_tables(A, seed)
    pri, v, r = consumed_scalars(pol, val, rew)
    net = FakeNet(pol, val, rew)
    rng = np.random.default_rng(seed + 1)
    np.random.seed(seed + 2)

    dmax = sims + 1
    out = dict(A=A, sims=sims, wave=wave, noise_on=int(noise_on),
               table_priors=pri, table_value=v, table_reward=r,
               hash_mod=HASH_MOD, c_base=cfg.pb_c_base, c_init=cfg.pb_c_init,
               gamma=cfg.discount, frac=cfg.root_exploration_fraction)
    root_hash = rng.integers(1, HASH_MOD, size=games)
    noise = np.zeros((games, A), np.float64)
    visits = np.zeros((games, A), np.int32)
    child_w = np.zeros((games, A), np.float64)
    root_n = np.zeros(games, np.int32)
    root_w = np.zeros(games, np.float64)
    root_v = np.zeros(games, np.float64)
    lo = np.zeros(games, np.float64)
    hi = np.zeros(games, np.float64)
    probs = np.zeros((games, A), np.float64)
    path_len = np.zeros((games, sims), np.int32)
    path_act = -np.ones((games, sims, dmax), np.int16)

    real_dirichlet = np.random.dirichlet
    real_select = M.MCTS.select_child
    for gi in range(games):
        h0 = int(root_hash[gi])
        row = h0 % T_ROWS
        root = M.Node(0, None)
        root.expand(list(range(A)), M.NetworkOutput(
            val[row:row + 1], None, pol[row:row + 1],
            torch.tensor([h0], dtype=torch.long).view(1, 1, 1, 1)))
        if noise_on:
            drawn = []

            def rec_dirichlet(alpha, *a, **k):
                x = real_dirichlet(alpha, *a, **k)
                drawn.append(x)
                return x
            np.random.dirichlet = rec_dirichlet
            try:
                root.add_exploration_noise(cfg.root_dirichlet_alpha,
                                           cfg.root_exploration_fraction)
            finally:
                np.random.dirichlet = real_dirichlet
            noise[gi] = drawn[0]
        paths = []

        def rec_select(self, node, mm, _root=root, _paths=paths):
            a, kid = real_select(self, node, mm)
            if node is _root:
                _paths.append([])
            _paths[-1].append(a)
            return a, kid
        M.MCTS.select_child = rec_select
        try:
            mcts = M.MCTS(cfg)
            mm = M.MinMaxStats()
            for _ in range(sims // wave):
                ap = mcts.run(root, A, net, mm)
        finally:
            M.MCTS.select_child = real_select
        assert len(paths) == sims
        for si, p in enumerate(paths):
            path_len[gi, si] = len(p)
            path_act[gi, si, :len(p)] = p
        visits[gi] = [root.children[a].visit_count for a in range(A)]
        child_w[gi] = [root.children[a].value_sum for a in range(A)]
        root_n[gi] = root.visit_count
        root_w[gi] = root.value_sum
        root_v[gi] = root.value()
        lo[gi], hi[gi] = mm.minimum, mm.maximum
        probs[gi] = [ap[a] for a in range(A)]
        # the fake network must have fed the tree exactly the table scalars
        for a in range(A):
            kid = root.children[a]
            if kid.is_expanded():
                krow = child_hash(h0, a) % T_ROWS
                assert kid.reward == float(r[krow])
                assert all(kid.children[b].prior == pri[krow, b] for b in range(A))
    out.update(root_hash=root_hash, noise=noise, visits=visits, child_w=child_w,
               root_n=root_n, root_w=root_w, root_value=root_v, lo=lo, hi=hi,
               probs=probs, path_len=path_len, path_act=path_act)
    np.savez_compressed(os.path.join(HERE, f"g1_tree_{name}.npz"), **out)
    print(name, "visits[0] =", visits[0].tolist(), "depth max =", int(path_len.max()))


def small_cfg(A=5):
    cfg = M.MuZeroConfig()
    cfg.observation_shape = (12, 32, 32)
    cfg.action_space_size = A
    cfg.num_channels = 16
    cfg.downsample_steps = [(8, 2), (16, 3)]
    cfg.fc_reward_layers = [16]
    cfg.fc_value_layers = [16]
    cfg.fc_policy_layers = [16]
    cfg.support_size = SMALL_S
    return cfg


def nets_small():
    cfg = small_cfg()
    torch.manual_seed(7)
    net = M.MuZeroNetwork(cfg).eval()
    onets.randomize_bn(net, seed=3)
    g = torch.Generator().manual_seed(11)
    obs = torch.rand(3, 12, 32, 32, generator=g)
    act = torch.tensor([[0], [4], [2]])
    with torch.no_grad():
        o0 = net.initial_inference(obs)
        o1 = net.recurrent_inference(o0.hidden_state, act)
    out = {f"w::{k}": v.numpy() for k, v in net.state_dict().items()}
    out.update(obs=obs.numpy(), act=act.numpy(),
               h0=o0.hidden_state.numpy(), p0=o0.policy_logits.numpy(), v0=o0.value.numpy(),
               h1=o1.hidden_state.numpy(), p1=o1.policy_logits.numpy(), v1=o1.value.numpy(),
               r1=o1.reward.numpy())
    np.savez_compressed(os.path.join(HERE, "g2_nets_small.npz"), **out)
    print("g2 ok", o0.hidden_state.shape)


def nets_full():
    out = {}
    for A in (6, 18):
        cfg = M.MuZeroConfig()
        cfg.observation_shape = (96, 96, 12)  # as train.py:632-634 leaves it: [0] = 96 input channels
        cfg.action_space_size = A
        torch.manual_seed(0)
        net = M.MuZeroNetwork(cfg).eval()
        out[f"digest_init_A{A}"] = onets.state_digest(net)
        onets.randomize_bn(net, seed=1)
        out[f"digest_bn_A{A}"] = onets.state_digest(net)
        frames = np.random.default_rng(1234).integers(0, 256, size=(2, 96, 96, 96), dtype=np.uint8)
        obs = torch.tensor(np.transpose(frames, (0, 3, 1, 2)) / 255.0, dtype=torch.float32)
        act = torch.tensor([[1], [A - 1]])
        with torch.no_grad():
            o0 = net.initial_inference(obs)
            o1 = net.recurrent_inference(o0.hidden_state, act)
            o2 = net.recurrent_inference(o1.hidden_state, act.flip(0))
        S = cfg.support_size
        out.update({
            f"A{A}_h0": o0.hidden_state.numpy()[:1], f"A{A}_p0": o0.policy_logits.numpy(),
            f"A{A}_v0": o0.value.numpy(),
            f"A{A}_h1": o1.hidden_state.numpy()[:1], f"A{A}_p1": o1.policy_logits.numpy(),
            f"A{A}_v1": o1.value.numpy(), f"A{A}_r1": o1.reward.numpy(),
            f"A{A}_p2": o2.policy_logits.numpy(), f"A{A}_v2": o2.value.numpy(),
            f"A{A}_r2": o2.reward.numpy(),
            f"A{A}_sv0": M.support_to_scalar(o0.value, S).numpy(),
            f"A{A}_sv1": M.support_to_scalar(o1.value, S).numpy(),
            f"A{A}_sr1": M.support_to_scalar(o1.reward, S).numpy(),
            f"A{A}_h0_absmean": np.array(o0.hidden_state.abs().mean().item()),
        })
        print("g3", A, out[f"digest_init_A{A}"][:16], float(o0.hidden_state.abs().mean()))
    np.savez_compressed(os.path.join(HERE, "g3_nets_full.npz"), **out)


def decode_cases():
    g = torch.Generator().manual_seed(5)
    S = 300
    nb = 2 * S + 1
    x = torch.randn(96, nb, generator=g)
    x[:32] *= 4.0
    for i in range(32, 64):  # single dominant bin, incl. the two ends and the centre
        x[i, (i - 32) * (nb - 1) // 31] += 30.0
    x[64] = 0.0              # uniform
    x[65] = -80.0; x[65, 0] = 80.0
    x[66] = -80.0; x[66, nb - 1] = 80.0
    x[67] = torch.linspace(-20, 20, nb)
    y = M.support_to_scalar(x, S)
    xs = torch.randn(40, 2 * SMALL_S + 1, generator=g) * 3
    ys = M.support_to_scalar(xs, SMALL_S)
    np.savez_compressed(os.path.join(HERE, "g4_decode.npz"), logits=x.numpy(), scalar=y.numpy(),
                        logits_small=xs.numpy(), scalar_small=ys.numpy())
    print("g4 range", float(y.min()), float(y.max()))


def ucb_cases(n=4000):
    rng = np.random.default_rng(42)
    cfg = M.MuZeroConfig()
    mcts = M.MCTS(cfg)
    pn = rng.integers(0, 60, n)
    cn = rng.integers(0, 12, n)
    cn[rng.random(n) < 0.3] = 0
    cw = rng.normal(0, 5, n) * np.maximum(cn, 1)
    lo = rng.normal(-2, 2, n)
    hi = lo + np.abs(rng.normal(0, 3, n))
    same = rng.random(n) < 0.1
    hi[same] = lo[same]
    unset = rng.random(n) < 0.05
    p32 = rng.random(n).astype(np.float32)
    p64 = rng.random(n)
    s32 = np.zeros(n, np.float32)
    s64 = np.zeros(n, np.float64)
    for i in range(n):
        mm = M.MinMaxStats()
        if not unset[i]:
            mm.minimum, mm.maximum = float(lo[i]), float(hi[i])
        par = M.Node(0, None); par.visit_count = int(pn[i])
        for prior, dst in ((p32[i], s32), (p64[i], s64)):
            kid = M.Node(prior, None)
            kid.visit_count = int(cn[i]); kid.value_sum = float(cw[i])
            sc = mcts.ucb_score(par, kid, mm)
            assert type(sc) is (np.float32 if dst is s32 else np.float64), type(sc)
            dst[i] = sc
    np.savez_compressed(os.path.join(HERE, "g5_ucb.npz"), pn=pn, cn=cn, cw=cw, lo=lo, hi=hi,
                        unset=unset, p32=p32, p64=p64, s32=s32, s64=s64,
                        c_base=cfg.pb_c_base, c_init=cfg.pb_c_init)
    print("g5 ok")


def action_cases(n=600):
    rng = np.random.default_rng(77)
    cfg = M.MuZeroConfig()
    A_list = rng.choice([4, 6, 9, 18], n)
    temps = rng.choice([0.0, 1.0, 0.5, 0.25], n)
    counts = np.zeros((n, 18), np.int32)
    u = np.zeros(n, np.float64)
    act = np.zeros(n, np.int32)
    for i in range(n):
        A = int(A_list[i])
        c = rng.multinomial(50, rng.dirichlet(np.ones(A) * 0.5))
        if i % 7 == 0:  # ties
            c[:] = 0; c[rng.integers(0, A, 2)] = 25
        counts[i, :A] = c
        root = M.Node(0, None)
        for a in range(A):
            root.children[a] = M.Node(0.0, None)
            root.children[a].visit_count = int(c[a])
        np.random.seed(1000 + i)
        u[i] = np.random.random_sample()
        np.random.seed(1000 + i)
        act[i] = M.select_action(cfg, root, float(temps[i]))
        if temps[i] != 0:  # exactly one uniform draw was consumed
            nxt = np.random.random_sample()
            np.random.seed(1000 + i); np.random.random_sample()
            assert nxt == np.random.random_sample()
    np.savez_compressed(os.path.join(HERE, "g6_action.npz"), A=A_list, temps=temps,
                        counts=counts, u=u, action=act)
    print("g6 ok")


TREE_CASES = {
    # name: (A, sims, wave, noise, games, seed)
    "a6_w1_noise": (6, 50, 1, True, 8, 100),
    "a18_w1_noise": (18, 50, 1, True, 8, 200),
    "a9_w1_plain": (9, 50, 1, False, 8, 300),
    "a6_w50_noise": (6, 50, 50, True, 4, 400),
    "a18_w5_noise": (18, 50, 5, True, 4, 500),
    "a4_w1_deep": (4, 200, 1, True, 4, 600),
    "a9_w2_noise": (9, 50, 2, True, 4, 700),
    "a64_w1_noise": (64, 80, 1, True, 3, 800),
}

if __name__ == "__main__":
    torch.set_num_threads(8)
    only = sys.argv[1:]            # names of tree cases to (re)generate; none = everything
    for name, spec in TREE_CASES.items():
        if not only or name in only:
            tree_case(name, *spec)
    if not only:
        decode_cases()
        ucb_cases()
        action_cases()
        nets_small()
        nets_full()
