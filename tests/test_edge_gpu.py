"""Edge cases of the device path: ragged batch sizes, extreme action counts, full trees,
error behaviour of the reference surface."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.helpers import TableNet, child_hash, load

pytestmark = pytest.mark.gpu


def _fake_search(B, A, sims, wave, seed, noisy=True):
    """Device trees driven by a hash-table network with A actions (table built here)."""
    from muzero_amd.tree import DeviceTrees
    from oracle import tree as otree
    rng = np.random.default_rng(seed)
    rows = 257
    logits = rng.normal(0, 1.5, (rows, A)).astype(np.float32)
    pri = np.exp(logits - logits.max(1, keepdims=True))
    pri = (pri / pri.sum(1, keepdims=True)).astype(np.float32)
    val = rng.normal(0, 3, rows).astype(np.float32)
    rew = rng.normal(0, 1, rows).astype(np.float32)
    root_hash = rng.integers(1, 2147483647, size=B).astype(np.int64)
    noise = rng.dirichlet(np.full(A, 0.3), size=B) if noisy and A > 1 else None
    steps = sims // wave
    trees = DeviceTrees(B, A, steps + 1, max_visits=sims)
    node_hash = np.zeros((B, steps + 1), np.int64)
    node_hash[:, 0] = root_hash
    trees.root_init(torch.from_numpy(pri[root_hash % rows]).cuda(),
                    None if noise is None else torch.from_numpy(noise).cuda(), 0.25)
    ar = np.arange(B)
    for s in range(steps):
        trees.select()
        parent = trees.array("leaf_parent").cpu().numpy()
        act = trees.array("leaf_action").cpu().numpy()
        kid = child_hash(node_hash[ar, parent], act.astype(np.int64))
        node_hash[:, s + 1] = kid
        r = kid % rows
        trees.expand_backup(torch.from_numpy(pri[r]).cuda(), torch.from_numpy(val[r]).cuda(),
                            torch.from_numpy(rew[r]).cuda(), wave, 0.997)
    visits, root_value, probs = (t.cpu().numpy() for t in trees.readout())
    assert int(trees.array("status").abs().sum()) == 0

    def infer(hiddens, actions):
        out = []
        for h, a in zip(hiddens, actions):
            k = int(child_hash(int(h), int(a)))
            out.append((pri[k % rows], float(val[k % rows]), float(rew[k % rows]), k))
        return out
    for b in range(B):
        root, _ = otree.search(pri[root_hash[b] % rows], int(root_hash[b]), infer, sims, wave,
                               noise=None if noise is None else noise[b])
        assert otree.visit_counts(root).tolist() == visits[b].tolist(), (B, A, b)
        assert root.q() == root_value[b]
        assert otree.action_probs(root).tolist() == probs[b].tolist()


@pytest.mark.parametrize("B,A,sims,wave", [(1, 6, 50, 1), (3, 18, 20, 1), (5, 1, 12, 1), (7, 64, 30, 1),
                                           (2, 2, 64, 1), (6, 33, 40, 4)])
def test_ragged_batches_and_extreme_action_counts(B, A, sims, wave):
    _fake_search(B, A, sims, wave, seed=B * 100 + A)


def test_full_tree_is_reported_not_overrun():
    from muzero_amd.tree import DeviceTrees
    B, A = 2, 4
    trees = DeviceTrees(B, A, 3, max_visits=10)   # room for the root + 2 expansions
    pri = torch.full((B, A), 0.25, device="cuda")
    z = torch.zeros(B, device="cuda")
    trees.root_init(pri, None)
    for _ in range(2):
        trees.select()
        trees.expand_backup(pri, z, z, 1, 0.997)
    assert trees.array("num_nodes").tolist() == [3, 3] and trees.array("status").tolist() == [0, 0]
    trees.select()
    trees.expand_backup(pri, z, z, 1, 0.997)      # no free node: flagged, nothing written out of range
    torch.cuda.synchronize()
    assert trees.array("status").tolist() == [2, 2]
    assert trees.array("num_nodes").tolist() == [3, 3]
    assert trees.array("root_visit").tolist() == [2, 2]


def test_select_action_ties_and_zero_temperature():
    from muzero_amd.tree import select_actions
    counts = torch.tensor([[0, 5, 5, 0], [7, 7, 7, 7], [0, 0, 0, 9], [1, 0, 0, 0]], dtype=torch.int32, device="cuda")
    assert select_actions(counts, 0.0).tolist() == [1, 0, 3, 0]              # first maximum (np.argmax)
    u = torch.tensor([0.0, 0.999999, 0.5, 0.3], dtype=torch.float64, device="cuda")
    assert select_actions(counts, 1.0, u).tolist() == [1, 3, 3, 0]           # zero-count actions never drawn


def test_run_on_unexpanded_root_raises_like_the_reference():
    from muzero_amd.config import MuZeroConfig
    from muzero_amd.search import MCTS, MinMaxStats, Node
    cfg = MuZeroConfig()
    with pytest.raises(IndexError):
        MCTS(cfg).run(Node(0, None), 6, None, MinMaxStats())


@pytest.mark.parametrize("B", [1, 37, 300])
def test_chain_kernel_any_batch(B):
    """Persistent workgroups: fewer images than CUs, and a count that does not divide evenly."""
    from muzero_amd import _lib, mfma
    g = torch.Generator().manual_seed(B)
    w = [mfma.pack_weights(torch.randn(256, 256, 3, 3, generator=g) / 48.0, 256).cuda() for _ in range(3)]
    bias = [(torch.randn(256, generator=g) * 0.1).cuda() for _ in range(3)]
    x = torch.randn(B, 12, 12, 256, generator=g).to(torch.bfloat16).cuda()
    arr = (_lib.ChainLayer * 3)()
    for i in range(3):
        arr[i] = _lib.ChainLayer(w[i].data_ptr(), bias[i].data_ptr(), 1, 1 if i == 2 else 0)
    out = torch.empty_like(x)
    _lib.call("mz_conv_chain_bf16", _lib.ptr(x), None, 0, arr, 3, _lib.ptr(out), None, None, None, B,
              _lib.current_stream())
    # layer by layer with the single-layer kernel
    class L:
        pass
    t = [torch.empty_like(x) for _ in range(3)]
    src, res = x, None
    for i in range(3):
        layer = L()
        layer.cin = layer.cout = 256
        layer.stride = 1
        layer.dev = {"w": w[i], "bias": bias[i]}
        mfma.conv(layer, src, t[i], B, 12, 12, True, t[0] if i == 2 else None, None, 0, None, None, None)
        src = t[i]
    torch.cuda.synchronize()
    assert torch.equal(out, t[2])      # same arithmetic in both kernels: bit-identical


@pytest.mark.parametrize("B", [257, 301, 512, 600, 1024])
def test_chain_kernel_two_images_per_pass_equals_one_per_pass(B):
    """With more images than compute units, workgroups take the images two at a time (shared weight
    fragments, layers written over their own input): the same arithmetic in the same order, so the
    result equals, bit for bit, that of the same images sent in batches small enough for the
    one-image kernel - with and without residual layers, with a single image left over (B = 600),
    spatial means included."""
    from muzero_amd import _lib, mfma
    n = 5
    g = torch.Generator().manual_seed(B)
    w = [mfma.pack_weights(torch.randn(256, 256, 3, 3, generator=g) / 48.0, 256).cuda() for _ in range(n)]
    bias = [(torch.randn(256, generator=g) * 0.1).cuda() for _ in range(n)]
    x = torch.randn(B, 12, 12, 256, generator=g).to(torch.bfloat16).cuda()

    def chain(inp, res_flags):
        arr = (_lib.ChainLayer * n)()
        for i in range(n):
            arr[i] = _lib.ChainLayer(w[i].data_ptr(), bias[i].data_ptr(), 1 if i != 1 else 0, res_flags[i])
        out, mean = torch.empty_like(inp), torch.empty(inp.shape[0], 256, device="cuda")
        _lib.call("mz_conv_chain_bf16", _lib.ptr(inp), None, 0, arr, n, _lib.ptr(out), None, None, _lib.ptr(mean),
                  inp.shape[0], _lib.current_stream())
        torch.cuda.synchronize()
        return out, mean

    for flags in ([0] * n, [0, 0, 1, 0, 1]):
        out, mean = chain(x, flags)
        parts = [chain(x[i:i + 200].contiguous(), flags) for i in range(0, B, 200)]      # 200 < CUs: one image per pass
        assert torch.equal(out, torch.cat([p[0] for p in parts]))
        assert torch.equal(mean, torch.cat([p[1] for p in parts]))
        assert out.float().abs().max() > 0


def test_chain_kernel_repeats_bit_for_bit():
    """Twenty launches of the dynamics-shaped chain at the benchmark's batch size on the same input:
    every result equals the first (no race between wavefronts, no read of stale LDS)."""
    from muzero_amd import _lib, mfma
    B, n = 1024, 5
    g = torch.Generator().manual_seed(7)
    w = [mfma.pack_weights(torch.randn(256, 256, 3, 3, generator=g) / 48.0, 256).cuda() for _ in range(n)]
    bias = [(torch.randn(256, generator=g) * 0.1).cuda() for _ in range(n)]
    x = torch.randn(B, 12, 12, 256, generator=g).to(torch.bfloat16).cuda()
    arr = (_lib.ChainLayer * n)()
    for i in range(n):
        arr[i] = _lib.ChainLayer(w[i].data_ptr(), bias[i].data_ptr(), 1, 1 if (i % 2 == 0 and i > 0) else 0)
    outs = [torch.empty_like(x) for _ in range(20)]
    means = [torch.empty(B, 256, device="cuda") for _ in range(20)]
    for o, m in zip(outs, means):
        _lib.call("mz_conv_chain_bf16", _lib.ptr(x), None, 0, arr, n, _lib.ptr(o), None, None, _lib.ptr(m), B,
                  _lib.current_stream())
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0].float()).all() and outs[0].float().abs().max() > 0
    for o, m in zip(outs[1:], means[1:]):
        assert torch.equal(o, outs[0]) and torch.equal(m, means[0])


def test_rotating_observation_equals_the_stacked_one():
    """The rotating, tiled form of the frame stack (3 values rewritten per pixel and move) is the
    stacked observation with its channels permuted, bit for bit, after a reset and after any number
    of pushes (ring wrap-around included); the first convolution reading it directly through equally
    permuted input-channel weights gives the same hidden state up to the order of its float32
    partial sums."""
    from muzero_amd.prep import FrameRing, untile
    from tests.test_bf16_gpu import _nets
    rng = np.random.default_rng(5)
    B = 3
    ring = FrameRing(B, 32, 96, 96)
    ring.enable_rotating(torch.bfloat16)
    frames = torch.from_numpy(rng.integers(0, 256, (B, 96, 96, 3), dtype=np.uint8))
    ring.reset(frames)
    cfg, ref, net = _nets(6)
    for n_push in (0, 1, 30, 5):          # 0, 1, 31 and 36 pushes in total: the last wraps the ring
        for _ in range(n_push):
            ring.push(torch.from_numpy(rng.integers(0, 256, (B, 96, 96, 3), dtype=np.uint8)))
        rot, newest = ring.rotating_observation()
        true = ring.observation(torch.bfloat16)
        perm = ring.rotation_permutation().cuda()
        assert torch.equal(true[..., perm], untile(rot))
        _check_rot_halo(rot)
        h_true = torch.empty(B, 12, 12, 256, dtype=torch.bfloat16, device="cuda")
        h_rot = torch.empty_like(h_true)
        mean = torch.empty(B, 256, dtype=torch.float32, device="cuda")
        net.represent(true, h_true, mean)
        net.represent(rot, h_rot, mean, rotation=newest)
        d = (h_true.float() - h_rot.float()).abs().max().item()
        assert d <= 0.02 * h_true.float().abs().max().item()


@pytest.mark.parametrize("rot_dtype", [torch.bfloat16, torch.float32])
def test_per_game_restart_refills_only_the_flagged_stacks(rot_dtype):
    """push(frames, restarted): flagged games get the reset rule (every slot = the new frame), the
    others an ordinary push; both the u8 ring's stacked observation and the rotating one agree
    with a host model of 5 independent frame stacks."""
    from muzero_amd.prep import FrameRing, untile
    rng = np.random.default_rng(9)
    B, D = 5, 32
    ring = FrameRing(B, D, 96, 96)
    ring.enable_rotating(rot_dtype)
    new = lambda: rng.integers(0, 256, (B, 96, 96, 3), dtype=np.uint8)
    f = new()
    ring.reset(torch.from_numpy(f))
    stacks = [[f[b]] * D for b in range(B)]                  # host model: oldest first
    for step in range(40):
        f = new()
        mask = rng.random(B) < 0.15 if step % 3 else np.zeros(B, bool)
        ring.push(torch.from_numpy(f), torch.from_numpy(mask) if mask.any() else None)
        for b in range(B):
            stacks[b] = [f[b]] * D if mask[b] else stacks[b][1:] + [f[b]]
    want = np.stack([np.concatenate(s, axis=2) for s in stacks])            # [B, 96, 96, 96] uint8
    got = ring.observation(torch.float32).cpu().numpy()
    assert np.array_equal(got, (want.astype(np.float32) / np.float32(255.0)))
    rot, newest = ring.rotating_observation()
    perm = ring.rotation_permutation().cuda()
    assert rot.dtype == rot_dtype
    assert torch.equal(ring.observation(rot_dtype)[..., perm], untile(rot))
    _check_rot_halo(rot)


def _check_rot_halo(rot):
    """Every (row, tile, slot) run of the rotating observation starts with the last pixel of the tile to its
    left (zero for a row's first tile) and ends with five values of padding that stay zero."""
    assert rot.shape[-1] == 56
    assert torch.equal(rot[:, :, 1:, :, 0:3], rot[:, :, :-1, :, 48:51])
    assert not rot[:, :, 0, :, 0:3].any() and not rot[..., 51:].any()


def test_graph_replay_of_the_rotating_path_and_stale_graph_guard():
    """A HIP graph of the search on the rotating observation keeps working while the frame ring
    advances (the stem's weight copy is chosen on the device from the ring's newest slot), gives
    the eager result bit for bit, follows weights reloaded in place, and refuses to run once the
    network had to allocate new weight buffers (the graph holds the old addresses)."""
    from muzero_amd import _lib
    from muzero_amd.config import MuZeroConfig
    from muzero_amd.network import MuZeroNetwork, init_state_dict
    from muzero_amd.selfplay import SelfPlayWorker, SyntheticEnv
    cfg = MuZeroConfig()
    cfg.action_space_size, cfg.num_simulations = 6, 10
    sd = init_state_dict(cfg, seed=2, randomize_bn=True)
    net = MuZeroNetwork(cfg, precision="bf16", state_dict=sd)
    B = 7
    worker = SelfPlayWorker(cfg, net, B, 10)
    env = SyntheticEnv(B, device="cuda", pool=8)
    worker.reset(env.reset())
    worker.draw_randomness()
    obs, _ = worker.ring.rotating_observation()
    replay = worker.mcts.capture(obs, worker.noise, rotation=worker.ring.newest_dev)
    for step in range(35):                                    # wraps the ring of 32
        worker.ring.push(env.step())
        if step % 6 and step < 30:
            continue
        obs, newest = worker.ring.rotating_observation()
        eager = worker.mcts.run(obs, worker.noise, rotation=newest)
        v, q = eager.visit_counts.cpu().numpy().copy(), eager.root_values.cpu().numpy().copy()
        worker.mcts.visits.zero_()
        res = replay()
        torch.cuda.synchronize()
        assert np.array_equal(res.visit_counts.cpu().numpy(), v), step
        assert np.array_equal(res.root_values.cpu().numpy(), q), step
    # new weights of the same shapes are written into the same device buffers: the graph stays valid
    # and computes with them
    net.load_state_dict(init_state_dict(cfg, seed=3, randomize_bn=True))
    eager = worker.mcts.run(obs, worker.noise, rotation=worker.ring.newest)
    v, q = eager.visit_counts.cpu().numpy().copy(), eager.root_values.cpu().numpy().copy()
    worker.mcts.visits.zero_()
    res = replay()
    assert np.array_equal(res.visit_counts.cpu().numpy(), v) and np.array_equal(res.root_values.cpu().numpy(), q)
    # a load that has to allocate new buffers (here forced) makes the old graph refuse to run
    net._structure = None
    net.load_state_dict(init_state_dict(cfg, seed=4, randomize_bn=True))
    with pytest.raises(_lib.MuZeroAmdError, match="capture"):
        replay()
    fresh = worker.mcts.capture(obs, worker.noise, rotation=worker.ring.newest_dev)
    eager = worker.mcts.run(obs, worker.noise, rotation=worker.ring.newest)
    v = eager.visit_counts.cpu().numpy().copy()
    assert np.array_equal(fresh().visit_counts.cpu().numpy(), v)


def test_pair_observation_for_the_float32_accurate_precision():
    """FrameRing.observation(pairs=True) = the float32 observation split into (hi, lo) bf16 pairs, bit for bit;
    a self-play step in the bf16x3 precision consumes it and gives the stacked-float32-observation result."""
    from muzero_amd.config import MuZeroConfig
    from muzero_amd.network import MuZeroNetwork, init_state_dict
    from muzero_amd.search import BatchedMCTS
    from muzero_amd.selfplay import SelfPlayWorker, SyntheticEnv
    cfg = MuZeroConfig()
    cfg.action_space_size, cfg.num_simulations = 6, 8
    net = MuZeroNetwork(cfg, precision="bf16x3", state_dict=init_state_dict(cfg, seed=2, randomize_bn=True))
    B = 5
    worker = SelfPlayWorker(cfg, net, B, 8)
    assert worker.pairs and not worker.rotating
    env = SyntheticEnv(B, device="cuda", pool=8)
    worker.reset(env.reset())
    for _ in range(35):
        worker.ring.push(env.step())
    f32 = worker.ring.observation(torch.float32)
    pairs = worker.ring.observation(pairs=True)
    assert pairs.shape == (B, 2, 96, 96, 96) and pairs.dtype == torch.bfloat16
    want = net._split(f32)
    assert torch.equal(pairs.view(torch.int16), want.view(torch.int16))
    worker.draw_randomness()
    actions, res = worker.step(None, temperature=0, add_noise=True)
    v = res.visit_counts.cpu().numpy().copy()
    other = BatchedMCTS(cfg, net, B, 8).run(f32, worker.noise)
    assert np.array_equal(other.visit_counts.cpu().numpy(), v)
    assert np.array_equal(other.root_values.cpu().numpy(), res.root_values.cpu().numpy())


def test_half_results_saturate_instead_of_overflowing():
    """IEEE-half kernels store a result beyond the largest finite half as 65 504 (-65 504), never as infinity:
    single trunk layer and the two-image chain, with and without ReLU."""
    from muzero_amd import _lib, mfma
    B = 300                                     # more images than compute units: the chain takes the two-image kernel
    w = torch.full((256, 256, 3, 3), 4.0)
    w[1::2] = -4.0                              # odd output channels: large negative sums
    wp = mfma.pack_weights(w, 256, torch.float16).cuda()
    bias = torch.zeros(256, device="cuda")
    x = torch.full((B, 12, 12, 256), 32.0, dtype=torch.float16, device="cuda")     # |sum| >= 4 taps * 256 * 32 * 4 = 131 072

    class _L:
        cin, cout, stride = 256, 256, 1
        dev = {"w": wp, "bias": bias}
    out = torch.empty_like(x)
    mfma.conv(_L, x, out, B, 12, 12, False, None, None, 0, None, None, None, half=True)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    assert float(out[..., 0::2].min()) == 65504.0 and float(out[..., 1::2].max()) == -65504.0
    arr = (_lib.ChainLayer * 1)()
    arr[0] = _lib.ChainLayer(wp.data_ptr(), bias.data_ptr(), 0, 0)
    out2 = torch.empty_like(x)
    mean = torch.empty(B, 256, device="cuda")
    _lib.call("mz_conv_chain_f16", _lib.ptr(x), None, 0, arr, 1, _lib.ptr(out2), None, None, _lib.ptr(mean), B,
              _lib.current_stream())
    torch.cuda.synchronize()
    assert torch.equal(out2, out)
    arr[0] = _lib.ChainLayer(wp.data_ptr(), bias.data_ptr(), 1, 0)            # with ReLU: the negative half is zero
    _lib.call("mz_conv_chain_f16", _lib.ptr(x), None, 0, arr, 1, _lib.ptr(out2), None, None, _lib.ptr(mean), B,
              _lib.current_stream())
    torch.cuda.synchronize()
    assert float(out2[..., 0::2].min()) == 65504.0 and float(out2[..., 1::2].abs().max()) == 0.0


@pytest.mark.parametrize("precision", ["f16", "f16x2"])
def test_half_precisions_through_the_worker_and_a_captured_graph(precision):
    """The IEEE-half precisions on the self-play worker's own path: f16 reads the rotating half observation (first
    convolution through permuted weight copies), f16x2 the (hi, lo) half pairs, which equal the split of the float32
    observation bit for bit; a HIP graph of the search replays the eager result exactly while the ring advances."""
    from muzero_amd.config import MuZeroConfig
    from muzero_amd.network import MuZeroNetwork, init_state_dict
    from muzero_amd.selfplay import SelfPlayWorker, SyntheticEnv
    cfg = MuZeroConfig()
    cfg.action_space_size, cfg.num_simulations = 6, 8
    net = MuZeroNetwork(cfg, precision=precision, state_dict=init_state_dict(cfg, seed=2, randomize_bn=True))
    B = 5
    worker = SelfPlayWorker(cfg, net, B, 8)
    assert worker.rotating == (precision == "f16") and worker.pairs == (precision == "f16x2")
    env = SyntheticEnv(B, device="cuda", pool=8)
    worker.reset(env.reset())
    for _ in range(33):
        worker.ring.push(env.step())
    worker.draw_randomness()
    if precision == "f16x2":
        pairs = worker.ring.observation(torch.float16, pairs=True)
        want = net._split(worker.ring.observation(torch.float32))
        assert pairs.dtype == torch.float16 and torch.equal(pairs.view(torch.int16), want.view(torch.int16))
        obs, rotation, rot_dev = pairs, None, None
    else:
        obs, rotation = worker.ring.rotating_observation()
        assert obs.dtype == torch.float16
        rot_dev = worker.ring.newest_dev
    replay = worker.mcts.capture(obs, worker.noise, rotation=rot_dev) if precision == "f16" else \
        worker.mcts.capture(worker.obs, worker.noise)
    for _ in range(3):
        actions, res = worker.step(env.step(), temperature=0, add_noise=True)
        v, q = res.visit_counts.cpu().numpy().copy(), res.root_values.cpu().numpy().copy()
        assert (v.sum(1) == 8).all() and np.isfinite(q).all()
        worker.mcts.visits.zero_()
        if precision == "f16x2":
            worker.ring.observation(torch.float16, out=worker.obs, pairs=True)
        r2 = replay()
        torch.cuda.synchronize()
        assert np.array_equal(r2.visit_counts.cpu().numpy(), v) and np.array_equal(r2.root_values.cpu().numpy(), q)
