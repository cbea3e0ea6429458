"""The convolution kernels of the search path write nowhere but their outputs: every output tensor sits in the
middle of a larger buffer whose margins hold a canary, at batches that leave workgroups a partial share (a ragged
count, a lone last image for the two-image kernel, a second pass)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = {"stem": (96, 96, 256, 2), "down0": (48, 256, 128, 2), "mid": (24, 128, 128, 1),
          "down1": (24, 128, 256, 2), "trunk": (12, 256, 256, 1)}


def guarded(shape, dtype, guard=8192):
    """A tensor of `shape` between two canary margins of one buffer; returns (tensor, check)."""
    n = int(np.prod(shape))
    canary = 1234.5 if dtype.is_floating_point else 77
    buf = torch.full((n + 2 * guard,), canary, dtype=dtype, device="cuda")
    t = buf[guard:guard + n].view(shape)
    t.zero_()

    def check():
        assert bool((buf[:guard] == canary).all()) and bool((buf[guard + n:] == canary).all()), "write outside the tensor"
    return t, check


@pytest.mark.parametrize("name", list(SHAPES))
@pytest.mark.parametrize("B", [1, 5])
def test_single_layer_kernels(name, B):
    from muzero_amd import _lib, mfma
    H, cin, cout, s = SHAPES[name]
    g = torch.Generator().manual_seed(B)
    chunk, whole = mfma.kernel_info(H, H, cin, cout, s)
    wpk = mfma.pack_weights(torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5), chunk).cuda()
    bias = torch.zeros(cout, device="cuda")
    x = torch.randn(B, H, H, cin, generator=g).to(torch.bfloat16).cuda()
    checks = []
    out, c = guarded((B, H // s, H // s, cout), torch.bfloat16); checks.append(c)
    mean = None
    if whole:
        mean, c = guarded((B, cout), torch.float32); checks.append(c)
    res = torch.randn(B, H // s, H // s, cout, generator=g).to(torch.bfloat16).cuda()
    _lib.call("mz_conv3x3_bf16", _lib.ptr(x), None, 0, _lib.ptr(wpk), _lib.ptr(bias), _lib.ptr(res), _lib.ptr(out), None,
              None, _lib.ptr(mean), B, H, H, cin, cout, s, 1, _lib.current_stream())
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out.float()).all())
    for c in checks:
        c()


@pytest.mark.parametrize("B", [3, 257, 301])
def test_chain_kernels(B):
    """The fused dynamics chain writing into a slot of a guarded pool (3: one image per workgroup; 257 and 301:
    the two-image kernel with a lone last image and second passes), and the 24x24 chain."""
    from muzero_amd import _lib, mfma
    g = torch.Generator().manual_seed(B)
    layers = []
    for _ in range(5):
        w = torch.randn(256, 256, 3, 3, generator=g) / 48.0
        layers.append((mfma.pack_weights(w, 256).cuda(), (torch.randn(256, generator=g) * 0.1).cuda()))
    specs = [(0, 1, 0), (1, 1, 0), (2, 1, 1), (3, 1, 0), (4, 1, 1)]
    arr = (_lib.ChainLayer * 5)()
    for i, (ci, relu, res) in enumerate(specs):
        arr[i] = _lib.ChainLayer(layers[ci][0].data_ptr(), layers[ci][1].data_ptr(), relu, res)
    pool, check_pool = guarded((3, B, 12, 12, 256), torch.bfloat16)
    pool[:2] = torch.randn(2, B, 12, 12, 256, generator=g).to(torch.bfloat16).cuda()
    slot = torch.randint(0, 2, (B,), generator=g, dtype=torch.int32).cuda()
    acts = torch.randint(0, 5, (B,), generator=g, dtype=torch.int32).cuda()
    avec = (torch.randn(5, 256, generator=g) * 0.5).cuda()
    mean, check_mean = guarded((B, 256), torch.float32)
    before = pool[:2].clone()
    _lib.call("mz_conv_chain_bf16", _lib.ptr(pool), _lib.ptr(slot), B, arr, 5, _lib.ptr(pool[2]), _lib.ptr(avec),
              _lib.ptr(acts), _lib.ptr(mean), B, _lib.current_stream())
    torch.cuda.synchronize()
    check_pool(); check_mean()
    assert torch.equal(pool[:2], before)                     # the parents' slots are read, never written
    assert bool(torch.isfinite(pool[2].float()).all()) and bool((pool[2].float().abs().sum() > 0))
    if B <= 5:
        mids = []
        for _ in range(4):
            w = torch.randn(128, 128, 3, 3, generator=g) / 34.0
            mids.append((mfma.pack_weights(w, 128).cuda(), torch.zeros(128, device="cuda")))
        marr = (_lib.ChainLayer * 4)()
        for i in range(4):
            marr[i] = _lib.ChainLayer(mids[i][0].data_ptr(), mids[i][1].data_ptr(), 1, i % 2)
        x = torch.randn(B, 24, 24, 128, generator=g).to(torch.bfloat16).cuda()
        out, check_out = guarded((B, 24, 24, 128), torch.bfloat16)
        _lib.call("mz_conv_chain_mid_bf16", _lib.ptr(x), marr, 4, _lib.ptr(out), B, _lib.current_stream())
        torch.cuda.synchronize()
        check_out()


@pytest.mark.parametrize("B", [1, 7])
def test_rotating_observation_and_first_convolution(B):
    """mz_obs_slot_write into a guarded rotating observation for every slot, then the first convolution reading it
    into a guarded output."""
    from muzero_amd import _lib, mfma
    from muzero_amd.prep import FrameRing
    g = torch.Generator().manual_seed(B)
    ring = FrameRing(B)
    ring.enable_rotating(torch.bfloat16)
    shape = tuple(ring.rot.shape)
    rot, check_rot = guarded(shape, torch.bfloat16)
    ring.rot = rot                                            # the ring writes into the guarded tensor from here on
    frames = torch.randint(0, 256, (B, 96, 96, 3), generator=g, dtype=torch.uint8)
    ring.reset(frames)
    for _ in range(33):
        ring.push(torch.randint(0, 256, (B, 96, 96, 3), generator=g, dtype=torch.uint8))
    obs, newest = ring.rotating_observation()
    assert obs.data_ptr() == rot.data_ptr()
    wpk = mfma.pack_weights(torch.randn(256, 96, 3, 3, generator=g) / 29.0, 96).cuda()
    bias = torch.zeros(256, device="cuda")
    out, check_out = guarded((B, 48, 48, 256), torch.bfloat16)
    _lib.call("mz_conv3x3_bf16_rot", _lib.ptr(obs), _lib.ptr(wpk), _lib.ptr(bias), _lib.ptr(out), B, 1, _lib.current_stream())
    torch.cuda.synchronize()
    check_rot(); check_out()
    assert bool(torch.isfinite(out.float()).all()) and float(out.float().abs().sum()) > 0


@pytest.mark.parametrize("B,A", [(1, 4), (37, 18), (130, 64)])
def test_tree_workspace_and_heads(B, A):
    """A 50-simulation search on random priors with the trees' workspace between canaries (the staged
    backup-and-descent launch and the plain kernels), and the heads kernel writing into guarded outputs."""
    import ctypes as C
    from muzero_amd import _lib
    from muzero_amd.tree import DeviceTrees
    from tests.test_bf16_gpu import _nets
    g = torch.Generator().manual_seed(B)
    sims = 50
    trees = DeviceTrees(B, A, sims + 1)
    ws, check_ws = guarded((trees.layout.total_bytes,), torch.uint8, guard=1 << 16)
    trees.ws = ws
    pri = torch.softmax(torch.randn(B, A, generator=g), dim=1).cuda()
    noise = torch.distributions.Dirichlet(torch.full((A,), 0.3)).sample((B,)).double().cuda()
    trees.root_init(pri, noise, 0.25)
    trees.select()
    for s in range(sims):
        p = torch.softmax(torch.randn(B, A, generator=g), dim=1).cuda()
        v = torch.randn(B, generator=g).cuda()
        r = torch.randn(B, generator=g).cuda()
        if s + 1 < sims:
            trees.expand_backup_select(p, v, r, 1, 0.997)
        else:
            trees.expand_backup(p, v, r, 1, 0.997)
    visits, check_v = guarded((B, A), torch.int32, guard=4096)
    rootv, check_r = guarded((B,), torch.float64, guard=4096)
    probs, check_p = guarded((B, A), torch.float64, guard=4096)
    trees.readout(visits, rootv, probs)
    torch.cuda.synchronize()
    for c in (check_ws, check_v, check_r, check_p):
        c()
    assert bool((visits.sum(1) == sims).all())
    if A in (6, 18):                                    # the heads of a full-size network (the fixture's action counts)
        cfg, ref, net = _nets(A)
        mean = torch.randn(B, 256, generator=g).cuda()
        O = 2 * net.S + 1
        outs = [guarded(shape, torch.float32, guard=4096) for shape in ((B, A), (B,), (B,), (B, A), (B, O), (B, O))]
        _lib.call("mz_heads", _lib.ptr(mean), B, net.W, net.policy.hidden, net.A, net.S, C.byref(net.policy.c),
                  C.byref(net.value_head.c), C.byref(net.reward_head.c), *[_lib.ptr(t) for t, _ in outs],
                  _lib.current_stream())
        torch.cuda.synchronize()
        for t, c in outs:
            c()
            assert bool(torch.isfinite(t).all())


def test_weight_packing_and_replay_kernels():
    """The device-side weight packing (forward fragments with their zero k-steps, the stride-2 data-gradient form,
    the heads' linear layers) and the replay sampler / priority write-back with every output and workspace between
    canaries, at sizes that are not multiples of the kernels' block shapes."""
    from muzero_amd import _lib
    g = torch.Generator().manual_seed(9)
    checks = []
    for cout, cin, chunk in ((256, 96, 96), (128, 256, 64), (256, 256, 256)):
        w = torch.randn(cout, cin, 3, 3, generator=g).cuda()
        n = _lib.lib().mz_packed_conv_elems(cout, cin, 0)
        out, c = guarded((n,), torch.bfloat16); checks.append(c)
        bias_out, c = guarded((cout,), torch.float32); checks.append(c)
        cb = torch.zeros(cout, device="cuda")
        _lib.call("mz_weights_fold_pack", _lib.ptr(w), _lib.ptr(cb), None, None, None, None, 1e-5, cout, cin, chunk, 0, 0, 0,
                  _lib.ptr(out), _lib.ptr(bias_out), _lib.current_stream())
    for cout, cin in ((128, 256), (256, 128)):
        w = torch.randn(cout, cin, 3, 3, generator=g).cuda()
        out, c = guarded((_lib.lib().mz_packed_dgrad_s2_elems(cout, cin),), torch.bfloat16); checks.append(c)
        _lib.call("mz_weights_pack_dgrad_s2", _lib.ptr(w), cout, cin, _lib.ptr(out), _lib.current_stream())
    for O, K in ((18, 256), (601, 256)):
        w = torch.randn(O, K, generator=g).cuda()
        b = torch.randn(O, generator=g).cuda()
        tiles = (O + 15) // 16
        wp, c = guarded((tiles * 16 * K,), torch.float16); checks.append(c)
        bp, c = guarded((tiles * 16,), torch.float32); checks.append(c)
        _lib.call("mz_pack_linear", _lib.ptr(w), _lib.ptr(b), O, K, 2, _lib.ptr(wp), _lib.ptr(bp), _lib.current_stream())
    # replay: 100 003 positions in 37 games, 1000 draws
    n, games, batch = 100003, 37, 1000
    prio, c = guarded((n,), torch.float64); checks.append(c)
    prio.copy_(torch.rand(n, generator=g, dtype=torch.float64).cuda())
    cuts = np.sort(np.random.default_rng(1).choice(np.arange(1, n), games - 1, replace=False))
    starts = torch.tensor(np.concatenate([[0], cuts, [n]]), dtype=torch.int64, device="cuda")
    u = torch.rand(batch, generator=g, dtype=torch.float64).cuda()
    flat, c = guarded((batch,), torch.int32); checks.append(c)
    game, c = guarded((batch,), torch.int32); checks.append(c)
    pos, c = guarded((batch,), torch.int32); checks.append(c)
    wts, c = guarded((batch,), torch.float64); checks.append(c)
    need = _lib.lib().mz_replay_workspace_bytes(n)
    ws, c = guarded(((need + 7) // 8,), torch.float64); checks.append(c)
    _lib.call("mz_replay_sample", _lib.ptr(prio), n, _lib.ptr(u), batch, 1.0, _lib.ptr(starts), games, _lib.ptr(flat),
              _lib.ptr(game), _lib.ptr(pos), _lib.ptr(wts), _lib.ptr(ws), need, _lib.current_stream())
    vals = torch.rand(batch, generator=g, dtype=torch.float64).cuda()
    _lib.call("mz_replay_update", _lib.ptr(prio), n, _lib.ptr(flat), _lib.ptr(vals), batch, 1.0, _lib.current_stream())
    torch.cuda.synchronize()
    for c in checks:
        c()
    assert bool(((flat >= 0) & (flat < n)).all())
