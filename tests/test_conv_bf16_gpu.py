"""bf16 MFMA convolution kernels against a plain PyTorch fp32 reference of the
same op (inputs and weights rounded to bf16 first, fp32 accumulation)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (H, Cin, Cout, stride)
SHAPES = {"stem": (96, 96, 256, 2), "down0": (48, 256, 128, 2), "mid": (24, 128, 128, 1),
          "down1": (24, 128, 256, 2), "trunk": (12, 256, 256, 1)}


class _Layer:
    pass


def _run(name, B, relu=True, residual=False, gather=False, action=False, mean=False, seed=0, dtype=torch.bfloat16):
    """dtype: the 16-bit operand format, bfloat16 (mz_conv3x3_bf16) or float16 (mz_conv3x3_f16); the reference
    rounds inputs and weights to the same format.  Bound: 1 % + 1 % of the mean magnitude for bf16 (an output is
    itself rounded to 8 bits), 0.15 % + 0.15 % for half (11 bits)."""
    from muzero_amd import mfma
    half = dtype == torch.float16
    tol = 0.0015 if half else 0.01
    H, cin, cout, stride = SHAPES[name]
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    bias = torch.randn(cout, generator=g) * 0.1
    n_img = 3 * B if gather else B
    x = torch.randn(n_img, H, H, cin, generator=g).to(dtype)
    HO = (H - 1) // stride + 1
    layer = _Layer()
    layer.cin, layer.cout, layer.stride = cin, cout, stride
    chunk, whole = mfma.kernel_info(H, H, cin, cout, stride)
    layer.dev = {"w": mfma.pack_weights(w, chunk, dtype).cuda(), "bias": bias.cuda()}
    slot = torch.randint(0, 3, (B,), generator=g, dtype=torch.int32) if gather else None
    res = torch.randn(B, HO, HO, cout, generator=g).to(dtype) if residual else None
    acts = torch.randint(0, 7, (B,), generator=g, dtype=torch.int32) if action else None
    avec = torch.randn(7, cin, generator=g) * 0.5 if action else None
    out = torch.empty(B, HO, HO, cout, dtype=dtype, device="cuda")
    mean_out = torch.empty(B, cout, dtype=torch.float32, device="cuda") if mean else None
    cu = lambda t: None if t is None else t.cuda()
    mfma.conv(layer, x.cuda(), out, B, H, H, relu, cu(res), cu(slot), B, cu(avec), cu(acts), mean_out, half=half)
    torch.cuda.synchronize()

    # reference
    xin = x.float()
    if gather:
        xin = torch.stack([xin[int(slot[b]) * B + b] for b in range(B)])
    if action:
        xin = (xin + avec[acts.long()].view(B, 1, 1, cin)).to(dtype).float()
    ref = F.conv2d(xin.permute(0, 3, 1, 2), w.to(dtype).float(), bias, stride=stride, padding=1)
    ref = ref.permute(0, 2, 3, 1)
    if residual:
        ref = ref + res.float()
    if relu:
        ref = F.relu(ref)
    got = out.float().cpu()
    scale = ref.abs().mean().item()
    err = (got - ref).abs()
    assert (err <= tol * ref.abs() + tol * scale).all(), (name, err.max().item(), scale)
    if mean:
        np.testing.assert_allclose(mean_out.cpu().numpy(), ref.mean(dim=(1, 2)).numpy(), rtol=2e-3, atol=2e-3 * scale)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("name", list(SHAPES))
def test_conv_shapes(name, dtype):
    _run(name, B=3 if name != "stem" else 2, dtype=dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_conv_trunk_fused_options(dtype):
    _run("trunk", B=5, relu=True, residual=True, mean=True, seed=1, dtype=dtype)
    _run("trunk", B=4, relu=True, gather=True, action=True, seed=2, dtype=dtype)
    _run("trunk", B=2, relu=False, seed=3, dtype=dtype)
    _run("mid", B=2, relu=True, residual=True, seed=4, dtype=dtype)


def test_unsupported_shape_is_an_error():
    from muzero_amd import _lib, mfma
    with pytest.raises(_lib.MuZeroAmdError, match="no MFMA kernel"):
        mfma.kernel_info(10, 10, 64, 64, 1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B", [6, 300, 515])
def test_fused_chain_against_torch_fp32(B, dtype):
    """mz_conv_chain_bf16 / mz_conv_chain_f16 (images resident in LDS across the dynamics network's five layers,
    gather from a pool, action vectors, spatial mean) against a PyTorch float32 reference of the chain with the
    activations rounded to the operand format between layers.  B = 6: the one-image kernel; B = 300 and 515:
    more images than compute units, i.e. the TWO-image kernel that the benchmark's batch runs on - 515 makes
    workgroups take a second pass and leaves a last image without a partner."""
    from muzero_amd import _lib, mfma
    half = dtype == torch.float16
    g = torch.Generator().manual_seed(7)
    n_blocks = 2
    convs = []
    for _ in range(1 + 2 * n_blocks):
        w = torch.randn(256, 256, 3, 3, generator=g) / (3 * 16.0)
        b = torch.randn(256, generator=g) * 0.1
        convs.append((w.cuda(), b.cuda(), mfma.pack_weights(w, 256, dtype).cuda()))
    pool = torch.randn(3 * B, 12, 12, 256, generator=g).to(dtype).cuda()
    slot = torch.randint(0, 3, (B,), generator=g, dtype=torch.int32).cuda()
    acts = torch.randint(0, 5, (B,), generator=g, dtype=torch.int32).cuda()
    avec = (torch.randn(5, 256, generator=g) * 0.5).cuda()
    specs = [(0, 1, 0)] + [(1 + 2 * j + k, 1, k) for j in range(n_blocks) for k in (0, 1)]
    arr = (_lib.ChainLayer * len(specs))()
    for i, (ci, relu, res) in enumerate(specs):
        arr[i] = _lib.ChainLayer(convs[ci][2].data_ptr(), convs[ci][1].data_ptr(), relu, res)
    out = torch.empty(B, 12, 12, 256, dtype=dtype, device="cuda")
    mean = torch.empty(B, 256, dtype=torch.float32, device="cuda")
    _lib.call("mz_conv_chain_f16" if half else "mz_conv_chain_bf16", _lib.ptr(pool), _lib.ptr(slot), B, arr, len(specs),
              _lib.ptr(out), _lib.ptr(avec), _lib.ptr(acts), _lib.ptr(mean), B, _lib.current_stream())
    torch.cuda.synchronize()

    # PyTorch fp32 reference with the operand format's rounding of activations between layers
    idx = slot.long() * B + torch.arange(B, device="cuda")
    x = pool[idx].float()
    rnd = lambda t: t.to(dtype).float()
    x = rnd(x + avec[acts.long()].view(B, 1, 1, 256))
    conv = lambda t, i: F.conv2d(t.permute(0, 3, 1, 2), rnd(convs[i][0]), convs[i][1], padding=1).permute(0, 2, 3, 1)
    x = rnd(F.relu(conv(x, 0)))
    for j in range(n_blocks):
        t = rnd(F.relu(conv(x, 1 + 2 * j)))
        y = F.relu(conv(t, 2 + 2 * j) + x)
        last_f32 = y
        x = rnd(y)
    got = out.float()
    scale = x.abs().mean().item()
    tol = 0.004 if half else 0.02      # a rounding flip in one layer moves the next layers' inputs by an ulp
    assert bool(((got - x).abs() <= tol * x.abs() + tol * scale).all()), (got - x).abs().max().item()
    np.testing.assert_allclose(mean.cpu().numpy(), last_f32.mean(dim=(1, 2)).cpu().numpy(), rtol=tol / 4,
                               atol=tol / 4 * scale)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,n_blocks", [(1, 2), (5, 2), (3, 1), (300, 2)])
def test_mid_chain_matches_layer_by_layer(B, n_blocks, dtype):
    """mz_conv_chain_mid_bf16 (a 24x24x128 image resident in LDS through the residual blocks of the first
    downsample stage) against the same layers run one kernel at a time (the per-layer kernel adds the
    block input after the bias, the chain starts its accumulation from it: equal up to one rounding of
    the float32 sum, i.e. a bf16 ulp here and there) and against a PyTorch fp32 reference of the
    blocks; more images than compute units (B = 300) makes workgroups take a second image; the result
    of an image does not depend on the batch it is in, and a repeated launch repeats bit for bit."""
    from muzero_amd import _lib, mfma
    g = torch.Generator().manual_seed(11 + B)
    convs = []
    for _ in range(2 * n_blocks):
        w = torch.randn(128, 128, 3, 3, generator=g) / (3 * 128 ** 0.5)
        b = torch.randn(128, generator=g) * 0.1
        layer = _Layer()
        layer.cin, layer.cout, layer.stride = 128, 128, 1
        layer.dev = {"w": mfma.pack_weights(w, 128, dtype).cuda(), "bias": b.cuda()}
        convs.append((w, b, layer))
    x0 = torch.randn(B, 24, 24, 128, generator=g).to(dtype).cuda()
    half = dtype == torch.float16
    tol = 0.004 if half else 0.02
    arr = (_lib.ChainLayer * len(convs))()
    for i, (_, _, layer) in enumerate(convs):
        arr[i] = _lib.ChainLayer(layer.dev["w"].data_ptr(), layer.dev["bias"].data_ptr(), 1, i % 2)

    def chain(inp):
        out = torch.empty_like(inp)
        _lib.call("mz_conv_chain_mid_f16" if half else "mz_conv_chain_mid_bf16", _lib.ptr(inp), arr, len(convs), _lib.ptr(out), inp.shape[0],
                  _lib.current_stream())
        torch.cuda.synchronize()
        return out

    out = chain(x0)
    assert torch.equal(out, chain(x0))
    if B > 4:
        assert torch.equal(out[3:5], chain(x0[3:5].contiguous()))

    # one kernel per layer
    x = x0
    for j in range(n_blocks):
        t, y = torch.empty_like(x), torch.empty_like(x)
        mfma.conv(convs[2 * j][2], x, t, B, 24, 24, True, None, None, 0, None, None, None, half=half)
        mfma.conv(convs[2 * j + 1][2], t, y, B, 24, 24, True, x, None, 0, None, None, None, half=half)
        x = y
    torch.cuda.synchronize()
    a, b = out.float(), x.float()
    scale = b.abs().mean().item()
    assert ((a - b).abs() <= tol * b.abs() + tol * scale).all(), (a - b).abs().max().item()
    assert (a != b).float().mean().item() < 0.05          # a few last-bit differences, no more

    # PyTorch fp32 reference with bf16 rounding of the activations between layers
    n_ref = min(B, 6)
    rnd = lambda t: t.to(dtype).float()
    conv = lambda t, i: F.conv2d(t.permute(0, 3, 1, 2), rnd(convs[i][0]), convs[i][1], padding=1).permute(0, 2, 3, 1)
    r = x0[:n_ref].float().cpu()
    for j in range(n_blocks):
        t = rnd(F.relu(conv(r, 2 * j)))
        r = rnd(F.relu(conv(t, 2 * j + 1) + r))
    got = out[:n_ref].float().cpu()
    scale = r.abs().mean().item()
    assert ((got - r).abs() <= tol * r.abs() + tol * scale).all(), (got - r).abs().max().item()
    assert got.abs().max() > 0


def test_mid_chain_layer_options():
    """mz_conv_chain_mid_bf16 with a single layer, with eight layers (its maximum), and with a layer that has no
    ReLU (negative outputs survive into the next layer's input), each against the per-layer kernel."""
    from muzero_amd import _lib, mfma
    g = torch.Generator().manual_seed(21)
    B = 4
    layers = []
    for _ in range(8):
        layer = _Layer()
        layer.cin, layer.cout, layer.stride = 128, 128, 1
        layer.dev = {"w": mfma.pack_weights(torch.randn(128, 128, 3, 3, generator=g) / (3 * 128 ** 0.5), 128).cuda(),
                     "bias": (torch.randn(128, generator=g) * 0.1).cuda()}
        layers.append(layer)
    x0 = torch.randn(B, 24, 24, 128, generator=g).to(torch.bfloat16).cuda()
    for specs in ([(1, 0)], [(0, 0), (1, 0), (1, 1)], [(1, 0), (1, 1)] * 4):       # (relu, add_residual) per layer
        arr = (_lib.ChainLayer * len(specs))()
        for i, (relu, res) in enumerate(specs):
            arr[i] = _lib.ChainLayer(layers[i].dev["w"].data_ptr(), layers[i].dev["bias"].data_ptr(), relu, res)
        out = torch.empty_like(x0)
        _lib.call("mz_conv_chain_mid_bf16", _lib.ptr(x0), arr, len(specs), _lib.ptr(out), B, _lib.current_stream())
        x, prev = x0, None
        for i, (relu, res) in enumerate(specs):
            y = torch.empty_like(x)
            mfma.conv(layers[i], x, y, B, 24, 24, bool(relu), prev if res else None, None, 0, None, None, None)
            prev, x = x, y
        torch.cuda.synchronize()
        a, b = out.float(), x.float()
        scale = b.abs().mean().item()
        assert ((a - b).abs() <= 0.03 * b.abs() + 0.03 * scale).all(), (len(specs), (a - b).abs().max().item())
    with pytest.raises(_lib.MuZeroAmdError, match="block input"):
        arr = (_lib.ChainLayer * 1)()
        arr[0] = _lib.ChainLayer(layers[0].dev["w"].data_ptr(), layers[0].dev["bias"].data_ptr(), 1, 1)
        _lib.call("mz_conv_chain_mid_bf16", _lib.ptr(x0), arr, 1, _lib.ptr(torch.empty_like(x0)), B, _lib.current_stream())


# --------------------------------------------------------------- bf16x3: float32-accurate MFMA kernels
def _split(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return torch.stack([hi, lo], dim=1).contiguous()


def _join(x):
    return x[:, 0].float() + x[:, 1].float()


def _pack_pair(w, chunk):
    from muzero_amd import mfma
    return (mfma.pack_weights(w, chunk).cuda(),
            mfma.pack_weights(w - w.to(torch.bfloat16).float(), chunk).cuda())


@pytest.mark.parametrize("name,H,cin,cout,stride", [("stem", 96, 96, 256, 2), ("down0", 48, 256, 128, 2),
                                                     ("mid", 24, 128, 128, 1), ("down1", 24, 128, 256, 2)])
def test_x3_conv_shapes_against_torch_fp32(name, H, cin, cout, stride):
    """mz_conv3x3_bf16x3 against torch's float32 convolution (double precision as the arbiter)."""
    from muzero_amd import _lib, mfma
    g = torch.Generator().manual_seed(3)
    B = 3
    x = torch.randn(B, H, H, cin, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    HO = (H - 1) // stride + 1
    res = torch.randn(B, HO, HO, cout, generator=g) if stride == 1 else None
    chunk, _ = mfma.kernel_info(H, H, cin, cout, stride)
    wh, wl = _pack_pair(w, chunk)
    xd, bd = _split(x).cuda(), bias.cuda()
    rd = _split(res).cuda() if res is not None else None
    out = torch.empty(B, 2, HO, HO, cout, dtype=torch.bfloat16, device="cuda")
    _lib.call("mz_conv3x3_bf16x3", _lib.ptr(xd), _lib.ptr(wh), _lib.ptr(wl), _lib.ptr(bd), _lib.ptr(rd), _lib.ptr(out),
              B, H, H, cin, cout, stride, 1, _lib.current_stream())
    got = _join(out.cpu())
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), bias.double(), stride, 1)
    ref = ref.permute(0, 2, 3, 1)
    if res is not None:
        ref = ref + res.double()
    ref = torch.relu(ref)
    err = (got.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 1e-4 * scale, (name, err, scale)     # measured ~1e-5 of the output range


@pytest.mark.parametrize("B,use_slots", [(5, False), (300, True)])
def test_x3_chain_against_torch_fp32(B, use_slots):
    """The split-precision chain (dynamics shape: conv + two residual blocks, gather by slot, action
    vector, spatial mean) against the same layers in torch float64."""
    from muzero_amd import _lib, mfma
    g = torch.Generator().manual_seed(11)
    n_layers, A = 5, 6
    ws = [torch.randn(256, 256, 3, 3, generator=g) / 48 for _ in range(n_layers)]
    bs = [torch.randn(256, generator=g) * 0.1 for _ in range(n_layers)]
    avec = torch.randn(A, 256, generator=g) * 0.2
    slots = 3 if use_slots else 1
    pool = torch.randn(slots, B, 12, 12, 256, generator=g)
    in_slot = torch.randint(0, slots, (B,), generator=g, dtype=torch.int32)
    actions = torch.randint(0, A, (B,), generator=g, dtype=torch.int32)
    packs = [_pack_pair(w, 256) for w in ws]
    biases = [b.cuda() for b in bs]
    arr = (_lib.ChainLayerX3 * n_layers)()
    for i in range(n_layers):
        arr[i] = _lib.ChainLayerX3(packs[i][0].data_ptr(), packs[i][1].data_ptr(), biases[i].data_ptr(), 1,
                                   1 if i in (2, 4) else 0)
    pool_d = _split(pool.view(slots * B, 12, 12, 256)).cuda()        # [slots*B, 2, 12, 12, 256]
    out = torch.empty(B, 2, 12, 12, 256, dtype=torch.bfloat16, device="cuda")
    mean = torch.empty(B, 256, dtype=torch.float32, device="cuda")
    avec_d, slot_d, act_d = avec.cuda(), in_slot.cuda(), actions.cuda()
    _lib.call("mz_conv_chain_bf16x3", _lib.ptr(pool_d), _lib.ptr(slot_d) if use_slots else None, B, arr, n_layers,
              _lib.ptr(out), _lib.ptr(avec_d), _lib.ptr(act_d), _lib.ptr(mean), B, _lib.current_stream())
    torch.cuda.synchronize()
    x = pool[in_slot.long(), torch.arange(B)] if use_slots else pool[0]
    x = (x + avec[actions.long()][:, None, None, :]).permute(0, 3, 1, 2).double()
    conv = lambda t, i: torch.nn.functional.conv2d(t, ws[i].double(), bs[i].double(), 1, 1)
    x = torch.relu(conv(x, 0))
    for i in (1, 3):
        y = torch.relu(conv(x, i))
        x = torch.relu(conv(y, i + 1) + x)
    ref = x.permute(0, 2, 3, 1)
    got = _join(out.cpu()).double()
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 2e-4 * scale
    assert (mean.cpu().double() - ref.mean(dim=(1, 2))).abs().max().item() <= 2e-4 * scale
