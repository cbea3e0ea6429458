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


def _run(name, B, relu=True, residual=False, gather=False, action=False, mean=False, seed=0):
    from muzero_amd import mfma
    H, cin, cout, stride = SHAPES[name]
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    bias = torch.randn(cout, generator=g) * 0.1
    n_img = 3 * B if gather else B
    x = torch.randn(n_img, H, H, cin, generator=g).to(torch.bfloat16)
    HO = (H - 1) // stride + 1
    layer = _Layer()
    layer.cin, layer.cout, layer.stride = cin, cout, stride
    chunk, whole = mfma.kernel_info(H, H, cin, cout, stride)
    layer.dev = {"w": mfma.pack_weights(w, chunk).cuda(), "bias": bias.cuda()}
    slot = torch.randint(0, 3, (B,), generator=g, dtype=torch.int32) if gather else None
    res = torch.randn(B, HO, HO, cout, generator=g).to(torch.bfloat16) if residual else None
    acts = torch.randint(0, 7, (B,), generator=g, dtype=torch.int32) if action else None
    avec = torch.randn(7, cin, generator=g) * 0.5 if action else None
    out = torch.empty(B, HO, HO, cout, dtype=torch.bfloat16, device="cuda")
    mean_out = torch.empty(B, cout, dtype=torch.float32, device="cuda") if mean else None
    cu = lambda t: None if t is None else t.cuda()
    mfma.conv(layer, x.cuda(), out, B, H, H, relu, cu(res), cu(slot), B, cu(avec), cu(acts), mean_out)
    torch.cuda.synchronize()

    # reference
    xin = x.float()
    if gather:
        xin = torch.stack([xin[int(slot[b]) * B + b] for b in range(B)])
    if action:
        xin = (xin + avec[acts.long()].view(B, 1, 1, cin)).to(torch.bfloat16).float()
    ref = F.conv2d(xin.permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), bias, stride=stride, padding=1)
    ref = ref.permute(0, 2, 3, 1)
    if residual:
        ref = ref + res.float()
    if relu:
        ref = F.relu(ref)
    got = out.float().cpu()
    scale = ref.abs().mean().item()
    err = (got - ref).abs()
    assert (err <= 0.01 * ref.abs() + 0.01 * scale).all(), (name, err.max().item(), scale)
    if mean:
        np.testing.assert_allclose(mean_out.cpu().numpy(), ref.mean(dim=(1, 2)).numpy(), rtol=2e-3, atol=2e-3 * scale)


@pytest.mark.parametrize("name", list(SHAPES))
def test_conv_shapes(name):
    _run(name, B=3 if name != "stem" else 2)


def test_conv_trunk_fused_options():
    _run("trunk", B=5, relu=True, residual=True, mean=True, seed=1)
    _run("trunk", B=4, relu=True, gather=True, action=True, seed=2)
    _run("trunk", B=2, relu=False, seed=3)
    _run("mid", B=2, relu=True, residual=True, seed=4)


def test_unsupported_shape_is_an_error():
    from muzero_amd import _lib, mfma
    with pytest.raises(_lib.MuZeroAmdError, match="no MFMA kernel"):
        mfma.kernel_info(10, 10, 64, 64, 1)


def test_fused_chain_matches_layer_by_layer():
    """mz_conv_chain_bf16 (image resident in LDS across layers) against the same layers run
    one kernel at a time, and against a PyTorch fp32 reference of the chain."""
    import ctypes as C
    from muzero_amd import _lib, mfma
    g = torch.Generator().manual_seed(7)
    B, n_blocks = 6, 2
    convs = []
    for _ in range(1 + 2 * n_blocks):
        w = torch.randn(256, 256, 3, 3, generator=g) / (3 * 16.0)
        b = torch.randn(256, generator=g) * 0.1
        convs.append((w, b, mfma.pack_weights(w, 256).cuda(), b.cuda()))
    pool = torch.randn(3 * B, 12, 12, 256, generator=g).to(torch.bfloat16).cuda()
    slot = torch.randint(0, 3, (B,), generator=g, dtype=torch.int32).cuda()
    acts = torch.randint(0, 5, (B,), generator=g, dtype=torch.int32).cuda()
    avec = (torch.randn(5, 256, generator=g) * 0.5).cuda()
    specs = [(0, 1, 0)] + [(1 + 2 * j + k, 1, k) for j in range(n_blocks) for k in (0, 1)]
    arr = (_lib.ChainLayer * len(specs))()
    for i, (ci, relu, res) in enumerate(specs):
        arr[i] = _lib.ChainLayer(convs[ci][2].data_ptr(), convs[ci][3].data_ptr(), relu, res)
    out = torch.empty(B, 12, 12, 256, dtype=torch.bfloat16, device="cuda")
    mean = torch.empty(B, 256, dtype=torch.float32, device="cuda")
    _lib.call("mz_conv_chain_bf16", _lib.ptr(pool), _lib.ptr(slot), B, arr, len(specs), _lib.ptr(out),
              _lib.ptr(avec), _lib.ptr(acts), _lib.ptr(mean), B, _lib.current_stream())
    torch.cuda.synchronize()

    # PyTorch fp32 reference with bf16 rounding of activations between layers
    import torch.nn.functional as F
    x = torch.stack([pool[int(slot[b]) * B + b] for b in range(B)]).float().cpu()
    x = (x + avec.cpu()[acts.cpu().long()].view(B, 1, 1, 256)).to(torch.bfloat16).float()
    rnd = lambda t: t.to(torch.bfloat16).float()
    conv = lambda t, i: F.conv2d(t.permute(0, 3, 1, 2), rnd(convs[i][0]), convs[i][1], padding=1).permute(0, 2, 3, 1)
    x = rnd(F.relu(conv(x, 0)))
    for j in range(n_blocks):
        t = rnd(F.relu(conv(x, 1 + 2 * j)))
        y = F.relu(conv(t, 2 + 2 * j) + x)
        last_f32 = y
        x = rnd(y)
    got = out.float().cpu()
    scale = x.abs().mean().item()
    assert ((got - x).abs() <= 0.02 * x.abs() + 0.02 * scale).all(), (got - x).abs().max().item()
    np.testing.assert_allclose(mean.cpu().numpy(), last_f32.mean(dim=(1, 2)).numpy(), rtol=5e-3, atol=5e-3 * scale)
