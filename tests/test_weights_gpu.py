"""Device-side weight hand-over (mz_weights_fold_pack, mz_pack_linear, mz_action_vectors) against the
same arithmetic written in torch on the host: bit for bit."""
import numpy as np
import pytest
import torch

from tests.test_oracle_golden import full_cfg

pytestmark = pytest.mark.gpu

BN_EPS = 1e-5


def host_fold(sd, conv, bn):
    """BatchNorm (running statistics) folded into the convolution, in double, rounded to float32."""
    w, b = sd[conv + ".weight"].double(), sd[conv + ".bias"].double()
    scale = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + BN_EPS)
    return (w * scale.view(-1, 1, 1, 1)).float(), ((b - sd[bn + ".running_mean"].double()) * scale + sd[bn + ".bias"].double()).float()


def host_linear(w, bf16):
    """bf16: 0 float32 fragments, 1 bf16, 2 IEEE half (mz_pack_linear's format codes)."""
    O, K = w.shape
    tiles = (O + 15) // 16
    wp = torch.zeros(tiles * 16, K)
    wp[:O] = w.float()
    if bf16:
        dt = torch.bfloat16 if bf16 == 1 else torch.float16
        return wp.to(dt).view(tiles, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1)
    return wp.view(tiles, 16, K // 4, 4).permute(0, 2, 3, 1).contiguous().view(-1)


def same_bits(a, b):
    a, b = a.detach().cpu().contiguous(), b.detach().cpu().contiguous()
    assert a.dtype == b.dtype and a.shape == b.shape, (a.dtype, b.dtype, a.shape, b.shape)
    two = (torch.bfloat16, torch.float16)
    return torch.equal(a.view(torch.int16 if a.dtype in two else torch.int32),
                       b.view(torch.int16 if b.dtype in two else torch.int32))


@pytest.mark.parametrize("precision", ["bf16", "f16", "bf16x3", "fp32"])
def test_device_packed_weights_equal_host_packed(precision):
    from muzero_amd import mfma
    from muzero_amd.network import MuZeroNetwork, init_state_dict
    from muzero_amd.prep import rotation_permutation
    cfg = full_cfg(6)
    sd = init_state_dict(cfg, seed=5, randomize_bn=True)
    net = MuZeroNetwork(cfg, precision=precision, state_dict=sd)
    first = net.rep_plan[0][1]
    op_dtype = torch.float16 if precision == "f16" else torch.bfloat16
    if precision != "bf16x3":
        net._rotated_weights(first, 0)                      # builds the table of rotated copies
    for layer in net.all_convs():
        w, b = host_fold(sd, layer.name, layer.bn)
        assert same_bits(layer.dev["bias"], b), layer.name
        if precision == "fp32":
            assert same_bits(layer.dev["w"], w.permute(2, 3, 1, 0).reshape(-1)), layer.name
        else:
            assert same_bits(layer.dev["w"], mfma.pack_weights(w, layer.chunk, op_dtype)), layer.name
            if precision == "bf16x3":
                assert same_bits(layer.dev["w_lo"], mfma.pack_weights(w - w.to(torch.bfloat16).float(), layer.chunk))
    if precision != "bf16x3":
        w, _ = host_fold(sd, first.name, first.bn)
        D = first.cin // 3
        for r in (0, 1, 17, D - 1):
            wr = w[:, rotation_permutation(D, r)]
            want = wr.permute(2, 3, 1, 0).reshape(-1) if precision == "fp32" else mfma.pack_weights(wr, first.chunk, op_dtype)
            assert same_bits(first.dev["rot"][r], want), r
    wa = sd["dynamics.conv_action.weight"].float()[:, :, 0, 0]
    assert same_bits(net.action_vec, (wa.t() + sd["dynamics.conv_action.bias"].float()).contiguous())
    for head in (net.policy, net.value_head, net.reward_head):
        w1, w2 = sd[f"{head.name}.{head.k1}.weight"], sd[f"{head.name}.{head.k2}.weight"]
        assert head.bf16 == (2 if precision in ("bf16", "f16") else 0)      # both 16-bit precisions: IEEE-half heads
        assert same_bits(head.w1p, host_linear(w1, head.bf16)) and same_bits(head.w2p, host_linear(w2, head.bf16))
        assert same_bits(head.b1, sd[f"{head.name}.{head.k1}.bias"].float())
        b2 = torch.zeros((head.out + 15) // 16 * 16)
        b2[:head.out] = sd[f"{head.name}.{head.k2}.bias"].float()
        assert same_bits(head.b2, b2)


def test_reload_is_in_place_and_takes_device_tensors():
    """A second load with the same shapes keeps every buffer's address (what a recorded HIP graph holds),
    accepts device tensors without a host round trip, and is fast enough to time here."""
    import time
    from muzero_amd.network import MuZeroNetwork, init_state_dict
    cfg = full_cfg(6)
    net = MuZeroNetwork(cfg, precision="bf16", state_dict=init_state_dict(cfg, seed=5, randomize_bn=True))
    first = net.rep_plan[0][1]
    net._rotated_weights(first, 0)
    ptrs = [t.data_ptr() for l in net.all_convs() for t in l.dev.values()] + [net.action_vec.data_ptr(), net.policy.w1p.data_ptr()]
    gen = net.generation
    sd2 = {k: v.cuda() for k, v in init_state_dict(cfg, seed=6, randomize_bn=True).items()}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    net.load_state_dict(sd2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"fold + pack of the whole network on the device (incl. 32 rotated stem copies): {dt * 1e3:.2f} ms")
    assert net.generation == gen
    assert ptrs == [t.data_ptr() for l in net.all_convs() for t in l.dev.values()] + [net.action_vec.data_ptr(), net.policy.w1p.data_ptr()]
    w, b = host_fold({k: v.cpu() for k, v in sd2.items()}, net.dyn_first.name, net.dyn_first.bn)
    from muzero_amd import mfma
    assert same_bits(net.dyn_first.dev["w"], mfma.pack_weights(w, 256)) and same_bits(net.dyn_first.dev["bias"], b)
    assert dt < 0.05


def test_transposed_packing_for_the_data_gradient():
    """transpose=1 packs W'[o][i][tap] = w[i][o][8 - tap]."""
    from muzero_amd import _lib, mfma
    g = torch.Generator().manual_seed(1)
    w = torch.randn(256, 256, 3, 3, generator=g)
    wd = w.cuda()
    out = torch.empty(mfma.packed_elems(256, 256), dtype=torch.bfloat16, device="cuda")
    _lib.call("mz_weights_fold_pack", _lib.ptr(wd), None, None, None, None, None, 0.0, 256, 256, 256, 0, 1, 0,
              _lib.ptr(out), None, _lib.current_stream())
    want = mfma.pack_weights(w.permute(1, 0, 2, 3).flip(2, 3).contiguous(), 256)
    assert same_bits(out, want)


def test_rccl_weight_broadcast_entry_points_single_rank():
    """mz_comm_* with one rank on the one GPU of this box: the library binds RCCL at run time, creates a
    communicator and broadcasts a flat buffer in place (a 1-rank broadcast leaves it unchanged).  More ranks
    need more GPUs; the rank plumbing around it is rehearsed with gloo in the CPU tests."""
    from muzero_amd import dist as mzdist
    comm = mzdist.WeightComm()
    assert comm.world == 1 and comm.rank == 0
    flat = torch.arange(1 << 20, dtype=torch.float32, device="cuda")
    want = flat.clone()
    comm.broadcast(flat, 0)
    torch.cuda.synchronize()
    assert torch.equal(flat, want)
    with pytest.raises(Exception):
        comm.broadcast(flat, 3)              # root outside the communicator
    comm.close()


def test_pack_many_equals_single_packs():
    """mz_weights_pack_many (several weights of one shape, forward and data-gradient forms, one launch) writes
    exactly what mz_weights_fold_pack writes for each of them."""
    import ctypes as C
    from muzero_amd import _lib, mfma
    g = torch.Generator().manual_seed(3)
    n = 5
    ws = [(torch.randn(256, 256, 3, 3, generator=g) / 48.0).cuda() for _ in range(n)]
    forms = [0, 1, 1, 0, 1]
    elems = mfma.packed_elems(256, 256)
    many = [torch.full((elems,), 7.0, dtype=torch.bfloat16, device="cuda") for _ in range(n)]
    single = [torch.full((elems,), 9.0, dtype=torch.bfloat16, device="cuda") for _ in range(n)]
    wp = (C.c_void_p * n)(*[w.data_ptr() for w in ws])
    op = (C.c_void_p * n)(*[o.data_ptr() for o in many])
    tr = (C.c_int32 * n)(*forms)
    _lib.call("mz_weights_pack_many", wp, op, tr, n, 256, 256, 256, _lib.current_stream())
    for w, o, t in zip(ws, single, forms):
        _lib.call("mz_weights_fold_pack", _lib.ptr(w), None, None, None, None, None, 0.0, 256, 256, 256, 0, t, 0,
                  _lib.ptr(o), None, _lib.current_stream())
    torch.cuda.synchronize()
    for a, b in zip(many, single):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    with pytest.raises(_lib.MuZeroAmdError, match="null"):
        bad = (C.c_void_p * n)(*([w.data_ptr() for w in ws[:-1]] + [None]))
        _lib.call("mz_weights_pack_many", bad, op, tr, n, 256, 256, 256, _lib.current_stream())


def test_loaded_parameters_are_a_snapshot_and_key_order_is_free():
    """load_state_dict on float32 DEVICE tensors (a learner's live parameters) keeps its own copy: training on
    after the hand-over does not change what state_dict() returns; and the same parameters in another key order
    refill the existing buffers (no new generation, a captured search stays valid)."""
    from collections import OrderedDict
    from muzero_amd.config import MuZeroConfig
    from muzero_amd.network import MuZeroNetwork, init_state_dict
    cfg = MuZeroConfig()
    cfg.action_space_size = 6
    live = OrderedDict((k, v.cuda()) for k, v in init_state_dict(cfg, seed=4, randomize_bn=True).items())
    net = MuZeroNetwork(cfg, precision="bf16", state_dict=live)
    gen = net.generation
    before = {k: v.clone() for k, v in net.state_dict().items()}
    for v in live.values():
        if v.is_floating_point():
            v.add_(1.0)                                   # the learner moves on
    after = net.state_dict()
    assert all(torch.equal(before[k], after[k]) for k in before)
    shuffled = OrderedDict((k, before[k]) for k in reversed(list(before)))
    net.load_state_dict(shuffled)
    assert net.generation == gen
    assert all(torch.equal(before[k], net.state_dict()[k]) for k in before)
