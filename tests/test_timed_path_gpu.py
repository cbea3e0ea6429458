"""The path the benchmark times - SelfPlayWorker.step() on the ROTATING observation (a move rewrites one
slot of the tiled frame stack, the first convolution reads it through one of 32 channel-permuted weight
copies) - held to references that do not share its code: torch float32 conv2d on the stacked observation of
a host frame-stack model (train.py:22-63, :529-540; model.py:182,197), the reference's own network outputs
(fixture g3) and the CPU oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import load
from tests.test_bf16_gpu import _nets

pytestmark = pytest.mark.gpu


class HostStacks:
    """CustomFrameStack (train.py:22-63) for B games at once in numpy: reset fills every slot with the
    frame, a push drops the oldest 3 channels and appends the new frame; channel = slot*3 + rgb, oldest first."""

    def __init__(self, frames):
        self.obs = np.concatenate([frames] * 32, axis=-1)

    def push(self, frames):
        self.obs = np.concatenate([self.obs[..., 3:], frames], axis=-1)


def _folded_stem(ref):
    """conv1 + bn1 of the representation folded in float64 (eval-mode BatchNorm), then float32."""
    c, bn = ref.representation.conv1, ref.representation.bn1
    s = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
    w = (c.weight.double() * s.view(-1, 1, 1, 1)).float()
    b = ((c.bias.double() - bn.running_mean.double()) * s + bn.bias.double()).float()
    return w.detach(), b.detach()


def test_rotating_stem_matches_torch_fp32_for_every_rotation():
    """mz_conv3x3_bf16_rot (and its device-indexed form _rot_sel) on the rotating observation for each of the
    32 positions of the newest slot, 301 games (a ragged count: more patches than the launch's workgroup
    granularity divides): every output within 1 % + 1 % of the mean magnitude of torch's float32 conv2d on the
    stacked observation in the reference's channel order, inputs and weights rounded to bf16 as the kernel
    sees them."""
    from muzero_amd import _lib
    from muzero_amd.prep import FrameRing
    B = 301
    cfg, ref, net = _nets(18)
    w, b = _folded_stem(ref)
    w_ref, b_ref = w.to(torch.bfloat16).float().cuda(), b.cuda()
    first = net.rep_plan[0][1]
    rng = np.random.default_rng(11)
    new = lambda: rng.integers(0, 256, (B, 96, 96, 3), dtype=np.uint8)
    ring = FrameRing(B)
    ring.enable_rotating(torch.bfloat16)
    f = new()
    ring.reset(torch.from_numpy(f))
    host = HostStacks(f)
    table = net._rotated_weights(first, None)
    seen = set()
    worst = 0.0
    for step in range(32):
        f = new()
        if step % 4 == 1:          # mostly-constant frames: long runs of equal bytes, zero borders
            f[:, :8] = 0
            f[:, :, -7:] = 255
        ring.push(torch.from_numpy(f))
        host.push(f)
        rot, newest = ring.rotating_observation()
        seen.add(newest)
        y = torch.empty(B, 48, 48, 256, dtype=torch.bfloat16, device="cuda")
        _lib.call("mz_conv3x3_bf16_rot", _lib.ptr(rot), _lib.ptr(table[newest]), _lib.ptr(first.dev["bias"]),
                  _lib.ptr(y), B, 1, _lib.current_stream())
        y2 = torch.empty_like(y)
        _lib.call("mz_conv3x3_bf16_rot_sel", _lib.ptr(rot), _lib.ptr(table), table.shape[1], _lib.ptr(ring.newest_dev),
                  _lib.ptr(first.dev["bias"]), _lib.ptr(y2), B, 1, _lib.current_stream())
        x = (torch.from_numpy(host.obs).cuda().float() / 255.0).to(torch.bfloat16).float()   # train.py:534-540
        want = F.relu(F.conv2d(x.permute(0, 3, 1, 2), w_ref, b_ref, stride=2, padding=1)).permute(0, 2, 3, 1)
        assert torch.equal(y, y2), step
        err = (y.float() - want).abs()
        scale = want.abs().mean()
        bound = 0.01 * want.abs() + 0.01 * scale
        worst = max(worst, float((err / bound).max()))
        assert bool((err <= bound).all()), (step, newest, float(err.max()), float(scale))
        del x, want, err, bound
    assert seen == set(range(32))
    print(f"rotating stem vs torch fp32: worst error {worst:.3f} of the 1 % + 1 % bound over 32 rotations x {B} games")


@pytest.mark.parametrize("A", [6, 18])
def test_rotating_initial_inference_against_the_reference_fixture(A):
    """The reference's own outputs for its two fixture observations (g3: hidden state, policy logits, value
    logits, decoded value of model.py:332-338 in float32), reached through the timed path: the 32 frames of each
    observation pushed into the ring behind j frames of something else, so that the newest slot - and with it the
    weight copy the first convolution uses - takes every one of its 32 values; representation on the rotating
    tensor, then the heads.  The bound stated for the bf16 precision: logits 0.8 % of the output's RMS, the
    decoded value 2 %, the hidden state 6 %."""
    from muzero_amd.network import support_to_scalar
    from muzero_amd.prep import FrameRing
    g = load("g3_nets_full")
    cfg, ref, net = _nets(A)
    frames = np.random.default_rng(1234).integers(0, 256, size=(2, 96, 96, 96), dtype=np.uint8)
    junk = np.random.default_rng(77).integers(0, 256, size=(2, 96, 96, 3), dtype=np.uint8)
    worst = {}
    seen = set()
    for j in range(32):
        ring = FrameRing(2)
        ring.enable_rotating(torch.bfloat16)
        ring.reset(torch.from_numpy(junk))
        for _ in range(j):
            ring.push(torch.from_numpy(junk))
        for k in range(32):
            ring.push(torch.from_numpy(np.ascontiguousarray(frames[..., 3 * k:3 * k + 3])))
        rot, newest = ring.rotating_observation()
        seen.add(newest)
        hidden = torch.empty(2, 12, 12, 256, dtype=torch.bfloat16, device="cuda")
        mean = torch.empty(2, 256, dtype=torch.float32, device="cuda")
        net.represent(rot, hidden, mean, rotation=newest)
        priors, value, _, pl, vl, _ = net._heads(mean, 2, False, True)
        torch.cuda.synchronize()
        got = {"h0": hidden[:1].permute(0, 3, 1, 2).float(), "p0": pl, "v0": vl, "sv0": support_to_scalar(vl, 300)}
        for key, frac in (("h0", 0.06), ("p0", 0.008), ("v0", 0.008), ("sv0", 0.02)):
            want = g[f"A{A}_{key}"]
            rms = max(float(np.sqrt((want.astype(np.float64) ** 2).mean())), 1e-3)
            err = float(np.abs(got[key].cpu().numpy() - want).max()) / rms
            worst[key] = max(worst.get(key, 0.0), err)
            assert err <= frac, (j, newest, key, err)
        want_pri = torch.softmax(torch.from_numpy(g[f"A{A}_p0"]), dim=1).numpy()        # model.py:397
        assert float(np.abs(priors.cpu().numpy() - want_pri).max()) <= 0.02 * float(want_pri.max())
        assert float((value.cpu() - torch.from_numpy(g[f"A{A}_sv0"])).abs().max()) <= 0.02 * max(
            1.0, float(np.abs(g[f"A{A}_sv0"]).max()))
    assert seen == set(range(32))
    print(f"rotating initial inference A={A}, worst max-error / RMS over 32 rotations:",
          {k: round(v, 4) for k, v in worst.items()})


def test_worker_step_root_outputs_against_the_cpu_oracle():
    """SelfPlayWorker.step() - the call the benchmark makes - at a ragged batch across a ring wrap: the root
    priors and root network values the search consumed against oracle.nets.Net.initial (float32, CPU) on the
    host model's stacked observation, in the bf16 bound; and the trees it built from those outputs replayed
    bit-exactly by the oracle tree for every game."""
    from muzero_amd.selfplay import SelfPlayWorker
    from oracle import nets as onets
    from oracle import prep as oprep
    from oracle import tree as otree
    A, B, sims = 18, 11, 12
    cfg, ref, net = _nets(A)
    worker = SelfPlayWorker(cfg, net, B, sims)
    assert worker.rotating
    rng = np.random.default_rng(3)
    new = lambda: rng.integers(0, 256, (B, 96, 96, 3), dtype=np.uint8)
    f = new()
    worker.reset(torch.from_numpy(f))
    host = HostStacks(f)
    for move in range(36):
        f = new()
        host.push(f)
        if move not in (0, 30, 31, 35):        # newest slot 0, 30, 31 and, past the wrap, 3
            worker.ring.push(torch.from_numpy(f))
            continue
        worker.draw_randomness()
        actions, res = worker.step(torch.from_numpy(f))
        m = worker.mcts
        pri = m.root_priors.cpu().numpy().copy()
        val = m.root_net_value.cpu().numpy().copy()
        with torch.no_grad():
            h, p, v = ref.initial(torch.tensor(oprep.prepare(host.obs)))
        want_pri = torch.softmax(p, dim=1).numpy()
        want_val = onets.decode_support(v, cfg.support_size).numpy()
        assert np.abs(pri - want_pri).max() <= 0.02 * want_pri.max(), (move, np.abs(pri - want_pri).max())
        assert np.abs(val - want_val).max() <= 0.02 * max(1.0, np.abs(want_val).max()), (move, val, want_val)
        visits = res.visit_counts.cpu().numpy()
        assert (visits.sum(1) == sims).all()
        a = actions.cpu().numpy()
        assert all(visits[i, a[i]] > 0 for i in range(B))
    # the last search again, stepwise, recording what the trees consumed: the oracle tree replays it exactly
    obs, newest = worker.ring.rotating_observation()
    pri0, _ = net.initial_to_pool(obs, m.pool[0], newest)
    pri0 = pri0.clone()
    assert np.array_equal(pri0.cpu().numpy(), pri)
    m.trees.root_init(pri0, worker.noise, cfg.root_exploration_fraction)
    log = []
    for s in range(sims):
        m.trees.select()
        p, v, r = net.recurrent_in_pool(m.pool, m.leaf_parent, B, m.leaf_action, m.pool[s + 1])
        log.append((p.cpu().numpy(), v.cpu().numpy(), r.cpu().numpy()))
        m.trees.expand_backup(p, v, r, 1, cfg.discount)
    stepwise = m.trees.readout()[0].cpu().numpy()
    assert np.array_equal(stepwise, visits)
    noise = worker.noise.cpu().numpy()
    for b in range(B):
        it = iter([(l[0][b], float(l[1][b]), float(l[2][b]), None) for l in log])
        root, _ = otree.search(pri[b], None, lambda h, a, it=it: [next(it)], sims, 1, noise=noise[b])
        assert otree.visit_counts(root).tolist() == visits[b].tolist()
