#!/usr/bin/env python3
"""Which 16-bit operand formats could meet the reference's float32 outputs to 1e-3 at the matrix cores' full or
2/3 rate?  Emulation in torch float32 on the CPU (no GPU needed): a matrix-core convolution with operands in a
16-bit format and float32 accumulation differs from a float32 convolution of the ROUNDED operands only by the
order of the float32 partial sums (~1e-7), so rounding the operands and calling conv2d reproduces each scheme's
error.  For every scheme: the whole network (eval-mode BatchNorm folded, model.py:178-281) on the two fixture
observations, initial + two recurrent inferences, every output's worst error as a multiple of the bound the
float32-accurate precisions are held to (|got - want| <= 2e-4 + 1e-3 |want|) and as a fraction of its RMS.

Schemes (w = weights, x = activations; hi = rounding to the format, lo = rounding of the remainder):
  bf16      w_hi x_hi                         1 MFMA / product   (the throughput precision; checks the emulation
                                                                  against the errors measured on the kernels)
  f16       w_hi x_hi in IEEE half            1 MFMA
  bf16 wx2  (w_hi + w_lo) x_hi                2 MFMAs  (drops w_hi x_lo)
  bf16 xw2  w_hi (x_hi + x_lo)                2 MFMAs  (drops w_lo x_hi)
  f16 wx2 / f16 xw2                           2 MFMAs
  bf16x3    w_hi x_hi + w_lo x_hi + w_hi x_lo 3 MFMAs  (the float32-accurate precision)
usage: operand_precision_emulation.py [A]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import nets as onets  # noqa: E402
from oracle import prep as oprep  # noqa: E402
from tests.test_oracle_golden import full_cfg  # noqa: E402

A = int(sys.argv[1]) if len(sys.argv) > 1 else 18
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g3_nets_full.npz"))
torch.manual_seed(0)
ref = onets.Net(full_cfg(A)).eval()
onets.randomize_bn(ref, seed=1)
assert onets.state_digest(ref) == str(g[f"digest_bn_A{A}"])


def rounder(fmt, parts):
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[fmt]

    def r(t):
        hi = t.to(dt).float()
        return hi if parts == 1 else hi + (t - hi).to(dt).float()
    return r


def fold(conv, bn):
    s = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
    return (conv.weight.double() * s.view(-1, 1, 1, 1)).float(), ((conv.bias.double() - bn.running_mean.double()) * s
                                                                  + bn.bias.double()).float()


class Emu:
    def __init__(self, rw, rx, head_r=None):
        self.rw, self.rx, self.head_r = rw, rx, head_r

    def conv(self, x, conv, bn, stride=1, residual=None):
        w, b = fold(conv, bn)
        if self.product == "x3":      # hi*hi + lo*hi + hi*lo of bf16 pairs: only lo*lo is missing
            r1 = rounder("bf16", 1)
            wh, xh = r1(w), r1(x)
            wl, xl = r1(w - wh), r1(x - xh)
            y = (F.conv2d(xh, wh, None, stride, 1) + F.conv2d(xh, wl, None, stride, 1) + F.conv2d(xl, wh, None, stride, 1))
            y = y + b.view(1, -1, 1, 1)
        else:
            y = F.conv2d(self.rx(x), self.rw(w), b, stride, 1)
        if residual is not None:
            y = y + residual
        return F.relu(y)

    def res(self, x, blk):
        t = self.store(self.conv(x, blk.conv1, blk.bn1))
        return self.conv(t, blk.conv2, blk.bn2, residual=x)

    def store(self, y):
        """Activations are kept in HBM / LDS in the scheme's activation format."""
        return self.rx(y) if self.product != "x3" else rounder("bf16", 2)(y)

    def represent(self, obs):
        rep = ref.representation
        x = self.store(self.conv(self.store(obs), rep.conv1, rep.bn1, 2))
        for d in rep.downsample:
            x = self.store(self.conv(x, d.conv, d.bn, 2))
            for blk in d.blocks:
                x = self.store(self.res(x, blk))
        for blk in rep.resblocks:
            last = self.res(x, blk)
            x = self.store(last)
        return x, last.mean(dim=(2, 3))      # the kernels take the spatial mean of the float32 values

    def dynamics(self, h, action):
        d = ref.dynamics
        vec = d.conv_action.weight[:, :, 0, 0].t()[action.view(-1)] + d.conv_action.bias      # [B, 256]
        x = self.store(h + vec.view(h.shape[0], -1, 1, 1))
        x = self.store(self.conv(x, d.conv_state, d.bn))
        for blk in d.resblocks:
            last = self.res(x, blk)
            x = self.store(last)
        return x, last.mean(dim=(2, 3))

    def head(self, seq, m):
        if self.head_r is None:
            return seq(m)
        r = self.head_r
        h = F.relu(F.linear(r(m), r(seq[0].weight), seq[0].bias))
        return F.linear(r(h), r(seq[2].weight), seq[2].bias)


def run(name, rw, rx, product="plain", head_r=None):
    e = Emu(rw, rx, head_r)
    e.product = product
    frames = np.random.default_rng(1234).integers(0, 256, size=(2, 96, 96, 96), dtype=np.uint8)
    obs = torch.tensor(oprep.prepare(frames))
    act = torch.tensor([[1], [A - 1]])
    with torch.no_grad():
        h0, m0 = e.represent(obs)
        out = {"h0": h0[:1], "p0": e.head(ref.prediction.policy_head, m0), "v0": e.head(ref.prediction.value_head, m0)}
        h1, m1 = e.dynamics(h0, act)
        out.update(h1=h1[:1], p1=e.head(ref.prediction.policy_head, m1), v1=e.head(ref.prediction.value_head, m1),
                   r1=e.head(ref.dynamics.reward_head, m1))
        h2, m2 = e.dynamics(h1, act.flip(0))
        out.update(p2=e.head(ref.prediction.policy_head, m2), v2=e.head(ref.prediction.value_head, m2),
                   r2=e.head(ref.dynamics.reward_head, m2))
        out["sv0"] = onets.decode_support(out["v0"], 300)
        out["sv1"] = onets.decode_support(out["v1"], 300)
        out["sr1"] = onets.decode_support(out["r1"], 300)
    row = {"scheme": name}
    for k, got in out.items():
        want = g[f"A{A}_{k}"]
        err = np.abs(got.numpy() - want)
        row[k] = {"x_bound": round(float((err / (2e-4 + 1e-3 * np.abs(want))).max()), 2),
                  "of_rms": round(float(err.max() / max(np.sqrt((want.astype(np.float64) ** 2).mean()), 1e-3)), 5)}
    row["worst_x_bound"] = max(v["x_bound"] for k, v in row.items() if isinstance(v, dict))
    print(json.dumps(row), flush=True)
    return row


ident = lambda t: t
rows = [
    run("fp32 (emulation's own floor)", ident, ident),
    run("bf16, bf16 heads", rounder("bf16", 1), rounder("bf16", 1), head_r=rounder("bf16", 1)),
    run("bf16, fp32 heads", rounder("bf16", 1), rounder("bf16", 1)),
    run("f16, f16 heads", rounder("f16", 1), rounder("f16", 1), head_r=rounder("f16", 1)),
    run("f16, fp32 heads", rounder("f16", 1), rounder("f16", 1)),
    run("bf16 wx2: (w_hi + w_lo) x_hi", rounder("bf16", 2), rounder("bf16", 1)),
    run("bf16 xw2: w_hi (x_hi + x_lo)", rounder("bf16", 1), rounder("bf16", 2)),
    run("f16 wx2: (w_hi + w_lo) x_hi", rounder("f16", 2), rounder("f16", 1)),
    run("f16 xw2: w_hi (x_hi + x_lo)", rounder("f16", 1), rounder("f16", 2)),
    run("bf16x3, fp32 heads", None, None, product="x3"),
]
print("\n| scheme | worst error / 1e-3 bound | hidden h0 | policy p0 | value v0 | p1 | v1 | r1 | decoded sv1 |")
print("|---|---|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| {r['scheme']} | {r['worst_x_bound']} | " + " | ".join(
        f"{r[k]['x_bound']} ({r[k]['of_rms'] * 100:.3f} % RMS)" for k in ("h0", "p0", "v0", "p1", "v1", "r1", "sv1")) + " |")
