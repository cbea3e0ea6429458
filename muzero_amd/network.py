"""The representation / dynamics / prediction networks on HIP kernels, behind the
reference's call surface (model.py:284-348).

``MuZeroNetwork(config).initial_inference(obs)`` and
``.recurrent_inference(hidden, action)`` take and return the reference's NCHW
float32 tensors.  The search driver (``muzero_amd.search``) uses the
``*_to_pool`` methods instead, which keep hidden states in a device pool in the
kernels' own NHWC layout and return only what the tree consumes (softmax priors,
decoded value and reward).

Inference semantics: BatchNorm uses its running statistics (eval mode), folded
into the convolution weights when a ``state_dict`` is loaded.  Parameter names
are the reference's, so a reference checkpoint's ``model_state_dict`` loads
directly.

Precisions: ``"bf16"`` - bf16 MFMA kernels with float32 accumulation for the reference's
default architecture (throughput); ``"f16"`` - the same kernels with operands in IEEE half (11 significant
bits instead of 8, the same matrix rate: the policy / value / reward outputs land within the 1e-3 bound of the
reference's float32 path on its fixtures, hidden states within ~0.5 % of their RMS; values beyond 65 504
saturate); ``"f16x2"`` - float32-accurate at two MFMAs per product: activations as (hi, lo) pairs of IEEE half,
weights rounded to half once (1e-3 against the reference's float32 path like bf16x3, at 2/3 of its matrix work);
``"bf16x3"`` - the same architecture with every float32 value
carried as a (hi, lo) pair of bf16 values and three MFMAs per product: float32-accurate outputs
(1e-3 against the reference's float32 path) at a third of the bf16 matrix rate; ``"fp32"`` - exact
float32 kernels on the vector units for any layer shape (validation).
"""
import ctypes as C
import math
from collections import OrderedDict
from typing import Dict, List, Optional

import torch

from . import _lib

BN_EPS = 1e-5


class NetworkOutput:
    """Same fields and helpers as model.py:284-304."""

    def __init__(self, value, reward, policy_logits, hidden_state):
        self.value = value
        self.reward = reward
        self.policy_logits = policy_logits
        self.hidden_state = hidden_state

    def scalar_value(self) -> float:
        return support_to_scalar(self.value, self.value.shape[-1] // 2).item()

    def scalar_reward(self) -> Optional[float]:
        if self.reward is None:
            return None
        return support_to_scalar(self.reward, self.reward.shape[-1] // 2).item()


def support_to_scalar(logits: torch.Tensor, support_size: int) -> torch.Tensor:
    """model.py:98-109 on the device: float32 logits [n, 2S+1] -> float32 [n]."""
    _lib.expect(logits, "logits", torch.float32, (logits.shape[0], 2 * support_size + 1), contiguous=False)
    logits = logits.contiguous()
    out = torch.empty(logits.shape[0], dtype=torch.float32, device=logits.device)
    _lib.call("mz_support_to_scalar", _lib.ptr(logits), logits.shape[0], support_size, _lib.ptr(out),
              _lib.current_stream())
    return out


# --------------------------------------------------------------------------- weights
def init_state_dict(cfg, seed: int = 0, randomize_bn: bool = False) -> "OrderedDict[str, torch.Tensor]":
    """Random parameters with the reference's names and shapes (torch default
    initialisation laws: U(+-1/sqrt(fan_in)) for conv / linear weights and biases,
    identity BatchNorm).  ``randomize_bn`` draws non-trivial BatchNorm statistics so
    that folding is exercised."""
    g = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def uni(shape, fan_in):
        k = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * k

    def conv(name, cin, cout, ks=3):
        sd[name + ".weight"] = uni((cout, cin, ks, ks), cin * ks * ks)
        sd[name + ".bias"] = uni((cout,), cin * ks * ks)

    def bn(name, c):
        if randomize_bn:
            sd[name + ".weight"] = torch.rand(c, generator=g) * 0.5 + 0.75
            sd[name + ".bias"] = torch.randn(c, generator=g) * 0.1
            sd[name + ".running_mean"] = torch.randn(c, generator=g) * 0.1
            sd[name + ".running_var"] = torch.rand(c, generator=g) + 0.5
        else:
            sd[name + ".weight"] = torch.ones(c)
            sd[name + ".bias"] = torch.zeros(c)
            sd[name + ".running_mean"] = torch.zeros(c)
            sd[name + ".running_var"] = torch.ones(c)
        sd[name + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    def res(name, c):
        conv(name + ".conv1", c, c); bn(name + ".bn1", c)
        conv(name + ".conv2", c, c); bn(name + ".bn2", c)

    def mlp(name, cin, hidden, cout):
        i = 0
        for h in hidden:
            sd[f"{name}.{i}.weight"] = uni((h, cin), cin)
            sd[f"{name}.{i}.bias"] = uni((h,), cin)
            cin = h
            i += 2
        sd[f"{name}.{i}.weight"] = uni((cout, cin), cin)
        sd[f"{name}.{i}.bias"] = uni((cout,), cin)

    W, A, S = cfg.num_channels, cfg.action_space_size, cfg.support_size
    conv("representation.conv1", cfg.observation_shape[0], W); bn("representation.bn1", W)
    c = W
    for i, (cout, n) in enumerate(cfg.downsample_steps):
        conv(f"representation.downsample.{i}.conv", c, cout); bn(f"representation.downsample.{i}.bn", cout)
        for j in range(n):
            res(f"representation.downsample.{i}.blocks.{j}", cout)
        c = cout
    for j in range(6):
        res(f"representation.resblocks.{j}", c)
    conv("dynamics.conv_action", A, W, 1)
    conv("dynamics.conv_state", W, W); bn("dynamics.bn", W)
    for j in range(cfg.num_blocks):
        res(f"dynamics.resblocks.{j}", W)
    mlp("dynamics.reward_head", W, cfg.fc_reward_layers, 2 * S + 1)
    mlp("prediction.policy_head", W, cfg.fc_policy_layers, A)
    mlp("prediction.value_head", W, cfg.fc_value_layers, 2 * S + 1)
    return sd


class ConvLayer:
    """One 3x3 convolution + the BatchNorm behind it, as the kernels see them: shapes and the device
    buffers (``dev``) holding the folded bias and the weights in the precision's layout."""

    def __init__(self, sd, conv: str, bn: str, stride: int):
        w = sd[conv + ".weight"]
        self.cout, self.cin = int(w.shape[0]), int(w.shape[1])
        self.stride = stride
        self.name, self.bn = conv, bn
        self.dev: Dict[str, torch.Tensor] = {}


class _HeadWeights(C.Structure):
    _fields_ = [("w1p", C.c_void_p), ("b1", C.c_void_p), ("w2p", C.c_void_p), ("b2", C.c_void_p),
                ("bf16", C.c_int32), ("reserved", C.c_int32)]


class Head:
    """One fully connected head with a single hidden layer; its weights live in persistent device
    buffers in MFMA fragment order (``mz_pack_linear`` refills them in place)."""

    def __init__(self, sd, name: str, device, bf16: int = 0):
        """bf16: weight format - 0 float32, 1 bf16, 2 IEEE half (mz_head_weights.bf16)."""
        keys = sorted({int(k.split(".")[-2]) for k in sd if k.startswith(name + ".") and k.endswith(".weight")})
        if len(keys) != 2:
            raise NotImplementedError(f"{name}: exactly one hidden layer is supported (got {len(keys) - 1})")
        self.name, (self.k1, self.k2) = name, keys
        w1, w2 = sd[f"{name}.{self.k1}.weight"], sd[f"{name}.{self.k2}.weight"]
        self.hidden, self.out, self.cin = int(w1.shape[0]), int(w2.shape[0]), int(w1.shape[1])
        if self.hidden % 16 or self.cin % 4:
            raise NotImplementedError(f"{name}: hidden width must be a multiple of 16 and the input of 4")
        self.bf16 = int(bf16) if (bf16 and self.hidden % 32 == 0 and self.cin % 32 == 0) else 0
        dt = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}[self.bf16]
        tiles = (self.out + 15) // 16
        self.w1p = torch.empty(self.hidden * self.cin, dtype=dt, device=device)
        self.b1 = torch.empty(self.hidden, dtype=torch.float32, device=device)
        self.w2p = torch.empty(tiles * 16 * self.hidden, dtype=dt, device=device)
        self.b2 = torch.empty(tiles * 16, dtype=torch.float32, device=device)
        self.c = _HeadWeights(self.w1p.data_ptr(), self.b1.data_ptr(), self.w2p.data_ptr(), self.b2.data_ptr(),
                              self.bf16, 0)

    def fill(self, dsd) -> None:
        """dsd: the state_dict as float32 device tensors."""
        n = self.name
        w1, b1 = dsd[f"{n}.{self.k1}.weight"], dsd[f"{n}.{self.k1}.bias"]
        w2, b2 = dsd[f"{n}.{self.k2}.weight"], dsd[f"{n}.{self.k2}.bias"]
        st = _lib.current_stream()
        _lib.call("mz_pack_linear", _lib.ptr(w1), _lib.ptr(b1), self.hidden, self.cin, int(self.bf16),
                  _lib.ptr(self.w1p), _lib.ptr(self.b1), st)
        _lib.call("mz_pack_linear", _lib.ptr(w2), _lib.ptr(b2), self.out, self.hidden, int(self.bf16),
                  _lib.ptr(self.w2p), _lib.ptr(self.b2), st)


# --------------------------------------------------------------------------- network
class MuZeroNetwork:
    def __init__(self, config, precision: str = "bf16", device="cuda", state_dict=None, seed: int = 0,
                 input_hw=(96, 96)):
        if precision not in ("bf16", "f16", "bf16x3", "f16x2", "fp32"):
            raise ValueError("precision must be 'bf16', 'f16', 'bf16x3', 'f16x2' or 'fp32'")
        _lib.lib()  # fail now, loudly, if the HIP library is missing
        self.config = config
        self.precision = precision
        self.device = torch.device(device)
        self.split = precision in ("bf16x3", "f16x2")   # activations are (hi, lo) 16-bit pairs: [B, 2, H, W, C]
        self.half = precision in ("f16", "f16x2")       # 16-bit operands in IEEE half instead of bf16
        self.two_products = precision == "f16x2"        # split activations, weights hi only
        self.act_dtype = {"fp32": torch.float32, "f16": torch.float16, "f16x2": torch.float16}.get(precision, torch.bfloat16)
        self.obs_dtype = {"bf16": torch.bfloat16, "f16": torch.float16}.get(precision, torch.float32)   # what represent() takes
        self.op16 = precision in ("bf16", "f16")   # one 16-bit value per activation: the throughput kernels
        self.A = config.action_space_size
        self.S = config.support_size
        self.W = config.num_channels
        self.cin = config.observation_shape[0]
        self._buffers: Dict[tuple, torch.Tensor] = {}
        self._chains: Dict[str, tuple] = {}
        self.dynamics_events = None  # set to a list to time every fused dynamics launch with HIP events
        self.heads_events = None     # the same for the heads launch of every recurrent step
        self.input_hw = tuple(input_hw)
        self.load_state_dict(state_dict if state_dict is not None else init_state_dict(config, seed))

    # ---- parameters
    def load_state_dict(self, sd) -> None:
        """Take the parameters of a state_dict (the reference's names; host or device tensors).  All
        arithmetic happens on the device: BatchNorm folding, rounding, fragment packing
        (``mz_weights_fold_pack``, ``mz_pack_linear``, ``mz_action_vectors``) - device tensors, e.g.
        the views into a broadcast buffer, never visit the host.  The kernels' weight buffers are
        allocated by the first call and REFILLED IN PLACE by later calls with the same shapes, so
        their addresses - in a recorded HIP graph of the search, say - stay valid; ``generation``
        counts the (re)allocations."""
        def own(v):
            # a float32 device tensor would come through .to() unchanged, i.e. as an alias of the caller's
            # (possibly live, still training) parameter: keep a copy, so that state_dict() goes on returning
            # what was loaded - the weights the search is using
            t = v.detach().to(self.device, torch.float32).contiguous()
            return t.clone() if t.data_ptr() == v.data_ptr() else t
        dsd = {k: (own(v) if v.is_floating_point() else v.detach().clone()) for k, v in sd.items()}
        structure = {k: tuple(v.shape) for k, v in sd.items()}       # compared as a mapping: key order is free
        if structure != getattr(self, "_structure", None):
            self._build(dsd)
            self._structure = structure
            self.generation = getattr(self, "generation", 0) + 1
        self._dsd = dsd
        self._fill()

    def state_dict(self):
        """The parameters last loaded (host copies, reference names)."""
        return {k: v.detach().cpu() for k, v in self._dsd.items()}

    def _build(self, sd) -> None:
        """Layer plan and device buffers for the shapes in ``sd``."""
        cfg = self.config
        rep: List[tuple] = [("conv", ConvLayer(sd, "representation.conv1", "representation.bn1", 2))]
        for i, (_, n) in enumerate(cfg.downsample_steps):
            p = f"representation.downsample.{i}"
            rep.append(("conv", ConvLayer(sd, p + ".conv", p + ".bn", 2)))
            for j in range(n):
                rep.append(("res", self._res(sd, f"{p}.blocks.{j}")))
        j = 0
        while f"representation.resblocks.{j}.conv1.weight" in sd:
            rep.append(("res", self._res(sd, f"representation.resblocks.{j}")))
            j += 1
        self.rep_plan = rep
        self.dyn_first = ConvLayer(sd, "dynamics.conv_state", "dynamics.bn", 1)
        self.dyn_blocks = [self._res(sd, f"dynamics.resblocks.{j}") for j in range(cfg.num_blocks)]
        self.action_vec = torch.empty(self.A, self.W, dtype=torch.float32, device=self.device)     # [A][W]
        # The heads of both 16-bit precisions run in IEEE half: their input is the float32 spatial mean, their FLOPs
        # are negligible, and with bf16 weights and activations their own rounding - not the trunk's, which
        # averages out in the mean - was three quarters of the bf16 precision's logit error
        hb = 2 if self.op16 else 0
        self.policy = Head(sd, "prediction.policy_head", self.device, hb)
        self.value_head = Head(sd, "prediction.value_head", self.device, hb)
        self.reward_head = Head(sd, "dynamics.reward_head", self.device, hb)
        if not (self.policy.hidden == self.value_head.hidden == self.reward_head.hidden):
            raise NotImplementedError("the three heads must share one hidden width")
        self._chains = {}
        H, W = self.input_hw
        for kind, item in self.rep_plan:
            for layer in ([item] if kind == "conv" else item):
                layer.hw = (H, W)
                H, W = (H - 1) // layer.stride + 1, (W - 1) // layer.stride + 1
        for layer in [self.dyn_first] + [l for blk in self.dyn_blocks for l in blk]:
            layer.hw = (H, W)
        for layer in self.all_convs():
            layer.dev["bias"] = torch.empty(layer.cout, dtype=torch.float32, device=self.device)
            if self.precision == "fp32":
                layer.chunk = 0
                layer.dev["w"] = torch.empty(9 * layer.cin * layer.cout, dtype=torch.float32, device=self.device)
            else:
                from . import mfma
                layer.chunk, layer.whole_image = mfma.kernel_info(layer.hw[0], layer.hw[1], layer.cin, layer.cout,
                                                                  layer.stride)
                n = mfma.packed_elems(layer.cout, layer.cin)
                layer.dev["w"] = torch.empty(n, dtype=torch.float16 if self.half else torch.bfloat16, device=self.device)
                if self.split and not self.two_products:   # the part of the float32 weight its bf16 rounding leaves out
                    layer.dev["w_lo"] = torch.empty(n, dtype=torch.bfloat16, device=self.device)

    def _fold_pack(self, layer: ConvLayer, mode: int, out: torch.Tensor, bias_out=None, rot_depth: int = 0) -> None:
        d = self._dsd
        bn = layer.bn
        _lib.call("mz_weights_fold_pack", _lib.ptr(d[layer.name + ".weight"]), _lib.ptr(d[layer.name + ".bias"]),
                  _lib.ptr(d[bn + ".weight"]), _lib.ptr(d[bn + ".bias"]), _lib.ptr(d[bn + ".running_mean"]),
                  _lib.ptr(d[bn + ".running_var"]), BN_EPS, layer.cout, layer.cin, layer.chunk, mode, 0, rot_depth,
                  _lib.ptr(out), _lib.ptr(bias_out), _lib.current_stream())

    def _fill(self) -> None:
        """Device copies of the folded weights in the layout the kernels read, into the existing buffers."""
        for layer in self.all_convs():
            self._fold_pack(layer, self._pack_mode, layer.dev["w"], layer.dev["bias"])
            if "w_lo" in layer.dev:
                self._fold_pack(layer, 1, layer.dev["w_lo"])
            if "rot" in layer.dev:
                self._fold_pack(layer, self._pack_mode, layer.dev["rot"], None, rot_depth=layer.cin // 3)
        d = self._dsd
        _lib.call("mz_action_vectors", _lib.ptr(d["dynamics.conv_action.weight"]),
                  _lib.ptr(d["dynamics.conv_action.bias"]), self.W, self.A, _lib.ptr(self.action_vec),
                  _lib.current_stream())
        for head in (self.policy, self.value_head, self.reward_head):
            head.fill(d)

    @property
    def _pack_mode(self) -> int:
        """mz_weights_fold_pack mode of this precision's main weight buffers."""
        return {"fp32": 2, "f16": 3, "f16x2": 3}.get(self.precision, 0)

    @staticmethod
    def _res(sd, p):
        return (ConvLayer(sd, p + ".conv1", p + ".bn1", 1), ConvLayer(sd, p + ".conv2", p + ".bn2", 1))

    def all_convs(self):
        for kind, item in self.rep_plan:
            yield from ([item] if kind == "conv" else item)
        yield self.dyn_first
        for blk in self.dyn_blocks:
            yield from blk

    # ---- scratch
    def _buf(self, tag, shape, dtype=None):
        dtype = dtype or self.act_dtype
        key = (tag, tuple(shape), dtype)
        t = self._buffers.get(key)
        if t is None:
            t = torch.empty(*shape, dtype=dtype, device=self.device)
            self._buffers[key] = t
        return t

    def act_shape(self, B, H, W, Cc):
        """Shape of an activation tensor of B images in this precision's layout."""
        return (B, 2, H, W, Cc) if self.split else (B, H, W, Cc)

    def hidden_shape(self, h, w):
        """Per-game shape of a hidden state in the device pool."""
        return (2, h, w, self.W) if self.split else (h, w, self.W)

    # ---- kernels
    def _conv(self, layer: ConvLayer, x, out, H, W, relu=True, residual=None, in_slot=None, slot_stride=0,
              actions=None, mean_out=None, weights=None):
        B = out.shape[0]
        if self.split:
            if in_slot is not None or actions is not None or mean_out is not None or weights is not None:
                raise NotImplementedError("bf16x3: gather / action / mean are fused only in the 12x12 chain kernel")
            if (H, W, layer.cin, layer.cout, layer.stride) == (12, 12, 256, 256, 1):
                raise NotImplementedError("bf16x3: 12x12 layers run through the chain kernel")
            if self.two_products:
                _lib.call("mz_conv3x3_f16x2", _lib.ptr(x), _lib.ptr(layer.dev["w"]), _lib.ptr(layer.dev["bias"]),
                          _lib.ptr(residual), _lib.ptr(out), B, H, W, layer.cin, layer.cout, layer.stride,
                          1 if relu else 0, _lib.current_stream())
                return
            _lib.call("mz_conv3x3_bf16x3", _lib.ptr(x), _lib.ptr(layer.dev["w"]), _lib.ptr(layer.dev["w_lo"]),
                      _lib.ptr(layer.dev["bias"]), _lib.ptr(residual), _lib.ptr(out), B, H, W, layer.cin, layer.cout,
                      layer.stride, 1 if relu else 0, _lib.current_stream())
            return
        weights = layer.dev["w"] if weights is None else weights
        if self.precision == "fp32":
            _lib.call("mz_conv3x3_f32", _lib.ptr(x), _lib.ptr(in_slot), slot_stride, _lib.ptr(weights),
                      _lib.ptr(layer.dev["bias"]), _lib.ptr(residual), _lib.ptr(out),
                      _lib.ptr(self.action_vec) if actions is not None else None, _lib.ptr(actions),
                      B, H, W, layer.cin, layer.cout, layer.stride, 1 if relu else 0, _lib.current_stream())
            if mean_out is not None:
                HO = (H - 1) // layer.stride + 1
                WO = (W - 1) // layer.stride + 1
                _lib.call("mz_spatial_mean_f32", _lib.ptr(out), _lib.ptr(mean_out), B, HO * WO, layer.cout,
                          _lib.current_stream())
        else:
            from . import mfma
            mfma.conv(layer, x, out, B, H, W, relu, residual, in_slot, slot_stride,
                      self.action_vec if actions is not None else None, actions, mean_out, weights, half=self.half)

    def _chain_ok(self, convs, H, W) -> bool:
        """True when the fused LDS-resident chain kernel serves these layers."""
        return (self.precision != "fp32" and H == 12 and W == 12 and 1 <= len(convs) <= 20
                and all(c.cin == 256 and c.cout == 256 and c.stride == 1 for c in convs))

    def _mid_blocks(self, plan, i, H, W, last) -> int:
        """Number of consecutive residual blocks from plan[i] on that the fused 24x24x128 kernel
        (mz_conv_chain_mid_bf16) serves: at most 4 (its 8 layers), never the network's last block (the
        kernel does not produce the spatial mean)."""
        n = 0
        while (self.op16 and H == 24 and W == 24 and n < 4 and i + n < len(plan) and i + n != last
               and plan[i + n][0] == "res"
               and all(c.cin == 128 and c.cout == 128 and c.stride == 1 for c in plan[i + n][1])):
            n += 1
        return n

    def _chain(self, key, blocks, first, x, out, in_slot=None, slot_stride=0, actions=None, mean_out=None):
        """[first conv] + residual blocks as ONE kernel (mz_conv_chain_bf16)."""
        cached = self._chains.get(key)
        if cached is None:
            specs = []
            if first is not None:
                specs.append((first, 1, 0))
            for c1, c2 in blocks:
                specs += [(c1, 1, 0), (c2, 1, 1)]
            if self.split and not self.two_products:
                arr = (_lib.ChainLayerX3 * len(specs))()
                for i, (c, relu, res) in enumerate(specs):
                    arr[i] = _lib.ChainLayerX3(c.dev["w"].data_ptr(), c.dev["w_lo"].data_ptr(),
                                               c.dev["bias"].data_ptr(), relu, res)
            else:
                arr = (_lib.ChainLayer * len(specs))()
                for i, (c, relu, res) in enumerate(specs):
                    arr[i] = _lib.ChainLayer(c.dev["w"].data_ptr(), c.dev["bias"].data_ptr(), relu, res)
            cached = self._chains[key] = (arr, len(specs))
        arr, n = cached
        entry = {"bf16x3": "mz_conv_chain_bf16x3", "f16x2": "mz_conv_chain_f16x2", "f16": "mz_conv_chain_f16"}.get(
            self.precision, "mz_conv_chain_bf16")
        _lib.call(entry, _lib.ptr(x), _lib.ptr(in_slot), slot_stride, arr, n, _lib.ptr(out),
                  _lib.ptr(self.action_vec) if actions is not None else None, _lib.ptr(actions),
                  _lib.ptr(mean_out), out.shape[0], _lib.current_stream())

    def _res_block(self, blk, x, tmp, out, H, W, mean_out=None):
        """out = relu(conv2(relu(conv1(x))) + x); out may alias nothing but x's pool slot."""
        self._conv(blk[0], x, tmp, H, W, relu=True)
        self._conv(blk[1], tmp, out, H, W, relu=True, residual=x, mean_out=mean_out)

    def _heads(self, mean, B, with_reward: bool, want_logits: bool, tag: str = ""):
        """Search mode (want_logits False) writes into persistent buffers, so a steady-state
        search step allocates nothing; the reference-surface mode returns fresh tensors."""
        dev = self.device
        O = 2 * self.S + 1
        if want_logits:
            new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
            priors, value = new(B, self.A), new(B)
            reward = new(B) if with_reward else None
            pl, vl = new(B, self.A), new(B, O)
            rl = new(B, O) if with_reward else None
        else:
            priors = self._buf("priors" + tag, (B, self.A), torch.float32)
            value = self._buf("value" + tag, (B,), torch.float32)
            reward = self._buf("reward", (B,), torch.float32) if with_reward else None
            pl = vl = rl = None
        _lib.call("mz_heads", _lib.ptr(mean), B, self.W, self.policy.hidden, self.A, self.S,
                  C.byref(self.policy.c), C.byref(self.value_head.c),
                  C.byref(self.reward_head.c) if with_reward else None,
                  _lib.ptr(priors), _lib.ptr(value), _lib.ptr(reward), _lib.ptr(pl), _lib.ptr(vl), _lib.ptr(rl),
                  _lib.current_stream())
        return priors, value, reward, pl, vl, rl

    # ---- representation / dynamics on NHWC activations
    def hidden_hw(self, H: int, W: int):
        n = 1 + len(self.config.downsample_steps)
        for _ in range(n):
            H, W = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        return H, W

    def _rotated_weights(self, layer: ConvLayer, newest) -> torch.Tensor:
        """Device weights of the first convolution for an observation in rotating form
        (prep.FrameRing.rotating_observation): input channels permuted to the slot order.  All
        rotations are packed at once on first use (32 x 0.44 MB for the reference's stem) and
        refilled by every later load_state_dict.  newest None: the whole [D, packed] table."""
        cache = layer.dev.get("rot")
        if cache is None:
            if layer.cin % 3:
                raise ValueError("rotating observations need 3 channels per frame")
            D = layer.cin // 3
            if self.precision == "fp32":
                cache = torch.empty(D, 9 * layer.cin * layer.cout, dtype=torch.float32, device=self.device)
            else:
                from . import mfma
                cache = torch.empty(D, mfma.packed_elems(layer.cout, layer.cin),
                                    dtype=torch.float16 if self.half else torch.bfloat16, device=self.device)
            layer.dev["rot"] = cache
            self._fold_pack(layer, self._pack_mode, cache, None, rot_depth=D)
        return cache if newest is None else cache[newest]

    def represent(self, obs_nhwc: torch.Tensor, hidden_out: torch.Tensor, mean_out: torch.Tensor,
                  rotation: Optional[int] = None) -> None:
        """obs_nhwc [B,H,W,Cin] (activation dtype) -> hidden_out [B,h,w,W], mean_out float32 [B,W].
        rotation: newest slot of an observation in rotating form (None = reference channel order)."""
        rot_kernel = False
        if rotation is not None:   # [B, H, W/16, D, 56], see prep.FrameRing.enable_rotating
            from .prep import ROT_RUN
            if obs_nhwc.dim() != 5 or obs_nhwc.shape[3] * 3 != self.cin or obs_nhwc.shape[4] != ROT_RUN:
                raise ValueError("a rotating observation is [B, H, W/16, frames, 56]")
            B, H, W = obs_nhwc.shape[0], obs_nhwc.shape[1], obs_nhwc.shape[2] * 16
            _lib.expect(obs_nhwc, "observation", self.obs_dtype)
            first = self.rep_plan[0][1]
            rot_kernel = (self.op16 and (H, W, self.cin, first.cout) == (96, 96, 96, 256))
            if not rot_kernel:     # validation precision / other shapes: undo the tiling, keep the permuted weights
                from .prep import untile
                obs_nhwc = untile(obs_nhwc).contiguous()
        else:
            B, H, W = obs_nhwc.shape[0], obs_nhwc.shape[1], obs_nhwc.shape[2]
        pairs_in = self.split and rotation is None and obs_nhwc.dim() == 5     # already (hi, lo) pairs
        if pairs_in:
            H, W = obs_nhwc.shape[2], obs_nhwc.shape[3]
            _lib.expect(obs_nhwc, "observation", self.act_dtype, (B, 2, H, W, self.cin))
        hh, hw = self.hidden_hw(H, W)
        if not rot_kernel and not pairs_in:
            _lib.expect(obs_nhwc, "observation", self.obs_dtype, (B, H, W, self.cin))
        _lib.expect(hidden_out, "hidden_out", self.act_dtype, self.act_shape(B, hh, hw, self.W))
        _lib.expect(mean_out, "mean_out", torch.float32, (B, self.W))
        x = obs_nhwc
        if self.split:
            if rotation is not None:
                raise NotImplementedError("bf16x3 takes the observation in the reference's channel order")
            x = obs_nhwc if pairs_in else self._split(obs_nhwc)
        plan = self.rep_plan
        # the trailing residual blocks at the hidden resolution run as one fused chain
        n_tail = 0
        while n_tail < len(plan) and plan[len(plan) - 1 - n_tail][0] == "res":
            n_tail += 1
        tail = [item for _, item in plan[len(plan) - n_tail:]]
        fuse = n_tail > 0 and self._chain_ok([c for blk in tail for c in blk], hh, hw)
        head = plan[:len(plan) - n_tail] if fuse else plan
        last = len(plan) - 1
        skip = 0
        for i, (kind, item) in enumerate(head):
            final = i == last
            if skip:                   # a residual block that the fused kernel before it already covered
                skip -= 1
                continue
            if kind == "conv":
                HO, WO = (H - 1) // 2 + 1, (W - 1) // 2 + 1
                y = hidden_out if final else self._buf(("rep", i % 3), self.act_shape(B, HO, WO, item.cout))
                on_device = torch.is_tensor(rotation)     # the newest slot as a device int32 (HIP-graph replay)
                if on_device and not (i == 0 and rot_kernel):
                    if i == 0:
                        raise NotImplementedError("a device-side rotation index needs the rotating stem kernel")
                w = None
                if i == 0 and rotation is not None:
                    w = self._rotated_weights(item, None if on_device else rotation)
                if i == 0 and rot_kernel and on_device:
                    _lib.expect(rotation, "rotation", torch.int32, (1,))
                    _lib.call("mz_conv3x3_f16_rot_sel" if self.half else "mz_conv3x3_bf16_rot_sel", _lib.ptr(x), _lib.ptr(w),
                              w.shape[1], _lib.ptr(rotation),
                              _lib.ptr(item.dev["bias"]), _lib.ptr(y), B, 1, _lib.current_stream())
                elif i == 0 and rot_kernel:
                    _lib.call("mz_conv3x3_f16_rot" if self.half else "mz_conv3x3_bf16_rot", _lib.ptr(x), _lib.ptr(w),
                              _lib.ptr(item.dev["bias"]), _lib.ptr(y), B, 1, _lib.current_stream())
                else:
                    self._conv(item, x, y, H, W, relu=True, mean_out=mean_out if final else None, weights=w)
                x, H, W = y, HO, WO
            elif self._mid_blocks(head, i, H, W, last):
                # consecutive residual blocks at 24x24x128 (the first downsample stage's): one kernel,
                # the image resident in LDS
                n_blk = self._mid_blocks(head, i, H, W, last)
                skip = n_blk - 1
                y = self._buf(("rep", i % 3), self.act_shape(B, H, W, item[0].cout))
                key = ("mid", i)
                cached = self._chains.get(key)
                if cached is None:
                    convs = [c for _, blk in head[i:i + n_blk] for c in blk]
                    arr = (_lib.ChainLayer * len(convs))()
                    for j, c in enumerate(convs):
                        arr[j] = _lib.ChainLayer(c.dev["w"].data_ptr(), c.dev["bias"].data_ptr(), 1, j % 2)
                    cached = self._chains[key] = (arr, len(convs))
                _lib.call("mz_conv_chain_mid_f16" if self.half else "mz_conv_chain_mid_bf16", _lib.ptr(x), cached[0],
                          cached[1], _lib.ptr(y), B, _lib.current_stream())
                x = y
            else:
                c = item[0].cout
                tmp = self._buf(("rep_tmp", H), self.act_shape(B, H, W, c))
                y = hidden_out if final else self._buf(("rep", i % 3), self.act_shape(B, H, W, c))
                self._res_block(item, x, tmp, y, H, W, mean_out=mean_out if final else None)
                x = y
        if fuse:
            self._chain("rep_tail", tail, None, x, hidden_out, mean_out=mean_out)

    def dynamics(self, pool: torch.Tensor, in_slot: Optional[torch.Tensor], slot_stride: int,
                 actions: torch.Tensor, hidden_out: torch.Tensor, mean_out: torch.Tensor) -> None:
        """hidden' = dynamics(pool[in_slot[b]*slot_stride + b], actions[b]) for every b."""
        B, (H, W, c) = hidden_out.shape[0], hidden_out.shape[-3:]
        shape = self.act_shape(B, H, W, self.W)
        _lib.expect(hidden_out, "hidden_out", self.act_dtype, shape)
        _lib.expect(actions, "actions", torch.int32, (B,))
        _lib.expect(mean_out, "mean_out", torch.float32, (B, self.W))
        if in_slot is None:
            _lib.expect(pool, "hidden_state", self.act_dtype, shape)
        else:
            _lib.expect(in_slot, "in_slot", torch.int32, (B,))
            _lib.expect(pool, "pool", self.act_dtype)
            if pool.dim() != len(shape) + 1 or tuple(pool.shape[1:]) != shape or slot_stride != B:
                raise ValueError(f"pool must be [slots, {', '.join(map(str, shape))}] with slot_stride {B}")
        if self._chain_ok([self.dyn_first] + [l for blk in self.dyn_blocks for l in blk], H, W):
            timing = self.dynamics_events  # optional list: (start, end) event pairs around each launch
            if timing is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            self._chain("dynamics", self.dyn_blocks, self.dyn_first, pool, hidden_out, in_slot, slot_stride,
                        actions, mean_out)
            if timing is not None:
                e1.record()
                timing.append((e0, e1))
            return
        if self.split:
            raise NotImplementedError("bf16x3 serves the reference architecture's 12x12x256 dynamics network only")
        a = self._buf("dyn_a", (B, H, W, c))
        t = self._buf("dyn_t", (B, H, W, c))
        cur, spare = a, self._buf("dyn_b", (B, H, W, c))
        n = len(self.dyn_blocks)
        first_out = hidden_out if n == 0 else cur
        self._conv(self.dyn_first, pool, first_out, H, W, relu=True, in_slot=in_slot, slot_stride=slot_stride,
                   actions=actions, mean_out=mean_out if n == 0 else None)
        for j, blk in enumerate(self.dyn_blocks):
            y = hidden_out if j == n - 1 else spare
            self._res_block(blk, cur, t, y, H, W, mean_out=mean_out if j == n - 1 else None)
            cur, spare = y, cur

    # ---- search-facing entry points (hidden states stay in the pool)
    def initial_to_pool(self, obs_nhwc: torch.Tensor, hidden_out: torch.Tensor, rotation: Optional[int] = None):
        B = obs_nhwc.shape[0]
        mean = self._buf("mean", (B, self.W), torch.float32)
        self.represent(obs_nhwc, hidden_out, mean, rotation)
        priors, value, _, _, _, _ = self._heads(mean, B, False, False, tag="_root")  # own buffers: they
        return priors, value                                # outlive the recurrent steps that follow

    def recurrent_in_pool(self, pool, in_slot, slot_stride, actions, hidden_out):
        B = hidden_out.shape[0]
        mean = self._buf("mean", (B, self.W), torch.float32)
        self.dynamics(pool, in_slot, slot_stride, actions, hidden_out, mean)
        timing = self.heads_events
        if timing is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        priors, value, reward, _, _, _ = self._heads(mean, B, True, False)
        if timing is not None:
            e1.record()
            timing.append((e0, e1))
        return priors, value, reward

    # ---- the reference's call surface (NCHW float32 in and out)
    def _split(self, x: torch.Tensor) -> torch.Tensor:
        """float32 [B, ...] -> (hi, lo) 16-bit pairs [B, 2, ...] (bf16, or IEEE half for f16x2)."""
        x = x.contiguous()
        B, per = x.shape[0], x[0].numel()
        y = torch.empty((B, 2) + tuple(x.shape[1:]), dtype=self.act_dtype, device=self.device)
        _lib.call("mz_split_f32_half" if self.half else "mz_split_f32", _lib.ptr(x), _lib.ptr(y), B, per, _lib.current_stream())
        return y

    def _join(self, x: torch.Tensor) -> torch.Tensor:
        """(hi, lo) bf16 pairs [B, 2, ...] -> float32 [B, ...]."""
        B, per = x.shape[0], x[0, 0].numel()
        y = torch.empty((B,) + tuple(x.shape[2:]), dtype=torch.float32, device=self.device)
        _lib.call("mz_join_f32_half" if self.half else "mz_join_f32", _lib.ptr(x.contiguous()), _lib.ptr(y), B, per,
                  _lib.current_stream())
        return y

    def _to_nhwc(self, x_nchw: torch.Tensor, as_obs: bool = False) -> torch.Tensor:
        """NCHW float32 -> NHWC in the activation layout (as_obs: in the layout represent() takes)."""
        B, Cc, H, W = x_nchw.shape
        x = x_nchw.contiguous().float()
        y = torch.empty(B, H, W, Cc, dtype=torch.float32, device=self.device)
        _lib.call("mz_nchw_to_nhwc_f32", _lib.ptr(x), _lib.ptr(y), B, Cc, H * W, _lib.current_stream())
        if as_obs:
            return y if self.obs_dtype == torch.float32 else y.to(self.obs_dtype)
        if self.split:
            return self._split(y)
        return y if self.act_dtype == torch.float32 else y.to(self.act_dtype)

    def _to_nchw(self, x_nhwc: torch.Tensor) -> torch.Tensor:
        if self.split:
            x_nhwc = self._join(x_nhwc)
        B, H, W, Cc = x_nhwc.shape
        x = x_nhwc if x_nhwc.dtype == torch.float32 else x_nhwc.float()
        y = torch.empty(B, Cc, H, W, dtype=torch.float32, device=self.device)
        _lib.call("mz_nhwc_to_nchw_f32", _lib.ptr(x.contiguous()), _lib.ptr(y), B, Cc, H * W, _lib.current_stream())
        return y

    def initial_inference(self, observation: torch.Tensor) -> NetworkOutput:
        """model.py:332-338.  observation: float32 [B, C, H, W] on the device."""
        obs = self._to_nhwc(observation.to(self.device), as_obs=True)
        B, H, W, _ = obs.shape
        h, w = self.hidden_hw(H, W)
        hidden = torch.empty(self.act_shape(B, h, w, self.W), dtype=self.act_dtype, device=self.device)
        mean = self._buf("mean", (B, self.W), torch.float32)
        self.represent(obs, hidden, mean)
        _, _, _, pl, vl, _ = self._heads(mean, B, False, True)
        return NetworkOutput(vl, None, pl, self._to_nchw(hidden))

    def recurrent_inference(self, hidden_state: torch.Tensor, action: torch.Tensor) -> NetworkOutput:
        """model.py:340-348.  hidden_state float32 [B, C, h, w]; action integer [B, 1]."""
        hs = self._to_nhwc(hidden_state.to(self.device))
        B, (h, w) = hs.shape[0], hs.shape[-3:-1]
        actions = action.to(self.device).view(B).to(torch.int32).contiguous()
        if B and (int(actions.min()) < 0 or int(actions.max()) >= self.A):    # the kernel gathers by action
            raise IndexError(f"action outside [0, {self.A})")
        out = torch.empty(self.act_shape(B, h, w, self.W), dtype=self.act_dtype, device=self.device)
        mean = self._buf("mean", (B, self.W), torch.float32)
        self.dynamics(hs, None, 0, actions, out, mean)
        _, _, _, pl, vl, rl = self._heads(mean, B, True, True)
        return NetworkOutput(vl, rl, pl, self._to_nchw(out))
