"""The training side of the loop: episode records and n-step targets, the replay buffer, and
the K-step unrolled training step that produces the weights the search path consumes.

Counterparts in the reference:
  GameHistory          Game                        train.py:66-129
  ReplayBuffer         ReplayBuffer                model.py:549-618, compute_priorities train.py:542-550
  Learner              compute_losses              train.py:330-431
                       update_weights              train.py:295-328 (SGD + momentum + weight decay,
                                                   StepLR, gradient clipping; train.py:156-166, :309-313)
  support_ce_loss      scalar_to_support_loss      model.py:132-140

Division of labour: the learner is the one place where gradients are needed.  It keeps the
network as torch modules (same parameter names as the reference, BatchNorm in training mode
as train.py:155 leaves it) under torch autograd; with precision "bf16" the 3x3 convolutions
(forward, data gradient, weight gradient) and the BatchNorm + ReLU around them are this
library's kernels, wrapped as autograd functions; the support cross-entropy (forward and
gradient) and the n-step targets are HIP kernels (``mz_support_ce``, ``mz_nstep_targets``).  After an update ``Learner.state_dict()`` is
broadcast (``muzero_amd.dist.broadcast_state_dict``) and loaded into the inference engine
(``MuZeroNetwork.load_state_dict``), which folds BatchNorm for the search.
"""
from typing import Dict, List, Optional, Sequence, Tuple

import ctypes as C

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, checkpoint


# --------------------------------------------------------------------------- native convolutions
_PACK_EPOCH = [0]     # bumped at the start of every unrolled forward pass (Learner.losses_from_tensors)


# The reference architecture's 3x3 layers (model.py:146-148, :165, :182), (cin, cout, stride) -> input side: the
# shapes this library has forward, data-gradient and weight-gradient kernels for.
_NATIVE_LAYERS = {(96, 256, 2): 96, (256, 128, 2): 48, (128, 128, 1): 24, (128, 256, 2): 24, (256, 256, 1): 12}


class _PackedWeights:
    """Fragment-packed bf16 copies of one convolution's float32 weight (forward form and the form its data
    gradient reads: transposed and mirrored for a stride-1 layer, split by output parity for a stride-2 layer),
    repacked on the device when the parameter has changed (optimizer steps bump its version counter)."""

    def __init__(self, cin=256, cout=256, stride=1):
        self._slots = {}
        self.pending = None      # a list while the layer's weight gradients are deferred (see Conv3x3.defer_wgrad)
        self.cin, self.cout, self.stride = cin, cout, stride
        self._chunk = None

    @property
    def chunk(self) -> int:
        """Input-channel chunk of the forward kernel's fragment order."""
        if self._chunk is None:
            from . import mfma
            side = _NATIVE_LAYERS[(self.cin, self.cout, self.stride)]
            self._chunk = mfma.kernel_info(side, side, self.cin, self.cout, self.stride)[0]
        return self._chunk

    def key(self, weight: torch.Tensor):
        # while a HIP graph of the update is being recorded the packing must be part of it (a replay runs no
        # Python, so the version check cannot happen then): pack in every forward / backward pass
        epoch = _PACK_EPOCH[0] if torch.cuda.is_current_stream_capturing() else -1
        return (weight.data_ptr(), weight._version, epoch)

    def buffer(self, weight: torch.Tensor, transpose: bool) -> torch.Tensor:
        """The (possibly stale) packed buffer of one form, allocated on first use."""
        from . import mfma
        slot = self._slots.get(transpose)
        if slot is None:
            elems = (_lib.lib().mz_packed_dgrad_s2_elems(self.cout, self.cin) if transpose and self.stride == 2
                     else mfma.packed_elems(self.cout, self.cin))
            slot = self._slots[transpose] = (None, torch.empty(elems, dtype=torch.bfloat16, device=weight.device))
        return slot[1]

    def mark_current(self, weight: torch.Tensor, transpose: bool) -> None:
        self._slots[transpose] = (self.key(weight), self._slots[transpose][1])

    def get(self, weight: torch.Tensor, transpose: bool) -> torch.Tensor:
        """The packed form, repacked first if the weight has changed (a layer applied at several unroll steps
        packs once per pass; Learner packs all layers at the start of a pass in one launch)."""
        key = self.key(weight)
        slot = self._slots.get(transpose)
        if slot is None or slot[0] != key:
            buf = self.buffer(weight, transpose)
            if transpose and self.stride == 2:
                _lib.call("mz_weights_pack_dgrad_s2", _lib.ptr(weight.detach()), self.cout, self.cin, _lib.ptr(buf),
                          _lib.current_stream())
            else:
                _lib.call("mz_weights_fold_pack", _lib.ptr(weight.detach()), None, None, None, None, None, 0.0, self.cout,
                          self.cin, self.chunk, 0, 1 if transpose else 0, 0, _lib.ptr(buf), None, _lib.current_stream())
            self._slots[transpose] = (key, buf)
        return self._slots[transpose][1]


class _Conv3x3Native(torch.autograd.Function):
    """conv2d(x, w, b, stride, padding 1) for the layer shapes of _NATIVE_LAYERS on this library's kernels: forward
    = mz_conv3x3_bf16; data gradient = the same kernel on the transposed, mirrored weights (stride 1) or
    mz_conv_dgrad_s2_bf16 (stride 2); weight and bias gradient = mz_conv_wgrad_bf16.
    x: bf16 [N, Cin, H, W] in channels_last memory (= the kernels' NHWC)."""

    @staticmethod
    def _conv(x_nhwc, wpk, bias, cout, stride):
        n, h, w, cin = x_nhwc.shape
        out = torch.empty(n, h // stride, w // stride, cout, dtype=torch.bfloat16, device=x_nhwc.device)
        _lib.call("mz_conv3x3_bf16", _lib.ptr(x_nhwc), None, 0, _lib.ptr(wpk), _lib.ptr(bias), None, _lib.ptr(out), None,
                  None, None, n, h, w, cin, cout, stride, 0, _lib.current_stream())
        return out

    @staticmethod
    def forward(ctx, x, weight, bias, packed, want_stats=False):
        """want_stats (12x12 layers): also return double [N, 2, 256], per image the sum and the sum of squares of
        the outputs - the column sums a following training-mode BatchNorm would otherwise get from a pass of its own."""
        xh = x.permute(0, 2, 3, 1)
        if not xh.is_contiguous() or x.dtype != torch.bfloat16:
            raise ValueError("native convolution: bf16 channels_last input expected")
        ctx.save_for_backward(x, weight)
        ctx.packed = packed
        ctx.set_materialize_grads(False)     # no zero-filled "gradient" of the statistics output per backward call
        if not want_stats:
            out = _Conv3x3Native._conv(xh, packed.get(weight, False), bias.detach().float(), packed.cout, packed.stride)
            return out.permute(0, 3, 1, 2)
        n = xh.shape[0]
        out = torch.empty_like(xh)
        stats = torch.empty(n, 2, 256, dtype=torch.float64, device=x.device)
        _lib.call("mz_conv3x3_bf16_stats", _lib.ptr(xh), _lib.ptr(packed.get(weight, False)),
                  _lib.ptr(bias.detach().float()), _lib.ptr(out), _lib.ptr(stats), n, _lib.current_stream())
        ctx.mark_non_differentiable(stats)
        return out.permute(0, 3, 1, 2), stats

    @staticmethod
    def backward(ctx, grad, *unused):
        if grad is None:
            return None, None, None, None, None
        x, weight = ctx.saved_tensors
        pk = ctx.packed
        g = grad.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        gh, xh = g.permute(0, 2, 3, 1), x.permute(0, 2, 3, 1)
        n, h, w, cin = xh.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if pk.stride == 1:
                dx = _Conv3x3Native._conv(gh, pk.get(weight, True), _zero_bias(x.device, cin), cin, 1).permute(0, 3, 1, 2)
            elif cin == 96:      # the first layer: in the network its input is the observation and needs no gradient
                dx = torch.ops.aten.convolution_backward(g, x, weight.detach().to(torch.bfloat16), None, [2, 2], [1, 1],
                                                         [1, 1], False, [0, 0], 1, [True, False, False])[0]
            else:
                dxh = torch.empty_like(xh)
                _lib.call("mz_conv_dgrad_s2_bf16", _lib.ptr(gh), _lib.ptr(pk.get(weight, True)), _lib.ptr(dxh), n, h, w,
                          cin, pk.cout, _lib.current_stream())
                dx = dxh.permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1] and pk.pending is not None:
            pk.pending.append((gh, xh))       # summed over the unroll steps by Conv3x3.flush_wgrad
        elif ctx.needs_input_grad[1]:
            need = _lib.lib().mz_conv_wgrad_workspace_bytes_for(n, h, w, cin, pk.cout)
            ws = _wgrad_workspace(need, x.device)
            dw = torch.empty(pk.cout, cin, 3, 3, dtype=torch.float32, device=x.device)
            if ctx.needs_input_grad[2]:      # the bias gradient comes out of the same pass over grad
                db = torch.empty(pk.cout, dtype=torch.float32, device=x.device)
            _lib.call("mz_conv_wgrad_bf16", _lib.ptr(gh), _lib.ptr(xh), n, h, w, cin, pk.cout, _lib.ptr(dw), _lib.ptr(db),
                      _lib.ptr(ws), ws.numel() * 4, _lib.current_stream())
        elif ctx.needs_input_grad[2]:
            db = g.sum(dim=(0, 2, 3), dtype=torch.float32)
        return dx, dw, db, None, None


def pack_dgrad_s2(weight: torch.Tensor) -> torch.Tensor:
    """A stride-2 layer's float32 device weight [Cout][Cin][3][3] in the form mz_conv_dgrad_s2_bf16 reads."""
    cout, cin = weight.shape[0], weight.shape[1]
    out = torch.empty(_lib.lib().mz_packed_dgrad_s2_elems(cout, cin), dtype=torch.bfloat16, device=weight.device)
    _lib.call("mz_weights_pack_dgrad_s2", _lib.ptr(weight.detach().float().contiguous()), cout, cin, _lib.ptr(out),
              _lib.current_stream())
    return out


_WGRAD_WS: Dict[tuple, torch.Tensor] = {}
_ZERO_BIAS: Dict[tuple, torch.Tensor] = {}


def _zero_bias(device, channels: int = 256) -> torch.Tensor:
    """The all-zero bias of a data-gradient convolution (one per device and width, never written)."""
    key = (device.type, device.index, channels)
    t = _ZERO_BIAS.get(key)
    if t is None:
        t = _ZERO_BIAS[key] = torch.zeros(channels, dtype=torch.float32, device=device)
    return t



def _wgrad_workspace(nbytes: int, device) -> torch.Tensor:
    """One scratch buffer per device for the weight-gradient partial sums (stream-ordered reuse)."""
    key = (device.type, device.index)
    t = _WGRAD_WS.get(key)
    if t is None or t.numel() * 4 < nbytes:
        t = _WGRAD_WS[key] = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
    return t


class Conv3x3(nn.Conv2d):
    """nn.Conv2d(c_in, c_out, 3, stride, 1) whose instances with one of the reference architecture's layer shapes
    (_NATIVE_LAYERS) run on this library's kernels (forward, data gradient, weight gradient) when the learner
    computes in bf16; every other shape, the CPU and the float32 precision use torch's convolution.  Parameter
    names are nn.Conv2d's."""

    native = True        # Learner(native_convs=False) clears this on its modules

    def __init__(self, cin, cout, stride=1):
        super().__init__(cin, cout, 3, stride, 1)
        self._packed = _PackedWeights(cin, cout, stride)

    def defer_wgrad(self, on: bool = True) -> None:
        """A layer applied at several unroll steps (the dynamics network's): its backward passes only record
        their (grad_out, x) pairs, and ``flush_wgrad`` - called once after ``loss.backward()`` - turns all of
        them into the weight and bias gradients with ONE kernel pass (mz_conv_wgrad_bf16_multi)."""
        self._packed.pending = [] if on else None

    def flush_wgrad(self) -> None:
        pend = self._packed.pending
        if not pend:
            return
        self._packed.pending = []
        dev = self.weight.device
        k = 0
        while k < len(pend):
            n = pend[k][0].shape[0]
            seg = [pend[k]]
            while len(seg) < 8 and k + len(seg) < len(pend) and pend[k + len(seg)][0].shape[0] == n:
                seg.append(pend[k + len(seg)])
            k += len(seg)
            gp = (C.c_void_p * len(seg))(*[g.data_ptr() for g, _ in seg])
            xp = (C.c_void_p * len(seg))(*[x.data_ptr() for _, x in seg])
            ws = _wgrad_workspace(_lib.lib().mz_conv_wgrad_workspace_bytes(n * len(seg)), dev)
            dw = torch.empty(256, 256, 3, 3, dtype=torch.float32, device=dev)
            db = torch.empty(256, dtype=torch.float32, device=dev)
            _lib.call("mz_conv_wgrad_bf16_multi", gp, xp, len(seg), n, 12, 12, 256, 256, _lib.ptr(dw), _lib.ptr(db),
                      _lib.ptr(ws), ws.numel() * 4, _lib.current_stream())
            for p, g in ((self.weight, dw), (self.bias, db)):
                if p.requires_grad:
                    p.grad = g if p.grad is None else p.grad + g

    @staticmethod
    def flush_wgrad_group(convs) -> None:
        """The deferred weight (and bias) gradients of SEVERAL 12x12 256->256 layers from one kernel pass and one
        reduction (mz_conv_wgrad_bf16_group): each layer's recorded (grad_out, x) pairs are its segments.  Layers
        the group form cannot take (more than 8 segments, unequal batches) go through ``flush_wgrad``."""
        todo = []
        for m in convs:
            pend = m._packed.pending
            if not pend:
                continue
            same = all(g.shape[0] == pend[0][0].shape[0] for g, _ in pend)
            if (m.in_channels, m.out_channels, m.stride[0]) == (256, 256, 1) and len(pend) <= 8 and same:
                todo.append((m, pend))
                m._packed.pending = []
            else:
                m.flush_wgrad()
        while todo:                       # at most 32 layers and 64 (layer, segment) pairs per launch
            batch, pairs = [], 0
            while todo and len(batch) < 32 and pairs + len(todo[0][1]) <= 64:
                pairs += len(todo[0][1])
                batch.append(todo.pop(0))
            dev = batch[0][0].weight.device
            n = len(batch)
            arr = (_lib.WgradLayer * n)()
            keep, outs = [], []
            images = (C.c_int32 * n)()
            for i, (m, pend) in enumerate(batch):
                gp = (C.c_void_p * len(pend))(*[g.data_ptr() for g, _ in pend])
                xp = (C.c_void_p * len(pend))(*[x.data_ptr() for _, x in pend])
                dw = torch.empty(256, 256, 3, 3, dtype=torch.float32, device=dev)
                db = torch.empty(256, dtype=torch.float32, device=dev)
                keep += [gp, xp]
                outs.append((m, dw, db))
                arr[i] = _lib.WgradLayer(gp, xp, len(pend), pend[0][0].shape[0], dw.data_ptr(), db.data_ptr())
                images[i] = len(pend) * pend[0][0].shape[0]
            ws = _wgrad_workspace(_lib.lib().mz_conv_wgrad_group_workspace_bytes(images, n), dev)
            _lib.call("mz_conv_wgrad_bf16_group", arr, n, _lib.ptr(ws), ws.numel() * 4, _lib.current_stream())
            for m, dw, db in outs:
                for p, g in ((m.weight, dw), (m.bias, db)):
                    if p.requires_grad:
                        p.grad = g if p.grad is None else p.grad + g

    def runs_native(self, x) -> bool:
        side = _NATIVE_LAYERS.get((self.in_channels, self.out_channels, self.stride[0]))
        return (self.native and x.is_cuda and torch.is_autocast_enabled() and side is not None
                and tuple(x.shape[1:]) == (self.in_channels, side, side))

    def has_stats(self, x) -> bool:
        """Whether the forward kernel of this layer can hand a following BatchNorm its column sums."""
        return self.runs_native(x) and (self.in_channels, self.out_channels, self.stride[0]) == (256, 256, 1)

    def forward(self, x, want_stats: bool = False):
        """want_stats (native path only): returns (y, per-image column sums of y), see _Conv3x3Native.forward."""
        if self.runs_native(x):
            with torch.autocast("cuda", enabled=False):
                xb = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
                return _Conv3x3Native.apply(xb, self.weight, self.bias, self._packed, want_stats)
        if want_stats:
            raise ValueError("column sums come from the native convolution only")
        return super().forward(x)


def _nhwc(t):
    th = t.permute(0, 2, 3, 1)
    if not th.is_contiguous() or t.dtype != torch.bfloat16:
        raise ValueError("bf16 channels_last activations expected")
    return th


class _StepInput(torch.autograd.Function):
    """One pass over a hidden state h (mz_step_input_bf16): x = bf16(h + table[actions]) on every pixel - the dynamics
    network's hidden state plus the 1x1 convolution of the one-hot action planes (model.py:231-236) - and the
    spatial mean of h that the prediction heads read (model.py:273-281).  h: bf16 [N, C, H, W] channels_last;
    table: float32 [A, C]; actions: int32 [N].  Returns (x, mean float32 [N, C])."""

    @staticmethod
    def forward(ctx, h, table, actions):
        hh = _nhwc(h)
        n, hgt, wid, c = hh.shape
        x = torch.empty_like(hh)
        mean = torch.empty(n, c, dtype=torch.float32, device=h.device)
        _lib.call("mz_step_input_bf16", _lib.ptr(hh), _lib.ptr(table), _lib.ptr(actions), _lib.ptr(x), _lib.ptr(mean), n,
                  hgt * wid, c, _lib.current_stream())
        ctx.save_for_backward(actions)
        ctx.shape, ctx.n_actions = (n, hgt, wid, c), table.shape[0]
        ctx.set_materialize_grads(False)
        return x.permute(0, 3, 1, 2), mean

    @staticmethod
    def backward(ctx, dx, dmean):
        (actions,) = ctx.saved_tensors
        n, hgt, wid, c = ctx.shape
        if dx is None and dmean is None:
            return None, None, None
        dev = actions.device
        if dx is not None:
            dx = _nhwc(dx.to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
        if dmean is not None:
            dmean = dmean.float().contiguous()
        dh = torch.empty(n, hgt, wid, c, dtype=torch.bfloat16, device=dev)
        sums = dtable = None
        if ctx.needs_input_grad[1]:
            dtable = torch.empty(ctx.n_actions, c, dtype=torch.float32, device=dev)
            if dx is None:
                dtable.zero_()
            else:
                sums = torch.empty(n, c, dtype=torch.float32, device=dev)
        _lib.call("mz_step_input_grad_bf16", _lib.ptr(dx), _lib.ptr(dmean), _lib.ptr(actions), n, hgt * wid, c,
                  ctx.n_actions, _lib.ptr(dh), _lib.ptr(sums), _lib.ptr(dtable if sums is not None else None),
                  _lib.current_stream())
        return dh.permute(0, 3, 1, 2), dtable, None


class _PixelMean(torch.autograd.Function):
    """The spatial mean of a bf16 channels_last hidden state as float32 [N, C] (the same kernel without the
    action vector)."""

    @staticmethod
    def forward(ctx, h):
        hh = _nhwc(h)
        n, hgt, wid, c = hh.shape
        mean = torch.empty(n, c, dtype=torch.float32, device=h.device)
        _lib.call("mz_step_input_bf16", _lib.ptr(hh), None, None, None, _lib.ptr(mean), n, hgt * wid, c,
                  _lib.current_stream())
        ctx.shape = (n, hgt, wid, c)
        return mean

    @staticmethod
    def backward(ctx, dmean):
        n, hgt, wid, c = ctx.shape
        dh = torch.empty(n, hgt, wid, c, dtype=torch.bfloat16, device=dmean.device)
        _lib.call("mz_step_input_grad_bf16", None, _lib.ptr(dmean.float().contiguous()), None, n, hgt * wid, c, 0,
                  _lib.ptr(dh), None, None, _lib.current_stream())
        return dh.permute(0, 3, 1, 2)


_BN_WS: Dict[tuple, torch.Tensor] = {}


def _bn_workspace(C: int, device) -> torch.Tensor:
    key = (device.type, device.index, C)
    t = _BN_WS.get(key)
    if t is None:
        t = _BN_WS[key] = torch.empty(_lib.lib().mz_bn_workspace_bytes(C) // 8 + 1, dtype=torch.float64, device=device)
    return t


class _BNGradSum:
    """The running sums of one BatchNorm layer's weight / bias gradients over the unroll steps of a pass
    (mz_bn_relu_train_bwd_acc adds each step's into them); ``Learner._backward`` hands them to the parameters."""

    def __init__(self):
        self.dgamma = self.dbeta = None
        self.active = False          # set around loss.backward() by Learner._backward; otherwise autograd sums

    def take(self):
        out, self.dgamma, self.dbeta = (self.dgamma, self.dbeta), None, None
        return out


class _BNActNative(torch.autograd.Function):
    """relu(batch_norm(x) [+ residual]) in training mode on this library's kernels (mz_bn_relu_train_fwd / _bwd).
    x, residual: bf16 [N, C, H, W] in channels_last memory.  Running statistics are updated in place.
    grad_sum (a _BNGradSum or None): where the affine parameters' gradients are summed instead of being returned."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, running_mean, running_var, eps, momentum, stats=None, grad_sum=None):
        n, c, h, w = x.shape
        rows = n * h * w
        y = torch.empty_like(x)                       # keeps the channels_last strides
        save_mean = torch.empty(c, dtype=torch.float32, device=x.device)
        save_invstd = torch.empty_like(save_mean)
        ws = _bn_workspace(c, x.device)
        if stats is None:
            _lib.call("mz_bn_relu_train_fwd", _lib.ptr(x), _lib.ptr(residual), _lib.ptr(gamma.detach()),
                      _lib.ptr(beta.detach()), _lib.ptr(running_mean), _lib.ptr(running_var), float(eps), float(momentum),
                      1, rows, c, _lib.ptr(y), _lib.ptr(save_mean), _lib.ptr(save_invstd), _lib.ptr(ws), ws.numel() * 8,
                      _lib.current_stream())
        else:                                         # column sums of x came with x (per image, from the convolution)
            _lib.expect(stats, "stats", torch.float64, (stats.shape[0], 2, c))
            _lib.call("mz_bn_relu_train_fwd_stats", _lib.ptr(x), _lib.ptr(residual), _lib.ptr(gamma.detach()),
                      _lib.ptr(beta.detach()), _lib.ptr(running_mean), _lib.ptr(running_var), float(eps), float(momentum),
                      1, rows, c, _lib.ptr(y), _lib.ptr(save_mean), _lib.ptr(save_invstd), _lib.ptr(stats),
                      stats.shape[0], _lib.ptr(ws), ws.numel() * 8, _lib.current_stream())
        ctx.save_for_backward(x, y, gamma, save_mean, save_invstd)
        ctx.has_residual = residual is not None
        ctx.grad_sum = grad_sum
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, save_mean, save_invstd = ctx.saved_tensors
        n, c, h, w = x.shape
        dy = dy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_residual else None
        acc = ctx.grad_sum if ctx.grad_sum is not None and ctx.grad_sum.active else None
        first = acc is None or acc.dgamma is None
        if first:
            dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
            dbeta = torch.empty_like(dgamma)
            if acc is not None:
                acc.dgamma, acc.dbeta = dgamma, dbeta
        else:
            dgamma, dbeta = acc.dgamma, acc.dbeta
        ws = _bn_workspace(c, x.device)
        _lib.call("mz_bn_relu_train_bwd_acc", _lib.ptr(x), _lib.ptr(dy), _lib.ptr(y), _lib.ptr(gamma.detach()),
                  _lib.ptr(save_mean), _lib.ptr(save_invstd), 1, n * h * w, c, _lib.ptr(dx), _lib.ptr(dres),
                  _lib.ptr(dgamma), _lib.ptr(dbeta), 0 if first else 1, _lib.ptr(ws), ws.numel() * 8,
                  _lib.current_stream())
        if acc is not None:
            dgamma = dbeta = None            # handed over by Learner._backward once the pass is through
        return dx, dres, dgamma, dbeta, None, None, None, None, None, None


_BN_COUNTS: Optional[dict] = None


class count_bn_applications:
    """Context manager around a forward pass: the fused BatchNorm path's ``num_batches_tracked += 1`` of every
    application (one tiny launch each, ~50 per unrolled update) becomes one multi-tensor add at the end."""

    def __enter__(self):
        global _BN_COUNTS
        self._outer, _BN_COUNTS = _BN_COUNTS, {}
        return self

    def __exit__(self, *exc):
        global _BN_COUNTS
        counts, _BN_COUNTS = _BN_COUNTS, self._outer
        if counts and exc[0] is None:
            torch._foreach_add_([bn.num_batches_tracked for bn in counts], [int(n) for n in counts.values()])
        return False


def _bn_native(bn: nn.BatchNorm2d, x) -> bool:
    return (getattr(bn, "native", False) and bn.training and x.is_cuda and x.dtype == torch.bfloat16
            and torch.is_autocast_enabled() and x.dim() == 4 and x.shape[1] in (64, 128, 256, 512)
            and x.is_contiguous(memory_format=torch.channels_last) and bn.track_running_stats and bn.momentum is not None)


def conv_bn_act(conv: "Conv3x3", bn: nn.BatchNorm2d, x, residual=None):
    """relu(bn(conv(x)) [+ residual]): where both run on this library's kernels the convolution hands the
    BatchNorm its column sums, and the BatchNorm's own statistics pass over the activations falls away."""
    if conv.has_stats(x) and getattr(bn, "native", False) and bn.training and bn.track_running_stats \
            and bn.momentum is not None:
        y, stats = conv(x, want_stats=True)
        return bn_act(bn, y, residual, stats=stats)
    return bn_act(bn, conv(x), residual)


def bn_act(bn: nn.BatchNorm2d, x, residual=None, stats=None):
    """relu(bn(x) [+ residual]) (model.py:155-158, :166, :237).  The fused kernels serve training-mode BatchNorm on
    bf16 channels_last activations - what the native convolutions produce - when the learner computes in bf16;
    everything else goes through torch."""
    if _bn_native(bn, x):
        with torch.autocast("cuda", enabled=False):
            if residual is not None:
                residual = residual.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            if _BN_COUNTS is not None:          # counted once per pass by count_bn_applications
                _BN_COUNTS[bn] = _BN_COUNTS.get(bn, 0) + 1
            else:
                bn.num_batches_tracked.add_(1)
            return _BNActNative.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                                      bn.momentum, stats, getattr(bn, "grad_sum", None))
    if stats is not None:
        raise ValueError("column sums are for the fused BatchNorm path only")
    y = bn(x)
    if residual is not None:
        y = y + residual
    return F.relu(y)


# --------------------------------------------------------------------------- modules
class _Res(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv1 = Conv3x3(c, c)
        self.bn1 = nn.BatchNorm2d(c)
        self.conv2 = Conv3x3(c, c)
        self.bn2 = nn.BatchNorm2d(c)

    def forward(self, x):
        y = conv_bn_act(self.conv1, self.bn1, x)
        return conv_bn_act(self.conv2, self.bn2, y, residual=x)


class _Down(nn.Module):
    def __init__(self, cin, cout, n):
        super().__init__()
        self.conv = Conv3x3(cin, cout, 2)
        self.bn = nn.BatchNorm2d(cout)
        self.blocks = nn.ModuleList([_Res(cout) for _ in range(n)])

    def forward(self, x):
        x = bn_act(self.bn, self.conv(x))
        for b in self.blocks:
            x = b(x)
        return x


def _mlp(cin, hidden, cout):
    layers = []
    for h in hidden:
        layers += [nn.Linear(cin, h), nn.ReLU()]
        cin = h
    layers.append(nn.Linear(cin, cout))
    return nn.Sequential(*layers)


class _Representation(nn.Module):
    def __init__(self, cin, width, steps):
        super().__init__()
        self.conv1 = Conv3x3(cin, width, 2)
        self.bn1 = nn.BatchNorm2d(width)
        stages, c = [], width
        for cout, n in steps:
            stages.append(_Down(c, cout, n))
            c = cout
        self.downsample = nn.Sequential(*stages)
        self.resblocks = nn.ModuleList([_Res(c) for _ in range(6)])

    def forward(self, x):
        if self.conv1.native and x.is_cuda and torch.is_autocast_enabled():
            # NHWC bf16 from the first layer on: the library's kernels (and the fused BatchNorm) want it, and
            # MIOpen's NHWC kernels then run without a layout transpose around every call
            x = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        x = self.downsample(bn_act(self.bn1, self.conv1(x)))
        for b in self.resblocks:
            x = b(x)
        return x


class _Dynamics(nn.Module):
    def __init__(self, width, n_blocks, n_actions, fc_reward, support):
        super().__init__()
        self.n_actions = n_actions
        self.conv_action = nn.Conv2d(n_actions, width, 1)
        self.conv_state = Conv3x3(width, width)
        self.bn = nn.BatchNorm2d(width)
        self.resblocks = nn.ModuleList([_Res(width) for _ in range(n_blocks)])
        self.reward_head = _mlp(width, fc_reward, 2 * support + 1)

    def fused_input(self, h) -> bool:
        """Whether h can go through the one-pass kernel (native bf16 path, channels_last hidden state)."""
        return (self.conv_state.runs_native(h) and h.dtype == torch.bfloat16
                and h.is_contiguous(memory_format=torch.channels_last))

    def action_table(self):
        """float32 [A][width]: what conv_action adds to every pixel for each action (weight column + bias)."""
        with torch.autocast("cuda", enabled=False):
            w = self.conv_action.weight[:, :, 0, 0]                                   # [width][A]
            return (w.t() + self.conv_action.bias).float().contiguous()

    def step_input(self, h, action, table=None):
        """(hidden + conv1x1(one-hot planes), spatial mean of hidden) in one pass (see _StepInput).  table: the
        result of ``action_table()`` when the caller applies the network several times per pass."""
        if table is None:
            table = self.action_table()
        with torch.autocast("cuda", enabled=False):
            return _StepInput.apply(h, table, action.reshape(-1).to(torch.int32))

    def trunk_from(self, x):
        x = conv_bn_act(self.conv_state, self.bn, x)
        for b in self.resblocks:
            x = b(x)
        return x

    def trunk(self, h, action):
        """The next hidden state (model.py:231-241 without the reward head)."""
        # hidden + conv1x1(one-hot planes) == hidden + W[:, action] + bias on every pixel
        if self.fused_input(h):
            return self.trunk_from(self.step_input(h, action)[0])
        w = self.conv_action.weight[:, :, 0, 0]                      # [width][A]
        x = h + (w.t()[action.view(-1)] + self.conv_action.bias)[:, :, None, None]
        return self.trunk_from(x)

    def forward(self, h, action):
        x = self.trunk(h, action)
        return x, self.reward_head(x.mean(dim=(2, 3)))


class _Prediction(nn.Module):
    def __init__(self, width, n_actions, fc_policy, fc_value, support):
        super().__init__()
        self.policy_head = _mlp(width, fc_policy, n_actions)
        self.value_head = _mlp(width, fc_value, 2 * support + 1)

    def forward(self, h):
        m = h.mean(dim=(2, 3))
        return self.policy_head(m), self.value_head(m)


class TrainableNetwork(nn.Module):
    """The network with gradients; ``state_dict`` keys equal the reference's."""

    def __init__(self, cfg):
        super().__init__()
        self.representation = _Representation(cfg.observation_shape[0], cfg.num_channels, cfg.downsample_steps)
        self.dynamics = _Dynamics(cfg.num_channels, cfg.num_blocks, cfg.action_space_size, cfg.fc_reward_layers,
                                  cfg.support_size)
        self.prediction = _Prediction(cfg.num_channels, cfg.action_space_size, cfg.fc_policy_layers,
                                      cfg.fc_value_layers, cfg.support_size)

    def initial(self, obs):
        h = self.representation(obs)
        p, v = self.prediction(h)
        return h, p, v

    def recurrent(self, h, action):
        h2, r = self.dynamics(h, action)
        p, v = self.prediction(h2)
        return h2, r, p, v

    def unroll(self, obs, actions):
        """initial + K x recurrent (train.py:343-372) with the heads applied ONCE to the hidden states of all
        K + 1 steps: the heads are per-sample MLPs on the spatial mean, so (K + 1) x B rows through each head
        equal K + 1 applications to B rows - one sixth of the small GEMMs, bias / ReLU kernels and
        gradient-accumulation adds of an update.  actions: int64 [B, K].  Returns the policy and value logits
        [(K+1) B, .] (step-major: rows t B .. (t+1) B - 1 are step t) and the reward logits [K B, .] (steps 1..K)."""
        h = self.representation(obs)
        dyn = self.dynamics
        means = []
        table = dyn.action_table() if dyn.fused_input(h) else None      # once per pass: its gradient is summed first
        for t in range(actions.shape[1]):
            if dyn.fused_input(h):           # the step's input and the heads' mean from one pass over h
                x, mean = dyn.step_input(h, actions[:, t], table)
                means.append(mean)
                h = dyn.trunk_from(x)
            else:
                means.append(h.mean(dim=(2, 3)))
                h = dyn.trunk(h, actions[:, t])
        means.append(_PixelMean.apply(h) if dyn.fused_input(h) else h.mean(dim=(2, 3)))
        m = torch.cat(means)
        policy, value = self.prediction.policy_head(m), self.prediction.value_head(m)
        reward = self.dynamics.reward_head(m[means[0].shape[0]:]) if len(means) > 1 else None
        return policy, value, reward


# --------------------------------------------------------------------------- loss
class _SupportCEHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, support_size):
        x = logits.contiguous()
        t = targets.contiguous().float()
        loss = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        dlog = torch.empty_like(x)
        _lib.call("mz_support_ce", _lib.ptr(x), _lib.ptr(t), x.shape[0], support_size, _lib.ptr(loss),
                  _lib.ptr(dlog), _lib.current_stream())
        ctx.save_for_backward(dlog)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        (dlog,) = ctx.saved_tensors
        return dlog * grad_out.unsqueeze(1), None, None


def support_ce_loss(logits, targets, support_size: int, impl="hip"):
    """-(two_hot(transform(target)) * log_softmax(logits)).sum(1) (model.py:132-140).  impl "hip": the fused
    kernel (device tensors only; there is no CPU implementation in this package).  A callable
    ``impl(logits, targets, support_size)`` is used instead when given: the tests run the learner's host logic
    on the CPU with the reference's op sequence that way."""
    if callable(impl):
        return impl(logits, targets, support_size)
    if impl != "hip":
        raise ValueError("impl must be 'hip' or a callable")
    if not logits.is_cuda:
        raise _lib.MuZeroAmdError("support_ce_loss: the HIP kernel needs device tensors")
    return _SupportCEHip.apply(logits, targets, support_size)


# --------------------------------------------------------------------------- episodes and targets
class GameHistory:
    """One episode (train.py:66-129).  Observations are kept as the single new frame per
    step; the stacked observation of a position is rebuilt from the 32 frames before it."""

    def __init__(self, action_space_size: int, discount: float, frame_stack: int = 32, frame_channels: int = 3):
        self.action_space_size = action_space_size
        self.discount = discount
        self.frame_stack = frame_stack
        self.frame_channels = frame_channels
        self.done = False
        self.actions: List[int] = []
        self.frames: List[np.ndarray] = []
        self.rewards: List[float] = []
        self.values: List[float] = []
        self.policies: List[np.ndarray] = []
        self.environment_steps = 0

    def store_search_stats(self, root, action_probs: Dict[int, float]) -> None:
        self.values.append(root.value())
        policy = np.zeros(self.action_space_size)
        for a, p in action_probs.items():
            policy[a] = p
        self.policies.append(policy)

    def store_transition(self, action: Optional[int], observation: np.ndarray, reward: float) -> None:
        """``observation`` is either the new frame or the whole stacked observation the
        reference passes (train.py:246, :281); of a stack only the newest frame is kept."""
        self.actions.append(action if action is not None else 0)   # dummy action 0 at index 0 (train.py:123)
        if observation.ndim == 3 and observation.shape[2] == self.frame_channels * self.frame_stack:
            observation = np.ascontiguousarray(observation[:, :, -self.frame_channels:])
        self.frames.append(observation)
        self.rewards.append(reward)
        self.environment_steps += 1

    def get_observation(self, index: int) -> np.ndarray:
        """(H, W, 3*frame_stack) uint8, oldest first; the first frame pads the start (train.py:49-50)."""
        idx = [max(0, index - self.frame_stack + 1 + k) for k in range(self.frame_stack)]
        return np.concatenate([self.frames[i] for i in idx], axis=2)

    def compute_target_value(self, state_index: int, td_steps: int) -> float:
        value, discount = 0.0, 1.0
        for i in range(td_steps):
            idx = state_index + i
            if idx >= len(self.rewards):
                break
            value += self.rewards[idx] * discount
            discount *= self.discount
        if state_index + td_steps < len(self.values):
            value += self.values[state_index + td_steps] * discount
        return value

    def make_target(self, state_index: int, num_unroll_steps: int, td_steps: int):
        out = []
        for cur in range(state_index, state_index + num_unroll_steps + 1):
            v = self.compute_target_value(cur, td_steps)
            if cur < len(self.rewards):
                out.append((v, self.rewards[cur], self.policies[cur]))
            else:
                out.append((v, 0, np.zeros(self.action_space_size)))
        return out


class EpisodeStore:
    """Device-resident rewards, search values and search policies of the episodes in the replay
    window, stored back to back, so that ``make_target`` for a whole training batch is one
    kernel launch (``mz_nstep_targets``) with only the (game, position) pairs crossing PCIe.
    Games are addressed by their index in the window; ``pop_front`` follows the replay
    buffer's eviction of its oldest game."""

    def __init__(self, action_space_size: int, discount: float, device="cuda", capacity: int = 1 << 16,
                 keep_frames: bool = True):
        """keep_frames: also keep every position's new frame (uint8) on the device, so that the stacked
        observations of a training batch are gathered by one kernel (``observations``) instead of being
        concatenated on the host and copied (27 648 bytes per position for 96x96 RGB)."""
        self.A, self.discount, self.device = action_space_size, float(discount), torch.device(device)
        self.keep_frames = keep_frames
        self.frames = None                   # uint8 [positions, H*W, 3], allocated with the first game
        self.frame_stack = None
        self.rewards = torch.empty(capacity, dtype=torch.float64, device=self.device)
        self.values = torch.empty(capacity, dtype=torch.float64, device=self.device)
        self.policies = torch.empty(capacity, self.A, dtype=torch.float64, device=self.device)
        self.r_off: List[int] = [0]          # host copies of the offsets; the device copy is refreshed lazily
        self.v_off: List[int] = [0]
        self.base = 0                        # games [0, base) have been evicted
        self._dev_off = None

    def __len__(self):
        return len(self.r_off) - 1 - self.base

    @staticmethod
    def _grow(t: torch.Tensor, need: int) -> torch.Tensor:
        if need <= t.shape[0]:
            return t
        new = torch.empty((max(need, 2 * t.shape[0]),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        new[:t.shape[0]] = t
        return new

    def add_game(self, game: GameHistory) -> int:
        nr, nv = len(game.rewards), len(game.values)
        if len(game.policies) != nv:
            raise ValueError("an episode holds one search policy per search value")
        r0, v0 = self.r_off[-1], self.v_off[-1]
        self.rewards = self._grow(self.rewards, r0 + nr)
        self.values = self._grow(self.values, v0 + nv)
        self.policies = self._grow(self.policies, v0 + nv)
        if nr:
            self.rewards[r0:r0 + nr] = torch.from_numpy(np.asarray(game.rewards, np.float64)).to(self.device)
        if nv:
            self.values[v0:v0 + nv] = torch.from_numpy(np.asarray(game.values, np.float64)).to(self.device)
            self.policies[v0:v0 + nv] = torch.from_numpy(np.asarray(game.policies, np.float64)).to(self.device)
        if self.keep_frames and nr and len(getattr(game, "frames", ())) == nr and game.frames[0].ndim == 3 \
                and game.frames[0].shape[2] == 3:
            fr = torch.from_numpy(np.stack(game.frames))                 # [nr, H, W, 3] uint8: one frame per position
            if self.frames is None:
                self.frame_hw = tuple(fr.shape[1:3])
                self.frame_stack = game.frame_stack
                self.frames = torch.empty((max(r0 + nr, 1024),) + (fr.shape[1] * fr.shape[2], 3), dtype=torch.uint8,
                                          device=self.device)
            self.frames = self._grow(self.frames, r0 + nr)
            self.frames[r0:r0 + nr] = fr.view(nr, -1, 3).to(self.device, non_blocking=True)
        elif self.frames is not None:
            raise ValueError("every stored game must bring one HxWx3 frame per position once frames are kept")
        self.r_off.append(r0 + nr)
        self.v_off.append(v0 + nv)
        self._dev_off = None
        return len(self) - 1

    def pop_front(self) -> None:
        self.base += 1
        if self.base > 1024 and self.base > (len(self.r_off) - 1) // 2:     # compact: drop the evicted prefix
            r0, v0 = self.r_off[self.base], self.v_off[self.base]
            self.rewards = self.rewards[r0:].clone()
            if self.frames is not None:
                self.frames = self.frames[r0:].clone()
            self.values = self.values[v0:].clone()
            self.policies = self.policies[v0:].clone()
            self.r_off = [x - r0 for x in self.r_off[self.base:]]
            self.v_off = [x - v0 for x in self.v_off[self.base:]]
            self.base = 0
        self._dev_off = None

    def observations(self, q_game, q_pos) -> torch.Tensor:
        """float32 [n, 3 * frame_stack, H, W]: the prepared, stacked observation of every queried position
        (Game.get_observation + prepare_observation, train.py:128-129, :529-540), gathered on the device."""
        if self.frames is None:
            raise ValueError("this store keeps no frames")
        if self._dev_off is None:
            self._dev_off = (torch.tensor(self.r_off, dtype=torch.int64, device=self.device),
                             torch.tensor(self.v_off, dtype=torch.int64, device=self.device))
        qg = np.asarray(q_game, np.int64) + self.base
        qp = np.asarray(q_pos, np.int64)
        n = qg.shape[0]
        lens = np.diff(np.asarray(self.r_off))[qg] if n else np.zeros(0)
        if n == 0 or qg.min() < self.base or qg.max() >= len(self.r_off) - 1 or qp.min() < 0 or (qp >= lens).any():
            raise IndexError("observation query outside the stored episodes")
        d_g = torch.from_numpy(qg.astype(np.int32)).to(self.device)
        d_p = torch.from_numpy(qp.astype(np.int32)).to(self.device)
        H, W = self.frame_hw
        out = torch.empty(n, 3 * self.frame_stack, H, W, dtype=torch.float32, device=self.device)
        _lib.call("mz_gather_stacked_obs", _lib.ptr(self.frames), _lib.ptr(self._dev_off[0]), _lib.ptr(d_g), _lib.ptr(d_p),
                  n, self.frame_stack, H * W, _lib.ptr(out), _lib.current_stream())
        return out

    def targets(self, q_game, q_pos, K: int, td_steps: int):
        """-> float64 device tensors target_values [n, K+1], target_rewards [n, K+1],
        target_policies [n, K+1, A] (train.py:103-118 for every query)."""
        if self._dev_off is None:
            self._dev_off = (torch.tensor(self.r_off, dtype=torch.int64, device=self.device),
                             torch.tensor(self.v_off, dtype=torch.int64, device=self.device))
        qg = np.asarray(q_game, np.int64) + self.base
        qp = np.asarray(q_pos, np.int64)
        n = qg.shape[0]
        if n == 0 or qg.min() < self.base or qg.max() >= len(self.r_off) - 1 or qp.min() < 0:
            raise IndexError("target query outside the stored episodes")
        d_g = torch.from_numpy(qg.astype(np.int32)).to(self.device)
        d_p = torch.from_numpy(qp.astype(np.int32)).to(self.device)
        tv = torch.empty(n, K + 1, dtype=torch.float64, device=self.device)
        tr = torch.empty(n, K + 1, dtype=torch.float64, device=self.device)
        tp = torch.empty(n, K + 1, self.A, dtype=torch.float64, device=self.device)
        _lib.call("mz_nstep_targets", _lib.ptr(self.rewards), _lib.ptr(self.values), _lib.ptr(self.policies),
                  _lib.ptr(self._dev_off[0]), _lib.ptr(self._dev_off[1]), len(self.r_off) - 1, self.A, _lib.ptr(d_g),
                  _lib.ptr(d_p), n, K, td_steps, self.discount, _lib.ptr(tv), _lib.ptr(tr), _lib.ptr(tp),
                  _lib.current_stream())
        return tv, tr, tp


def batched_targets(games: Sequence[GameHistory], queries: Sequence[Tuple[int, int]], K: int, td_steps: int,
                    device="cuda"):
    """make_target for many (game, position) pairs of a list of episodes: builds a one-off
    ``EpisodeStore``.  A replay buffer keeps one instead (``ReplayBuffer(config, store)``)."""
    store = EpisodeStore(games[0].action_space_size, games[0].discount, device,
                         capacity=max(1, sum(max(len(g.rewards), len(g.values)) for g in games)))
    for g in games:
        store.add_game(g)
    q = np.asarray(queries, np.int64).reshape(-1, 2)
    return store.targets(q[:, 0], q[:, 1], K, td_steps)


class DevicePriorities:
    """The replay window's priorities on the device, one float64 per stored position, oldest game first,
    plus the first flat index of every game.  Draws (model.py:580-596) and priority updates
    (model.py:612-614) run as kernels (``mz_replay_sample``, ``mz_replay_update``), so a training batch
    is chosen and re-prioritised without the priorities crossing PCIe."""

    def __init__(self, device="cuda", capacity: int = 1 << 16):
        self.device = torch.device(device)
        self._prio = torch.empty(max(capacity, 1), dtype=torch.float64, device=self.device)
        self._head = self._tail = 0
        self._lengths: List[int] = []
        self._starts = None          # device int64 [n_games + 1], refreshed lazily
        self._ws = None

    def __len__(self):
        return self._tail - self._head

    @property
    def values(self) -> torch.Tensor:
        """A view of the live priorities (writes go through)."""
        return self._prio[self._head:self._tail]

    def append_game(self, priorities) -> None:
        v = torch.as_tensor(np.asarray(priorities, np.float64))
        n = v.numel()
        if self._tail + n > self._prio.numel():        # compact, then grow if still short
            live = self._prio[self._head:self._tail]
            fresh = torch.empty(max(self._prio.numel(), 2 * (live.numel() + n)), dtype=torch.float64,
                                device=self.device)
            fresh[:live.numel()] = live
            self._prio, self._head, self._tail = fresh, 0, live.numel()
        self._prio[self._tail:self._tail + n] = v.to(self.device)
        self._tail += n
        self._lengths.append(n)
        self._starts = None

    def pop_front(self) -> None:
        self._head += self._lengths.pop(0)
        self._starts = None

    def sample(self, uniforms, beta: float):
        """uniforms: [batch] float64 draws in [0, 1) (array or device tensor).  -> device tensors
        (flat int32, game int32, position int32, weights float64)."""
        n = len(self)
        if n == 0:
            raise IndexError("sampling from an empty replay window")
        u = torch.as_tensor(uniforms, dtype=torch.float64).to(self.device).contiguous()
        batch = u.numel()
        if self._starts is None:
            self._starts = torch.tensor(np.concatenate([[0], np.cumsum(self._lengths)]), dtype=torch.int64,
                                        device=self.device)
        need = _lib.lib().mz_replay_workspace_bytes(n)
        if self._ws is None or self._ws.numel() * 8 < need:
            self._ws = torch.empty(max(need // 8 * 2, 64), dtype=torch.float64, device=self.device)
        flat = torch.empty(batch, dtype=torch.int32, device=self.device)
        game, pos = torch.empty_like(flat), torch.empty_like(flat)
        w = torch.empty(batch, dtype=torch.float64, device=self.device)
        prio = self.values
        _lib.call("mz_replay_sample", _lib.ptr(prio), n, _lib.ptr(u), batch, float(beta), _lib.ptr(self._starts),
                  len(self._lengths), _lib.ptr(flat), _lib.ptr(game), _lib.ptr(pos), _lib.ptr(w), _lib.ptr(self._ws),
                  self._ws.numel() * 8, _lib.current_stream())
        return flat, game, pos, w

    def update(self, flat, values, alpha: float) -> None:
        f = torch.as_tensor(flat).to(self.device, torch.int32).contiguous()
        v = torch.as_tensor(values, dtype=torch.float64).to(self.device).contiguous()
        if f.numel() != v.numel():
            raise ValueError("one priority per index")
        if f.numel() == 0:
            return
        _lib.call("mz_replay_update", _lib.ptr(self.values), len(self), _lib.ptr(f), _lib.ptr(v), f.numel(),
                  float(alpha), _lib.current_stream())


class ReplayBuffer:
    """Prioritised replay over positions (model.py:549-618), host side.  One departure from
    the reference, on purpose: ``sample_batch`` returns the FLAT position index of every
    sample, so ``update_priorities`` writes the slots that were sampled (the reference keeps
    the index after it was reduced to an in-game offset, model.py:595 / :607, and so updates
    the wrong slots)."""

    def __init__(self, config, episode_store: Optional[EpisodeStore] = None,
                 device_priorities: Optional[DevicePriorities] = None):
        self.window_size = config.replay_buffer_size
        self.batch_size = config.batch_size
        self.alpha = config.priority_alpha
        self.beta = config.priority_beta
        self.buffer: List[GameHistory] = []
        self._prio = np.empty(1024, np.float64)   # one priority per position, live part [_head, _tail)
        self._head = self._tail = 0
        self.store = episode_store           # device mirror for batched targets (optional)
        self.device_priorities = device_priorities   # device mirror of the priorities (optional): when given,
        #                                              sample_positions_device / update_priorities_device use it

    def __len__(self):
        return len(self.buffer)

    @property
    def priorities(self) -> np.ndarray:
        """One priority per stored position, oldest game first (a view: writes go through)."""
        return self._prio[self._head:self._tail]

    @priorities.setter
    def priorities(self, values) -> None:
        values = np.asarray(values, np.float64)
        self._prio = values.copy() if len(values) else np.empty(1024, np.float64)
        self._head, self._tail = 0, len(values)

    def _append(self, values: np.ndarray) -> None:
        n = len(values)
        if self._tail + n > len(self._prio):           # compact, then grow if still short
            live = self._prio[self._head:self._tail]
            cap = max(len(self._prio), 2 * (len(live) + n))
            fresh = np.empty(cap, np.float64)
            fresh[:len(live)] = live
            self._prio, self._head, self._tail = fresh, 0, len(live)
        self._prio[self._tail:self._tail + n] = values
        self._tail += n

    def save_game(self, game: GameHistory, priorities: Optional[np.ndarray] = None) -> None:
        if priorities is None:
            priorities = np.full(len(game.rewards), self.priorities.max() if self._tail > self._head else 1)
        if len(priorities) != len(game.rewards):
            raise ValueError(f"{len(priorities)} priorities for a game of {len(game.rewards)} positions: the buffer "
                             "keeps exactly one per position (sampling maps flat indices through game lengths)")
        self.buffer.append(game)
        self._append(np.asarray(priorities, np.float64))
        if self.store is not None:
            self.store.add_game(game)
        if self.device_priorities is not None:
            self.device_priorities.append_game(priorities)
        while len(self.buffer) > self.window_size:
            old = self.buffer.pop(0)
            self._head += len(old.rewards)
            if self.store is not None:
                self.store.pop_front()
            if self.device_priorities is not None:
                self.device_priorities.pop_front()

    def sample_positions(self, uniforms: Optional[np.ndarray] = None):
        """-> (game index, in-game position, flat index) arrays and importance weights.
        ``uniforms`` are the [0,1) draws np.random.choice would consume (global generator if None).
        Same arithmetic as model.py:580-588: sequential sum of the priorities, p / total, inversion
        of the cumulative distribution as np.random.choice does it."""
        prio = self.priorities
        n = len(prio)
        total = np.cumsum(prio)[-1]                     # left-to-right, as Python's sum()
        probs = np.full(n, 1 / n) if total == 0 else prio / total
        u = np.random.random_sample(self.batch_size) if uniforms is None else np.asarray(uniforms)
        cdf = np.cumsum(probs)
        cdf /= cdf[-1]
        flat = np.searchsorted(cdf, u, side="right")
        weights = (n * probs[flat]) ** -self.beta
        weights = weights / weights.max()
        starts = np.cumsum([0] + [len(g.rewards) for g in self.buffer])
        gi = np.searchsorted(starts, flat, side="right") - 1
        return gi, flat - starts[gi], flat, weights

    def update_priorities(self, flat_indices, priorities) -> None:
        idx = np.asarray(flat_indices, np.int64)
        pr = np.asarray(priorities, np.float64)
        for i, v in zip(idx, pr):                      # list order: the last of several entries for a slot wins
            self.priorities[i] = (v + 1e-6) ** self.alpha
        if self.device_priorities is not None:
            self.device_priorities.update(idx, pr, self.alpha)

    def sample_positions_device(self, uniforms=None):
        """``sample_positions`` drawn by the device kernel from the device copy of the priorities:
        -> device tensors (game, position, flat index, weights)."""
        if self.device_priorities is None:
            raise ValueError("this buffer was built without device_priorities")
        u = np.random.random_sample(self.batch_size) if uniforms is None else uniforms
        flat, game, pos, w = self.device_priorities.sample(u, self.beta)
        return game, pos, flat, w

    def update_priorities_device(self, flat_indices, priorities) -> None:
        """New priorities for sampled positions from device tensors (e.g. the per-sample losses of the
        update that consumed them); the host copy of the priorities is refreshed from the device."""
        self.device_priorities.update(flat_indices, priorities, self.alpha)
        self.priorities[:] = self.device_priorities.values.cpu().numpy()


def compute_priorities(game: GameHistory, td_steps: int) -> np.ndarray:
    """|search value - n-step target| + 1e-6 per position (train.py:542-550)."""
    return np.array([abs(game.values[i] - game.compute_target_value(i, td_steps)) + 1e-6
                     for i in range(len(game.values))])


# --------------------------------------------------------------------------- the training step
def unroll_actions(action_lists, K: int) -> np.ndarray:
    """[n, K] actions in the order the unroll consumes them.  The reference feeds the dynamics
    network at unroll step t the action list's entry t-1, and action 0 at step 0 and past the
    end of the list (train.py:370-373 builds the tensor, :410-413 uses it in the same
    iteration)."""
    out = np.zeros((len(action_lists), max(K, 1)), np.int64)
    for i, acts in enumerate(action_lists):
        for t in range(1, K):
            if t - 1 < len(acts):
                out[i, t] = acts[t - 1]
    return out[:, :K]


class Learner:
    def __init__(self, config, device="cuda", state_dict=None, loss_impl=None,
                 precision: str = "fp32", autotune: bool = True, data_parallel: bool = False,
                 native_convs: bool = True):
        """precision "fp32" is the reference's arithmetic (parity tests run in it); "bf16" runs the
        convolutions and linear layers under bfloat16 autocast with float32 master weights,
        BatchNorm statistics and losses - and, with native_convs, the 3x3 convolutions (forward, data
        gradient, weight gradient) on this library's MFMA kernels instead of MIOpen.
        autotune: see below."""
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        if autotune and torch.device(device).type == "cuda":
            # let MIOpen time its convolution solvers once per shape (process-wide switch): the
            # default heuristic picks slow ones for these 12x12 and 24x24 layers (28 -> 22 ms per
            # bf16 update at batch 128)
            torch.backends.cudnn.benchmark = True
        self.config = config
        self.data_parallel = data_parallel   # every rank trains on its own batch; gradients are averaged
        self.precision = precision
        self.device = torch.device(device)
        self.loss_impl = loss_impl or "hip"    # or a callable, see support_ce_loss
        self.network = TrainableNetwork(config).to(self.device)      # stays in training mode (train.py:155)
        if state_dict is not None:
            self.network.load_state_dict(state_dict)
        self.native_convs = bool(native_convs and precision == "bf16" and self.device.type == "cuda")
        for m in self.network.modules():
            if isinstance(m, (Conv3x3, nn.BatchNorm2d)):
                m.native = self.native_convs
        if self.native_convs:       # the dynamics network's BatchNorms run once per unroll step (see _backward)
            for m in self.network.dynamics.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.grad_sum = _BNGradSum()
        # bf16 learner on a GPU: torch's single-kernel SGD (weight decay, momentum and step in one pass over the
        # parameters instead of three); the float32 learner keeps the default, the reference's own (train.py:156-159)
        opt_kw = {"fused": True} if (precision == "bf16" and self.device.type == "cuda") else {}
        self.optimizer = torch.optim.SGD(self.network.parameters(), lr=config.lr_init, momentum=config.momentum,
                                         weight_decay=config.weight_decay, **opt_kw)
        self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=config.lr_decay_steps,
                                                         gamma=config.lr_decay_rate)
        self.training_step = 0

    def state_dict(self):
        return {k: v.detach().cpu() for k, v in self.network.state_dict().items()}

    def device_state_dict(self):
        """The parameters where they are (no copy): what ``loop.publish`` broadcasts and the search engine
        folds and packs on the device."""
        return {k: v.detach() for k, v in self.network.state_dict().items()}

    def save_checkpoint(self, path: str) -> None:
        """The reference's checkpoint dictionary (train.py:481-495)."""
        checkpoint.write(path, self.training_step, self.network, self.optimizer, self.scheduler, self.config)

    @classmethod
    def from_checkpoint(cls, path: str, device="cuda", loss_impl=None) -> "Learner":
        """Resume from a checkpoint written by the reference or by ``save_checkpoint``:
        weights, BatchNorm statistics, momentum buffers, learning-rate schedule and step."""
        ck = checkpoint.read(path)
        lrn = cls(checkpoint.config_from(ck), device=device, state_dict=ck["model_state_dict"], loss_impl=loss_impl)
        if ck.get("optimizer_state_dict") is not None:
            lrn.optimizer.load_state_dict(ck["optimizer_state_dict"])
        if ck.get("scheduler_state_dict") is not None:
            lrn.scheduler.load_state_dict(ck["scheduler_state_dict"])
        lrn.training_step = int(ck.get("training_step", 0))
        return lrn

    def compute_losses(self, batch, weights):
        """The reference's signature (train.py:330-333): ``batch`` is a list of (stacked uint8
        observation HWC, action list, [(value, reward, policy)] * (K+1)) as ``sample_batch``
        returns it."""
        K, A = self.config.num_unroll_steps, self.config.action_space_size
        n = len(batch)
        obs = torch.stack([torch.from_numpy(np.ascontiguousarray(o)) for o, _, _ in batch]).to(self.device)
        obs = obs.permute(0, 3, 1, 2).float() / 255.0                       # prepare_observation, train.py:529-540
        tv = np.zeros((n, K + 1)); tr = np.zeros((n, K + 1)); tp = np.zeros((n, K + 1, A))
        for i, (_, _, targets) in enumerate(batch):
            for t, (v, r, p) in enumerate(targets[:K + 1]):                 # missing entries stay 0 (train.py:363-368)
                tv[i, t], tr[i, t], tp[i, t] = v, r, p
        unroll = unroll_actions([a for _, a, _ in batch], K)
        return self.losses_from_tensors(obs, unroll, tv, tr, tp, weights)

    def losses_from_tensors(self, observations, actions, target_values, target_rewards, target_policies, weights):
        """train.py:343-431.  observations: float32 [B, C, H, W] (already prepared); actions
        int64 [B, K] in unroll order (see ``unroll_actions``); target_* [B, K+1(, A)];
        weights [B].  Returns (value_loss, reward_loss, policy_loss, per-sample loss) with the
        reference's averaging conventions."""
        cfg = self.config
        K, S = cfg.num_unroll_steps, cfg.support_size
        dev = self.device
        _PACK_EPOCH[0] += 1
        self._pack_native_weights()
        f32 = lambda x: torch.as_tensor(x, device=dev).float()
        tv, tr, tp, w = f32(target_values), f32(target_rewards), f32(target_policies), f32(weights)
        act = torch.as_tensor(actions, device=dev).long()
        # ONE autocast region around the whole unroll (its cache then casts every float32 weight to bf16 once per
        # update, not once per unroll step); the losses are computed in float32 with autocast switched off
        amp = torch.autocast(dev.type, dtype=torch.bfloat16, enabled=self.precision == "bf16")
        f32_math = torch.autocast(dev.type, enabled=False)
        B = tv.shape[0]
        with amp, count_bn_applications():
            policy_logits, value_logits, reward_logits = self.network.unroll(observations.to(dev), act[:, :K])
        with f32_math:
            # every step's loss in one go; row t B + b of the logits is step t of sample b
            step_major = lambda x: x.transpose(0, 1).reshape(x.shape[1] * B, *x.shape[2:])
            v_loss = support_ce_loss(value_logits.float(), step_major(tv), S, self.loss_impl).view(K + 1, B)
            p_loss = -(step_major(tp) * torch.log_softmax(policy_logits.float(), dim=1)).sum(1).view(K + 1, B)
            r_loss = torch.zeros(1, B, device=dev)                   # no reward prediction at the root
            if K > 0:
                r_loss = torch.cat([r_loss, support_ce_loss(reward_logits.float(), step_major(tr[:, 1:]), S,
                                                            self.loss_impl).view(K, B)])
            total = cfg.value_loss_weight * v_loss + cfg.reward_loss_weight * r_loss + cfg.policy_loss_weight * p_loss
            per_sample = (total * w).sum(0) / (K + 1)
            total_v, total_p = v_loss.mean(), p_loss.mean()
            total_r = r_loss[1:].mean() if K > 0 else torch.zeros(1, device=dev)
        return total_v, total_r, total_p, per_sample

    def update_weights(self, batch, weights):
        """One optimisation step (train.py:303-313).  Returns the metrics and the per-sample
        losses that become the new replay priorities (train.py:318-319)."""
        return self._step(*self.compute_losses(batch, weights))

    def update_from_tensors(self, observations, actions, target_values, target_rewards, target_policies, weights):
        """``update_weights`` for a batch already in tensor form (see ``losses_from_tensors``)."""
        return self._step(*self.losses_from_tensors(observations, actions, target_values, target_rewards,
                                                    target_policies, weights))

    # ---- the whole update as one HIP graph
    def update_graphed(self, observations, actions, target_values, target_rewards, target_policies, weights):
        """``update_from_tensors`` replayed from a HIP graph: losses, backward pass, gradient clipping and the
        SGD step of one batch shape are recorded once and then replayed - an eager update is ~2000 kernel
        launches and bound by the host that issues them.  Recording needs three eager passes first
        (allocations, solver choices, momentum buffers); they run on the caller's batch, and everything they
        change - parameters, BatchNorm statistics and counters, momentum - is put back before the recording, so
        that every call, the first included, is exactly ONE update per batch as train.py:303-313 (neither the
        schedule nor ``training_step`` moves during the warm-up).  The graph is recorded again when the batch shape
        or the learning rate changes.  Returns the same dictionary; the metrics are read back after the replay."""
        if self.data_parallel:
            raise NotImplementedError("the graphed update does not include the gradient all-reduce")
        dev = self.device
        ins = [torch.as_tensor(observations).to(dev, torch.float32), torch.as_tensor(actions).to(dev, torch.long),
               torch.as_tensor(target_values).to(dev, torch.float32), torch.as_tensor(target_rewards).to(dev, torch.float32),
               torch.as_tensor(target_policies).to(dev, torch.float32), torch.as_tensor(weights).to(dev, torch.float32)]
        key = (tuple(tuple(t.shape) for t in ins), tuple(g["lr"] for g in self.optimizer.param_groups))
        st = getattr(self, "_graph_state", None)
        if st is None or st["key"] != key:
            static = [t.clone() for t in ins]

            def body():
                v, r, p, per = self.losses_from_tensors(*static)
                loss = per.mean()
                self.optimizer.zero_grad(set_to_none=True)
                self._backward(loss)
                torch.nn.utils.clip_grad_norm_(self.network.parameters(), self.config.max_grad_norm)
                self.optimizer.step()
                return torch.stack([v.detach().reshape(()), r.detach().reshape(()), p.detach().reshape(()), loss.detach()]), per.detach()

            snapshot = self._snapshot_training_state()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):                         # allocations, solver choices, momentum buffers
                    body()
            torch.cuda.current_stream().wait_stream(side)
            self._restore_training_state(snapshot)
            graph = torch.cuda.CUDAGraph()
            self.optimizer.zero_grad(set_to_none=True)
            with torch.cuda.graph(graph):
                outs = body()
            st = self._graph_state = {"key": key, "graph": graph, "static": static, "outs": outs}
            return self._graph_replay(st, ins)             # (the recording itself executed nothing)
        return self._graph_replay(st, ins)

    def _snapshot_training_state(self):
        """Copies of everything an update changes: parameters and buffers, momentum."""
        net = {k: v.detach().clone() for k, v in self.network.state_dict().items()}
        momentum = {}
        for group in self.optimizer.param_groups:
            for p in group["params"]:
                buf = self.optimizer.state.get(p, {}).get("momentum_buffer")
                momentum[p] = None if buf is None else buf.detach().clone()
        return net, momentum

    def _restore_training_state(self, snapshot) -> None:
        """Write a snapshot back IN PLACE (the tensors keep their addresses: a graph recorded afterwards, and the
        optimizer, go on using them).  A momentum buffer that did not exist at the snapshot becomes zero, from
        which the next step leaves it equal to the gradient - what SGD's first step does."""
        net, momentum = snapshot
        with torch.no_grad():
            for k, v in self.network.state_dict().items():       # detached aliases of the live tensors
                v.copy_(net[k])
            for p, buf in momentum.items():
                cur = self.optimizer.state.get(p, {}).get("momentum_buffer")
                if cur is not None:
                    if buf is None:
                        cur.zero_()
                    else:
                        cur.copy_(buf)

    def _graph_replay(self, st, ins):
        for dst, src in zip(st["static"], ins):
            dst.copy_(src, non_blocking=True)
        st["graph"].replay()
        # the replay stepped the weights in place without touching their version counters, and refilled the shared
        # packed buffers from the weights as they were BEFORE its own step: forget what is packed, so that the
        # next eager pass packs again
        for m in getattr(self, "_native_conv_list", None) or ():
            for t, (_, buf) in list(m._packed._slots.items()):
                m._packed._slots[t] = (None, buf)
        self.scheduler.step()
        self.training_step += 1
        m, per = st["outs"]
        m = m.cpu().numpy()
        return {"value_loss": float(m[0]), "reward_loss": float(m[1]), "policy_loss": float(m[2]),
                "total_loss": float(m[3]), "priorities": per.cpu().numpy()}

    def _pack_native_weights(self) -> None:
        """Both packed forms (forward, data gradient) of every native convolution whose weight has changed since
        it was last packed, in ONE launch (mz_weights_pack_many) instead of one per layer and form at first use."""
        if not self.native_convs:
            return
        convs = getattr(self, "_native_conv_list", None)
        if convs is None:
            convs = self._native_conv_list = [m for m in self.network.modules() if isinstance(m, Conv3x3) and
                                              (m.in_channels, m.out_channels, m.stride[0]) in _NATIVE_LAYERS]
        todo = [(m, t) for m in convs for t in (False, True)
                if not (t and m is self.network.representation.conv1)       # its input needs no gradient
                and (m._packed._slots.get(t) is None or m._packed._slots[t][0] != m._packed.key(m.weight))]
        groups: Dict[tuple, list] = {}
        for m, t in todo:
            if t and m.stride[0] == 2:            # the parity-split form has a packing of its own
                m._packed.get(m.weight, True)
            else:
                groups.setdefault((m.out_channels, m.in_channels, m._packed.chunk), []).append((m, t))
        for (cout, cin, chunk), items in groups.items():
            n = len(items)
            w = (C.c_void_p * n)(*[m.weight.data_ptr() for m, _ in items])
            out = (C.c_void_p * n)(*[m._packed.buffer(m.weight, t).data_ptr() for m, t in items])
            tr = (C.c_int32 * n)(*[1 if t else 0 for _, t in items])
            _lib.call("mz_weights_pack_many", w, out, tr, n, cout, cin, chunk, _lib.current_stream())
            for m, t in items:
                m._packed.mark_current(m.weight, t)

    def _backward(self, loss) -> None:
        """loss.backward() with the weight gradients of all 12x12 256->256 convolutions computed after the pass by one
        kernel launch (the dynamics network's layers, applied once per unroll step, summed over their steps in it),
        and the dynamics BatchNorms adding each step's affine gradients into a running sum in their kernel - instead
        of a launch per layer and step and autograd's additions."""
        mods = list(self.network.dynamics.modules()) if self.native_convs else []
        norms = [m for m in mods if isinstance(m, nn.BatchNorm2d) and getattr(m, "grad_sum", None) is not None]
        # every 12x12 256->256 convolution of the network - the dynamics network's (applied once per unroll step) and
        # the representation network's - records its operands during the pass; one launch computes all of them after it
        convs = [m for m in self.network.modules() if isinstance(m, Conv3x3)
                 and (m.in_channels, m.out_channels, m.stride[0]) == (256, 256, 1)] if self.native_convs else []
        for m in convs:
            m.defer_wgrad(True)
        for m in norms:
            m.grad_sum.active = True
        try:
            loss.backward()
            Conv3x3.flush_wgrad_group(convs)
            for m in norms:
                for p, g in zip((m.weight, m.bias), m.grad_sum.take()):
                    if g is not None and p.requires_grad:
                        p.grad = g if p.grad is None else p.grad + g
        finally:
            for m in convs:
                m.defer_wgrad(False)
            for m in norms:
                m.grad_sum.take()
                m.grad_sum.active = False

    def _step(self, v, r, p, per):
        loss = per.mean()
        self.optimizer.zero_grad()
        self._backward(loss)
        if self.data_parallel:
            from . import dist as mzdist
            mzdist.average_gradients(self.network.parameters())     # one all-reduce per update
        torch.nn.utils.clip_grad_norm_(self.network.parameters(), self.config.max_grad_norm)
        self.optimizer.step()
        self.scheduler.step()
        self.training_step += 1
        return {"value_loss": v.item(), "reward_loss": r.item(), "policy_loss": p.item(), "total_loss": loss.item(),
                "priorities": per.detach().cpu().numpy()}
