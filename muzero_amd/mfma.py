"""Host side of the bf16 MFMA convolution (mz_conv3x3_bf16): weight packing into
MFMA fragment order and the launch wrapper."""
import ctypes as C

import torch

from . import _lib


def kernel_info(H: int, W: int, cin: int, cout: int, stride: int):
    """(cin_chunk, whole_image) of the kernel that serves this layer shape."""
    chunk, whole = C.c_int32(0), C.c_int32(0)
    _lib.call("mz_conv3x3_bf16_info", H, W, cin, cout, stride, C.byref(chunk), C.byref(whole))
    return chunk.value, bool(whole.value)


def pack_weights(w: torch.Tensor, chunk: int) -> torch.Tensor:
    """w: float32 [Cout][Cin][3][3] (BatchNorm already folded) -> bf16 fragments
    [cin_chunk][tap][k32][n16][lane = kq*16 + r][8], padded by two k-steps of zeros."""
    cout, cin = w.shape[0], w.shape[1]
    ncc, kc, nt = cin // chunk, chunk // 32, cout // 16
    x = w.to(torch.bfloat16).permute(2, 3, 1, 0).reshape(9, ncc, kc, 4, 8, nt, 16)
    #    dims:  tap, cc, kc, kq, j, nt, r   ->   cc, tap, kc, nt, kq, r, j
    x = x.permute(1, 0, 2, 5, 3, 6, 4).contiguous().view(-1)
    pad = torch.zeros(2 * nt * 512, dtype=torch.bfloat16)
    return torch.cat([x, pad])


def conv(layer, x, out, B, H, W, relu, residual, in_slot, slot_stride, action_vec, actions, mean_out, weights=None):
    """weights: packed fragments to use instead of the layer's own (same shape, e.g. permuted input channels)."""
    weights = layer.dev["w"] if weights is None else weights
    _lib.call("mz_conv3x3_bf16", _lib.ptr(x), _lib.ptr(in_slot), slot_stride, _lib.ptr(weights),
              _lib.ptr(layer.dev["bias"]), _lib.ptr(residual), _lib.ptr(out), _lib.ptr(action_vec),
              _lib.ptr(actions), _lib.ptr(mean_out), B, H, W, layer.cin, layer.cout, layer.stride,
              1 if relu else 0, _lib.current_stream())
