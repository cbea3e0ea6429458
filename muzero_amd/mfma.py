"""Host side of the bf16 MFMA convolution (mz_conv3x3_bf16): the launch wrapper, and the fragment
packing written out in torch (``pack_weights``) - the network packs on the device
(``mz_weights_fold_pack``); the torch form serves tests and tools, and checks the kernel."""
import ctypes as C

import torch

from . import _lib


def kernel_info(H: int, W: int, cin: int, cout: int, stride: int):
    """(cin_chunk, whole_image) of the kernel that serves this layer shape."""
    chunk, whole = C.c_int32(0), C.c_int32(0)
    _lib.call("mz_conv3x3_bf16_info", H, W, cin, cout, stride, C.byref(chunk), C.byref(whole))
    return chunk.value, bool(whole.value)


def packed_elems(cout: int, cin: int) -> int:
    """bf16 elements of one packed 3x3 layer, including the two k-steps of zeros behind the fragments."""
    return cout * cin * 9 + 2 * (cout // 16) * 512


def pack_weights(w: torch.Tensor, chunk: int, dtype=torch.bfloat16) -> torch.Tensor:
    """w: float32 [Cout][Cin][3][3] (BatchNorm already folded) -> fragments of ``dtype`` (bfloat16 or float16)
    [cin_chunk][tap][k32][n16][lane = kq*16 + r][8], padded by two k-steps of zeros."""
    cout, cin = w.shape[0], w.shape[1]
    ncc, kc, nt = cin // chunk, chunk // 32, cout // 16
    x = w.to(dtype).permute(2, 3, 1, 0).reshape(9, ncc, kc, 4, 8, nt, 16)
    #    dims:  tap, cc, kc, kq, j, nt, r   ->   cc, tap, kc, nt, kq, r, j
    x = x.permute(1, 0, 2, 5, 3, 6, 4).contiguous().view(-1)
    pad = torch.zeros(2 * nt * 512, dtype=dtype)
    return torch.cat([x, pad])


def pack_weights_batch(w: torch.Tensor, chunk: int) -> torch.Tensor:
    """pack_weights for a stack of weight tensors [n][Cout][Cin][3][3] on whatever device they are
    on -> [n][packed length]; row i equals pack_weights(w[i], chunk)."""
    n, cout, cin = w.shape[0], w.shape[1], w.shape[2]
    ncc, kc, nt = cin // chunk, chunk // 32, cout // 16
    x = w.to(torch.bfloat16).permute(0, 3, 4, 2, 1).reshape(n, 9, ncc, kc, 4, 8, nt, 16)
    x = x.permute(0, 2, 1, 3, 6, 4, 7, 5).reshape(n, -1)
    pad = torch.zeros(n, 2 * nt * 512, dtype=torch.bfloat16, device=w.device)
    return torch.cat([x, pad], dim=1).contiguous()


def conv(layer, x, out, B, H, W, relu, residual, in_slot, slot_stride, action_vec, actions, mean_out, weights=None,
         half=False):
    """weights: packed fragments to use instead of the layer's own (same shape, e.g. permuted input channels).
    half: activations and fragments are IEEE half (mz_conv3x3_f16) instead of bf16."""
    weights = layer.dev["w"] if weights is None else weights
    _lib.call("mz_conv3x3_f16" if half else "mz_conv3x3_bf16", _lib.ptr(x), _lib.ptr(in_slot), slot_stride, _lib.ptr(weights),
              _lib.ptr(layer.dev["bias"]), _lib.ptr(residual), _lib.ptr(out), _lib.ptr(action_vec),
              _lib.ptr(actions), _lib.ptr(mean_out), B, H, W, layer.cin, layer.cout, layer.stride,
              1 if relu else 0, _lib.current_stream())
