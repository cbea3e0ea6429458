#!/usr/bin/env python3
"""Where the cycles of the convolution chain go: run the dynamics-shaped chain (5 layers, B=2048:
four passes of two images per workgroup; with MZ_CHAIN_NO_PAIRS=1 eight passes of one image) on a
diagnostic build of the library whose kernels record the shader clock at the phase boundaries of
every layer, and print per-layer averages.  Phases of the one-image kernel: main loop, epilogue,
barrier.  Of the two-image kernel: main loop, wait until every wavefront has finished reading
(printed as "epilogue"), epilogue + barrier (printed as "barrier wait").

    make -C muzero_amd/csrc stamps            # builds muzero_amd/csrc/libmuzero_amd_stamps.so
    MUZERO_AMD_LIB=$PWD/muzero_amd/csrc/libmuzero_amd_stamps.so python tools/chain_phase_stamps.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from muzero_amd import _lib, mfma  # noqa: E402

B, n_layers = 2048, 5
keep = []
arr = (_lib.ChainLayer * n_layers)()
for l in range(n_layers):
    w = mfma.pack_weights(torch.randn(256, 256, 3, 3) / 48.0, 256).cuda(); b = (torch.randn(256) * 0.1).cuda()
    keep += [w, b]
    arr[l] = _lib.ChainLayer(w.data_ptr(), b.data_ptr(), 1, 1 if (l % 2 == 0 and l > 0) else 0)
x = torch.randn(B, 12, 12, 256, device="cuda").to(torch.bfloat16); out = torch.empty_like(x)
mean = torch.empty(B, 256, device="cuda")
for _ in range(5):
    _lib.call("mz_conv_chain_bf16", _lib.ptr(x), None, 0, arr, n_layers, _lib.ptr(out), None, None, _lib.ptr(mean), B, _lib.current_stream())
torch.cuda.synchronize()
buf = (C.c_longlong * 8192)()
_lib.lib().mz_debug_read_stamps.restype = C.c_int
print("rc", _lib.lib().mz_debug_read_stamps(buf, 8192))
st = np.array(buf[:2 * 2 * 4 * 8 * 4]).reshape(2, 2, 4, 8, 4)[:, :, :, :5]
loop = st[..., 1] - st[..., 0]; epi = st[..., 2] - st[..., 1]; bar = st[..., 3] - st[..., 2]
print("cycles per layer (mean over 2 WGs x 2 waves x 4 images), by layer: main loop", loop.mean((0, 1, 2)).round(0))
print("epilogue", epi.mean((0, 1, 2)).round(0)); print("barrier wait", bar.mean((0, 1, 2)).round(0))
gap = st[:, :, :, 1:, 0] - st[:, :, :, :-1, 3]
print("after-barrier to next layer start", gap.mean().round(0))
img = st[:, :, 1:, 0, 0] - st[:, :, :-1, 4, 3]
print("between images (store + load):", img.mean().round(0))
print("whole image:", (st[:, :, :, 4, 3] - st[:, :, :, 0, 0]).mean().round(0), " first stamp to last:", st.max() - st.min())
print("wave0 vs wave5 main loop", loop[:, 0].mean().round(0), loop[:, 1].mean().round(0))
for w, name in enumerate(("wave 0", "wave 5")):
    print(name, "by layer: main loop", loop[:, w].mean((0, 1)).round(0), "epilogue", epi[:, w].mean((0, 1)).round(0),
          "barrier wait", bar[:, w].mean((0, 1)).round(0))
