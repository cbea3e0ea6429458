"""CPU-side checks of the C-ABI library: it loads, exports every symbol the
header declares, and validates arguments without touching a GPU."""
import ctypes as C
import os
import re

import pytest

from muzero_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "muzero_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mz_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    handle = _lib.lib()
    names = header_symbols()
    assert "mz_tree_select" in names and len(names) >= 8
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/muzero_amd.h but not exported"
    assert set(_lib.declared_symbols()) == set(names)


def test_layout_query_and_argument_errors():
    L = _lib.TreeLayout()
    _lib.call("mz_tree_layout_query", 1024, 18, 51, C.byref(L))
    assert L.total_bytes > 1024 * 51 * 18 * 20
    offs = [getattr(L, f) for f, _ in _lib.TreeLayout._fields_[5:]]
    assert len(set(offs)) == len(offs) and all(o % 256 == 0 for o in offs)
    with pytest.raises(_lib.MuZeroAmdError, match="1<=A<=64"):
        _lib.call("mz_tree_layout_query", 4, 65, 51, C.byref(L))
    with pytest.raises(_lib.MuZeroAmdError, match="null"):
        _lib.call("mz_tree_select", None, C.byref(L), None, None, 10, None)


def test_argument_validation_needs_no_gpu():
    """Every entry point rejects bad arguments before touching the device."""
    L = _lib.TreeLayout()
    _lib.call("mz_tree_layout_query", 4, 6, 11, C.byref(L))
    cases = [
        ("mz_tree_root_init", (None, C.byref(L), None, None, 0.75, 0.25, None), "null"),
        ("mz_tree_expand_backup", (C.c_void_p(8), C.byref(L), None, None, None, 1, 0.997, None), "null network output"),
        ("mz_select_action", (None, 4, 6, 1.0, None, None, None), "bad arguments"),
        ("mz_conv3x3_f32", (None, None, 0) + (None,) * 6 + (1, 12, 12, 8, 8, 1, 1, None), "null tensor"),
        ("mz_conv3x3_bf16", (None, None, 0) + (None,) * 7 + (1, 12, 12, 256, 256, 1, 1, None), "null tensor"),
        ("mz_conv_chain_bf16", (C.c_void_p(8), None, 0, None, 5, C.c_void_p(8), None, None, None, 4, None), "null"),
        ("mz_conv_chain_mid_bf16", (C.c_void_p(8), None, 4, C.c_void_p(8), 4, None), "null"),
        ("mz_conv_chain_mid_bf16", (C.c_void_p(8), C.c_void_p(8), 9, C.c_void_p(8), 4, None), "n_layers"),
        ("mz_heads", (None, 4, 256, 256, 6, 300) + (None,) * 10, "null argument"),
        ("mz_ring_push", (C.c_void_p(8), C.c_void_p(8), 4, 32, 27647, 0, 0, None, None), "multiple of 16"),
        ("mz_ring_to_obs", (C.c_void_p(8), C.c_void_p(8), 1, 4, 32, 9216, 32, None), "outside the ring"),
        ("mz_prepare_observation", (None, None, 1, 96, 9216, None), "bad arguments"),
        ("mz_support_to_scalar", (None, 4, 300, None, None), "bad arguments"),
    ]
    for name, args, msg in cases:
        with pytest.raises(_lib.MuZeroAmdError, match=msg):
            _lib.call(name, *args)
    chunk, whole = C.c_int32(), C.c_int32()
    with pytest.raises(_lib.MuZeroAmdError, match="no MFMA kernel"):
        _lib.call("mz_conv3x3_bf16_info", 10, 10, 64, 64, 1, C.byref(chunk), C.byref(whole))
    _lib.call("mz_conv3x3_bf16_info", 12, 12, 256, 256, 1, C.byref(chunk), C.byref(whole))
    assert (chunk.value, whole.value) == (256, 1)
