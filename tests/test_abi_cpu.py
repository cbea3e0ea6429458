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
