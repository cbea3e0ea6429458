"""muzero_amd - MI355X-native MuZero self-play search path.

Host side (Python) of the C-ABI library in ``muzero_amd/csrc``; mirrors the call
surface of the reference's ``model.py`` for the search path.
"""
from ._lib import MuZeroAmdError  # noqa: F401
from .config import MuZeroConfig  # noqa: F401

__all__ = ["MuZeroAmdError", "MuZeroConfig"]
