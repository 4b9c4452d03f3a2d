"""Locates and loads libsfmloc_hip.so.  Fails loudly: there is no fallback implementation."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsfmloc_hip.so")
_handle = None


class LibraryMissing(ImportError):
    pass


def load():
    global _handle
    if _handle is None:
        if not os.path.exists(LIB_PATH):
            raise LibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C sfmlocalization_amd/csrc`). The HIP library is the only compute path.")
        _handle = ctypes.CDLL(LIB_PATH)
    return _handle
