"""Locates and loads libsfmloc_hip.so.  Fails loudly: there is no fallback implementation."""
import ctypes
import os

# One hardware queue per in-flight query context: the HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES
# hardware queues (4 by default), and two contexts that share one serialise their kernels behind each other's
# 2.5 ms Hamming scan (measured: 280 -> 381 queries/s at 3 in flight, profiles/r01_inflight_hwq_sweep.txt).
# The runtime reads the variable when it initialises, so this only takes effect when no HIP call came first;
# an explicit setting in the environment wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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
