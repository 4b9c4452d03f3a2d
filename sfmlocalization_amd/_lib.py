"""Locates and loads libsfmloc_hip.so.  Fails loudly: there is no fallback implementation."""
import ctypes
import importlib.util
import os
import sys

# One hardware queue per in-flight query context: the HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES
# hardware queues (4 by default), and two contexts that share one serialise their kernels behind each other's
# 2.5 ms Hamming scan (measured: 280 -> 381 queries/s at 3 in flight, profiles/r01_inflight_hwq_sweep.txt).
# The runtime reads the variable when it initialises, so this only takes effect when no HIP call came first;
# an explicit setting in the environment wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
# SFMLOC_LIB_PATH: an instrumented build of the same library (tools/ only; e.g. the time-stamp build of K3 / K5)
LIB_PATH = os.environ.get("SFMLOC_LIB_PATH") or os.path.join(_HERE, "lib", "libsfmloc_hip.so")
_handle = None


class LibraryMissing(ImportError):
    pass


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's).  Two HIP runtimes in one
    process do not work: whichever initialises second finds no device.  torch.distributed is this package's
    multi-GPU plumbing, so when torch is installed its copy is loaded first (by path, without importing torch) and
    libsfmloc_hip.so binds to it through the SONAME; SFMLOC_HIP_RUNTIME=system keeps the system runtime."""
    if os.environ.get("SFMLOC_HIP_RUNTIME", "") == "system" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    global _handle
    if _handle is None:
        if not os.path.exists(LIB_PATH):
            raise LibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C sfmlocalization_amd/csrc`). The HIP library is the only compute path.")
        _share_torch_hip_runtime()
        _handle = ctypes.CDLL(LIB_PATH)
    return _handle
