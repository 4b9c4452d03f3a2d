"""MI355X-native query localisation for hulop/SfMLocalization maps.

The compute path is libsfmloc_hip.so (hand-written HIP for gfx950) behind the C ABI of
include/sfmloc.h; this package is the thin ctypes shim plus the host-side mirror of the
reference's interface for that path.  There is no CPU fallback.
"""
from . import _lib  # noqa: F401
from .capi import (  # noqa: F401
    SfmlocError, Params, MapDesc, Map, Query, Context, BofModel, ImgBow, Akaze, KernelStats, Pose, default_params, device_count, NOMATCH, debug_math, dense_gray,
    Undistorter, image_read, image_decode, image_size,
)

__all__ = ["SfmlocError", "Params", "MapDesc", "Map", "Query", "Context", "BofModel", "ImgBow", "Akaze", "KernelStats", "default_params",
           "device_count", "NOMATCH", "Pose", "debug_math", "dense_gray", "Undistorter", "image_read", "image_decode",
           "image_size"]
