"""CPU: the C-ABI library loads, exports every symbol include/sfmloc.h declares, its structs have the
layout the ctypes mirror assumes, and compute entry points fail loudly without a GPU."""
import ctypes
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

import sfmlocalization_amd as S
from sfmlocalization_amd import capi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sfmloc.h")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sfmloc_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    names = declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/sfmloc.h but not exported"
    assert sorted(capi.SYMBOLS) == names, "capi.SYMBOLS out of date with include/sfmloc.h"
    assert capi._L().sfmloc_abi_version() == 2      # 2: sfmloc_params.guided_matching


def test_struct_layout_matches_header():
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "sfmloc.h"
int main(void){
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(sfmloc_params), sizeof(sfmloc_map_desc), sizeof(sfmloc_map_info), sizeof(sfmloc_kernel_stats), sizeof(sfmloc_pose), offsetof(sfmloc_pose, center));
  printf("%zu %zu %zu %zu\n", offsetof(sfmloc_params, geom_precision), offsetof(sfmloc_params, seed), offsetof(sfmloc_params, profile), offsetof(sfmloc_map_desc, bow));
  return 0; }
'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    sizes = [int(x) for x in out]
    assert sizes[0] == ctypes.sizeof(capi.Params)
    assert sizes[1] == ctypes.sizeof(capi.MapDesc)
    assert sizes[2] == ctypes.sizeof(capi.MapInfo)
    assert sizes[3] == ctypes.sizeof(capi.KernelStats)
    assert sizes[4] == ctypes.sizeof(capi.Pose)
    assert sizes[5] == capi.Pose.center.offset
    assert sizes[6] == capi.Params.geom_precision.offset
    assert sizes[7] == capi.Params.seed.offset
    assert sizes[8] == capi.Params.profile.offset
    assert sizes[9] == capi.MapDesc.bow.offset


def test_default_params_are_the_reference_defaults():
    p = S.default_params()
    assert abs(p.dist_ratio - 0.6) < 1e-7      # localization.cpp:70
    assert p.ransac_round == 200               # localization.cpp:71
    assert p.geom_precision == 4.0             # localization.cpp:81
    assert (p.min_putative, p.min_resection_points, p.min_inliers) == (16, 8, 10)  # localization.cpp:56-58
    assert p.p3p_max_iteration == 4096
    assert p.refine_pose == 0


def test_no_cpu_fallback():
    if S.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(S.SfmlocError) as ei:
        S.Map([0], [0, 1], np.zeros((1, 64), np.uint8))
    assert ei.value.code == capi.ENODEV
    assert "no CPU fallback" in str(ei.value)
