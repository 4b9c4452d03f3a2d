#!/usr/bin/env python3
"""Single-item latency of the f64 building blocks (run under rocprofv3 --kernel-trace; k_debug_math / k_debug_wave7
durations for n = 1 are one lane's / one wave's dependent-chain latency)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sfmlocalization_amd as S  # noqa: E402

rng = np.random.Generator(np.random.PCG64(5))
x1 = rng.uniform(-0.5, 0.5, (1, 7, 2))
x2 = x1 + rng.normal(0, 0.05, (1, 7, 2))
seven = np.concatenate([x1.reshape(1, 14), x2.reshape(1, 14)], 1)
x = rng.uniform(-0.4, 0.4, (1, 3, 2))
X = rng.uniform(-3, 3, (1, 3, 3)) + np.array([0, 0, 9.0])
p3p = np.concatenate([x.reshape(1, 6), X.reshape(1, 9)], 1)
for rep in range(3):
    S.debug_math(0, np.array([[3.7]]), 1)          # log10
    S.debug_math(1, np.array([[3.7, 1.3]]), 2)     # sqrt + div
    S.debug_math(2, rng.normal(size=(1, 4)), 4)    # cubic
    S.debug_math(3, rng.normal(size=(1, 5)), 4)    # quartic
    S.debug_math(4, seven, 28)                     # seven-point, one lane
    S.debug_math(5, p3p, 49)                       # P3P
    S.debug_math(8, seven, 28)                     # seven-point, one wave
print("ok")
