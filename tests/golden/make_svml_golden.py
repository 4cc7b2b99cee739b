#!/usr/bin/env python3
"""tests/golden/svml_f32_vectors.npz: outputs of NumPy's OWN float32 arctan2 / arcsin (what the reference calls at
src/utils/utils.py:86-87) on this machine -- AVX512_SKX, i.e. NumPy's SVML loops -- for seeded inputs that cover the LiDAR range,
the wide exponent range, the quadrant / octant boundaries, both asin branches and the special values.  The restatement
(oracle/svml_f32.c) and the HIP kernel (csrc/svml_f32.h) are compared with these vectors on any host.

    python tests/golden/make_svml_golden.py      (build container; refuses to run on a CPU without AVX512_SKX)
"""
import os

import numpy as np
from numpy._core._multiarray_umath import __cpu_features__ as feats

assert feats.get("AVX512_SKX"), "generate the vectors on an AVX512_SKX machine (NumPy's SVML dispatch)"
rng = np.random.default_rng(20260924)
n = 30000
parts_x, parts_y = [], []
parts_x.append(rng.normal(0, 20, n)); parts_y.append(rng.normal(0, 20, n))                                  # LiDAR-like
parts_x.append(rng.uniform(-1, 1, n) * np.exp(rng.uniform(-60, 60, n)))
parts_y.append(rng.uniform(-1, 1, n) * np.exp(rng.uniform(-60, 60, n)))                                      # wide exponents
a = rng.uniform(-80, 80, n)
parts_x.append(a); parts_y.append(a * (1 + rng.integers(-3, 4, n) * 2.0 ** -23) * rng.choice([-1, 1], n))    # |y| ~ |x|
sp = np.array([0, -0.0, 1, -1, 2.5, -2.5, 1e-30, -1e-30, 1e30, 1e-39, 3e38, 2.0 ** -125, 2.0 ** 123, 2.0 ** -126, np.inf, -np.inf,
               np.nan], np.float32)
yy, xx = np.meshgrid(sp, sp)
parts_x.append(xx.ravel()); parts_y.append(yy.ravel())
x = np.concatenate(parts_x).astype(np.float32)
y = np.concatenate(parts_y).astype(np.float32)
s = np.concatenate([rng.uniform(-1, 1, n), rng.uniform(-0.5, 0.5, n), np.sign(rng.uniform(-1, 1, n)) * (1 - np.exp(rng.uniform(-25, 0, n))),
                    np.array([0, -0.0, 1, -1, 0.5, -0.5, 1e-20, 1 - 2.0 ** -24, 0.49999997, 1.5, -2, np.nan])]).astype(np.float32)
with np.errstate(all="ignore"):
    out = dict(x=x, y=y, atan2=np.arctan2(y, x), s=s, asin=np.arcsin(s), numpy_version=np.__version__)
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "svml_f32_vectors.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, os.path.getsize(dst), "bytes;", len(x), "atan2 and", len(s), "asin vectors")
