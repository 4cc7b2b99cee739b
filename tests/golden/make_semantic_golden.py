#!/usr/bin/env python3
"""tests/golden/semantic_idx.npz: `proj_idx` of the reference's range_projection(points, max_range=inf) for its two
scans -- the correspondences gen_semantic_data.py:39 uses -- plus a run of the reference's gen_semantic_data itself on
seeded per-point probabilities, reduced to a checksum per scan (the (64,900,20) images are 4.6 MB each).
Run in the build container only:  python tests/golden/make_semantic_golden.py"""
import os
import shutil
import sys
import tempfile

import numpy as np

REF = os.environ.get("OVERLAPNET_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REF, "src", "utils"))
import utils as ref_utils  # noqa: E402
import gen_semantic_data as ref_sem  # noqa: E402

out = {}
tmp = tempfile.mkdtemp()
try:
    os.makedirs(os.path.join(tmp, "scans"))
    os.makedirs(os.path.join(tmp, "probs"))
    os.makedirs(os.path.join(tmp, "dst"))
    for i in range(2):
        name = "%06d" % i
        src = os.path.join(REF, "data", "scans", name + ".bin")
        shutil.copy(src, os.path.join(tmp, "scans", name + ".bin"))
        pts = np.fromfile(src, dtype=np.float32).reshape(-1, 4)
        probs = np.random.default_rng(100 + i).random((pts.shape[0], 20)).astype(np.float32)
        probs.tofile(os.path.join(tmp, "probs", name + ".label"))
        out["idx_%d" % i] = ref_utils.range_projection(pts, max_range=np.inf)[3]
    sem = ref_sem.gen_semantic_data(os.path.join(tmp, "probs"), os.path.join(tmp, "scans"), os.path.join(tmp, "dst"))
    for i in range(2):
        assert np.array_equal(np.load(os.path.join(tmp, "dst", "semantic", "%06d.npy" % i)), sem[i])
        out["sum_%d" % i] = np.float64(sem[i].astype(np.float64).sum())
        out["wsum_%d" % i] = np.float64((sem[i].astype(np.float64) * np.arange(sem[i].size).reshape(sem[i].shape) % 1000).sum())
finally:
    shutil.rmtree(tmp)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "semantic_idx.npz"), **out)
print({k: (v.shape if hasattr(v, "shape") and v.ndim else float(v)) for k, v in out.items()})
