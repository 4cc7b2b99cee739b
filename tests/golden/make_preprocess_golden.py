#!/usr/bin/env python3
"""Generate tests/golden/kitti_preprocess.npz by RUNNING the reference's own preprocessing code.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_preprocess_golden.py

What it stores (all produced by the reference's `src/utils/utils.py`, imported unmodified):
  points_{0,1}     raw KITTI scans  data/scans/00000{0,1}.bin          (N,4) f32
  range_{0,1}      range_projection(...)[0]   (utils.py:59-134)        (64,900) f32
  intensity_{0,1}  range_projection(...)[2]                             (64,900) f32
  idx_{0,1}        range_projection(...)[3]                             (64,900) i32
  normal_{0,1}     gen_normal_map(range, vertex) (utils.py:137-175)     (64,900,3) f32
(the vertex map is recoverable as points[kept][idx] and is not stored.)

It also asserts that these equal the `.npy` files the reference ships under
data/preprocess_data_demo/{depth,intensity,normal}/, i.e. the golden file is pinned twice:
by the reference code run here and by the reference's own shipped outputs.
"""
import os
import sys

import numpy as np

REF = os.environ.get("OVERLAPNET_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REF, "src", "utils"))
import utils as ref_utils  # noqa: E402  (the reference module, not ours)

out = {}
for i, name in enumerate(["000000", "000001"]):
    pts = np.fromfile(os.path.join(REF, "data", "scans", name + ".bin"), dtype=np.float32).reshape(-1, 4)
    rng, vtx, inten, idx = ref_utils.range_projection(pts)
    nrm = ref_utils.gen_normal_map(rng, vtx)
    demo = os.path.join(REF, "data", "preprocess_data_demo")
    assert np.array_equal(rng, np.load(os.path.join(demo, "depth", name + ".npy")))
    assert np.array_equal(inten, np.load(os.path.join(demo, "intensity", name + ".npy")))
    assert np.array_equal(nrm, np.load(os.path.join(demo, "normal", name + ".npy")))
    out["points_%d" % i] = pts
    out["range_%d" % i] = rng
    out["intensity_%d" % i] = inten
    out["idx_%d" % i] = idx
    out["normal_%d" % i] = nrm
    print(name, pts.shape, "valid range px %.4f" % np.mean(rng > 0), "valid normal px %.4f" % np.mean(nrm[..., 0] != -1))

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kitti_preprocess.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, os.path.getsize(dst), "bytes")
