#!/usr/bin/env python3
"""Generate tests/golden/nn_oracle.npz -- fp64 ORACLE outputs on the two reference scans.

PARITY UNPINNED: these vectors come from this repo's CPU restatement (oracle/overlapnet_oracle.py),
not from the reference's Keras/TensorFlow graph, which cannot run here (no tensorflow/keras/h5py, no
pretrained weights in the tree).  They pin the oracle against regressions and travel to the GPU box.

    python tests/golden/make_nn_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import overlapnet_oracle as O  # noqa: E402
from tools import synthetic as S  # noqa: E402

fx = S.load_fixture_images()
out = {}
pairs = np.array([[0, 1], [1, 0], [0, 0]])
for C in (1, 4, 5):
    flags = S.flags_of(C)
    imgs = np.stack([S.stack(fx["range_%d" % i], fx["normal_%d" % i], fx["intensity_%d" % i], flags) for i in range(2)])
    w = S.make_test_weights(C, seed=0)
    ov, yaw, lg, corr, fv = O.infer_pairs(imgs, pairs, w, S.REFERENCE_MODEL_CFG, np.float64)
    out["fv_c%d" % C] = fv.reshape(2, 360, 128).astype(np.float32)
    out["overlap_c%d" % C] = ov
    out["logit_c%d" % C] = lg
    out["yaw_c%d" % C] = yaw
    out["corr_c%d" % C] = corr
    print("C=%d logits %s overlaps %s yaws %s" % (C, lg, ov, yaw))
out["pairs"] = pairs
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nn_oracle.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, os.path.getsize(dst))
