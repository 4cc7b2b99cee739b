#!/usr/bin/env python3
"""fp64 ORACLE outputs of bench.py's `fullstack` step on its first 64 pairs -> tests/golden/parity_fullstack.npz.

The step starts from RAW clouds (tools/synthetic.fullstack_cloud: the shipped scans rotated about z): projection + normals +
channel stacking + leg + both heads.  The oracle starts from the same clouds with its own restatement of every stage
(range_projection with the restated NumPy float32 angles, gen_normal_map, fp64 leg and heads).  Query = cloud 1024 (the
candidate pool of the benchmark is clouds 0..1023).  bench.py compares its fullstack results with this file in the default run;
tests/test_parity_sweep.py re-runs the oracle on a sample and requires equality with the file.

    python tests/golden/make_fullstack_golden.py [--pairs 64]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import overlapnet_oracle as O  # noqa: E402
from tools import synthetic as S  # noqa: E402

QUERY_CLOUD = 1024


def oracle_images(fx, ids, channels=4):
    flags = S.flags_of(channels)
    rows = []
    for i in ids:
        rng, vtx, itn, _ = O.range_projection(S.fullstack_cloud(fx, i))
        rows.append(S.stack(rng, O.gen_normal_map(rng, vtx), itn, flags))
    return np.stack(rows)


def oracle_fullstack(ids, channels=4, weights=None):
    fx = S.load_fixture_images()
    w = weights or S.make_test_weights(channels, seed=0)
    qfv = O.leg_forward(oracle_images(fx, [QUERY_CLOUD], channels), w, S.REFERENCE_MODEL_CFG, np.float64)
    ov, lg, yaw = [], [], []
    for b in range(0, len(ids), 16):
        fv = O.leg_forward(oracle_images(fx, ids[b:b + 16], channels), w, S.REFERENCE_MODEL_CFG, np.float64)
        o, y, g, _ = O.heads_forward(fv, np.repeat(qfv, fv.shape[0], axis=0), w)
        ov.append(o), yaw.append(y), lg.append(g)
    return np.concatenate(ov), np.concatenate(yaw), np.concatenate(lg)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=64)
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 1)
    # the benchmark's weights (file name kept: bench.py reads it) and the trained-like dynamic range (tests/test_parity_sweep.py)
    for name, fname in (("glorot", "parity_fullstack.npz"), ("trained_like", "parity_fullstack_trained_like.npz")):
        ov, yaw, lg = oracle_fullstack(list(range(args.pairs)), weights=S.WEIGHT_SETS[name](4))
        path = os.path.join(ROOT, "tests", "golden", fname)
        np.savez_compressed(path, overlap=ov, yaw=yaw, logit=lg, query_cloud=np.array([QUERY_CLOUD]), channels=np.array([4]))
        print("wrote %s: %d pairs, logits [%.2f, %.2f], yaw %s ..." % (path, len(ov), lg.min(), lg.max(), yaw[:8]))


if __name__ == "__main__":
    main()
