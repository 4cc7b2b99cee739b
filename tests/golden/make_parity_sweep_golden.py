#!/usr/bin/env python3
"""fp64 ORACLE outputs of the benchmark's 1-vs-1024 sweep (BASELINE.json configs[1]) -> tests/golden/parity_sweep_<set>.npz.

The fp64 oracle needs ~0.15 s per head pair on 8 cores, so the 2 x 1024 pairs are evaluated once in the build
container and committed (a few hundred KB); `tests/test_parity_sweep.py` rebuilds the SAME inputs from the same seeds
on the GPU box, runs the HIP path on them and compares every pair.  The test also re-runs the oracle live on a few
pairs and asserts they equal this file (guards against a drift of the input recipe).

    python tests/golden/make_parity_sweep_golden.py [--pool 1024] [--channels 4] [--sets glorot trained_like]
    (committed: pool 1024 at C = 4, and pool 128 at C = 1 and C = 5 for the trained-like set: *_c1_p128.npz, *_c5_p128.npz)

Stored per weight set: overlap, logit (fp64), yaw, the top-2 gap of the correlation vector (relative, for the near-tie
rule of SURVEY.md section 8c), max |corr| per pair, and an fp64 checksum of the oracle's query feature volume.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import overlapnet_oracle as O  # noqa: E402
from tools import synthetic as S  # noqa: E402


def oracle_sweep(pool, channels, weights, log=print):
    cfg = S.REFERENCE_MODEL_CFG
    fx = S.load_fixture_images()
    qfv = O.leg_forward(S.sweep_query_image(channels, fx), weights, cfg, np.float64)          # (1,1,360,128)
    ov = np.zeros(pool)
    lg = np.zeros(pool)
    yaw = np.zeros(pool, np.int64)
    gap = np.zeros(pool)
    cmax = np.zeros(pool)
    fsum = np.zeros(pool)
    t0 = time.time()
    for s, imgs in S.sweep_pool_images(pool, channels, 0, fx):
        fv = O.leg_forward(imgs, weights, cfg, np.float64)
        n = fv.shape[0]
        for b in range(0, n, 16):
            m = min(16, n - b)
            o, y, g, c = O.heads_forward(fv[b:b + m], np.repeat(qfv, m, axis=0), weights)
            sl = slice(s + b, s + b + m)
            ov[sl], yaw[sl], lg[sl] = o, y, g
            srt = np.sort(c, axis=1)
            with np.errstate(all="ignore"):
                gap[sl] = np.nan_to_num((srt[:, -1] - srt[:, -2]) / np.abs(srt[:, -1]))
            cmax[sl] = np.max(np.abs(c), axis=1)
            fsum[sl] = fv[b:b + m].reshape(m, -1).sum(axis=1)
        log("  %d / %d pairs, %.0f s" % (s + n, pool, time.time() - t0))
    return {"overlap": ov, "logit": lg, "yaw": yaw, "corr_top2_gap": gap, "corr_absmax": cmax, "feat_sum": fsum,
            "query_feat_sum": np.array([qfv.sum()]), "query_feat_max": np.array([qfv.max()])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pool", type=int, default=1024)
    ap.add_argument("--channels", type=int, default=4)
    ap.add_argument("--sets", nargs="+", default=list(S.WEIGHT_SETS))
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 1)
    for name in args.sets:
        w = S.WEIGHT_SETS[name](args.channels)
        print("weight set %s" % name, flush=True)
        out = oracle_sweep(args.pool, args.channels, w, log=lambda m: print(m, flush=True))
        suffix = "" if (args.pool, args.channels) == (1024, 4) else "_c%d_p%d" % (args.channels, args.pool)
        path = os.path.join(ROOT, "tests", "golden", "parity_sweep_%s%s.npz" % (name, suffix))
        np.savez_compressed(path, pool=np.array([args.pool]), channels=np.array([args.channels]), **out)
        print("wrote %s: logits [%.2f, %.2f]" % (path, out["logit"].min(), out["logit"].max()), flush=True)


if __name__ == "__main__":
    main()
