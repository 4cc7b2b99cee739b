#!/usr/bin/env python3
"""fp64 ORACLE outputs for the tolerance BUDGET MAP (tests/test_parity_budget.py) -> tests/golden/parity_budget.npz.

Where does the 1e-4 overlap tolerance of the north star run out?  Cases (64 pairs each unless stated, query = the benchmark's):
  gain_0.5 / gain_2 / gain_4   trained-like weights (C = 4) with the LAST leg layer's kernel and bias times g: the feature volumes
                               scale exactly by g (ReLU is positively homogeneous), the Delta head's first (linear) layer with
                               them, logits by about g -- the dynamic range a differently trained `model_geo` could have
  depth50                      depth-only (C = 1) images with 90 % valid pixels uniform in (0, 50) m (max range), rest -1
  invalid                      scans that are -1 everywhere (an empty sweep): pairs (invalid, invalid), (invalid candidates vs the
                               benchmark query) and (benchmark candidates vs an invalid query), 8 pairs
The inputs are rebuilt from seeds by `budget_case`, which the GPU test imports from here.

    python tests/golden/make_parity_budget_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools import synthetic as S  # noqa: E402

CASE_NAMES = ["gain_0.5", "gain_2", "gain_4", "depth50", "invalid"]
LAST_LEG_LAYER = "s_conv10"


def scaled_last_layer(w, g):
    w = dict(w)
    for k in (LAST_LEG_LAYER + "/kernel", LAST_LEG_LAYER + "/bias"):
        assert k in w, sorted(w)
        w[k] = (w[k].astype(np.float64) * g).astype(np.float32)
    return w


def budget_case(name):
    """-> (channels, weights, candidate images (n, 64, 900, C), query images (1 or n, 64, 900, C))."""
    fx = S.load_fixture_images()
    if name.startswith("gain_"):
        g = float(name.split("_")[1])
        w = scaled_last_layer(S.make_trained_like_weights(4), g)
        _, imgs = next(S.sweep_pool_images(64, 4, 0, fx, chunk=64))
        return 4, w, imgs, S.sweep_query_image(4, fx)
    if name == "depth50":
        rng = np.random.default_rng(50)
        imgs = rng.uniform(0.05, 50.0, size=(65, 64, 900, 1)).astype(np.float32)
        imgs[rng.random(imgs.shape) < 0.10] = -1.0
        return 1, S.make_trained_like_weights(1), imgs[:64], imgs[64:]
    if name == "invalid":
        inv = np.full((1, 64, 900, 4), -1.0, np.float32)
        _, pool = next(S.sweep_pool_images(3, 4, 0, fx, chunk=3))
        q = S.sweep_query_image(4, fx)
        cands = np.concatenate([inv, inv, inv, pool, inv, pool[:1]], axis=0)            # 8 pairs
        quers = np.concatenate([inv, q, q, inv, inv, inv, inv, q], axis=0)
        return 4, S.make_trained_like_weights(4), cands, quers
    raise KeyError(name)


def main():
    from oracle import overlapnet_oracle as O
    torch.set_num_threads(os.cpu_count() or 1)
    out = {}
    for name in CASE_NAMES:
        C, w, cands, quers = budget_case(name)
        cfg = S.REFERENCE_MODEL_CFG
        fv = O.leg_forward(cands, w, cfg, np.float64)
        qf = O.leg_forward(quers, w, cfg, np.float64)
        n = fv.shape[0]
        if qf.shape[0] == 1:
            qf = np.repeat(qf, n, axis=0)
        ov, yaw, lg, gap = np.zeros(n), np.zeros(n, np.int64), np.zeros(n), np.zeros(n)
        for b in range(0, n, 16):
            o, y, l_, c = O.heads_forward(fv[b:b + 16], qf[b:b + 16], w)
            ov[b:b + 16], yaw[b:b + 16], lg[b:b + 16] = o, y, l_
            srt = np.sort(c, axis=1)
            with np.errstate(all="ignore"):
                gap[b:b + 16] = np.nan_to_num((srt[:, -1] - srt[:, -2]) / np.abs(srt[:, -1]))
        out.update({name + "/overlap": ov, name + "/yaw": yaw, name + "/logit": lg, name + "/corr_top2_gap": gap,
                    name + "/feat_max": np.array([fv.max(), qf.max()])})
        print("%-9s n %d  logits [%.2f, %.2f]  features up to %.1f  near-tie pairs (gap < 1e-5) %d"
              % (name, n, lg.min(), lg.max(), fv.max(), int((gap < 1e-5).sum())), flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "parity_budget.npz"), **out)


if __name__ == "__main__":
    main()
