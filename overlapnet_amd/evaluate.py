"""Error statistics of overlap / yaw predictions against ground truth.

Restates the metric part of the reference's evaluation script (src/two_heads/testing.py:276-318): mean / RMS / max
absolute overlap error, and the circular yaw error (in 1-degree bins) over the pairs whose ground-truth overlap
exceeds a threshold (0.7).  Ground-truth npz layout: `overlaps` (n,4) = [idx1, idx2, overlap, yaw_bin]
(demo/demo4_gen_gt_files.py:97-109).
"""
from __future__ import annotations

from typing import Dict

import numpy as np


def load_ground_truth(npz_path: str):
    """-> (idx1 (n,), idx2 (n,), overlap (n,), yaw_bin (n,)) from a reference ground-truth file."""
    with np.load(npz_path, allow_pickle=True) as z:
        o = np.asarray(z["overlaps"], np.float64)
    return o[:, 0].astype(np.int64), o[:, 1].astype(np.int64), o[:, 2], o[:, 3].astype(np.int64)


def yaw_bin_to_degrees(yaw_bin: np.ndarray) -> np.ndarray:
    """network output bin k <-> yaw = 180 - k degrees (infer.py:158; GT bin: com_overlap_yaw.py:54)."""
    return 180 - np.asarray(yaw_bin)


def circular_error_deg(a: np.ndarray, b: np.ndarray, period: int = 360) -> np.ndarray:
    d = np.abs(np.asarray(a, np.int64) - np.asarray(b, np.int64)) % period
    return np.minimum(d, period - d)


def error_statistics(pred_overlap, gt_overlap, pred_yaw_deg, gt_yaw_deg, yaw_overlap_thres: float = 0.7) -> Dict[str, float]:
    po, go = np.asarray(pred_overlap, np.float64), np.asarray(gt_overlap, np.float64)
    diff = np.abs(po - go)
    out = {"n": int(diff.size), "overlap_mae": float(diff.mean()) if diff.size else float("nan"),
           "overlap_rms": float(np.sqrt((diff ** 2).mean())) if diff.size else float("nan"),
           "overlap_max": float(diff.max()) if diff.size else float("nan")}
    sel = go > yaw_overlap_thres
    if np.any(sel):
        ye = circular_error_deg(np.asarray(pred_yaw_deg)[sel], np.asarray(gt_yaw_deg)[sel])
        out.update({"yaw_n": int(sel.sum()), "yaw_mean_err_deg": float(ye.mean()),
                    "yaw_rms_err_deg": float(np.sqrt((ye.astype(np.float64) ** 2).mean())), "yaw_max_err_deg": float(ye.max())})
    else:
        out.update({"yaw_n": 0})
    return out
