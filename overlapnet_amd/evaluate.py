"""Error statistics of overlap / yaw predictions against ground truth.

Restates the metric part of the reference's evaluation script (src/two_heads/testing.py:276-318): mean / RMS / max
absolute overlap error, and the circular yaw error (in 1-degree bins) over the pairs whose ground-truth overlap
exceeds a threshold (0.7).  Ground-truth npz layout: `overlaps` (n,4) = [idx1, idx2, overlap, yaw_bin]
(demo/demo4_gen_gt_files.py:97-109).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

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


def load_pairs(npz_files: Sequence[str], shuffle: bool = False, rng: Optional[np.random.Generator] = None
               ) -> Tuple[List[str], List[str], List[str], List[str], np.ndarray, np.ndarray]:
    """The reference's test/training pair reader (src/two_heads/overlap_orientation_npz_file2string_string_nparray.py:8-76):
    -> (imgf1, imgf2, dir1, dir2, overlap, orientation_bin).  Both layouts: the current one (`overlaps` (n,4) +
    `seq` (n,2) sequence names, demo4_gen_gt_files.py:97-109) and the old single-array one (directories = '')."""
    f1: List[str] = []
    f2: List[str] = []
    d1: List[str] = []
    d2: List[str] = []
    ov: List[float] = []
    ori: List[float] = []
    for name in npz_files:
        with np.load(name, allow_pickle=True) as h:
            if len(h.files) == 1:
                arr = np.asarray(h[h.files[0]])
                seq = np.full((arr.shape[0], 2), "", dtype=object)
            else:
                arr = np.asarray(h["overlaps"])
                seq = np.asarray(h["seq"])
        a1 = ["%06d" % int(v) for v in arr[:, 0]]
        a2 = ["%06d" % int(v) for v in arr[:, 1]]
        s1, s2 = [str(v) for v in seq[:, 0]], [str(v) for v in seq[:, 1]]
        o, r = np.asarray(arr[:, 2], np.float64), np.asarray(arr[:, 3], np.float64)
        if shuffle:
            perm = (rng or np.random.default_rng()).permutation(len(a1))
            a1, a2 = [a1[i] for i in perm], [a2[i] for i in perm]
            s1, s2 = [s1[i] for i in perm], [s2[i] for i in perm]
            o, r = o[perm], r[perm]
        f1 += a1
        f2 += a2
        d1 += s1
        d2 += s2
        ov += list(o)
        ori += list(r)
    return f1, f2, d1, d2, np.asarray(ov), np.asarray(ori)


def run_test(infer, npz_files: Sequence[str], no_test_pairs: Optional[int] = None, out_dir: Optional[str] = None
             ) -> Dict[str, float]:
    """The reference's evaluation run (src/two_heads/testing.py:207-352) on the HIP path: feature volumes of every scan
    that occurs in the test pairs once, heads over all pairs (imgf1 -> head-left, imgf2 -> head-right, testing.py:236-243),
    overlap / circular-yaw statistics, and `validation_results.npz` = (n,4) [img1, img2, overlap, argmax bin]
    (testing.py:337-345).  `infer` is an `overlapnet_amd.infer.Infer` (same sequence for all pairs, as the reference assumes)."""
    f1, f2, _d1, _d2, gt_ov, gt_bin = load_pairs(npz_files, shuffle=False)
    n = len(f1) if no_test_pairs is None else min(int(no_test_pairs), len(f1))
    f1, f2, gt_ov, gt_bin = f1[:n], f2[:n], gt_ov[:n], gt_bin[:n]
    names = sorted(set(f1) | set(f2))
    pos = {name: i for i, name in enumerate(names)}
    idx1 = [pos[v] for v in f1]
    idx2 = [pos[v] for v in f2]
    if n == 0:
        raise Exception("no test pairs")
    # infer_multiple_vs_multiple: head-left = second_idxs, head-right = first_idxs (infer.py:224-225)
    ov, yaw = infer.infer_multiple_vs_multiple(names, idx2, idx1)
    ov = np.atleast_1d(np.asarray(ov, np.float64))
    pred_bin = 180 - np.atleast_1d(np.asarray(yaw, np.int64))       # infer.py:158 inverted: the argmax bin itself
    stats = error_statistics(ov, gt_ov, pred_bin, gt_bin.astype(np.int64))
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        m = np.zeros((n, 4))
        m[:, 0] = np.asarray(f1, dtype=float)
        m[:, 1] = np.asarray(f2, dtype=float)
        m[:, 2] = ov
        m[:, 3] = pred_bin
        np.savez(os.path.join(out_dir, "validation_results.npz"), m)
    return stats
