"""Loop-closure candidate gating and decision -- the caller side of the 1-vs-N path.

Restates what the reference's demo3 does around `Infer.infer_multiple` (demo/demo3_lcd.py:85-123): candidates are
frames older than `inactive_time_thres`, travelled further than `inactive_dist_thres` ago, and inside the
n-sigma covariance ellipse around the current pose; the loop closure is the candidate with the largest overlap
if that exceeds `overlap_thres`.  Animation / plotting are out of scope.

Pinned on the reference's own code: the test suite holds, frame by frame, the ellipse, the candidate list and the reported loop
closure produced by `AnimatedLCD.get_cov_ellipse` / `get_predictions` (imported unmodified by the golden-file generator
make_lcd_golden.py) for three synthetic trajectories, and requires equality (test_host_logic.py).
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np


def travelled_distances(xy: np.ndarray) -> np.ndarray:
    """Cumulative path length per frame (demo3_lcd.py keeps it in `self.traj_length`)."""
    xy = np.asarray(xy, np.float64)
    if len(xy) == 0:
        return np.zeros(0)
    seg = np.linalg.norm(np.diff(xy, axis=0), axis=1)
    return np.concatenate([[0.0], np.cumsum(seg)])


def covariance_ellipse(cov: np.ndarray, nstd: float = 3.0) -> Tuple[float, float, float]:
    """(width, height, angle_deg) of the nstd-sigma ellipse of a 2x2 covariance, matplotlib convention
    (demo3_lcd.py:125-140: eigen-decomposition, width/height = 2*nstd*sqrt(eigenvalues), angle of the major axis)."""
    vals, vecs = np.linalg.eigh(np.asarray(cov, np.float64)[:2, :2])
    order = vals.argsort()[::-1]
    vals, vecs = vals[order], vecs[:, order]
    theta = np.degrees(np.arctan2(*vecs[:, 0][::-1]))
    width, height = 2 * nstd * np.sqrt(np.maximum(vals, 0))
    return float(width), float(height), float(theta)


def gate_candidates(idx: int, traj_xy: np.ndarray, traj_length: np.ndarray, ellipse: Tuple[float, float, float],
                    inactive_time_thres: int = 100, inactive_dist_thres: float = 50.0) -> np.ndarray:
    """Reference frame ids to compare frame `idx` against (demo3_lcd.py:92-115)."""
    if idx < inactive_time_thres:
        return np.zeros(0, dtype=np.int64)
    indices = np.arange(idx - inactive_time_thres)
    dist_delta = traj_length[idx] - np.asarray(traj_length)[indices]
    indices = indices[dist_delta > inactive_dist_thres]
    if len(indices) == 0:
        return indices
    width, height, angle = ellipse
    cos_a = np.cos(np.radians(180.0 - angle))
    sin_a = np.sin(np.radians(180.0 - angle))
    xc = traj_xy[idx, 0] - traj_xy[indices, 0]
    yc = traj_xy[idx, 1] - traj_xy[indices, 1]
    xct = xc * cos_a - yc * sin_a
    yct = xc * sin_a + yc * cos_a
    rad_cc = (xct ** 2 / (width / 2.0) ** 2) + (yct ** 2 / (height / 2.0) ** 2)
    return indices[rad_cc < 1]


def decide(reference_idx: Sequence[int], overlaps, yaws, overlap_thres: float = 0.3) -> Optional[Tuple[int, float, int]]:
    """(frame id, overlap, yaw) of the loop closure, or None (demo3_lcd.py:118-121)."""
    overlaps = np.atleast_1d(np.asarray(overlaps))
    yaws = np.atleast_1d(np.asarray(yaws))
    if len(reference_idx) == 0 or overlaps.size == 0:
        return None
    k = int(np.argmax(overlaps))
    if overlaps[k] > overlap_thres:
        return int(np.asarray(reference_idx)[k]), float(overlaps[k]), int(yaws[k])
    return None


def detect(infer, idx: int, traj_xy: np.ndarray, traj_length: np.ndarray, ellipse, **gate_kw):
    """One streaming step: always feeds frame `idx` to `infer.infer_multiple` (it caches the frame's feature
    volume, demo3_lcd.py:88-90,122) and returns the loop-closure decision."""
    overlap_thres = gate_kw.pop("overlap_thres", 0.3)
    ref = gate_candidates(idx, traj_xy, traj_length, ellipse, **gate_kw)
    if hasattr(infer, "infer_best_match"):  # decision taken on the GPU, one record returned
        return infer.infer_best_match(idx, list(ref), overlap_thres)
    res = infer.infer_multiple(idx, list(ref))
    if res is None:
        return None
    return decide(ref, res[0], res[1], overlap_thres)
