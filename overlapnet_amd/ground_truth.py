"""Ground-truth overlap / yaw labels on the GPU -- drop-in for the reference's `src/utils/com_overlap_yaw.py`.

`com_overlap_yaw(scan_paths, poses, frame_idx, leg_output_width=360)` has the reference's signature and return value
(rows `[current_frame_idx, reference_frame_idx, overlap, yaw_bin]`, com_overlap_yaw.py:10-68); the N float64 range
projections of the transformed reference clouds (the O(N x 125k points) part) run in `csrc/overlap_gt.hip`, the yaw bin
is scalar host arithmetic restated operator for operator (Python precedence included).  `OverlapGroundTruth` keeps the
scans resident in HBM so that labelling every frame of a sequence (demo4 labels one, training wants all) costs one
upload: 1101 KITTI scans = 2.2 GB.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import numpy as np
import torch

from .engine import OvnEngine


def load_vertex(scan_path: str) -> np.ndarray:
    """(n,4) float64 homogeneous points (x, y, z, 1) of a KITTI .bin scan (utils.py:217-230)."""
    cur = np.fromfile(scan_path, dtype=np.float32).reshape((-1, 4))
    out = np.ones((cur.shape[0], 4))
    out[:, :3] = cur[:, :3]
    return out


def euler_angles_from_rotation_matrix(R):
    """(roll, pitch, yaw) after Slabaugh, utils.py:186-214."""
    def isclose(x, y, rtol=1.e-5, atol=1.e-8):
        return abs(x - y) <= atol + rtol * abs(y)

    phi = 0.0
    if isclose(R[2, 0], -1.0):
        theta = math.pi / 2.0
        psi = math.atan2(R[0, 1], R[0, 2])
    elif isclose(R[2, 0], 1.0):
        theta = -math.pi / 2.0
        psi = math.atan2(-R[0, 1], -R[0, 2])
    else:
        theta = -math.asin(R[2, 0])
        cos_theta = math.cos(theta)
        psi = math.atan2(R[2, 1] / cos_theta, R[2, 2] / cos_theta)
        phi = math.atan2(R[1, 0] / cos_theta, R[0, 0] / cos_theta)
    return psi, theta, phi


def yaw_bin(current_pose: np.ndarray, reference_pose: np.ndarray, yaw_resolution: int = 360) -> int:
    """com_overlap_yaw.py:49-55.  Note `-(yaw / pi) * W // 2 + W // 2`: the floor division applies to the product."""
    relative_transform = np.linalg.inv(current_pose).dot(reference_pose)
    _, _, yaw = euler_angles_from_rotation_matrix(relative_transform[:3, :3])
    return int(-(yaw / np.pi) * yaw_resolution // 2 + yaw_resolution // 2)


class OverlapGroundTruth:
    """Scans resident on the device; `mapping(frame_idx)` = the reference's ground_truth_mapping for that frame."""

    def __init__(self, scans: Sequence[np.ndarray], poses: np.ndarray, engine: Optional[OvnEngine] = None,
                 leg_output_width: int = 360, proj_H: int = 64, proj_W: int = 900, fov_up: float = 3.0,
                 fov_down: float = -25.0, max_range: float = 50.0):
        if len(scans) != len(poses):
            raise Exception("need one pose per scan (%d scans, %d poses)" % (len(scans), len(poses)))
        self.engine = engine or OvnEngine(proj_H, proj_W, 1)
        self.poses = np.asarray(poses, np.float64).reshape(-1, 4, 4)
        self.n = len(scans)
        self.leg_output_width = leg_output_width
        self.proj = dict(proj_h=proj_H, proj_w=proj_W, fov_up=fov_up, fov_down=fov_down, max_range=max_range)
        sizes = [int(np.asarray(s).shape[0]) for s in scans]
        self.max_points = max(sizes) if sizes else 0
        off = np.zeros(self.n + 1, np.int64)
        off[1:] = np.cumsum(sizes)
        pts = np.zeros((int(off[-1]), 4), np.float32)
        for i, s in enumerate(scans):
            pts[off[i]:off[i + 1], :3] = np.asarray(s)[:, :3]      # load_vertex keeps x, y, z only (utils.py:226-229)
        dev = self.engine.device
        self._points = torch.from_numpy(pts).to(dev)
        self._offsets = torch.from_numpy(off).to(dev)
        self._ref_poses = torch.from_numpy(np.ascontiguousarray(self.poses)).to(dev)

    def overlaps(self, frame_idx: int) -> np.ndarray:
        """(n,) float64 overlap of every scan with frame `frame_idx` (com_overlap_yaw.py:28-46)."""
        e = self.engine
        lo, hi = int(self._offsets[frame_idx]), int(self._offsets[frame_idx + 1])
        one = torch.tensor([0, hi - lo], dtype=torch.int64, device=e.device)
        cur = e.gt_range_images(self._points[lo:hi], one, hi - lo, **self.proj)
        inv_cur = torch.from_numpy(np.ascontiguousarray(np.linalg.inv(self.poses[frame_idx]))).to(e.device)
        counts = None
        chunk = 512                                            # 512 range images = 118 MB of scratch per pass
        out = np.zeros(self.n)
        valid = 0
        for s0 in range(0, self.n, chunk):
            s1 = min(self.n, s0 + chunk)
            off = (self._offsets[s0:s1 + 1] - self._offsets[s0]).contiguous()
            p0, p1 = int(self._offsets[s0]), int(self._offsets[s1])
            imgs = e.gt_range_images(self._points[p0:p1], off, self.max_points, self._ref_poses[s0:s1].contiguous(), inv_cur,
                                     **self.proj)
            counts = e.gt_overlap_counts(imgs, cur).cpu().numpy()
            out[s0:s1] = counts[:-1]
            valid = int(counts[-1])
        if self.n and valid == 0:
            raise ZeroDivisionError("frame %d has no point inside the field of view" % frame_idx)
        return out / valid if self.n else out

    def mapping(self, frame_idx: int) -> np.ndarray:
        m = np.zeros((self.n, 4))
        m[:, 0] = np.ones(self.n) * frame_idx
        m[:, 1] = np.arange(self.n)
        m[:, 2] = self.overlaps(frame_idx)
        m[:, 3] = [yaw_bin(self.poses[frame_idx], self.poses[r], self.leg_output_width) for r in range(self.n)]
        return m


def com_overlap_yaw(scan_paths: List[str], poses, frame_idx: int, leg_output_width: int = 360,
                    engine: Optional[OvnEngine] = None) -> np.ndarray:
    """Same signature and result as the reference's com_overlap_yaw (com_overlap_yaw.py:10-68)."""
    print('Start to compute ground truth overlap and yaw ...')
    scans = [np.fromfile(p, dtype=np.float32).reshape((-1, 4)) for p in scan_paths]
    gt = OverlapGroundTruth(scans, np.asarray(poses), engine=engine, leg_output_width=leg_output_width)
    ground_truth_mapping = gt.mapping(frame_idx)
    print('Finish generating ground_truth_mapping!')
    return ground_truth_mapping
