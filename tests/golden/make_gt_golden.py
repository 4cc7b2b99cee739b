#!/usr/bin/env python3
"""Generate tests/golden/gt_overlap_yaw.npz by RUNNING the reference's ground-truth generator
(src/utils/com_overlap_yaw.py:10-68, imported unmodified) on the two scans it ships and a set of synthetic poses.

Run in the build container only (needs /root/reference):    python tests/golden/make_gt_golden.py

Stored: poses (n,4,4) f64, scan_of (n,) which fixture scan each "frame" is, and for frame_idx in FRAMES the
(n,4) mapping [current idx, reference idx, overlap, yaw bin] the reference returns.  The points themselves are
already in tests/golden/kitti_preprocess.npz (points_0 / points_1).
"""
import os
import shutil
import sys
import tempfile

import numpy as np

REF = os.environ.get("OVERLAPNET_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REF, "src", "utils"))
import com_overlap_yaw as ref_gt  # noqa: E402  (the reference module, not ours)


def pose(x, y, z, yaw_deg, pitch_deg=0.0, roll_deg=0.0):
    cy, sy = np.cos(np.radians(yaw_deg)), np.sin(np.radians(yaw_deg))
    cp, sp = np.cos(np.radians(pitch_deg)), np.sin(np.radians(pitch_deg))
    cr, sr = np.cos(np.radians(roll_deg)), np.sin(np.radians(roll_deg))
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = [x, y, z]
    return T


# frame i uses fixture scan i % 2; a short drive with turns, one exact revisit, one far-away frame, +-180 deg cases
specs = [(0, 0, 0, 0), (0.8, 0.05, 0, 1.5), (1.7, 0.1, 0.01, 3.0), (3.0, 0.5, 0.0, 20.0), (4.0, 1.5, 0.02, 45.0, 1.0, -0.5),
         (4.5, 3.0, 0.0, 90.0), (4.0, 5.0, 0.0, 135.0), (2.5, 6.0, 0.0, 180.0), (1.0, 6.2, 0.0, -179.6), (0.0, 5.0, 0.0, -135.0),
         (-0.5, 3.0, 0.0, -90.0, 0.0, 2.0), (0.0, 0.0, 0.0, 0.0), (60.0, -40.0, 0.0, 10.0), (0.2, -0.1, 0.0, -0.4), (0.1, 0.0, 0.0, 179.9)]
poses = np.stack([pose(*s) for s in specs])
n = len(specs)
scan_of = np.arange(n) % 2

tmp = tempfile.mkdtemp()
try:
    paths = []
    for i in range(n):
        dst = os.path.join(tmp, "%06d.bin" % i)
        shutil.copy(os.path.join(REF, "data", "scans", "%06d.bin" % scan_of[i]), dst)
        paths.append(dst)
    out = {"poses": poses, "scan_of": scan_of}
    for f in (0, 4, 7, 11):
        out["mapping_%d" % f] = ref_gt.com_overlap_yaw(paths, poses, frame_idx=f)
        print(f, np.round(out["mapping_%d" % f][:, 2], 4), out["mapping_%d" % f][:, 3].astype(int))
finally:
    shutil.rmtree(tmp)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gt_overlap_yaw.npz"), **out)
print("wrote gt_overlap_yaw.npz")
