"""Ground-truth (overlap, yaw bin) restatement against vectors produced by the reference's own
src/utils/com_overlap_yaw.py (tests/golden/make_gt_golden.py)."""
import os

import numpy as np
import pytest

from oracle import overlapnet_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gt():
    z = np.load(os.path.join(G, "gt_overlap_yaw.npz"))
    pts = np.load(os.path.join(G, "kitti_preprocess.npz"))
    scans = [pts["points_%d" % s] for s in z["scan_of"]]
    return z, scans


@pytest.mark.parametrize("frame", [0, 4, 7, 11])
def test_mapping_equals_reference(gt, frame):
    z, scans = gt
    m = O.com_overlap_yaw(scans, z["poses"], frame)
    ref = z["mapping_%d" % frame]
    assert np.array_equal(m[:, [0, 1, 3]], ref[:, [0, 1, 3]])          # ids and yaw bins: exact (incl. the bin-360 quirk)
    assert np.array_equal(m[:, 2], ref[:, 2]), np.max(np.abs(m[:, 2] - ref[:, 2]))


def test_yaw_bin_quirks():
    z = np.load(os.path.join(G, "gt_overlap_yaw.npz"))
    bins7 = z["mapping_7"][:, 3]
    assert bins7.max() == 360           # yaw = -180 deg maps to bin 360, outside [0, 359]: kept as the reference produces it
    eye = np.eye(4)
    assert O.yaw_bin_from_poses(eye, eye) == 180
