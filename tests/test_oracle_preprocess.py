"""CPU: the preprocessing oracle against the golden vectors produced by the REFERENCE's own code
(tests/golden/make_preprocess_golden.py ran src/utils/utils.py and checked the shipped .npy fixtures)."""
import numpy as np
import pytest

from oracle import overlapnet_oracle as O


@pytest.mark.parametrize("scan", [0, 1])
@pytest.mark.parametrize("trig64", [False, True])
def test_range_projection_matches_reference(fixture_npz, scan, trig64):
    pts = fixture_npz["points_%d" % scan]
    rng, vtx, inten, idx = O.range_projection(pts, trig64=trig64)
    ref_rng = fixture_npz["range_%d" % scan]
    # NumPy's float32 arctan2/arcsin differ by CPU dispatch in the last ulp: allow a handful of pixels,
    # every other pixel must be bit-identical (0 differ in the build container for both variants).
    diff = rng != ref_rng
    assert diff.sum() <= 4, "range image differs from the reference in %d pixels" % diff.sum()
    same = ~diff
    assert np.array_equal(inten[same], fixture_npz["intensity_%d" % scan][same])
    assert np.array_equal(idx[same], fixture_npz["idx_%d" % scan][same])
    assert rng.dtype == np.float32 and idx.dtype == np.int32 and vtx.shape == (64, 900, 4)
    # vertex map is the winning point with w = 1, -1 where empty
    valid = rng > 0
    assert np.all(vtx[valid][:, 3] == 1) and np.all(vtx[~valid] == -1)
    kept = pts[(np.linalg.norm(pts[:, :3], 2, axis=1) > 0) & (np.linalg.norm(pts[:, :3], 2, axis=1) < 50)]
    assert np.array_equal(vtx[valid][:, :3], kept[idx[valid]][:, :3])


@pytest.mark.parametrize("scan", [0, 1])
def test_normal_map_matches_reference_bit_exact(fixture_npz, scan):
    pts = fixture_npz["points_%d" % scan]
    rng, vtx, _, _ = O.range_projection(pts)
    if not np.array_equal(rng, fixture_npz["range_%d" % scan]):
        pytest.skip("range image differs by trig ulps on this CPU; normals compared only on identical input")
    nrm = O.gen_normal_map(rng, vtx)
    assert np.array_equal(nrm, fixture_npz["normal_%d" % scan])
    assert np.all(nrm[63] == -1)  # last row never gets a normal (utils.py:150)


def test_projection_edge_cases():
    # empty cloud, all-out-of-range cloud, origin points: everything stays -1
    for pts in (np.zeros((0, 4), np.float32), np.array([[100, 0, 0, 1], [0, 0, 0, 1]], np.float32)):
        rng, vtx, inten, idx = O.range_projection(pts)
        assert np.all(rng == -1) and np.all(idx == -1) and np.all(vtx == -1) and np.all(inten == -1)
    # two points in one pixel: the nearer one wins regardless of order; index is post-filter
    a = np.array([[60, 0, 0, 0.1], [10, 0, 0, 0.5], [5, 0, 0, 0.9]], np.float32)
    for perm in ([0, 1, 2], [0, 2, 1]):
        rng, vtx, inten, idx = O.range_projection(a[perm])
        assert (rng > 0).sum() == 1
        y, x = np.argwhere(rng > 0)[0]
        assert rng[y, x] == 5 and inten[y, x] == np.float32(0.9)
        assert idx[y, x] == (1 if perm == [0, 1, 2] else 0)  # the 60 m point is filtered before numbering
    # equal depth tie -> lowest index
    t = np.array([[5, 0, 0, 0.3], [5, 0, 0, 0.7]], np.float32)
    rng, vtx, inten, idx = O.range_projection(t)
    y, x = np.argwhere(rng > 0)[0]
    assert idx[y, x] == 0 and inten[y, x] == np.float32(0.3)


def test_stack_channels_order(fixture_npz):
    d, n, it = fixture_npz["range_0"], fixture_npz["normal_0"], fixture_npz["intensity_0"]
    x = O.stack_channels(d, n, it)
    assert x.shape == (64, 900, 5)
    assert np.array_equal(x[..., 0], d) and np.array_equal(x[..., 1:4], n) and np.array_equal(x[..., 4], it)
    assert O.stack_channels(d, n, None).shape == (64, 900, 4)
    assert O.stack_channels(d, None, None).shape == (64, 900, 1)
