"""CPU: the preprocessing oracle against the golden vectors produced by the REFERENCE's own code
(tests/golden/make_preprocess_golden.py ran src/utils/utils.py and checked the shipped .npy fixtures)."""
import numpy as np
import pytest

from oracle import overlapnet_oracle as O


@pytest.mark.parametrize("scan", [0, 1])
@pytest.mark.parametrize("trig", ["svml", "numpy", "f64"])
def test_range_projection_matches_reference(fixture_npz, scan, trig):
    pts = fixture_npz["points_%d" % scan]
    rng, vtx, inten, idx = O.range_projection(pts, trig=trig)
    ref_rng = fixture_npz["range_%d" % scan]
    # 'svml' (the restated NumPy float32 functions of the generating machine) must reproduce every pixel on any host; this host's
    # NumPy ('numpy') and the correctly rounded functions ('f64') may move a point across a pixel edge now and then
    diff = rng != ref_rng
    assert diff.sum() <= (0 if trig == "svml" else 4), "range image differs from the reference in %d pixels" % diff.sum()
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


def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_transformed_clouds_match_the_reference(fixture_npz):
    """24 rotated / translated / tilted copies of the two scans (tests/golden/preprocess_transformed.npz: outputs of the reference's
    own range_projection + gen_normal_map): the RANGE image is identical on every cloud; the index image differs only where two
    points of a pixel share the minimal depth bit for bit (the reference's unstable argsort picks either, utils.py:108; the lowest
    index wins here); the normal map restated here equals the reference's on the reference's own vertex map."""
    from tools import synthetic as S
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "preprocess_transformed.npz"))
    ties = f64_range_diffs = 0
    for i in range(S.N_TRANSFORMED):
        pts = S.transformed_cloud(fixture_npz, i)
        assert _sha(pts) == str(g["sha_cloud_%d" % i]), "cloud %d is not reproduced on this host" % i
        rng, vtx, inten, idx = O.range_projection(pts)            # trig='svml'
        assert _sha(rng) == str(g["sha_range_%d" % i]), "range image of cloud %d differs from the reference" % i
        gi = g["idx_%d" % i]
        x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
        depth = np.sqrt((x * x + y * y) + z * z)
        kept = pts[(depth > 0) & (depth < 50)]
        dk = depth[(depth > 0) & (depth < 50)]
        assert len(kept) == int(g["n_kept_%d" % i])
        bad = np.argwhere(idx != gi)
        for (r, c) in bad:                                          # only exact depth ties may differ, towards the lower index
            assert dk[idx[r, c]] == dk[gi[r, c]] and idx[r, c] < gi[r, c]
        ties += len(bad)
        # the reference's images rebuilt from ITS index image -> its intensity and (through the restated gen_normal_map) its normals
        valid = gi >= 0
        g_rng = np.where(valid, dk[np.maximum(gi, 0)], np.float32(-1))
        g_int = np.where(valid, kept[np.maximum(gi, 0), 3], np.float32(-1))
        g_vtx = np.full((64, 900, 4), -1, np.float32)
        g_vtx[valid, :3] = kept[gi[valid], :3]
        g_vtx[valid, 3] = 1
        assert _sha(g_rng) == str(g["sha_range_%d" % i]) and _sha(g_int) == str(g["sha_intensity_%d" % i])
        assert _sha(O.gen_normal_map(g_rng, g_vtx)) == str(g["sha_normal_%d" % i]), "normal map of cloud %d" % i
        if not len(bad):
            assert _sha(inten) == str(g["sha_intensity_%d" % i])
        f64_range_diffs += int((O.range_projection(pts, trig="f64")[0] != g_rng).sum())
    assert ties == 10            # measured: 20 tie pixels in the 24 clouds, the reference took the higher index in 10 of them
    assert f64_range_diffs == 12  # what rounds 1-3 of the HIP kernel (float64 functions rounded to float32) got wrong


def test_other_geometries_match_the_reference(fixture_npz):
    """The reference's keyword arguments (image size, field of view, maximum range; utils.py:59), incl. a steep field of view on a
    pitched cloud where 27 % of the points take arcsin's |x| >= 0.5 branch (VRSQRT14PS in NumPy's SVML kernel): identical range
    images; index differences only at exact depth ties."""
    from tools import synthetic as S
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "preprocess_transformed.npz"))
    for k, (ci, _, H, W, up, down, mr) in enumerate(S.GEOMETRY_CASES):
        pts = S.geometry_cloud(fixture_npz, k)
        assert _sha(pts) == str(g["geo_sha_cloud_%d" % k])
        rng, vtx, inten, idx = O.range_projection(pts, fov_up=up, fov_down=down, proj_H=H, proj_W=W, max_range=mr)
        assert rng.shape == (H, W) and _sha(rng) == str(g["geo_sha_range_%d" % k]), k
        gi = g["geo_idx_%d" % k]
        x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
        depth = np.sqrt((x * x + y * y) + z * z)
        dk = depth[(depth > 0) & (depth < mr)]
        for (r, c) in np.argwhere(idx != gi):
            assert dk[idx[r, c]] == dk[gi[r, c]]
        if np.array_equal(idx, gi):
            assert _sha(inten) == str(g["geo_sha_intensity_%d" % k])
            assert _sha(O.gen_normal_map(rng, vtx, H, W)) == str(g["geo_sha_normal_%d" % k])


def test_literal_normal_map_equals_the_vectorised_one_and_the_shipped_image(fixture_npz):
    """`gen_normal_map_literal` (the reference's per-pixel double loop restated, what bench.py times as the CPU baseline of the
    preprocessing stage) == the vectorised oracle == the reference's shipped `normal/000000.npy`, bit for bit."""
    rng, vtx, _, _ = O.range_projection(fixture_npz["points_0"])
    lit = O.gen_normal_map_literal(rng, vtx)
    assert np.array_equal(lit, O.gen_normal_map(rng, vtx))
    assert np.array_equal(lit, fixture_npz["normal_0"])

