"""CPU: host-side logic that needs no GPU -- weight containers, synthetic generators, loop-closure gating, metrics."""
import os

import numpy as np
import pytest

from overlapnet_amd import evaluate as E
from overlapnet_amd import lcd
from tools import synthetic as S
from overlapnet_amd import weights as W


def test_weight_container_roundtrip_and_checks(tmp_path):
    w = S.make_test_weights(4, seed=3)
    W.check_weights(w, 4, S.REFERENCE_MODEL_CFG)
    p = str(tmp_path / "w.npz")
    W.save_npz(p, w)
    w2 = W.load_weights_file(p)
    assert sorted(w2) == sorted(w) and all(np.array_equal(w[k], w2[k]) for k in w)
    assert len(w) == 30  # 15 layers x (kernel, bias): s_conv1..10 + s_conv3a, c_conv1..3, overlap_output
    bad = dict(w)
    bad.pop("s_conv3a/kernel")
    with pytest.raises(KeyError):
        W.check_weights(bad, 4, S.REFERENCE_MODEL_CFG)
    bad = dict(w, **{"c_conv1/kernel": np.zeros((1, 15, 128, 32), np.float32)})
    with pytest.raises(ValueError):
        W.check_weights(bad, 4, S.REFERENCE_MODEL_CFG)
    with pytest.raises(Exception, match="not found"):
        W.load_weights_file(str(tmp_path / "missing.weight"))
    junk = tmp_path / "junk.weight"
    junk.write_bytes(b"not a weight file")
    with pytest.raises(Exception, match="unrecognised"):
        W.load_weights_file(str(junk))
    # Keras default init: glorot-uniform kernels within their limit, zero biases (what infer.py:121-122 leaves)
    k = W.keras_default_init(4, S.REFERENCE_MODEL_CFG, seed=1)
    lim = np.sqrt(6.0 / (5 * 15 * 4 + 5 * 15 * 16))
    assert np.all(np.abs(k["s_conv1/kernel"]) <= lim) and np.all(k["s_conv1/bias"] == 0)


def test_synthetic_inputs_are_deterministic_and_shaped():
    a = S.candidate_images(5, 4, seed=11)
    b = S.candidate_images(5, 4, seed=11)
    assert a.shape == (5, 64, 900, 4) and a.dtype == np.float32 and np.array_equal(a, b)
    fx = S.load_fixture_images()
    # candidate 0 is fixture scan 0 unshifted: invalid mask preserved, only valid depths perturbed
    assert np.array_equal(a[0, ..., 0] == -1, fx["range_0"] == -1)
    assert np.array_equal(a[0, ..., 1:4], fx["normal_0"])
    valid = fx["range_0"] > 0
    assert np.max(np.abs(a[0, ..., 0][valid] - fx["range_0"][valid])) < 0.2
    # candidate 3 = fixture 1 shifted by 111 columns
    assert np.array_equal(a[3, ..., 1:4], np.roll(fx["normal_1"], 111, axis=1))
    assert S.flags_of(5) == (True, True, True) and S.channels_of(True, True, False) == 4
    with pytest.raises(ValueError):
        S.flags_of(7)


def test_loop_closure_gating_and_decision():
    # straight drive out along x, a sideways loop, and a return next to the start
    t = np.arange(400)
    xy = np.stack([np.where(t < 200, t * 1.0, 399.0 - t), np.where(t < 200, 0.0, 2.0)], axis=1)
    tl = lcd.travelled_distances(xy)
    assert tl[0] == 0 and abs(tl[199] - 199.0) < 1e-9
    ell = lcd.covariance_ellipse(np.diag([9.0, 4.0]), nstd=3.0)
    assert abs(ell[0] - 18.0) < 1e-9 and abs(ell[1] - 12.0) < 1e-9
    assert lcd.gate_candidates(50, xy, tl, ell).size == 0          # younger than the inactive window
    ref = lcd.gate_candidates(390, xy, tl, ell)                     # near x = 9 on the way back
    assert ref.size > 0 and np.all(ref < 290)
    assert np.all(np.abs(xy[ref, 0] - xy[390, 0]) <= 9.0 + 1e-9)   # inside the 3-sigma half-width along x
    assert np.all(tl[390] - tl[ref] > 50.0)
    assert lcd.decide(ref, np.full(ref.size, 0.1), np.zeros(ref.size, int)) is None
    ov = np.linspace(0.2, 0.9, ref.size)
    d = lcd.decide(ref, ov, np.arange(ref.size))
    assert d == (int(ref[-1]), float(ov[-1]), ref.size - 1)
    assert lcd.decide([], [], []) is None

    class FakeInfer:
        def __init__(self):
            self.calls = []

        def infer_multiple(self, cur, refs):
            self.calls.append((cur, list(refs)))
            if not refs:
                return None
            return np.linspace(0.2, 0.9, len(refs)), np.zeros(len(refs), int)

    f = FakeInfer()
    assert lcd.detect(f, 10, xy, tl, ell) is None and f.calls[-1] == (10, [])
    got = lcd.detect(f, 390, xy, tl, ell)
    assert got is not None and got[0] == int(ref[-1]) and f.calls[-1][0] == 390


def test_error_statistics(tmp_path):
    gt = np.array([[0, 1, 0.9, 180], [0, 2, 0.2, 170], [1, 2, 0.8, 5]], float)
    np.savez(tmp_path / "gt.npz", overlaps=gt, seq=np.array([["07", "07"]] * 3, dtype=object))
    i1, i2, ov, yb = E.load_ground_truth(str(tmp_path / "gt.npz"))
    assert list(i2) == [1, 2, 2] and list(yb) == [180, 170, 5]
    assert list(E.yaw_bin_to_degrees(yb)) == [0, 10, 175]
    assert list(E.circular_error_deg([179, -179, 0], [-179, 179, 10])) == [2, 2, 10]
    st = E.error_statistics([0.85, 0.3, 0.7], ov, [0, 0, -178], E.yaw_bin_to_degrees(yb))
    assert st["n"] == 3 and abs(st["overlap_mae"] - (0.05 + 0.1 + 0.1) / 3) < 1e-12 and abs(st["overlap_max"] - 0.1) < 1e-12
    assert st["yaw_n"] == 2 and st["yaw_max_err_deg"] == 7 and abs(st["yaw_mean_err_deg"] - 3.5) < 1e-12


def test_pair_reader_both_layouts_and_test_run(tmp_path):
    """load_pairs against files written the way demo4 writes them (demo4_gen_gt_files.py:97-109) and in the old
    single-array layout, cross-checked with the reference's own reader when the reference tree is present."""
    import os
    import sys
    rng = np.random.default_rng(3)
    n = 7
    arr = np.zeros((n, 4))
    arr[:, 0] = 5
    arr[:, 1] = rng.permutation(n)
    arr[:, 2] = rng.random(n)
    arr[:, 3] = rng.integers(0, 360, n)
    seq = np.empty((n, 2), dtype=object)
    seq[:] = "07"
    new = str(tmp_path / "gt.npz")
    old = str(tmp_path / "old.npz")
    np.savez_compressed(new, overlaps=arr, seq=seq)
    np.savez(old, arr)
    f1, f2, d1, d2, ov, ori = E.load_pairs([new, old])
    assert f1[:n] == ["000005"] * n and f2[:n] == ["%06d" % v for v in arr[:, 1]] and d1[:n] == ["07"] * n
    assert d1[n:] == [""] * n and np.array_equal(ov, np.tile(arr[:, 2], 2)) and np.array_equal(ori, np.tile(arr[:, 3], 2))
    ref_dir = "/root/reference/src/two_heads"
    if os.path.isdir(ref_dir):                      # same answer as the reference's reader
        sys.path.insert(0, ref_dir)
        try:
            from overlap_orientation_npz_file2string_string_nparray import overlap_orientation_npz_file2string_string_nparray as ref
        finally:
            sys.path.pop(0)
        r = ref([new, old], shuffle=False)
        assert list(r[0]) == f1 and list(r[1]) == f2 and list(r[2]) == d1 and list(r[3]) == d2
        assert np.array_equal(r[4], ov) and np.array_equal(r[5], ori)

    class FakeInfer:            # records the pair roles run_test asks for
        def infer_multiple_vs_multiple(self, names, first, second):
            self.names, self.first, self.second = list(names), list(first), list(second)
            k = len(first)
            return np.linspace(0.1, 0.9, k).astype(np.float32), (180 - np.arange(k) * 3).astype(np.int64)

    inf = FakeInfer()
    st = E.run_test(inf, [new], no_test_pairs=5, out_dir=str(tmp_path / "res"))
    assert inf.names == sorted(set(f1[:5]) | set(f2[:5]))
    assert [inf.names[i] for i in inf.second] == f1[:5] and [inf.names[i] for i in inf.first] == f2[:5]   # l = imgf1, r = imgf2
    m = np.load(str(tmp_path / "res" / "validation_results.npz"))["arr_0"]
    assert m.shape == (5, 4) and np.array_equal(m[:, 3], np.arange(5) * 3) and np.allclose(m[:, 2], np.linspace(0.1, 0.9, 5))
    assert st["n"] == 5 and abs(st["overlap_mae"] - np.mean(np.abs(np.linspace(0.1, 0.9, 5) - arr[:5, 2]))) < 1e-6


@pytest.mark.parametrize("name", ["out_and_back", "figure_eight", "random_walk"])
def test_loop_closure_gating_against_reference_golden(name):
    """PINNED on the reference's own code: tests/golden/lcd_gating.npz was produced by running `AnimatedLCD.get_cov_ellipse` /
    `get_predictions` of demo/demo3_lcd.py:85-140 (imported unmodified, tests/golden/make_lcd_golden.py) over synthetic
    trajectories with a recorder in place of the network.  Frame by frame: the same ellipse, the same candidate list handed to
    `infer_multiple`, the same reported loop closure."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_lcd_golden", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                                                                  "make_lcd_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)                       # only its RecorderInfer / fake_overlap helpers; no reference import
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lcd_gating.npz")) as z:
        g = {k[len(name) + 1:]: z[k] for k in z.files if k.startswith(name + "_")}
    xy, cov = g["xy"], g["cov"]
    tl = lcd.travelled_distances(xy)
    rec = gen.RecorderInfer()
    for idx in range(len(xy)):
        ell = lcd.covariance_ellipse(cov[idx], nstd=3.0)
        np.testing.assert_allclose(ell, g["ellipse"][idx], rtol=1e-12, atol=1e-12)
        ref = lcd.gate_candidates(idx, xy, tl, ell)
        want = g["refs"][g["refs_off"][idx]:g["refs_off"][idx + 1]]
        assert np.array_equal(ref, want), (name, idx)
        got = lcd.detect(rec, idx, xy, tl, ell)
        assert rec.calls[-1] == (idx, list(want))      # the frame is ALWAYS fed to infer_multiple (it caches the feature volume)
        assert (-2 if got is None else got[0]) == g["result"][idx], (name, idx, got)
        if got is not None:
            k = list(want).index(got[0])
            assert got[1] == float(gen.fake_overlap(idx, want)[k]) and got[2] == int(gen.fake_yaw(idx, want)[k])
    assert len(rec.calls) == len(xy) and int(np.sum(g["result"] >= 0)) > 50
