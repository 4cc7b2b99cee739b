"""CPU: the network oracle against the reference's known-answer vectors, against its own literal
second formulation, and against the committed golden vectors (regression pin; parity unpinned vs TF)."""
import numpy as np
import pytest

from oracle import overlapnet_oracle as O
from tools import synthetic as S
from overlapnet_amd import weights as W


def test_range_padding_kat():
    # header comment of the reference layer: pad([1 2 3 4], 2) -> [3, 4, 1, 2, 3, 4, 1]  (RangePadding2D.py:5)
    x = np.array([1, 2, 3, 4], float).reshape(1, 1, 4, 1)
    assert O.range_padding_literal(x, 2).reshape(-1).tolist() == [3, 4, 1, 2, 3, 4, 1]
    # compute_output_shape: W -> 2W-1 (RangePadding2D.py:40-41)
    assert O.range_padding_literal(np.zeros((2, 1, 360, 3)), 180).shape == (2, 1, 719, 3)


def test_correlation_ramp_demo_kat():
    # the reference's __main__ demo (NormalizedCorrelation2D.py:112-144): 1x6x1 ramp vs ramp rolled by 1,
    # 'euclidean' normalisation -> peak 1.0 at index W/2 - 1 = 2
    a = np.arange(6, dtype=float).reshape(1, 1, 6, 1)
    b = np.roll(a, 1, axis=2)
    c = O.correlation_literal(a, b, "euclidean").reshape(-1)
    np.testing.assert_allclose(c, [0.5636, 0.7273, 1.0, 0.7273, 0.5636, 0.5091], atol=5e-5)
    assert int(np.argmax(c)) == 2


def test_correlation_closed_form_equals_literal_and_self_peak():
    rng = np.random.default_rng(3)
    l = np.maximum(rng.normal(size=(2, 1, 40, 5)), 0)
    r = np.maximum(rng.normal(size=(2, 1, 40, 5)), 0)
    fast = O.correlation_head_forward(l, r, np.float64)
    lit = O.correlation_literal(l, r).reshape(2, 40)
    np.testing.assert_allclose(fast, lit, rtol=1e-12, atol=1e-12)
    # self-correlation peaks at W/2 -> yaw 0 for W=360 (infer.py:158)
    f = np.maximum(rng.normal(size=(1, 1, 360, 8)), 0)
    c = O.correlation_head_forward(f, f)
    assert int(np.argmax(c[0])) == 180 and O.yaw_from_orientation(c)[0] == 0
    # rolling the LEFT features by s columns moves the peak by exactly s
    for s in (1, 17, 200):
        c2 = O.correlation_head_forward(np.roll(f, s, axis=2), f)
        assert int(np.argmax(c2[0])) == (180 + s) % 360


def test_conv_fast_equals_literal_small():
    rng = np.random.default_rng(5)
    x = rng.normal(size=(9, 31, 3))
    k = rng.normal(size=(3, 5, 3, 4))
    b = rng.normal(size=4)
    import torch
    for stride in ((1, 1), (2, 1), (2, 2)):
        xt = torch.as_tensor(x).permute(2, 0, 1).unsqueeze(0)
        fast = O._conv_valid(xt, k, b, stride, True, torch.float64)[0].permute(1, 2, 0).numpy()
        lit = O.conv2d_valid_literal(x, k, b, stride, True)
        np.testing.assert_allclose(fast, lit, rtol=1e-12, atol=1e-12)


def test_delta_head_fast_equals_literal_small():
    """Delta head on a 30-column toy feature volume with conv1size=5: materialised DeltaLayer + literal
    convs (generateNet.py:45-59,96-114) vs the torch form used at full size."""
    rng = np.random.default_rng(7)
    Wd, C, s = 30, 6, 5
    l = np.maximum(rng.normal(size=(1, 1, Wd, C)), 0)
    r = np.maximum(rng.normal(size=(1, 1, Wd, C)), 0)
    g = Wd // s
    w = {"c_conv1/kernel": rng.normal(size=(1, s, C, 7)) * 0.3, "c_conv1/bias": rng.normal(size=7) * 0.1,
         "c_conv2/kernel": rng.normal(size=(s, 1, 7, 9)) * 0.3, "c_conv2/bias": rng.normal(size=9) * 0.1,
         "c_conv3/kernel": rng.normal(size=(3, 3, 9, 11)) * 0.3, "c_conv3/bias": rng.normal(size=11) * 0.1,
         "overlap_output/kernel": rng.normal(size=((g - 2) * (g - 2) * 11, 1)) * 0.1,
         "overlap_output/bias": np.array([0.2])}
    ov, lg, inter = O.delta_head_forward(l, r, w, conv1size=s, return_intermediates=True)
    diff = O.delta_layer_literal(l[0], r[0])  # (W*1, W*1, C)
    assert diff.shape == (Wd, Wd, C)
    i, j, c = 4, 11, 2
    assert diff[i, j, c] == abs(l[0, 0, i, c] - r[0, 0, j, c])
    o1 = O.conv2d_valid_literal(diff, w["c_conv1/kernel"], w["c_conv1/bias"], (1, s), relu=False)
    o2 = O.conv2d_valid_literal(o1, w["c_conv2/kernel"], w["c_conv2/bias"], (s, 1), relu=True)
    o3 = O.conv2d_valid_literal(o2, w["c_conv3/kernel"], w["c_conv3/bias"], (1, 1), relu=True)
    np.testing.assert_allclose(inter["o1"], o1, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(inter["o2"], o2, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(inter["o3"], o3, rtol=1e-11, atol=1e-11)
    logit = o3.reshape(-1) @ w["overlap_output/kernel"].reshape(-1) + 0.2
    np.testing.assert_allclose(lg[0], logit, rtol=1e-11)
    np.testing.assert_allclose(ov[0], 1 / (1 + np.exp(-logit)), rtol=1e-12)
    # the head is NOT symmetric in (l, r) (c_conv1 pools r-columns, c_conv2 pools l-columns)
    ov2, lg2 = O.delta_head_forward(r, l, w, conv1size=s)
    assert abs(lg2[0] - lg[0]) > 1e-6


def test_leg_shapes_and_layer_table():
    for C in (1, 4, 5):
        layers = W.leg_layers(C, S.REFERENCE_MODEL_CFG)
        assert [l.name for l in layers] == ["s_conv1", "s_conv2", "s_conv3", "s_conv3a", "s_conv4", "s_conv5",
                                            "s_conv6", "s_conv7", "s_conv8", "s_conv9", "s_conv10"]
        assert W.leg_output_shape(64, 900, layers) == (1, 360, 128)
    # parameter counts quoted in SURVEY.md section 8a: 1,104,112 (leg, C=4) and 665,025 (head)
    shp = W.expected_shapes(4, S.REFERENCE_MODEL_CFG)
    leg = sum(int(np.prod(v)) for k, v in shp.items() if k.startswith("s_"))
    head = sum(int(np.prod(v)) for k, v in shp.items() if not k.startswith("s_"))
    assert leg == 1104112 and head == 665025
    assert W.dense_in_features() == 123904
    # without layer 3a the leg does not end in a 1x360 strip
    assert W.leg_output_shape(64, 900, W.leg_layers(4, {"strides_layer1": [2, 2]}))[0] != 1


@pytest.mark.parametrize("C", [1, 4, 5])
def test_oracle_reproduces_committed_golden(fixture_npz, nn_golden, C):
    flags = S.flags_of(C)
    imgs = np.stack([S.stack(fixture_npz["range_%d" % i], fixture_npz["normal_%d" % i],
                             fixture_npz["intensity_%d" % i], flags) for i in range(2)])
    w = S.make_test_weights(C, seed=0)
    fv = O.leg_forward(imgs, w, S.REFERENCE_MODEL_CFG, np.float64)
    assert fv.shape == (2, 1, 360, 128)
    np.testing.assert_allclose(fv.reshape(2, 360, 128), nn_golden["fv_c%d" % C], rtol=1e-5, atol=1e-6)
    pairs = nn_golden["pairs"]
    ov, yaw, lg, corr = O.heads_forward(fv[pairs[:, 0]], fv[pairs[:, 1]], w)
    np.testing.assert_allclose(lg, nn_golden["logit_c%d" % C], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(ov, nn_golden["overlap_c%d" % C], rtol=1e-9)
    assert np.array_equal(yaw, nn_golden["yaw_c%d" % C])
    np.testing.assert_allclose(corr, nn_golden["corr_c%d" % C], rtol=1e-9)
    # fp32 run of the same oracle stays within the north-star tolerance of the fp64 one
    fv32 = O.leg_forward(imgs, w, S.REFERENCE_MODEL_CFG, np.float32)
    ov32, yaw32, lg32, _ = O.heads_forward(fv32[pairs[:, 0]], fv32[pairs[:, 1]], w, dtype=np.float32)
    assert np.max(np.abs(ov32 - ov)) < 1e-4 and np.array_equal(yaw32, yaw)


def test_delta_head_literal_at_full_size():
    """One pair through the LITERAL forms at the real size (generateNet.py:45-59 tile + abs on 360 x 360 x 128, then
    c_conv1 1x15 stride (1,15) linear, c_conv2 15x1 stride (15,1), c_conv3 3x3, Flatten, Dense, :96-114) against the fast
    oracle: pins the stride-15 pooling semantics (which axis c_conv1 / c_conv2 pool) at 360 x 360 x 128, not only on toys."""
    rng = np.random.default_rng(17)
    w = S.make_test_weights(4, seed=0)
    l = np.maximum(rng.normal(0.2, 1.0, size=(360, 1, 128)), 0)      # (w, h, c) = the leg output with its unit height
    r = np.maximum(rng.normal(0.2, 1.0, size=(360, 1, 128)), 0)
    r[40:200] = l[10:170]                                            # partly overlapping content, asymmetric roles
    diff = O.delta_layer_literal(l, r)                               # (360, 360, 128): [i (left), j (right), c]
    assert diff.shape == (360, 360, 128)
    o1 = O.conv2d_valid_literal(diff, w["c_conv1/kernel"], w["c_conv1/bias"], (1, 15), relu=False)
    o2 = O.conv2d_valid_literal(o1, w["c_conv2/kernel"], w["c_conv2/bias"], (15, 1), relu=True)
    o3 = O.conv2d_valid_literal(o2, w["c_conv3/kernel"], w["c_conv3/bias"], (1, 1), relu=True)
    assert o1.shape == (360, 24, 64) and o2.shape == (24, 24, 128) and o3.shape == (22, 22, 256)
    logit = float(o3.reshape(-1) @ w["overlap_output/kernel"].astype(np.float64).reshape(-1) + w["overlap_output/bias"][0])
    fl = np.transpose(l, (1, 0, 2))[None]                            # (1, 1, 360, 128)
    fr = np.transpose(r, (1, 0, 2))[None]
    ov, lg, inter = O.delta_head_forward(fl, fr, w, return_intermediates=True)
    np.testing.assert_allclose(inter["o1"], o1, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(inter["o2"], o2, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(inter["o3"], o3, rtol=1e-11, atol=1e-11)
    assert abs(lg[0] - logit) < 1e-10 and abs(ov[0] - 1.0 / (1.0 + np.exp(-logit))) < 1e-12
    # the head is NOT symmetric in (l, r): swapping the roles changes the result
    _, lg_sw = O.delta_head_forward(fr, fl, w)
    assert abs(lg_sw[0] - lg[0]) > 1e-6
    # correlation head, literal padded sliding window at full size vs the Gram-diagonal closed form
    corr = O.correlation_head_forward(fl, fr)
    np.testing.assert_allclose(corr, O.correlation_literal(fl, fr).reshape(1, 360), rtol=1e-11, atol=1e-9)


def test_oracle_convolution_against_scipy():
    """Third implementation of the one primitive the neural oracle stands on: Keras `Conv2D(padding='valid')` is a cross-correlation
    (no kernel flip) sub-sampled by the stride.  The oracle's fast form (torch conv2d) and its literal NumPy form are compared with
    scipy.signal.correlate on the layer shapes the leg and the head use (strides (2,2), (2,1), (1,s), (s,1))."""
    import torch
    from scipy.signal import correlate
    rng = np.random.default_rng(11)
    for (h, w, cin, cout, kh, kw, sh, sw) in ((11, 37, 4, 3, 5, 15, 2, 2), (9, 40, 3, 5, 3, 15, 2, 1), (1, 30, 6, 4, 1, 9, 1, 1),
                                              (12, 30, 2, 3, 1, 15, 1, 15), (30, 4, 3, 2, 15, 1, 15, 1), (8, 8, 5, 7, 3, 3, 1, 1)):
        x = rng.normal(size=(h, w, cin))
        k = rng.normal(size=(kh, kw, cin, cout))
        b = rng.normal(size=cout)
        want = np.stack([sum(correlate(x[:, :, c], k[:, :, c, o], mode="valid") for c in range(cin)) + b[o] for o in range(cout)], axis=-1)
        want = want[::sh, ::sw]
        fast = O._conv_valid(torch.from_numpy(x).permute(2, 0, 1)[None], k, b, (sh, sw), False, torch.float64)[0].permute(1, 2, 0).numpy()
        lit = O.conv2d_valid_literal(x, k, b, (sh, sw), relu=False)
        assert fast.shape == want.shape == lit.shape == ((h - kh) // sh + 1, (w - kw) // sw + 1, cout)
        np.testing.assert_allclose(fast, want, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(lit, want, rtol=1e-12, atol=1e-12)
