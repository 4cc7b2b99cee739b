"""GPU (MI355X): the HIP path, called through the C ABI, against the CPU oracle and the golden vectors.

Tolerances (north star: |d overlap| <= 1e-4, exact yaw bin):
  * activations / corr vectors: max |gpu - fp64 oracle| <= 2e-5 * max|oracle|, in fp32 mode and in the default f16x3 mode
    (scaled 3-term fp16 split, fp32 accumulate) alike
  * logit: |d| <= 1e-3 * (1 + |logit|);  overlap: |d| <= 1e-4;  yaw: identical bin unless the oracle's
    own top-2 gap is below 1e-5 relative (reported, not failed)
  * projection: bit-identical images on both shipped scans (measured: 0 differing pixels)
"""
import os

import numpy as np
import pytest
import torch

from oracle import overlapnet_oracle as O
from tools import synthetic as S
from overlapnet_amd import weights as W

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not torch.cuda.is_available(), reason="needs an MI355X")]

CFG = S.REFERENCE_MODEL_CFG


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))) / (np.max(np.abs(b)) + 1e-30))


@pytest.fixture(scope="module")
def engines():
    from overlapnet_amd.engine import OvnEngine
    out = {}
    for C in (1, 4, 5):
        e = OvnEngine(64, 900, C)
        e.load_weights(S.make_test_weights(C, seed=0), CFG)
        out[C] = e
    yield out
    for e in out.values():
        e.close()


@pytest.fixture(scope="module")
def fixture_images(fixture_npz):
    def make(C):
        flags = S.flags_of(C)
        return np.stack([S.stack(fixture_npz["range_%d" % i], fixture_npz["normal_%d" % i],
                                 fixture_npz["intensity_%d" % i], flags) for i in range(2)])
    return make


def test_native_library_is_loaded_and_mfma_layout(engines):
    import ctypes
    from overlapnet_amd import _lib
    assert isinstance(_lib.load(), ctypes.CDLL)
    maps = open("/proc/self/maps").read()
    assert "libovn_hip.so" in maps, "the in-tree HIP extension is not mapped into this process"
    engines[4].selftest()


@pytest.mark.parametrize("C", [1, 4, 5])
def test_each_leg_layer_against_oracle(C):
    """Every conv layer of the leg in isolation (vec4 and scalar gathers, K tails, M tails, all strides)."""
    import ctypes
    from overlapnet_amd import _lib
    from overlapnet_amd.engine import OvnEngine, _ptr
    rng = np.random.default_rng(100 + C)
    w = S.make_test_weights(C, seed=1)
    h, wd = 64, 900
    for l in W.leg_layers(C, CFG):
        eng = OvnEngine(h, wd, l.cin)
        lib = eng.lib
        # enough scans for the strip kernels to take the first two layers (>= 384 workgroups), a couple for the rest
        nb = 12 if l.name in ("s_conv1", "s_conv2") else 2
        x = rng.normal(size=(nb, h, wd, l.cin)).astype(np.float32)
        k = w[l.name + "/kernel"]
        b = w[l.name + "/bias"]
        kt = torch.from_numpy(k).cuda()
        bt = torch.from_numpy(b).cuda()
        st = eng._stream()
        _lib.check(lib.ovn_add_leg_layer(eng._h, l.name.encode(), _ptr(kt), _ptr(bt), l.kh, l.kw, l.cin, l.cout, l.sh,
                                         l.sw, st), "add")
        oh, ow = (h - l.kh) // l.sh + 1, (wd - l.kw) // l.sw + 1
        xt = torch.from_numpy(x).cuda()
        out = torch.empty((nb, oh, ow, l.cout), dtype=torch.float32, device="cuda")
        ref = O._conv_valid(torch.from_numpy(x.astype(np.float64)).permute(0, 3, 1, 2), k, b, (l.sh, l.sw), True,
                            torch.float64).permute(0, 2, 3, 1).numpy()
        for mode in ("f32", "f16x3"):           # both conv kernels (conv_f32.hip, conv_f16x3.hip)
            eng.set_leg_precision(mode)
            out.fill_(float("nan"))
            rc = lib.ovn_debug_conv(eng._h, 0, _ptr(xt), nb, h, wd, _ptr(out), st)
            _lib.check(rc, "ovn_debug_conv")
            torch.cuda.synchronize()
            err = _rel(out.cpu().numpy(), ref)
            assert err < 2e-5, "%s (C=%d, %s): rel err %.3g" % (l.name, C, mode, err)
        eng.close()
        h, wd = oh, ow
    assert (h, wd) == (1, 360)


@pytest.mark.parametrize("C", [1, 4, 5])
def test_leg_against_oracle_and_golden(engines, fixture_images, nn_golden, C):
    imgs = fixture_images(C)
    ref = O.leg_forward(imgs, S.make_test_weights(C, seed=0), CFG, np.float64).reshape(2, 360, 128)
    assert engines[C].leg_precision == "f16x3"           # the default arithmetic
    fv_default = engines[C].leg(torch.from_numpy(imgs).cuda()).cpu().numpy()
    assert _rel(fv_default, ref) < 2e-5 and _rel(fv_default, nn_golden["fv_c%d" % C]) < 2e-5
    engines[C].set_leg_precision("f32")
    try:
        fv = engines[C].leg(torch.from_numpy(imgs).cuda()).cpu().numpy()
    finally:
        engines[C].set_leg_precision("f16x3")
    assert fv.shape == (2, 360, 128)
    assert _rel(fv, ref) < 2e-5
    assert _rel(fv, nn_golden["fv_c%d" % C]) < 2e-5
    # zero pattern of the ReLU output agrees except where the oracle value is within fp32 noise of 0
    mism = (fv == 0) != (ref == 0)
    assert np.all(np.abs(ref[mism]) < 1e-5 * np.max(ref))


@pytest.mark.parametrize("C", [1, 4, 5])
def test_leg_f16x3_mode(engines, fixture_images, C):
    """Leg convolutions on the fp16 matrix cores with the scaled 3-term split: features within 2e-5 of the fp64 oracle
    (the fp32-mode bound), and the end-to-end overlap/yaw gates hold on features produced this way."""
    imgs = np.concatenate([fixture_images(C), S.candidate_images(4, C, seed=9)[2:]])
    w = S.make_test_weights(C, seed=0)
    e = engines[C]
    assert e.leg_precision == "f16x3"
    fv = e.leg(torch.from_numpy(imgs).cuda())
    ref = O.leg_forward(imgs, w, CFG, np.float64)
    err = _rel(fv.cpu().numpy(), ref.reshape(-1, 360, 128))
    assert err < 2e-5, "f16x3 leg rel err %.3g" % err
    pairs = np.array([[0, 1], [1, 0], [2, 0], [3, 1], [2, 3]])
    r = e.heads(fv, fv, lidx=pairs[:, 0], ridx=pairs[:, 1])
    ov, yaw, _, corr = O.heads_forward(ref[pairs[:, 0]], ref[pairs[:, 1]], w)
    assert np.max(np.abs(r["overlap"].cpu().numpy() - ov)) <= 1e-4
    srt = np.sort(corr, axis=1)
    gap = (srt[:, -1] - srt[:, -2]) / np.abs(srt[:, -1])
    bad = r["yaw"].cpu().numpy() != yaw
    assert not np.any(bad & (gap > 1e-5)), (r["yaw"].cpu().numpy(), yaw, gap)


def test_leg_batch_tail_and_slicing(engines, fixture_images):
    """n not a multiple of any tile, and > the internal 1024-scan slice (1027 scans = slices of 514 + 513): same result per scan."""
    imgs = fixture_images(4)
    e = engines[4]
    one = e.leg(torch.from_numpy(imgs[:1]).cuda())
    assert torch.equal(one, e.leg(torch.from_numpy(imgs[:1]).cuda()))      # run-to-run deterministic
    many = torch.from_numpy(imgs[:1]).cuda().expand(1027, -1, -1, -1).contiguous()
    out = e.leg(many)
    assert torch.equal(out, out[:1].expand(1027, -1, -1))                    # every batch position identical
    assert torch.equal(one, out[:1])                                         # ... and identical to the scan computed alone
    e.set_leg_precision("f32")
    try:
        assert torch.equal(e.leg(many[:20])[:3], e.leg(many[:1]).expand(3, -1, -1))   # fp32 mode: one kernel, bit-identical
    finally:
        e.set_leg_precision("f16x3")
    assert e.leg(torch.empty((0, 64, 900, 4), device="cuda")).shape == (0, 360, 128)


@pytest.mark.parametrize("C", [1, 4, 5])
def test_leg_result_does_not_depend_on_the_batch(engines, fixture_images, C):
    """A scan's feature volume depends on that scan alone (reference: `leg.predict_generator`, infer.py:262-265): computed alone, in a
    batch of 9 next to a 300x-scaled neighbour (which moves every call-wide activation maximum by 8 powers of two), at another
    position, and in a batch of 1027 that is cut into two slices (514 + 513 scans) -- the same bits, in the default (f16x3) arithmetic."""
    imgs = fixture_images(C)
    e = engines[C]
    a = torch.from_numpy(imgs[:1]).cuda()
    b = torch.from_numpy(imgs[1:2]).cuda()
    alone = e.leg(a)
    loud = (300.0 * b).contiguous()
    quiet = (b / 300.0).contiguous()
    batch9 = torch.cat([loud, a, b, quiet, loud, b, b, a, quiet]).contiguous()
    out9 = e.leg(batch9)
    assert torch.equal(out9[1:2], alone) and torch.equal(out9[7:8], alone)
    assert torch.equal(out9[2], out9[5]) and torch.equal(out9[0], out9[4]) and torch.equal(out9[3], out9[8])
    assert torch.equal(out9[2:3], e.leg(b))
    big = torch.cat([b.expand(513, -1, -1, -1), a, loud, b.expand(511, -1, -1, -1), a]).contiguous()   # `a`: last of slice 1, last of slice 2
    outb = e.leg(big)
    assert outb.shape[0] == 1027 and torch.equal(outb[513:514], alone) and torch.equal(outb[1026:1027], alone)
    assert torch.equal(outb[514:515], out9[0:1]) and torch.equal(outb[0:1], out9[2:3]) and torch.equal(outb[700:701], out9[2:3])
    # the scaled scans are still computed to fp32-equivalent accuracy relative to their own range (their own scales apply)
    ref = e.leg(b)
    assert torch.equal(e.leg(loud), out9[0:1])
    assert torch.isfinite(out9).all() and float(out9[0].abs().max()) > 10 * float(ref.abs().max())


PRECISIONS = ("f32", "f16x3")   # arithmetic of the Delta head contractions (ovn_set_head_precision)
ACT_TOL = {"f32": 2e-5, "f16x3": 2e-5}
DEFAULT_HEAD = "f16x3"


def _check_heads(eng, fv, pairs, w, oracle_cache=None):
    fv4 = fv.reshape(-1, 1, 360, 128).astype(np.float64)
    ov, yaw, lg, corr = O.heads_forward(fv4[pairs[:, 0]], fv4[pairs[:, 1]], w)
    out = None
    for mode in PRECISIONS:
        eng.set_head_precision(mode)
        try:
            out = _check_heads_mode(eng, fv, pairs, (ov, yaw, lg, corr), mode)
        finally:
            eng.set_head_precision(DEFAULT_HEAD)
    return out


def _check_heads_mode(eng, fv, pairs, oracle, mode):
    ov, yaw, lg, corr = oracle
    fl = torch.from_numpy(np.ascontiguousarray(fv)).cuda()
    r = eng.heads(fl, fl, lidx=pairs[:, 0], ridx=pairs[:, 1], want_logit=True, want_corr=True)
    torch.cuda.synchronize()
    g_ov, g_yaw = r["overlap"].cpu().numpy(), r["yaw"].cpu().numpy()
    g_lg, g_corr = r["logit"].cpu().numpy(), r["corr"].cpu().numpy()
    assert _rel(g_corr, corr) < 2e-5, "corr vector rel err %.3g" % _rel(g_corr, corr)
    srt = np.sort(corr, axis=1)
    with np.errstate(all="ignore"):
        gap = np.nan_to_num((srt[:, -1] - srt[:, -2]) / np.abs(srt[:, -1]))
    bad = (g_yaw != yaw)
    assert not np.any(bad & (gap > 1e-5)), "yaw bins differ: gpu %s oracle %s gaps %s" % (g_yaw[bad], yaw[bad], gap[bad])
    # the yaw the kernel reports is the first argmax of the corr vector it reports
    assert np.array_equal(g_yaw, 180 - np.argmax(g_corr, axis=1))
    assert np.all(np.abs(g_lg - lg) <= 1e-3 * (1 + np.abs(lg))), "[%s] logit: gpu %s oracle %s" % (mode, g_lg, lg)
    assert np.max(np.abs(g_ov - ov)) <= 1e-4, "[%s] overlap: max err %.3g" % (mode, np.max(np.abs(g_ov - ov)))
    print("[%s] max |d overlap| %.3g  max |d logit| %.3g" % (mode, np.max(np.abs(g_ov - ov)), np.max(np.abs(g_lg - lg))))
    return g_ov, g_yaw, g_lg


@pytest.mark.parametrize("C", [1, 4, 5])
def test_heads_on_real_features(engines, fixture_images, C):
    imgs = np.concatenate([fixture_images(C), S.candidate_images(6, C, seed=7)[2:]])
    w = S.make_test_weights(C, seed=0)
    fv = O.leg_forward(imgs, w, CFG, np.float32).reshape(-1, 360, 128)  # same fp32 inputs to both sides
    pairs = np.array([[0, 1], [1, 0], [0, 0], [2, 0], [3, 1], [4, 5], [5, 2], [1, 1]])
    _check_heads(engines[C], fv, pairs, w)


def test_head_intermediates_against_oracle(engines):
    """c_conv2 / c_conv3 activations of the fused Delta kernel (asymmetric random features so a swapped
    l/r role or a transposed tile cannot pass)."""
    rng = np.random.default_rng(11)
    fv = np.maximum(rng.normal(0.3, 1.0, size=(3, 360, 128)), 0).astype(np.float32)
    w = S.make_test_weights(4, seed=0)
    e = engines[4]
    fl = torch.from_numpy(fv).cuda()
    pairs = np.array([[0, 1], [2, 0]])
    inters = []
    for p in range(2):
        l = fv[pairs[p, 0]].reshape(1, 1, 360, 128).astype(np.float64)
        r = fv[pairs[p, 1]].reshape(1, 1, 360, 128).astype(np.float64)
        inters.append(O.delta_head_forward(l, r, w, return_intermediates=True)[2])
    for mode in PRECISIONS:
        e.set_head_precision(mode)
        try:
            e.heads(fl, fl, lidx=pairs[:, 0], ridx=pairs[:, 1])
            o2, o3 = e.debug_head_activations(2)
        finally:
            e.set_head_precision(DEFAULT_HEAD)
        for p in range(2):
            e2, e3 = _rel(o2[p].cpu().numpy(), inters[p]["o2"]), _rel(o3[p].cpu().numpy(), inters[p]["o3"])
            print("[%s] pair %d: c_conv2 rel err %.3g, c_conv3 rel err %.3g" % (mode, p, e2, e3))
            assert e2 < ACT_TOL[mode], "[%s] c_conv2 output, pair %d: %.3g" % (mode, p, e2)
            assert e3 < ACT_TOL[mode], "[%s] c_conv3 output, pair %d: %.3g" % (mode, p, e3)


def test_heads_random_features_many_pairs(engines):
    rng = np.random.default_rng(21)
    fv = np.maximum(rng.normal(0.2, 1.0, size=(9, 360, 128)), 0).astype(np.float32)
    fv[3] = np.roll(fv[0], 25, axis=0)      # rolled copy -> known yaw
    fv[4] *= 0                               # all-zero feature volume (dead scan)
    pairs = np.array([[i, j] for i in range(9) for j in (0, 4, 7)][:20])
    ov, yaw, lg = _check_heads(engines[4], fv, pairs, S.make_test_weights(4, seed=0))
    # self pair -> yaw 0; rolled-by-25 left vs original right -> argmax 205 -> yaw -25
    k = [tuple(p) for p in pairs.tolist()]
    assert yaw[k.index((0, 0))] == 0 and yaw[k.index((3, 0))] == -25
    # zero candidate vs zero query: corr is flat -> first maximum -> bin 0 -> yaw 180
    assert yaw[k.index((4, 4))] == 180


def test_heads_signed_and_large_features(engines):
    """The C ABI takes arbitrary floats, not only ReLU outputs: negative values go through the per-pair shift of the min form
    (|l - r| = l' + r' - 2 min(l', r'), l' = l - min), large magnitudes through the per-pair power-of-two scales."""
    rng = np.random.default_rng(23)
    fv = rng.normal(0.0, 1.0, size=(6, 360, 128)).astype(np.float32)       # signed
    fv[2] *= 300.0                                                           # one volume 300 x larger than its partners
    fv[5] = -np.abs(fv[5])                                                   # all negative
    pairs = np.array([[i, j] for i in range(6) for j in (0, 2, 5)])
    _check_heads(engines[4], fv, pairs, S.make_test_weights(4, seed=0))


def test_one_vs_n_equals_indexed_pairs_and_is_deterministic(engines):
    """1-vs-N sweep (ridx NULL -> query 0, lidx NULL -> identity) == the general indexed form, bit for bit,
    across two runs and across the 2048-pair chunk boundary."""
    rng = np.random.default_rng(31)
    base = np.maximum(rng.normal(0.2, 1.0, size=(5, 360, 128)), 0).astype(np.float32)
    n = 2051
    cands = torch.from_numpy(base).cuda()[torch.arange(n) % 5].contiguous()
    query = torch.from_numpy(base[1:2]).cuda().contiguous()
    e = engines[4]
    a = e.heads(cands, query, want_logit=True)
    b = e.heads(cands, query, want_logit=True)
    assert torch.equal(a["logit"], b["logit"]) and torch.equal(a["yaw"], b["yaw"])
    # (the query goes BEHIND the candidates: a candidate's slot in the left pool -- which the f16x3 kernel's summation order
    # follows -- is then the same in both forms)
    allf = torch.cat([cands, query])
    c = e.heads(allf, allf, lidx=np.arange(n), ridx=np.full(n, n, np.int64), want_logit=True)
    assert torch.equal(a["logit"], c["logit"]) and torch.equal(a["yaw"], c["yaw"]) and torch.equal(a["overlap"], c["overlap"])
    # periodic inputs -> periodic outputs.  The f16x3 Delta kernel rotates its K walk with the workgroup index (L2
    # locality), so the same pair at another batch position sees its rounding errors summed in another order: ~1e-6 on
    # the logit; the fp32 mode keeps one fixed order and is bit-identical at every position.
    lg = a["logit"].cpu().numpy()
    yw = a["yaw"].cpu().numpy()
    for r in range(5):
        assert np.all(np.abs(lg[r::5] - lg[r]) <= 2e-5 * (1 + abs(lg[r]))) and np.all(yw[r::5] == yw[r])
    e.set_head_precision("f32")
    try:
        f = e.heads(cands, query, want_logit=True)["logit"].cpu().numpy()
    finally:
        e.set_head_precision(DEFAULT_HEAD)
    for r in range(5):
        assert np.all(f[r::5] == f[r])
    assert e.heads(cands[:0], query)["overlap"].shape == (0,)


@pytest.mark.parametrize("scan", [0, 1])
def test_projection_against_reference_golden(engines, fixture_npz, scan):
    from overlapnet_amd import preprocess as P
    pts = fixture_npz["points_%d" % scan]
    r = P.project_scans([pts], engine=engines[4], want=("range", "vertex", "intensity", "idx", "normal"),
                        stacked_flags=(True, True, True))
    rng = r["range"][0].cpu().numpy()
    ref = fixture_npz["range_%d" % scan]
    diff = rng != ref
    print("scan %d: %d range pixels differ from the reference-generated image" % (scan, diff.sum()))
    assert diff.sum() == 0, "%d range pixels differ from the reference" % diff.sum()
    same = ~diff
    assert np.array_equal(r["intensity"][0].cpu().numpy()[same], fixture_npz["intensity_%d" % scan][same])
    assert np.array_equal(r["idx"][0].cpu().numpy()[same], fixture_npz["idx_%d" % scan][same])
    nrm = r["normal"][0].cpu().numpy()
    ndiff = np.any(nrm != fixture_npz["normal_%d" % scan], axis=-1)
    # a differing range pixel can disturb its own normal and its left / upper neighbour's
    assert ndiff.sum() <= 3 * diff.sum(), "%d normal pixels differ (range diffs: %d)" % (ndiff.sum(), diff.sum())
    stk = r["stacked"][0].cpu().numpy()
    assert np.array_equal(stk[..., 0], rng) and np.array_equal(stk[..., 1:4], nrm)
    assert np.array_equal(stk[..., 4], r["intensity"][0].cpu().numpy())
    vtx = r["vertex"][0].cpu().numpy()
    assert np.all(vtx[rng > 0][:, 3] == 1) and np.all(vtx[rng <= 0] == -1)
    # stand-alone normal entry point == fused one
    n2 = engines[4].normals(r["range"], r["vertex"])[0].cpu().numpy()
    assert np.array_equal(n2, nrm)


def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_projection_on_24_transformed_clouds_against_the_reference(engines, fixture_npz):
    """the HIP projection against outputs of the reference's OWN range_projection + gen_normal_map on 24 rotated /
    translated / tilted copies of the two scans (tests/golden/preprocess_transformed.npz; clouds 0-11 are the ones bench.py's
    fullstack leg feeds).  Gate: 0 differing RANGE pixels on every cloud (rounds 1-3: 12); the index image may differ only where
    two points of a pixel have bit-identical minimal depth (the reference's unstable argsort, utils.py:108, picks either; lower
    index here): 10 pixels of 1.38 M; every image equals the CPU oracle's (same tie rule) bit for bit, and on the clouds without
    such a pixel the intensity and normal images carry the reference's hashes."""
    from overlapnet_amd import preprocess as P
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "preprocess_transformed.npz"))
    clouds = [S.transformed_cloud(fixture_npz, i) for i in range(S.N_TRANSFORMED)]
    r = P.project_scans(clouds, engine=engines[4], want=("range", "vertex", "intensity", "idx", "normal"))
    out = {k: v.cpu().numpy() for k, v in r.items()}
    range_diffs, tie_diffs = [], []
    for i, pts in enumerate(clouds):
        assert _sha(pts) == str(g["sha_cloud_%d" % i])
        gi = g["idx_%d" % i]
        x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
        depth = np.sqrt((x * x + y * y) + z * z)
        dk = depth[(depth > 0) & (depth < 50)]
        g_rng = np.where(gi >= 0, dk[np.maximum(gi, 0)], np.float32(-1))
        range_diffs.append(int((out["range"][i] != g_rng).sum()))
        bad = np.argwhere(out["idx"][i] != gi)
        tie_diffs.append(len(bad))
        for (a, b) in bad:
            assert dk[out["idx"][i][a, b]] == dk[gi[a, b]], "cloud %d pixel (%d, %d): different winner that is not a depth tie" % (i, a, b)
        o_rng, o_vtx, o_int, o_idx = O.range_projection(pts)
        assert np.array_equal(out["range"][i], o_rng) and np.array_equal(out["idx"][i], o_idx)
        assert np.array_equal(out["intensity"][i], o_int) and np.array_equal(out["vertex"][i], o_vtx)
        assert np.array_equal(out["normal"][i], O.gen_normal_map(o_rng, o_vtx))
        if not len(bad):
            assert _sha(out["range"][i]) == str(g["sha_range_%d" % i]) and _sha(out["intensity"][i]) == str(g["sha_intensity_%d" % i])
            assert _sha(out["normal"][i]) == str(g["sha_normal_%d" % i])
    print("differing range pixels per cloud vs the reference:", range_diffs, "| index pixels (exact depth ties):", tie_diffs)
    assert sum(range_diffs) == 0 and sum(tie_diffs) == 10


def test_projection_other_geometries_against_the_reference(engines, fixture_npz):
    """`ovn_project` with the reference's other keyword arguments (32 x 1800, 128 x 2048, 64 x 1024 images, other fields of view and
    maximum ranges; a +-60 degree field of view on a pitched cloud: arcsin's second branch) against outputs of the reference's own
    code: identical range images, every image equal to the CPU oracle's."""
    from overlapnet_amd import preprocess as P
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "preprocess_transformed.npz"))
    for k, (ci, _, H, W, up, down, mr) in enumerate(S.GEOMETRY_CASES):
        pts = S.geometry_cloud(fixture_npz, k)
        r = P.project_scans([pts], engine=engines[4], proj_H=H, proj_W=W, fov_up=up, fov_down=down, max_range=mr,
                            want=("range", "vertex", "intensity", "idx", "normal"))
        out = {n: v[0].cpu().numpy() for n, v in r.items()}
        assert out["range"].shape == (H, W) and _sha(out["range"]) == str(g["geo_sha_range_%d" % k]), k
        o_rng, o_vtx, o_int, o_idx = O.range_projection(pts, fov_up=up, fov_down=down, proj_H=H, proj_W=W, max_range=mr)
        assert np.array_equal(out["range"], o_rng) and np.array_equal(out["idx"], o_idx) and np.array_equal(out["intensity"], o_int)
        assert np.array_equal(out["normal"], O.gen_normal_map(o_rng, o_vtx, H, W))
        if np.array_equal(out["idx"], g["geo_idx_%d" % k]):
            assert _sha(out["intensity"]) == str(g["geo_sha_intensity_%d" % k]) and _sha(out["normal"]) == str(g["geo_sha_normal_%d" % k])


def test_projection_angles_have_numpys_float32_bits(engines, fixture_npz):
    """The two angle functions of utils.py:86-87 on the GPU (csrc/svml_f32.h) against NumPy's own float32 results: the committed
    vectors (arctan2 directly; arcsin through points built to hit given sines) and, through the pinned CPU restatement, every
    point of two transformed clouds plus 2 M random points over a wide exponent range, quadrant edges and zero coordinates."""
    from oracle import build_oracle as B
    e = engines[4]
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "svml_f32_vectors.npz")) as z:
        vx, vy, vat = z["x"], z["y"], z["atan2"]
    ok = np.isfinite(vx) & np.isfinite(vy)
    vx, vy, vat = vx[ok], vy[ok], vat[ok]
    pts = np.zeros((len(vx), 4), np.float32)
    pts[:, 0], pts[:, 1] = vx, vy
    yaw, _, pix = (t.cpu().numpy() for t in e.projection_angles(torch.from_numpy(pts).cuda(), max_range=3.0e38))
    keep = pix >= 0                        # (0,0,0) and overflowing depths are dropped by the range filter before any angle
    assert keep.sum() > 0.75 * len(vx)
    assert np.array_equal(yaw[keep].view(np.uint32), (-vat[keep]).view(np.uint32))
    rng = np.random.default_rng(99)
    n = 2_000_000
    rnd = np.zeros((n, 4), np.float32)
    rnd[:, :3] = (rng.uniform(-1, 1, (n, 3)) * np.exp(rng.uniform(-30, 30, (n, 1)))).astype(np.float32)
    rnd[::7, 2] *= np.float32(30.0)        # steep points: the |sin| >= 0.5 branch of arcsin
    rnd[::1001, 0] = 0
    rnd[::1003, 1] = 0
    rnd[::1007, 2] = 0
    rnd[5::2002, 0] = -0.0
    for cloud in (S.transformed_cloud(fixture_npz, 9), S.transformed_cloud(fixture_npz, 21), rnd):
        yaw, pitch, pix = (t.cpu().numpy() for t in e.projection_angles(torch.from_numpy(np.ascontiguousarray(cloud)).cuda(), max_range=1.0e30))
        x, y, zz = cloud[:, 0], cloud[:, 1], cloud[:, 2]
        depth = np.sqrt((x * x + y * y) + zz * zz)
        keep = (depth > 0) & (depth < np.float32(1.0e30))
        assert np.array_equal(keep, pix >= 0)
        o_yaw = -B.svml_arctan2(y[keep], x[keep])
        o_pitch = B.svml_arcsin(zz[keep] / depth[keep])
        assert np.array_equal(yaw[keep].view(np.uint32), o_yaw.view(np.uint32))
        assert np.array_equal(pitch[keep].view(np.uint32), o_pitch.view(np.uint32))
        if cloud is rnd:
            assert (np.abs(zz[keep] / depth[keep]) >= 0.5).sum() > 100000      # both arcsin branches


def test_projection_trig_rounded_mode(engines, fixture_npz):
    """`ovn_set_projection_trig(1)` (config['projection_trig'] = 'rounded'): the correctly rounded float32 arctan2 / arcsin -- the
    reference's utils.py:86-87 as NumPy evaluates it where its float32 loops call a correctly rounded libm -- bit for bit against
    float64 NumPy rounded once, the whole projection against the oracle in its 'f64' trig mode, and back to the default."""
    from overlapnet_amd import preprocess as P
    e = engines[4]
    cloud = S.transformed_cloud(fixture_npz, 9)
    dpts = torch.from_numpy(np.ascontiguousarray(cloud)).cuda()
    yaw0, pitch0, _ = (t.cpu().numpy() for t in e.projection_angles(dpts))
    e.set_projection_trig("rounded")
    try:
        yaw, pitch, pix = (t.cpu().numpy() for t in e.projection_angles(dpts))
        x, y, z = (cloud[:, k].astype(np.float32) for k in range(3))
        depth = np.sqrt((x * x + y * y) + z * z)
        keep = (depth > 0) & (depth < np.float32(50.0))
        assert np.array_equal(keep, pix >= 0)
        o_yaw = (-np.arctan2(y[keep].astype(np.float64), x[keep].astype(np.float64))).astype(np.float32)
        o_pitch = np.arcsin((z[keep] / depth[keep]).astype(np.float64)).astype(np.float32)
        assert np.array_equal(yaw[keep].view(np.uint32), o_yaw.view(np.uint32))
        assert np.array_equal(pitch[keep].view(np.uint32), o_pitch.view(np.uint32))
        differs = float(np.mean(yaw[keep].view(np.uint32) != yaw0[keep].view(np.uint32)))
        assert 0.1 < differs < 0.7            # the SVML kernels are 1-4 ulp functions: a third of the angles differ in the last bit
        r = P.project_scans([cloud], engine=e, want=("range", "idx"))
        o_rng, _, _, o_idx = O.range_projection(cloud, trig="f64")
        assert np.array_equal(r["range"][0].cpu().numpy(), o_rng)
        ties = r["idx"][0].cpu().numpy() != o_idx
        assert ties.sum() <= 4                # (bit-identical depths in one pixel: the reference's unstable sort, see the test above)
    finally:
        e.set_projection_trig("numpy_avx512")
    yaw1, _, _ = (t.cpu().numpy() for t in e.projection_angles(dpts))
    assert np.array_equal(yaw1.view(np.uint32), yaw0.view(np.uint32))
    with pytest.raises(ValueError):
        e.set_projection_trig("libm")


def test_projection_batch_ragged_and_edge_cases(engines, fixture_npz):
    from overlapnet_amd import preprocess as P
    p0, p1 = fixture_npz["points_0"], fixture_npz["points_1"]
    empty = np.zeros((0, 4), np.float32)
    far = np.array([[100, 0, 0, 1], [0, 0, 0, 1]], np.float32)
    tie = np.array([[5, 0, 0, 0.3], [5, 0, 0, 0.7], [60, 1, 1, 0.1], [2, 0, 0, 0.9]], np.float32)
    r = P.project_scans([p0, empty, p1[:1000], far, tie, p1], engine=engines[4],
                        want=("range", "intensity", "idx", "normal"))
    rng = r["range"].cpu().numpy()
    single0 = P.project_scans([p0], engine=engines[4], want=("range", "normal", "idx"))
    assert np.array_equal(rng[0], single0["range"][0].cpu().numpy())
    assert np.array_equal(r["normal"][0].cpu().numpy(), single0["normal"][0].cpu().numpy())
    assert np.array_equal(r["idx"][0].cpu().numpy(), single0["idx"][0].cpu().numpy())
    assert np.all(rng[1] == -1) and np.all(rng[3] == -1) and np.all(r["idx"][1].cpu().numpy() == -1)
    o_rng, _, o_int, o_idx = O.range_projection(p1[:1000])
    assert np.array_equal(rng[2], o_rng) and np.array_equal(r["idx"][2].cpu().numpy(), o_idx)
    t_rng, _, t_int, t_idx = O.range_projection(tie)
    assert np.array_equal(rng[4], t_rng) and np.array_equal(r["intensity"][4].cpu().numpy(), t_int)
    assert np.array_equal(r["idx"][4].cpu().numpy(), t_idx)
    assert (rng[4] > 0).sum() == 1 and rng[4].max() == 2.0
    assert np.array_equal(rng[5], P.project_scans([p1], engine=engines[4], want=("range",))["range"][0].cpu().numpy())


def test_infer_class_end_to_end(tmp_path, fixture_npz):
    """The drop-in `Infer` on .npy inputs laid out like demo1 writes them, against the oracle wired the way
    the reference wires pair roles (infer.py:140-152,188-190,224-225)."""
    from overlapnet_amd.infer import Infer
    seq = tmp_path / "data" / "07"
    for sub in ("depth", "normal", "intensity"):
        os.makedirs(seq / sub)
    imgs4 = []
    for i in range(4):
        s = i % 2
        shift = 40 * (i // 2)
        d = np.roll(fixture_npz["range_%d" % s], shift, axis=1)
        nm = np.roll(fixture_npz["normal_%d" % s], shift, axis=1)
        np.save(seq / "depth" / ("%06d.npy" % i), d)
        np.save(seq / "normal" / ("%06d.npy" % i), nm)
        imgs4.append(S.stack(d, nm, None, (True, True, False)))
    imgs4 = np.stack(imgs4)
    cfg = {"model": dict(S.REFERENCE_MODEL_CFG, inputShape=[64, 900]), "infer_seqs": "07",
           "data_root_folder": str(tmp_path / "data"), "use_depth": True, "use_normals": True,
           "use_class_probabilities": False, "use_class_probabilities_pca": False, "use_intensity": False,
           "batch_size": 16, "pretrained_weightsfilename": ""}
    w = S.make_test_weights(4, seed=0)
    inf = Infer(cfg, weights=w)
    assert cfg["model"]["inputShape"] == [64, 900, 4]  # mutated in place like infer.py:78-79
    assert inf.no_input_channels == 4 and inf.batch_size == 16 and inf.network_output_size == 360

    ref_fv = O.leg_forward(imgs4, w, CFG, np.float64)

    def oracle(pairs):
        return O.heads_forward(ref_fv[pairs[:, 0]], ref_fv[pairs[:, 1]], w)

    # infer_one(f1, f2): filenames = [name2, name1]; l = fv[0] = name2, r = fv[1] = name1
    ov, yaw = inf.infer_one("x/000000.bin", "y/000001.bin")
    o_ov, o_yaw, _, _ = oracle(np.array([[1, 0]]))
    assert ov.shape == (1,) and yaw.shape == (1,) and ov.dtype == np.float32
    assert list(inf.filenames) == ["000001", "000000"]
    assert abs(ov[0] - o_ov[0]) < 1e-4 and yaw[0] == o_yaw[0]
    with pytest.raises(Exception, match="only works with .bin"):
        inf.infer_one("a.txt", "b.bin")

    fv = inf.create_feature_volumes(["000000", "000003"])
    assert fv.shape == (2, 1, 360, 128) and _rel(fv[1, 0], ref_fv[3, 0]) < 2e-5

    # infer_multiple: frames fed in order; l = reference frame, r = current frame
    assert inf.infer_multiple(0, []) is None
    r1 = inf.infer_multiple(1, [0])
    assert r1[0].shape == () and r1[1].shape == (1,)        # squeeze -> 0-d when N == 1 (infer.py:197)
    inf.infer_multiple(2, [])
    ov3, yaw3 = inf.infer_multiple(3, [0, 1, 2])
    o_ov, o_yaw, _, _ = oracle(np.array([[0, 3], [1, 3], [2, 3]]))
    assert ov3.shape == (3,) and np.max(np.abs(ov3 - o_ov)) < 1e-4 and np.array_equal(yaw3, o_yaw)
    assert len(inf.feature_volumes) == 4 and inf.feature_volumes[0].shape == (1, 360, 128)

    # extension: the same sweep with demo3's decision (demo3_lcd.py:117-120) taken on the device
    from overlapnet_amd import lcd
    inf2 = Infer(dict(cfg, model=dict(S.REFERENCE_MODEL_CFG, inputShape=[64, 900])), weights=w)
    for i in range(3):
        assert inf2.infer_best_match(i, []) is None
    want = lcd.decide([0, 1, 2], ov3, yaw3, 0.0)
    got = inf2.infer_best_match(3, [0, 1, 2], overlap_thres=0.0)
    assert got[0] == want[0] and got[2] == want[2] and got[1] == np.float32(want[1])
    assert len(inf2.feature_volumes) == 4
    assert inf2.infer_best_match(4 - 1, [2, 0], overlap_thres=2.0) is None    # nothing above the threshold

    # infer_multiple_vs_multiple: l = second_idxs, r = first_idxs
    names = ["000000", "000001.bin", "d/000003.bin"]
    ovm, yawm = inf.infer_multiple_vs_multiple(names, [0, 1, 2], [2, 1, 1])
    o_ov, o_yaw, _, _ = oracle(np.array([[3, 0], [1, 1], [1, 3]]))
    assert np.max(np.abs(ovm - o_ov)) < 1e-4 and np.array_equal(yawm, o_yaw)
    assert inf.feature_volumes.shape == (3, 1, 360, 128)
    assert inf.infer_multiple_vs_multiple(names, [], []) is None
    with pytest.raises(Exception, match="same size"):
        inf.infer_multiple_vs_multiple(names, [0], [])
    cfg_bad = dict(cfg, model=dict(S.REFERENCE_MODEL_CFG, inputShape=[64, 900], legsType="360OutputkLegs_smaller"))
    with pytest.raises(AttributeError):
        Infer(cfg_bad, weights=w)


def test_best_match_on_device(engines):
    """ovn_best_match == `np.argmax` + threshold of demo3_lcd.py:117-120 (first maximum wins), bit-exact."""
    from overlapnet_amd.engine import decode_match
    e = engines[4]
    rng = np.random.default_rng(5)
    for n in (1, 2, 63, 64, 65, 1023, 1024, 1025, 4097, 100003):
        ov = (rng.integers(0, 97, n) / 97.0).astype(np.float32)            # plenty of exact ties
        yaw = rng.integers(-179, 181, n).astype(np.int32)
        ids = rng.permutation(n).astype(np.int32)
        t_ov, t_yaw, t_ids = (torch.from_numpy(a).cuda() for a in (ov, yaw, ids))
        k = int(np.argmax(ov))
        for thr in (0.3, 0.999):
            want = (k, float(ov[k]), int(yaw[k])) if ov[k] > thr else None
            assert decode_match(e.best_match(t_ov, t_yaw, thr)) == want
            assert decode_match(e.best_match(t_ov, t_yaw, thr, host=True)) == want      # record written into pinned host memory
            rec = e.best_match(t_ov, t_yaw, thr, ids=t_ids).cpu().numpy()
            assert rec[0] == ids[k] and rec[1:2].view(np.float32)[0] == ov[k] and rec[2] == yaw[k] and rec[3] == int(ov[k] > thr)
        assert e.best_match(t_ov, t_yaw, 0.0, index_offset=1000).cpu().numpy()[0] == 1000 + k
        assert e.best_match(t_ov, None, 0.0).cpu().numpy()[2] == 0           # yaw optional
    # NaN never wins; all-NaN and empty inputs report "nothing"
    ov = np.array([np.nan, 0.4, np.nan, 0.7, 0.7], np.float32)
    rec = e.best_match(torch.from_numpy(ov).cuda(), torch.arange(5, dtype=torch.int32).cuda(), 0.3).cpu().numpy()
    assert rec.tolist()[0] == 3 and rec[3] == 1
    assert e.best_match(torch.full((7,), float("nan")).cuda(), None, 0.3).cpu().tolist() == [-1, 0, 0, 0]
    assert decode_match(e.best_match(torch.empty(0).cuda(), None, 0.3)) is None
    with pytest.raises(Exception, match="yaw must be"):
        e.best_match(torch.zeros(4).cuda(), torch.zeros(3, dtype=torch.int32).cuda())


def test_spectral_correlation_head(engines):
    """Spectral form of the correlation head (cached spectra) == direct form == fp64 oracle."""
    rng = np.random.default_rng(41)
    fv = np.maximum(rng.normal(0.2, 1.0, size=(7, 360, 128)), 0).astype(np.float32)
    fv[3] = np.roll(fv[0], 25, axis=0)
    fv[4] *= 0
    e = engines[4]
    ft = torch.from_numpy(fv).cuda()
    spec = e.spectrum(ft)
    assert tuple(spec.shape) == (7, 128, 368)
    sp = spec.cpu().numpy()
    # spectrum == rfft along the column axis, Re at [0..180], Im at [184..364], padding zero
    ref = np.fft.rfft(fv.astype(np.float64), axis=1)            # (7, 181, 128)
    assert _rel(sp[:, :, :181], np.transpose(ref.real, (0, 2, 1))) < 5e-6
    assert _rel(sp[:, :, 184:365], np.transpose(ref.imag, (0, 2, 1))) < 5e-6
    assert np.all(sp[:, :, 181:184] == 0) and np.all(sp[:, :, 365:] == 0)
    # big calls use one workgroup per (scan, channel half), small ones three: same spectrum bit for bit; fp32 head mode = the
    # fp32 conv-kernel form of the same transform
    big = e.spectrum(ft[torch.arange(70) % 7].contiguous())
    assert torch.equal(big[:7], spec) and torch.equal(big[63:70], spec)
    e.set_head_precision("f32")
    try:
        sp32 = e.spectrum(ft).cpu().numpy()
    finally:
        e.set_head_precision(DEFAULT_HEAD)
    assert _rel(sp32, sp) < 5e-6
    pairs = np.array([[i, j] for i in range(7) for j in (0, 2, 4)])
    r = e.corr_head_spectral(spec, spec, lidx=pairs[:, 0], ridx=pairs[:, 1], want_corr=True)
    d = e.corr_head(ft, ft, lidx=pairs[:, 0], ridx=pairs[:, 1], want_corr=True)
    fv4 = fv.reshape(-1, 1, 360, 128).astype(np.float64)
    corr = O.correlation_head_forward(fv4[pairs[:, 0]], fv4[pairs[:, 1]])
    g = r["corr"].cpu().numpy()
    scale = np.max(np.abs(corr)) + 1e-30
    assert np.max(np.abs(g - corr)) / scale < 2e-5, "spectral corr rel err %.3g" % (np.max(np.abs(g - corr)) / scale)
    yaw = O.yaw_from_orientation(corr)
    srt = np.sort(corr, axis=1)
    with np.errstate(all="ignore"):
        gap = np.nan_to_num((srt[:, -1] - srt[:, -2]) / np.abs(srt[:, -1]))
    g_yaw = r["yaw"].cpu().numpy()
    bad = g_yaw != yaw
    assert not np.any(bad & (gap > 1e-5)), (g_yaw[bad], yaw[bad], gap[bad])
    assert np.array_equal(g_yaw[gap > 1e-5], d["yaw"].cpu().numpy()[gap > 1e-5])
    k = [tuple(p) for p in pairs.tolist()]
    assert g_yaw[k.index((0, 0))] == 0 and g_yaw[k.index((3, 0))] == -25
    # 1-vs-N convenience form == indexed form
    a = e.corr_head_spectral(spec, spec[2:3].contiguous())
    b = e.corr_head_spectral(spec, spec, lidx=np.arange(7), ridx=np.full(7, 2))
    assert torch.equal(a["yaw"], b["yaw"])


def test_infer_with_intensity_channel_and_missing_files(tmp_path, fixture_npz):
    """C = 5 (depth + normals + intensity, ImagePairOverlapOrientationSequence.py:143-207 channel order) through `Infer`,
    and the reference's error behaviour for unreadable inputs (:148-149,159-160)."""
    from overlapnet_amd.infer import Infer
    seq = tmp_path / "data" / "s"
    for sub in ("depth", "normal", "intensity"):
        os.makedirs(seq / sub)
    imgs = []
    for i in range(2):
        np.save(seq / "depth" / ("%06d.npy" % i), fixture_npz["range_%d" % i])
        np.save(seq / "normal" / ("%06d.npy" % i), fixture_npz["normal_%d" % i])
        np.save(seq / "intensity" / ("%06d.npy" % i), fixture_npz["intensity_%d" % i])
        imgs.append(S.stack(fixture_npz["range_%d" % i], fixture_npz["normal_%d" % i], fixture_npz["intensity_%d" % i],
                            (True, True, True)))
    cfg = {"model": dict(S.REFERENCE_MODEL_CFG, inputShape=[64, 900]), "infer_seqs": "s",
           "data_root_folder": str(tmp_path / "data"), "use_depth": True, "use_normals": True,
           "use_class_probabilities": False, "use_class_probabilities_pca": False, "use_intensity": True,
           "batch_size": 1, "pretrained_weightsfilename": ""}
    w = S.make_test_weights(5, seed=0)
    inf = Infer(cfg, weights=w)
    assert inf.no_input_channels == 5 and cfg["model"]["inputShape"] == [64, 900, 5]
    fv = inf.create_feature_volumes(["000000", "000001"])       # batch_size 1 -> two leg launches
    ref = O.leg_forward(np.stack(imgs), w, CFG, np.float64)
    assert _rel(fv, ref) < 2e-5
    ov, yaw = inf.infer_one("a/000000.bin", "b/000001.bin")
    o_ov, o_yaw, _, _ = O.heads_forward(ref[[1]], ref[[0]], w)
    assert abs(ov[0] - o_ov[0]) < 1e-4 and yaw[0] == o_yaw[0]
    with pytest.raises(Exception, match="Could not read depth image"):
        inf.create_feature_volumes(["000007"])
    os.remove(seq / "normal" / "000001.npy")
    with pytest.raises(Exception, match="Could not read normal image"):
        inf.create_feature_volumes(["000001"])
    cfg2 = dict(cfg, infer_seqs="nowhere", model=dict(S.REFERENCE_MODEL_CFG, inputShape=[64, 900]))
    with pytest.raises(Exception, match="first generate preprocessed"):
        Infer(cfg2, weights=w).infer_one("x/000000.bin", "y/000001.bin")
    # a KeyError for a missing use_* key, exactly like infer.py:62-73
    cfg3 = {k: v for k, v in cfg.items() if k != "use_intensity"}
    cfg3["model"] = dict(S.REFERENCE_MODEL_CFG, inputShape=[64, 900])
    with pytest.raises(KeyError):
        Infer(cfg3, weights=w)


def test_yaw_argmax_on_16384_seeded_pairs(engines):
    """Exact yaw bin on 16384 seeded pairs (>= 1e4), both forms of the correlation head, against the fp64
    correlation evaluated per channel with numpy's FFT -- the same sum as the oracle's wrapped Gram diagonals, checked against
    `O.correlation_head_forward` on a sample -- with the contract's near-tie rule (top-2 gap below 1e-5 relative may differ)."""
    n, npool = 16384, 512
    rng = np.random.default_rng(2024)
    pool = np.maximum(rng.normal(0.15, 1.0, size=(npool, 360, 128)), 0).astype(np.float32)
    pool[7] = np.roll(pool[3], 111, axis=0)
    li = rng.integers(0, npool, size=n)
    ri = rng.integers(0, npool, size=n)
    F = np.fft.rfft(pool.astype(np.float64), axis=1)                          # (npool, 181, 128)

    def corr64(a, b):    # corr[k] = sum_{j,c} l[(k + j + 180) mod 360, c] r[j, c]
        c = np.fft.irfft((F[a] * np.conj(F[b])).sum(axis=2), n=360, axis=1)
        return np.roll(c, -180, axis=1)

    sample = np.arange(0, n, n // 6)
    lit = O.correlation_head_forward(pool[li[sample]][:, None].astype(np.float64), pool[ri[sample]][:, None].astype(np.float64))
    assert np.max(np.abs(corr64(li[sample], ri[sample]) - lit)) <= 1e-9 * np.max(np.abs(lit))
    e = engines[4]
    pt = torch.from_numpy(pool).cuda()
    spec = e.spectrum(pt)
    g_spec = e.corr_head_spectral(spec, spec, lidx=li, ridx=ri)["yaw"].cpu().numpy()
    g_dir = e.corr_head(pt, pt, lidx=li, ridx=ri)["yaw"].cpu().numpy()
    nbad = {"spectral": 0, "direct": 0}
    for s0 in range(0, n, 2048):
        sl = slice(s0, s0 + 2048)
        c = corr64(li[sl], ri[sl])
        yaw = 180 - np.argmax(c, axis=1)
        srt = np.sort(c, axis=1)
        gap = (srt[:, -1] - srt[:, -2]) / np.abs(srt[:, -1])
        for name, g in (("spectral", g_spec[sl]), ("direct", g_dir[sl])):
            bad = g != yaw
            nbad[name] += int(bad.sum())
            assert not np.any(bad & (gap > 1e-5)), (name, g[bad], yaw[bad], gap[bad])
    print("yaw over %d pairs: %s bins differ from the fp64 argmax, all with a top-2 gap <= 1e-5" % (n, nbad))
    assert nbad["spectral"] <= 16 and nbad["direct"] <= 16


def test_full_size_sweep_properties(engines):
    """BASELINE-sized 1-vs-1024 sweep, size-independent properties: candidate k = query rolled by k columns must
    come back with yaw bin 180 + k (mod 360) -> yaw = -k wrapped, and rolling leaves the overlap logit of the
    SELF pair's neighbourhood finite and deterministic; both correlation forms agree; both heads agree with the
    small-batch call on the same pairs (chunking / batching invariance of the spectral + Delta path)."""
    rng = np.random.default_rng(77)
    q = np.maximum(rng.normal(0.2, 1.0, size=(360, 128)), 0).astype(np.float32)
    shifts = (np.arange(1024) * 7) % 360
    cands = np.stack([np.roll(q, int(s), axis=0) for s in shifts])
    e = engines[4]
    ct = torch.from_numpy(cands).cuda()
    qt = torch.from_numpy(q[None]).cuda()
    spec, qspec = e.spectrum(ct), e.spectrum(qt)
    r = e.heads(ct, qt, spec_l=spec, spec_r=qspec, want_logit=True)
    yaw = r["yaw"].cpu().numpy()
    expect = 180 - ((180 + shifts) % 360)
    assert np.array_equal(yaw, expect)
    d = e.heads(ct, qt, want_logit=True)                      # direct correlation form
    assert np.array_equal(d["yaw"].cpu().numpy(), expect)
    lg = r["logit"].cpu().numpy()
    assert np.all(np.isfinite(lg))
    # same pairs in a small batch: same values up to the position jitter of the rotated K walk
    small = e.heads(ct[100:116].contiguous(), qt, want_logit=True)["logit"].cpu().numpy()
    assert np.all(np.abs(small - lg[100:116]) <= 2e-5 * (1 + np.abs(small)))
    # shift 0 is the self pair: identical features -> |L-R| contains the zero diagonal; oracle check on 3 pairs
    idx = np.array([0, 1, 513])
    o_ov, o_yaw, _, _ = O.heads_forward(cands[idx].reshape(3, 1, 360, 128).astype(np.float64),
                                        np.repeat(q.reshape(1, 1, 360, 128).astype(np.float64), 3, axis=0),
                                        S.make_test_weights(4, seed=0))
    assert np.max(np.abs(r["overlap"].cpu().numpy()[idx] - o_ov)) <= 1e-4 and np.array_equal(yaw[idx], o_yaw)


def test_ground_truth_overlap_yaw_against_reference_golden(fixture_npz):
    """csrc/overlap_gt.hip + ground_truth.py against the mapping the reference's own com_overlap_yaw.py produced
    (tests/golden/make_gt_golden.py): yaw bins and overlaps identical (float64 atan2/asin of the device library vs glibc
    could move a point that sits on a pixel edge; on these 60 pairs none does)."""
    from overlapnet_amd.ground_truth import OverlapGroundTruth, com_overlap_yaw
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "gt_overlap_yaw.npz"))
    scans = [fixture_npz["points_%d" % s] for s in z["scan_of"]]
    gt = OverlapGroundTruth(scans, z["poses"])
    worst = 0.0
    for f in (0, 4, 7, 11):
        m = gt.mapping(f)
        ref = z["mapping_%d" % f]
        assert np.array_equal(m[:, [0, 1, 3]], ref[:, [0, 1, 3]])
        d = np.abs(m[:, 2] - ref[:, 2])
        worst = max(worst, float(d.max()))
        assert d.max() == 0, "frame %d: overlaps differ by %.3g" % (f, d.max())          # measured: identical on all 60 pairs
        assert m[f, 2] == 1.0                                   # a scan overlaps itself completely
    print("ground-truth overlap: max |gpu - reference| = %.3g" % worst)
    # the range image the kernel builds for the untransformed frame equals the float64 oracle's
    e = gt.engine
    pts = torch.from_numpy(np.ascontiguousarray(scans[0], np.float32)).cuda()
    img = e.gt_range_images(pts, torch.tensor([0, pts.shape[0]], dtype=torch.int64).cuda(), pts.shape[0]).cpu().numpy()[0]
    h = np.ones((scans[0].shape[0], 4))
    h[:, :3] = scans[0][:, :3]
    ref_img = O.range_image_f64(h)
    print("untransformed GT range image: %d pixels differ from the float64 oracle" % np.count_nonzero(img != ref_img))
    assert np.count_nonzero(img != ref_img) <= 4
    # drop-in function on .bin files, ragged / empty inputs
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        paths = []
        for i in range(3):
            pth = os.path.join(tmp, "%06d.bin" % i)
            (scans[i][: 1000 * i] if i else scans[0]).astype(np.float32).tofile(pth)     # scan 1: 1000 points, scan 2: 2000
            paths.append(pth)
        m = com_overlap_yaw(paths, z["poses"][:3], 0)
        want = O.com_overlap_yaw([scans[0], scans[1][:1000], scans[2][:2000]], z["poses"][:3], 0)
        assert np.array_equal(m[:, 3], want[:, 3]) and np.max(np.abs(m[:, 2] - want[:, 2])) <= 3.0 / 40000


@pytest.mark.parametrize("pca", [True, False])
def test_infer_with_semantic_channels(tmp_path, fixture_npz, pca):
    """`use_class_probabilities[_pca]` inputs (ImagePairOverlapOrientationSequence.py:165-195): channel order
    depth -> normals -> probabilities (20, or 3 after PCA) -> intensity; C = 25 / 8 with every cue on."""
    from overlapnet_amd.infer import Infer
    k = 3 if pca else 20
    sub = "probability_pca" if pca else "probability"
    seq = tmp_path / "data" / "07"
    for d in ("depth", "normal", "intensity", sub):
        os.makedirs(seq / d)
    rng = np.random.default_rng(11)
    imgs = []
    for i in range(2):
        prob = rng.random((64, 900, k)).astype(np.float32)
        prob /= prob.sum(axis=2, keepdims=True)
        np.save(seq / "depth" / ("%06d.npy" % i), fixture_npz["range_%d" % i])
        np.save(seq / "normal" / ("%06d.npy" % i), fixture_npz["normal_%d" % i])
        np.save(seq / "intensity" / ("%06d.npy" % i), fixture_npz["intensity_%d" % i])
        np.save(seq / sub / ("%06d.npy" % i), prob)
        imgs.append(np.concatenate([fixture_npz["range_%d" % i][..., None], fixture_npz["normal_%d" % i], prob,
                                    fixture_npz["intensity_%d" % i][..., None]], axis=2).astype(np.float32))
    C = 1 + 3 + k + 1
    cfg = {"model": dict(S.REFERENCE_MODEL_CFG, inputShape=[64, 900]), "infer_seqs": "07",
           "data_root_folder": str(tmp_path / "data"), "use_depth": True, "use_normals": True,
           "use_class_probabilities": True, "use_class_probabilities_pca": pca, "use_intensity": True,
           "batch_size": 16, "pretrained_weightsfilename": ""}
    w = W.synthetic_weights(C, CFG, seed=2, kernel_gain=1.3)
    inf = Infer(cfg, weights=w)
    assert inf.no_input_channels == C and cfg["model"]["inputShape"] == [64, 900, C]
    fv = inf.create_feature_volumes(["000000", "000001"])
    ref = O.leg_forward(np.stack(imgs), w, CFG, np.float64)
    assert _rel(fv, ref) < 2e-5
    ov, yaw = inf.infer_one("a/000000.bin", "b/000001.bin")
    o_ov, o_yaw, _, _ = O.heads_forward(ref[[1]], ref[[0]], w)
    assert abs(ov[0] - o_ov[0]) < 1e-4 and yaw[0] == o_yaw[0]


def test_gen_semantic_data_against_reference_golden(tmp_path, fixture_npz):
    """preprocess.gen_semantic_data against a run of the reference's gen_semantic_data.py on the same seeded per-point
    probabilities (tests/golden/make_semantic_golden.py): correspondences = proj_idx with max_range = inf."""
    from overlapnet_amd import preprocess as P
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "semantic_idx.npz"))
    os.makedirs(tmp_path / "scans")
    os.makedirs(tmp_path / "probs")
    os.makedirs(tmp_path / "dst")
    probs = []
    for i in range(2):
        pts = fixture_npz["points_%d" % i]
        pts.astype(np.float32).tofile(tmp_path / "scans" / ("%06d.bin" % i))
        pr = np.random.default_rng(100 + i).random((pts.shape[0], 20)).astype(np.float32)
        pr.tofile(tmp_path / "probs" / ("%06d.label" % i))
        probs.append(pr)
    sem = P.gen_semantic_data(str(tmp_path / "probs"), str(tmp_path / "scans"), str(tmp_path / "dst"))
    for i in range(2):
        idx = z["idx_%d" % i]
        want = np.full((64, 900, 20), -1, np.float32)
        want[idx >= 0] = probs[i][idx[idx >= 0]]
        assert abs(want.astype(np.float64).sum() - float(z["sum_%d" % i])) < 1e-6       # the golden run itself
        diff_px = np.count_nonzero(np.any(sem[i] != want, axis=2))
        assert diff_px <= 8, "scan %d: %d pixels differ" % (i, diff_px)                 # same gate as the projection test
        assert np.array_equal(np.load(tmp_path / "dst" / "semantic" / ("%06d.npy" % i)), sem[i])


def test_small_sweeps_equal_the_same_pairs_in_a_big_sweep(engines):
    """Small sweeps spread a pair's 12 passes over several workgroups; same arithmetic, same result as in a big sweep."""
    e = engines[4]
    rng = np.random.default_rng(77)
    fv = torch.from_numpy(np.maximum(rng.normal(0.2, 1.0, size=(300, 360, 128)), 0).astype(np.float32)).cuda()
    q = fv[7:8].contiguous()
    big = e.heads(fv, q, want_logit=True)
    for n in (1, 5, 21, 100, 255):
        small = e.heads(fv[:n].contiguous(), q, want_logit=True)
        assert torch.equal(small["logit"], big["logit"][:n]) and torch.equal(small["yaw"], big["yaw"][:n]), n


def test_100k_candidate_sweep_properties(engines):
    """BASELINE config '1-vs-100k synthetic candidate pool' on ONE GPU (18.4 GB of feature volumes + 18.8 GB of cached spectra;
    an 8-GPU run gives each rank an eighth of it): candidate k = query rolled by (7k mod 360) columns must come back with
    yaw = 180 - ((180 + shift) mod 360); the 49 chunks of 2048 pairs agree with small calls on the same candidates; the
    on-device decision equals the host argmax."""
    import time
    from overlapnet_amd.engine import decode_match
    e = engines[4]
    N = 100000
    rng = np.random.default_rng(5)
    q = torch.from_numpy(np.maximum(rng.normal(0.2, 1.0, size=(1, 360, 128)), 0).astype(np.float32)).cuda()
    base = torch.cat([torch.roll(q, int(s), dims=1) for s in range(360)])          # every shift once: 66 MB
    shifts = (torch.arange(N, device="cuda") * 7) % 360
    cands = base[shifts].contiguous()                                                # (N, 360, 128): 18.4 GB
    del base
    spec = e.spectrum(cands)
    qspec = e.spectrum(q)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = e.heads(cands, q, spec_l=spec, spec_r=qspec, want_logit=True)
    rec = e.best_match(r["overlap"], r["yaw"], 0.0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("1-vs-100k sweep on one GPU: %.3f s = %.0f pairs/s" % (dt, N / dt))
    yaw = r["yaw"].cpu().numpy()
    sh = shifts.cpu().numpy()
    assert np.array_equal(yaw, 180 - ((180 + sh) % 360))
    lg = r["logit"]
    assert bool(torch.isfinite(lg).all())
    # candidates with the same shift are the same scan: same logit wherever they sit in the pool, up to the position-dependent
    # summation order of the rotated K walks (360 distinct values)
    ref = lg[:360]
    for lo in (360 * 100, N - N % 360 - 360):
        assert torch.allclose(lg[lo:lo + 360], ref, rtol=2e-5, atol=2e-5)
    for lo in (0, 2047, 2048, 51234, N - 5):                                         # chunk edges, middle, tail
        small = e.heads(cands[lo:lo + 5].contiguous(), q, want_logit=True)["logit"]
        assert torch.allclose(small, lg[lo:lo + 5], rtol=2e-5, atol=2e-5)
    ov = r["overlap"].cpu().numpy()
    k = int(np.argmax(ov))
    assert decode_match(rec) == (k, float(ov[k]), int(yaw[k]))
    # 16 pairs sampled over the whole pool against the fp64 oracle: overlap within 1e-4, exact yaw bin
    w = S.make_test_weights(4, seed=0)
    sel = np.sort(np.random.default_rng(11).choice(N, 16, replace=False))
    sel[0], sel[-1] = 1024, N - 1                                                    # first pair of the second chunk, last pair
    fv64 = cands[torch.from_numpy(sel).cuda()].cpu().numpy().reshape(16, 1, 360, 128).astype(np.float64)
    q64 = np.repeat(q.cpu().numpy().reshape(1, 1, 360, 128).astype(np.float64), 16, axis=0)
    o_ov, o_yaw, o_lg, _ = O.heads_forward(fv64, q64, w)
    assert np.max(np.abs(ov[sel] - o_ov)) <= 1e-4 and np.array_equal(yaw[sel], o_yaw)
    assert np.max(np.abs(lg.cpu().numpy()[sel] - o_lg) / (1 + np.abs(o_lg))) <= 1e-3
    # the same sweep with the candidates' Delta cache rows (what Infer and bench.py run; 19.7 GB more): same bits for all 100 k pairs,
    # no index list, 98 chunks -- the cache row of candidate p must be row p in every chunk
    dc = e.delta_cache(cands)
    r2 = e.heads(cands, q, spec_l=spec, spec_r=qspec, want_logit=True, dcache_l=dc)
    torch.cuda.synchronize()
    assert torch.equal(r2["logit"], r["logit"]) and torch.equal(r2["overlap"], r["overlap"]) and torch.equal(r2["yaw"], r["yaw"])


def test_small_sweeps_have_the_bits_of_the_big_sweep(engines):
    """The head kernels pick their workgroup decomposition by sweep size (contraction: 1 ... 45 workgroups per pair; c_conv3 + Dense 3 or 12
    per pair; small sweeps carry the query's A2 tasks in their yaw launch -- all for the single-pair latency of demo2 / gated demo3
    queries): a candidate's result must not depend on it -- every prefix of a
    300-candidate sweep has the bits of the full sweep, with and without the Delta cache, with and without an index list."""
    e = engines[4]
    rng = np.random.default_rng(8)
    fv = torch.from_numpy(np.maximum(rng.normal(0.2, 1.0, size=(300, 360, 128)), 0).astype(np.float32)).cuda()
    fv *= torch.from_numpy(rng.uniform(0.5, 2.0, size=(300, 1, 1)).astype(np.float32)).cuda()
    q = fv[7:8].contiguous()
    spec, qs, dc = e.spectrum(fv), e.spectrum(fv[7:8].contiguous()), e.delta_cache(fv)
    full = e.heads(fv, q, spec_l=spec, spec_r=qs, want_logit=True, dcache_l=dc)
    # either side of every size at which a kernel changes its decomposition: contraction 5 | 6 (8-row passes) and 10 | 11 (one row tile
    # per pass), c_conv3 + Dense 21 | 22 (four workgroups per band), c_conv2 28 | 29, the query's A2 tasks in the yaw launch 64 | 65
    for n in (1, 2, 3, 5, 6, 10, 11, 21, 22, 28, 29, 42, 43, 64, 65, 86, 129):
        for kw in ({}, {"dcache_l": dc[:n].contiguous()}):
            r = e.heads(fv[:n].contiguous(), q, spec_l=spec[:n].contiguous(), spec_r=qs, want_logit=True, **kw)
            assert torch.equal(r["logit"], full["logit"][:n]) and torch.equal(r["yaw"], full["yaw"][:n]), (n, bool(kw))
        idx = np.arange(n, dtype=np.int32)
        r = e.heads(fv, q, lidx=idx, spec_l=spec, spec_r=qs, want_logit=True, dcache_l=dc)
        assert torch.equal(r["logit"], full["logit"][:n]), n
    one = e.heads(fv, q, lidx=np.array([200], np.int32), spec_l=spec, spec_r=qs, want_logit=True)      # a single pair far into the pool
    assert torch.equal(one["logit"], full["logit"][200:201])


def test_head_results_do_not_depend_on_the_launch_structure(engines):
    """ovn_set_head_pipeline (chunk / sub-chunk sizes, one or two streams, yaw head on a side stream) changes WHEN kernels run,
    never what they compute: every pair gets the same bits; and ovn_heads_spectral == ovn_delta_head + ovn_corr_head_spectral."""
    from overlapnet_amd import _lib
    e = engines[4]
    rng = np.random.default_rng(123)
    n = 700
    fv = torch.from_numpy(np.maximum(rng.normal(0.2, 1.0, size=(n, 360, 128)), 0).astype(np.float32)).cuda()
    q = fv[11:12].contiguous()
    spec, qspec = e.spectrum(fv), e.spectrum(q)
    DEFAULT_PIPELINE = e.head_pipeline()
    try:
        e.set_head_pipeline(1024, 0, 1, False)                      # one chunk, one stream, everything in order
        ref = e.heads(fv, q, spec_l=spec, spec_r=qspec, want_logit=True, want_corr=True)
        ov = torch.empty(n, device="cuda")
        lg = torch.empty(n, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(e.lib.ovn_delta_head(e._h, fv.data_ptr(), None, q.data_ptr(), None, n, ov.data_ptr(), lg.data_ptr(), st), "ovn_delta_head")
        sep = e.corr_head_spectral(spec, qspec, want_corr=True)
        assert torch.equal(ov, ref["overlap"]) and torch.equal(lg, ref["logit"])
        assert torch.equal(sep["yaw"], ref["yaw"]) and torch.equal(sep["corr"], ref["corr"])
        idx = torch.from_numpy(rng.permutation(n).astype(np.int32)).cuda()
        ref_idx = e.heads(fv, q, lidx=idx, spec_l=spec, spec_r=qspec, want_logit=True)
        # the K-walk rotation follows the candidate's SLOT in the pool, not its place in the list: a permuted list gives the same bits
        assert torch.equal(ref_idx["logit"], ref["logit"][idx.long()]) and torch.equal(ref_idx["yaw"], ref["yaw"][idx.long()])
        for cfg in ((1024, 256, 1, True), (1024, 256, 2, True), (1024, 100, 2, False), (300, 64, 2, True), (128, 0, 1, True),
                    (256, 96, 2, True)):
            e.set_head_pipeline(*cfg)
            r = e.heads(fv, q, spec_l=spec, spec_r=qspec, want_logit=True, want_corr=True)
            torch.cuda.synchronize()
            for k in ("overlap", "logit", "yaw", "corr"):
                assert torch.equal(r[k], ref[k]), (cfg, k)
            r2 = e.heads(fv, q, lidx=idx, spec_l=spec, spec_r=qspec, want_logit=True)
            assert torch.equal(r2["logit"], ref_idx["logit"]) and torch.equal(r2["yaw"], ref_idx["yaw"]), cfg
            d = e.heads(fv, q, want_logit=True)                      # direct correlation form through the same pipeline
            assert torch.equal(d["logit"], ref["logit"]), cfg
        with pytest.raises(Exception):
            e.set_head_pipeline(0, 0, 1, True)
        with pytest.raises(Exception):
            e.set_head_pipeline(1024, 0, 3, True)
    finally:
        e.set_head_pipeline(*DEFAULT_PIPELINE)


def test_delta_cache_gives_the_same_bits_and_falls_back_per_pair(engines):
    """ovn_delta_cache rows (packed words at the candidate's own scale, TT + b2, range) replace the per-pair preparation of a 1-vs-N
    sweep wherever they are valid -- no negative value, the query's maximum below the candidate's next power of two -- and every
    other pair is prepared in scratch as before: identical bits either way, pair by pair."""
    e = engines[4]
    rng = np.random.default_rng(2024)
    n = 300
    fv = np.maximum(rng.normal(0.2, 1.0, size=(n, 360, 128)), 0).astype(np.float32)
    fv[5] *= 40.0            # a candidate two buckets above the rest (its own scale is coarser than the query's: version 5..6 of the query)
    fv[6] *= 1.0e-3          # far below: the QUERY sets the bucket -> fallback
    fv[7] *= 0.0             # empty volume -> fallback
    fv[8] = -fv[8]           # negative values -> fallback (shifted arithmetic)
    fv[9] *= 3.0e4           # more than 2^7 above the query: no packed query version -> fallback
    fvt = torch.from_numpy(fv).cuda()
    spec = e.spectrum(fvt)
    dc = e.delta_cache(fvt)
    assert tuple(dc.shape) == (n, e.DELTA_CACHE_ELEMS)
    meta = dc[:, 46080 + 3072:46080 + 3074].cpu().numpy()
    assert np.array_equal(meta[:, 0], fv.reshape(n, -1).max(axis=1)) and np.array_equal(meta[:, 1], fv.reshape(n, -1).min(axis=1))
    for qi, qscale in ((0, 1.0), (1, 0.25), (2, 5.0), (8, 1.0)):      # ordinary, smaller, larger (many candidates fall back), negative query
        q = (fvt[qi:qi + 1] * qscale).contiguous()
        qs = e.spectrum(q)
        ref = e.heads(fvt, q, spec_l=spec, spec_r=qs, want_logit=True)
        got = e.heads(fvt, q, spec_l=spec, spec_r=qs, want_logit=True, dcache_l=dc)
        torch.cuda.synchronize()
        assert torch.equal(got["logit"], ref["logit"]) and torch.equal(got["overlap"], ref["overlap"]) and torch.equal(got["yaw"], ref["yaw"]), qi
        idx = torch.from_numpy(rng.permutation(n)[:77].astype(np.int32)).cuda()
        a = e.heads(fvt, q, lidx=idx, spec_l=spec, spec_r=qs, want_logit=True, dcache_l=dc)
        b = e.heads(fvt, q, lidx=idx, spec_l=spec, spec_r=qs, want_logit=True)
        assert torch.equal(a["logit"], b["logit"]), qi
    # against the fp64 oracle too (the cache changes where the work is done, not the arithmetic that is checked everywhere else)
    w = S.make_test_weights(4, seed=0)
    q = fvt[0:1].contiguous()
    got = e.heads(fvt[:12].contiguous(), q, spec_l=spec[:12].contiguous(), spec_r=e.spectrum(q), want_logit=True, dcache_l=dc[:12].contiguous())
    fv4 = fv[:12].reshape(-1, 1, 360, 128).astype(np.float64)
    o_ov, o_yaw, o_lg, _ = O.heads_forward(fv4, np.repeat(fv4[0:1], 12, axis=0), w)
    assert np.max(np.abs(got["overlap"].cpu().numpy() - o_ov)) <= 1e-4 and np.array_equal(got["yaw"].cpu().numpy(), o_yaw)
    with pytest.raises(Exception):
        e.heads(fvt, q, spec_l=spec, spec_r=e.spectrum(q), dcache_l=dc[:10].contiguous())


def test_delta_cache_rows_follow_the_candidates_across_chunks(engines):
    """A sweep WITHOUT an index list that spans several chunks / sub-chunks (two streams) must address candidate p's cache row, not
    row p - chunk_start (the cache pointer did not move with the feature pointer): same bits as without the cache, and
    within tolerance of the fp64 oracle for pairs beyond the first chunk."""
    e = engines[4]
    rng = np.random.default_rng(4242)
    n = 300
    fv = np.maximum(rng.normal(0.2, 1.0, size=(n, 360, 128)), 0).astype(np.float32)
    fv *= rng.uniform(0.5, 2.0, size=(n, 1, 1)).astype(np.float32)      # rows differ in scale: a wrong row changes the result
    fvt = torch.from_numpy(fv).cuda()
    spec, dc = e.spectrum(fvt), e.delta_cache(fvt)
    q = fvt[17:18].contiguous()
    qs = e.spectrum(q)
    saved = e.head_pipeline()
    try:
        e.set_head_pipeline(1024, 0, 1, False)
        ref = e.heads(fvt, q, spec_l=spec, spec_r=qs, want_logit=True)
        for cfg in ((128, 48, 2, True), (128, 0, 1, False), (100, 0, 1, False), (1024, 64, 2, False)):
            e.set_head_pipeline(*cfg)
            got = e.heads(fvt, q, spec_l=spec, spec_r=qs, want_logit=True, dcache_l=dc)
            torch.cuda.synchronize()
            assert torch.equal(got["logit"], ref["logit"]) and torch.equal(got["overlap"], ref["overlap"]) and torch.equal(got["yaw"], ref["yaw"]), cfg
        w = S.make_test_weights(4, seed=0)
        sel = [130, 255, 299]
        fv4 = fv[sel].reshape(-1, 1, 360, 128).astype(np.float64)
        o_ov, o_yaw, _, _ = O.heads_forward(fv4, np.repeat(fv[17:18].reshape(1, 1, 360, 128).astype(np.float64), len(sel), axis=0), w)
        assert np.max(np.abs(got["overlap"].cpu().numpy()[sel] - o_ov)) <= 1e-4 and np.array_equal(got["yaw"].cpu().numpy()[sel], o_yaw)
    finally:
        e.set_head_pipeline(*saved)


def test_dead_channel_compaction_of_the_query(engines):
    """1-vs-N sweeps drop the channels that are zero in ALL 360 columns of the query from the Delta head's contraction
    (`ovn_set_head_compaction`, default on): exact, so the results agree with the uncompacted walk (compaction off, and the indexed-pairs
    form, which never compacts) to fp32 rounding and with the fp64 oracle within the north-star tolerance; a query WITHOUT a dead channel
    walks the very K of the uncompacted kernel (same bits); a query with a negative value is never compacted (its shifted words have
    no zeros); with and without the candidates' Delta cache rows the bits are the same in every case."""
    e = engines[4]
    w = S.make_test_weights(4, seed=0)
    rng = np.random.default_rng(77)
    n = 70
    cands = np.maximum(rng.normal(0.2, 1.0, size=(n, 360, 128)), 0).astype(np.float32)
    cands[:, :, [3, 40, 41, 100]] = 0                     # candidates have dead channels of their own (irrelevant to the list)
    dc = torch.from_numpy(cands).cuda()
    spec = e.spectrum(dc)
    cache = e.delta_cache(dc)

    def sweep(q, compaction, use_cache=True):
        e.set_head_compaction(compaction)
        try:
            dq = torch.from_numpy(q).cuda()
            r = e.heads(dc, dq, spec_l=spec, spec_r=e.spectrum(dq), dcache_l=cache if use_cache else None, want_logit=True)
            return r["logit"].cpu().numpy(), r["overlap"].cpu().numpy(), r["yaw"].cpu().numpy()
        finally:
            e.set_head_compaction(True)

    # live = 128 - dead; a last slice of <= 16 live channels is packed tap-major (3, 6 or 9 MFMA steps instead of 15): dead = 28, 61, 93
    # (4, 3, 3 channels in the last of 4 / 3 / 2 slices), 112, 120, 122, 127 (16, 8, 6, 1 channels in the only slice)
    for dead in (0, 5, 28, 33, 61, 70, 93, 112, 120, 122, 127, 128):
        q = np.maximum(rng.normal(0.2, 1.0, size=(1, 360, 128)), 0).astype(np.float32)
        kill = rng.permutation(128)[:dead]
        q[:, :, kill] = 0
        lg1, ov1, yw1 = sweep(q, True)
        lg0, ov0, yw0 = sweep(q, False)
        lgn, ovn, ywn = sweep(q, True, use_cache=False)
        assert np.array_equal(lg1, lgn) and np.array_equal(yw1, ywn), dead          # cache rows or scratch: same bits
        assert np.array_equal(yw1, yw0)
        if dead == 0:
            assert np.array_equal(lg1, lg0)                                           # nothing to drop: the uncompacted K walk
        else:
            assert np.max(np.abs(lg1 - lg0)) <= 2e-5 * (1 + np.max(np.abs(lg0))), dead
        allf = torch.cat([dc, torch.from_numpy(q).cuda()])
        ri = e.heads(allf, allf, lidx=np.arange(n), ridx=np.full(n, n, np.int64), want_logit=True)["logit"].cpu().numpy()
        assert np.array_equal(ri, lg0)                                                # indexed pairs never compact
        st = None
        if dead not in (0, 128):
            sweep(q, True)
            st = e.head_walk_stats()
            live = 128 - dead
            n_last = live - 32 * ((live + 31) // 32 - 1)
            want_steps = 0 if n_last > 16 else 3 * (-(-(-(-15 // (32 // n_last))) // 3))
            assert st["compacted"] and st["live_channels"] == live and st["packed_last_slice_steps"] == want_steps, (dead, st)
        if dead in (28, 33, 61, 93, 112, 120, 122, 127, 128):
            fv = cands[:8].reshape(8, 1, 360, 128).astype(np.float64)
            o_ov, o_yaw, o_lg, _ = O.heads_forward(fv, np.repeat(q.reshape(1, 1, 360, 128).astype(np.float64), 8, axis=0), w)
            assert np.max(np.abs(ov1[:8] - o_ov)) <= 1e-4 and np.max(np.abs(lg1[:8] - o_lg)) <= 1e-3 * (1 + np.max(np.abs(o_lg)))
    # channels that are alive only in SOME column-group pairs (the list is ordered by that count and every pass walks only up to its
    # own last live position): random patterns, incl. pairs without a single live channel
    # Every pattern is ALSO compared with the fp64 oracle directly (16 pairs each, on the cache-row route `Infer` and bench.py take):
    # against compaction-off alone the HIP path would only be checked against itself.
    NOR = 16
    fv_or = cands[:NOR].reshape(NOR, 1, 360, 128).astype(np.float64)
    for trial in range(7):
        q = np.maximum(rng.normal(0.2, 1.0, size=(1, 360, 128)), 0).astype(np.float32)
        if trial < 6:
            keep = rng.random((12, 128)) < rng.choice([0.15, 0.5, 0.85])
            keep[:, rng.permutation(128)[:20]] = False
            if trial == 5:
                keep[3] = False
                keep[11] = False
        else:
            # nested live sets: 20 channels alive in every column-group pair, 30 more from pair 3 on, 30 more from pair 6 on, 30 more
            # from pair 9 on -> the list orders them by that count and column-group pairs 0-2 / 3-5 / 6-8 / 9-11 need 1 / 2 / 3 / 4 slices
            perm = rng.permutation(128)
            keep = np.zeros((12, 128), bool)
            keep[:, perm[:20]] = True
            keep[3:, perm[20:50]] = True
            keep[6:, perm[50:80]] = True
            keep[9:, perm[80:110]] = True
            q[0, :, perm[:110]] += 0.01      # every kept channel really is alive in each of its pairs
        q[0] *= np.repeat(keep, 30, axis=0)
        lg1, ov1, yw1 = sweep(q, True)
        lg0, ov0, yw0 = sweep(q, False)
        lgn, _, ywn = sweep(q, True, use_cache=False)
        assert np.array_equal(lg1, lgn) and np.array_equal(yw1, ywn), trial
        assert np.array_equal(yw1, yw0) and np.max(np.abs(lg1 - lg0)) <= 2e-5 * (1 + np.max(np.abs(lg0))), trial
        o_ov, o_yaw, o_lg, _ = O.heads_forward(fv_or, np.repeat(q.reshape(1, 1, 360, 128).astype(np.float64), NOR, axis=0), w)
        assert np.max(np.abs(ov1[:NOR] - o_ov)) <= 1e-4, (trial, float(np.max(np.abs(ov1[:NOR] - o_ov))))
        assert np.max(np.abs(lg1[:NOR] - o_lg) / (1 + np.abs(o_lg))) <= 1e-3, trial
        assert np.array_equal(yw1[:NOR], o_yaw), trial
        if trial == 6:
            sweep(q, True)
            st = e.head_walk_stats()          # the walk the kernel really took (ovn_head_walk_stats)
            assert st["compacted"] and st["slices_per_group_pair"] == [1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4] and st["live_channels"] == 110, st
            # the 14 channels of the fourth slice are packed two taps per step: 9 steps instead of 15 for the waves that walk it
            assert st["max_slices"] == 4 and st["packed_last_slice_steps"] == 9, st
            assert abs(st["k_walk_frac"] - (15 + 15 + 30 + 30 + 45 + 45 + 54 + 54) / 480.0) < 1e-9, st
        # a sweep of 3 candidates (half-pass workgroups) has the bits of the big one
        e.set_head_compaction(True)
        dq = torch.from_numpy(q).cuda()
        r3 = e.heads(dc[:3].contiguous(), dq, spec_l=spec[:3].contiguous(), spec_r=e.spectrum(dq), dcache_l=cache[:3].contiguous(), want_logit=True)
        assert np.array_equal(r3["logit"].cpu().numpy(), lg1[:3]), trial
    # a negative value anywhere in the query: the pair needs a shift, nothing is dropped -> the bits of the uncompacted walk
    q = np.maximum(rng.normal(0.2, 1.0, size=(1, 360, 128)), 0).astype(np.float32)
    q[:, :, :40] = 0
    q[0, 17, 5] = -0.25
    lg1, _, yw1 = sweep(q, True)
    lg0, _, yw0 = sweep(q, False)
    assert np.array_equal(lg1, lg0) and np.array_equal(yw1, yw0)


def test_delta_cache_is_ignored_where_it_does_not_apply(engines):
    """fp32 head mode and pair lists with their own right-hand volumes never use the Delta cache rows: same results as without."""
    e = engines[4]
    rng = np.random.default_rng(77)
    fv = torch.from_numpy(np.maximum(rng.normal(0.2, 1.0, size=(20, 360, 128)), 0).astype(np.float32)).cuda()
    spec, dc = e.spectrum(fv), e.delta_cache(fv)
    li, ri = np.arange(20), (np.arange(20) * 7) % 20
    a = e.heads(fv, fv, lidx=li, ridx=ri, spec_l=spec, spec_r=spec, want_logit=True, dcache_l=dc)
    b = e.heads(fv, fv, lidx=li, ridx=ri, spec_l=spec, spec_r=spec, want_logit=True)
    assert torch.equal(a["logit"], b["logit"]) and torch.equal(a["yaw"], b["yaw"])
    q = fv[3:4].contiguous()
    e.set_head_precision("f32")
    try:
        qs = e.spectrum(q)
        s32 = e.spectrum(fv)
        a = e.heads(fv, q, spec_l=s32, spec_r=qs, want_logit=True, dcache_l=dc)
        b = e.heads(fv, q, spec_l=s32, spec_r=qs, want_logit=True)
        assert torch.equal(a["logit"], b["logit"])
    finally:
        e.set_head_precision(DEFAULT_HEAD)


@pytest.mark.parametrize("s", [10, 16, 24, 40])
def test_other_conv1sizes_run_the_general_path(s, tmp_path, fixture_npz):
    """`conv1NetworkHead_conv1size` != 15 (generateNet.py:88-89): the library's general fp32 Delta path, against the fp64 oracle built
    with the same s -- s = 16 and 40 do not divide 360 ('valid' convolutions drop the remainder: 22 and 9 groups) -- through the
    engine and through `Infer` with the network.yml key."""
    from overlapnet_amd.engine import OvnEngine
    from overlapnet_amd.infer import Infer
    cfg = dict(CFG, conv1NetworkHead_conv1size=s)
    w = S.make_test_weights(4, seed=3, model_cfg=cfg)
    g = 360 // s
    assert w["c_conv1/kernel"].shape == (1, s, 128, 64) and w["overlap_output/kernel"].shape == ((g - 2) ** 2 * 256, 1)
    rng = np.random.default_rng(1000 + s)
    fv = np.maximum(rng.normal(0.2, 1.0, size=(6, 360, 128)), 0).astype(np.float32)
    fv[4] = -fv[4]                                  # signed features too
    pairs = np.array([[0, 1], [1, 0], [2, 2], [3, 5], [4, 0], [5, 4], [0, 3]])
    e = OvnEngine(64, 900, 4)
    try:
        e.load_weights(w, cfg)
        assert e.conv1size == s and not e.has_delta_cache
        ft = torch.from_numpy(fv).cuda()
        r = e.heads(ft, ft, lidx=pairs[:, 0], ridx=pairs[:, 1], want_logit=True)
        spec = e.spectrum(ft)
        r2 = e.heads(ft, ft[1:2].contiguous(), spec_l=spec, spec_r=spec[1:2].contiguous(), want_logit=True)
        with pytest.raises(Exception, match="Delta cache"):
            e.delta_cache(ft)
        fv4 = fv.reshape(-1, 1, 360, 128).astype(np.float64)
        ov, yaw, lg, _ = O.heads_forward(fv4[pairs[:, 0]], fv4[pairs[:, 1]], w, conv1size=s)
        assert np.max(np.abs(r["overlap"].cpu().numpy() - ov)) <= 1e-4 and np.array_equal(r["yaw"].cpu().numpy(), yaw)
        assert np.all(np.abs(r["logit"].cpu().numpy() - lg) <= 1e-3 * (1 + np.abs(lg)))
        ov2, yaw2, lg2, _ = O.heads_forward(fv4, np.repeat(fv4[1:2], 6, axis=0), w, conv1size=s)
        assert np.max(np.abs(r2["overlap"].cpu().numpy() - ov2)) <= 1e-4 and np.array_equal(r2["yaw"].cpu().numpy(), yaw2)
        assert lg.max() - lg.min() > 0.5            # not a degenerate (constant) head
    finally:
        e.close()
    if s == 10:                                     # the network.yml route: Infer accepts the key and streams frames as usual
        seq = tmp_path / "data" / "07"
        for sub in ("depth", "normal"):
            os.makedirs(seq / sub)
        imgs = []
        for i in range(3):
            d, nm = np.roll(fixture_npz["range_%d" % (i % 2)], 30 * i, axis=1), np.roll(fixture_npz["normal_%d" % (i % 2)], 30 * i, axis=1)
            np.save(seq / "depth" / ("%06d.npy" % i), d)
            np.save(seq / "normal" / ("%06d.npy" % i), nm)
            imgs.append(S.stack(d, nm, None, (True, True, False)))
        conf = {"model": dict(cfg, legsType="360OutputkLegs", overlap_head="DeltaLayerConv1NetworkHead", orientation_head="CorrelationHead",
                              inputShape=[64, 900], leg_output_width=360),
                "infer_seqs": "07", "data_root_folder": str(tmp_path / "data"), "use_depth": True, "use_normals": True,
                "use_class_probabilities": False, "use_class_probabilities_pca": False, "use_intensity": False, "batch_size": 16,
                "pretrained_weightsfilename": ""}
        inf = Infer(conf, weights=w)
        for i in range(3):
            res = inf.infer_multiple(i, list(range(i)))
        ofv = O.leg_forward(np.stack(imgs), w, cfg, np.float64)
        o_ov, o_yaw, _, _ = O.heads_forward(ofv[[0, 1]], ofv[[2, 2]], w, conv1size=s)
        assert np.max(np.abs(res[0] - o_ov)) <= 1e-4 and np.array_equal(res[1], o_yaw)


def test_query_ahead_gives_the_bits_of_the_serial_order(engines, fixture_images):
    """`QueryAhead` (leg + spectrum of the next query on a second context and stream, beside the current query's head kernels:
    the streaming form of `Infer.infer_multiple`, infer.py:162-203) returns exactly what `engine.leg` / `engine.spectrum` return on
    the caller's stream, for a sequence of different queries, in both submit / take orders, and refuses a third query in flight."""
    from overlapnet_amd.engine import QueryAhead
    from overlapnet_amd._lib import OvnError
    e = engines[4]
    w = S.make_test_weights(4, seed=0)
    imgs = torch.from_numpy(fixture_images(4)).cuda()
    queries = [imgs[i % imgs.shape[0]:i % imgs.shape[0] + 1].roll(37 * i, dims=2).contiguous() for i in range(6)]
    cands = e.leg(torch.cat(queries[:4]))
    cspec, cdc = e.spectrum(cands), e.delta_cache(cands)
    want = []
    for q in queries:
        fv = e.leg(q)
        sp = e.spectrum(fv)
        r = e.heads(cands, fv, spec_l=cspec, spec_r=sp, dcache_l=cdc)
        want.append((fv.clone(), sp.clone(), r["overlap"].clone(), r["yaw"].clone()))
    qa = QueryAhead(e, w, S.REFERENCE_MODEL_CFG)
    try:
        qa.submit(queries[0])
        for k, q in enumerate(queries):
            if k + 1 < len(queries):
                qa.submit(queries[k + 1])             # beside the heads of query k
            fv, sp = qa.take()
            r = e.heads(cands, fv, spec_l=cspec, spec_r=sp, dcache_l=cdc)
            assert torch.equal(fv, want[k][0]) and torch.equal(sp, want[k][1])
            assert torch.equal(r["overlap"], want[k][2]) and torch.equal(r["yaw"], want[k][3])
        with pytest.raises(OvnError, match="nothing submitted"):
            qa.take()
        for k, q in enumerate(queries[:3]):            # take-then-submit order, one in flight
            qa.submit(q)
            fv, sp = qa.take()
            assert torch.equal(fv, want[k][0]) and torch.equal(sp, want[k][1])
        qa.submit(queries[0])
        qa.submit(queries[1])
        with pytest.raises(OvnError, match="already in flight"):
            qa.submit(queries[2])
        assert torch.equal(qa.take()[0], want[0][0]) and torch.equal(qa.take()[0], want[1][0])
    finally:
        qa.close()


def test_query_ahead_soak_under_a_full_head_sweep(engines, fixture_images):
    """120 streamed queries, the heads of each a 1-vs-1024 sweep that keeps every CU busy while the next query's leg runs beside it:
    every feature volume, spectrum and score equals the serial order's (a missing event or a reused buffer would show up here)."""
    from overlapnet_amd.engine import QueryAhead
    e = engines[4]
    w = S.make_test_weights(4, seed=0)
    imgs = torch.from_numpy(fixture_images(4)).cuda()
    queries = [imgs[i % imgs.shape[0]:i % imgs.shape[0] + 1].roll(53 * i + 7, dims=2).contiguous() for i in range(8)]
    cands = e.leg(torch.cat(queries)).repeat(128, 1, 1).contiguous()          # 1024 candidates
    cspec, cdc = e.spectrum(cands), e.delta_cache(cands)
    want = []
    for q in queries:
        fv = e.leg(q)
        sp = e.spectrum(fv)
        r = e.heads(cands, fv, spec_l=cspec, spec_r=sp, dcache_l=cdc)
        want.append((fv.clone(), sp.clone(), r["overlap"].clone(), r["yaw"].clone()))
    qa = QueryAhead(e, w, S.REFERENCE_MODEL_CFG)
    bad = 0
    try:
        n = 120
        qa.submit(queries[0])
        results = []
        for k in range(n):
            if k + 1 < n:
                qa.submit(queries[(k + 1) % 8])
            fv, sp = qa.take()
            r = e.heads(cands, fv, spec_l=cspec, spec_r=sp, dcache_l=cdc)
            results.append((fv.clone(), sp.clone(), r["overlap"], r["yaw"]))     # clones are ordered on the consumer's stream
        torch.cuda.synchronize()
        for k, got in enumerate(results):
            bad += int(not all(torch.equal(a, b) for a, b in zip(got, want[k % 8])))
        # the same loop with wait_current=False (images resident, nothing on the caller's stream produces them): the heads of query
        # k - 1 read the slot that submit(k + 1) overwrites straight out of the helper's buffers -- the side stream must still wait
        # for THEM (the release event used to be recorded before those heads were enqueued)
        qa.submit(queries[0], wait_current=False)
        results = []
        for k in range(n):
            if k + 1 < n:
                qa.submit(queries[(k + 1) % 8], wait_current=False)
            fv, sp = qa.take()
            r = e.heads(cands, fv, spec_l=cspec, spec_r=sp, dcache_l=cdc)
            results.append((r["overlap"], r["yaw"]))
        torch.cuda.synchronize()
        for k, got in enumerate(results):
            bad += int(not all(torch.equal(a, b) for a, b in zip(got, want[k % 8][2:])))
    finally:
        qa.close()
    assert bad == 0


@pytest.mark.parametrize("C,mode", [(1, "f16x3"), (5, "f16x3"), (4, "f32")])
def test_query_ahead_other_channel_counts_and_fp32_mode(engines, fixture_images, C, mode):
    """The second context of `QueryAhead` is built like the first: other input channel counts (the generic first-layer kernel, the
    absmax pass) and the all-fp32 arithmetic give the serial order's bits too."""
    from overlapnet_amd.engine import QueryAhead
    e = engines[C]
    w = S.make_test_weights(C, seed=0)
    imgs = torch.from_numpy(fixture_images(C)).cuda()
    queries = [imgs[i % imgs.shape[0]:i % imgs.shape[0] + 1].roll(41 * i, dims=2).contiguous() for i in range(3)]
    e.set_leg_precision(mode)
    e.set_head_precision(mode)
    try:
        want = []
        for q in queries:
            fv = e.leg(q)
            want.append((fv.clone(), e.spectrum(fv).clone()))
        qa = QueryAhead(e, w, S.REFERENCE_MODEL_CFG)
        try:
            qa.submit(queries[0])
            for k in range(3):
                if k + 1 < 3:
                    qa.submit(queries[k + 1])
                fv, sp = qa.take()
                assert torch.equal(fv, want[k][0]) and torch.equal(sp, want[k][1])
        finally:
            qa.close()
    finally:
        e.set_leg_precision("f16x3")
        e.set_head_precision(DEFAULT_HEAD)
