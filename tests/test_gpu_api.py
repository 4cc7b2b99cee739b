"""GPU (MI355X): the drop-in API rows around the hot path, end to end through the HIP library --
  * a3: the demo1 drivers gen_depth_data / gen_normal_data / gen_intensity_data (gen_depth_data.py:10-48, gen_normal_data.py:10-46,
        gen_intensity_data.py:10-43) write the .npy files the reference ships, bit for bit, under enumeration-index names;
  * f2: `Infer(config)` with `pretrained_weightsfilename` naming a Keras-layout HDF5 file (infer.py:117-120);
  * f3: the evaluation run of testing.py:207-352 (`evaluate.run_test`) on the real `Infer`."""
import os

import numpy as np
import pytest
import torch

from oracle import overlapnet_oracle as O
from tools import synthetic as S
from overlapnet_amd import weights as W

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not torch.cuda.is_available(), reason="needs an MI355X")]

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CFG = S.REFERENCE_MODEL_CFG


def _write_sequence(root, fx, n, with_intensity=False):
    """n frames laid out like demo1 writes them: frame i = fixture scan (i mod 2) rolled by 40 (i // 2) columns."""
    seq = os.path.join(root, "07")
    for sub in ("depth", "normal", "intensity"):
        os.makedirs(os.path.join(seq, sub), exist_ok=True)
    imgs = []
    for i in range(n):
        s, shift = i % 2, 40 * (i // 2)
        d = np.roll(fx["range_%d" % s], shift, axis=1)
        nm = np.roll(fx["normal_%d" % s], shift, axis=1)
        it = np.roll(fx["intensity_%d" % s], shift, axis=1)
        np.save(os.path.join(seq, "depth", "%06d.npy" % i), d)
        np.save(os.path.join(seq, "normal", "%06d.npy" % i), nm)
        np.save(os.path.join(seq, "intensity", "%06d.npy" % i), it)
        imgs.append(S.stack(d, nm, it, (True, True, with_intensity)))
    return np.stack(imgs)


def _config(root, weightfile="", **extra):
    cfg = {"model": dict(CFG, inputShape=[64, 900]), "infer_seqs": "07", "data_root_folder": str(root), "use_depth": True,
           "use_normals": True, "use_class_probabilities": False, "use_class_probabilities_pca": False, "use_intensity": False,
           "batch_size": 16, "pretrained_weightsfilename": weightfile}
    cfg.update(extra)
    return cfg


def test_gen_data_drivers_write_the_reference_npy_files(tmp_path, fixture_npz):
    from overlapnet_amd import preprocess as P
    scans = tmp_path / "scans"
    os.makedirs(scans)
    # file names that differ from the enumeration index: the reference names its outputs by INDEX (gen_depth_data.py:41)
    fixture_npz["points_0"].astype(np.float32).tofile(scans / "000010.bin")
    fixture_npz["points_1"].astype(np.float32).tofile(scans / "000025.bin")
    dst = tmp_path / "dst"
    os.makedirs(dst)
    depth = P.gen_depth_data(str(scans), str(dst))
    normal = P.gen_normal_data(str(scans), str(dst))
    inten = P.gen_intensity_data(str(scans), str(dst))
    diffs = {}
    for i in range(2):
        for sub, key, ret in (("depth", "range_%d", depth), ("normal", "normal_%d", normal), ("intensity", "intensity_%d", inten)):
            f = dst / sub / ("%06d.npy" % i)
            assert f.is_file(), "%s not written under its enumeration index" % f
            got = np.load(f)
            ref = fixture_npz[key % i]            # produced by the reference's own utils.py, equal to the .npy files it ships
            assert got.dtype == np.float32 and got.shape == ref.shape and np.array_equal(got, ret[i])
            d = got != ref
            diffs["%s_%d" % (sub, i)] = int(np.count_nonzero(d if d.ndim == 2 else np.any(d, axis=-1)))
    print("pixels differing from the reference-generated .npy files:", diffs)
    assert all(v == 0 for v in diffs.values()), diffs
    assert sorted(os.listdir(dst / "depth")) == ["000000.npy", "000001.npy"]
    # normalize=True divides by the image maximum (gen_depth_data.py:37-38)
    os.makedirs(tmp_path / "dst2")
    dn = P.gen_depth_data(str(scans), str(tmp_path / "dst2"), normalize=True)
    assert np.array_equal(dn[0], fixture_npz["range_0"] / np.max(fixture_npz["range_0"]))
    assert np.array_equal(np.load(tmp_path / "dst2" / "depth" / "000000.npy"), dn[0])


def test_infer_loads_a_keras_layout_weight_file(tmp_path, fixture_npz):
    from overlapnet_amd.infer import Infer
    wfile = os.path.join(G, "keras_layout_full_c4.weight")
    w = W.load_weights_file(wfile)                 # the framework's own HDF5 reader (pinned against h5py in test_hdf5_lite.py)
    sums = np.load(os.path.join(G, "keras_layout_full_c4_checksums.npz"))
    for k in sums.files:                           # checksums written by h5py's side of the generator script
        assert abs(float(w[k].astype(np.float64).sum()) - sums[k][0]) < 1e-9 and abs(float(np.abs(w[k]).astype(np.float64).sum()) - sums[k][1]) < 1e-9
    imgs = _write_sequence(tmp_path / "data", fixture_npz, 4)
    inf = Infer(_config(tmp_path / "data", wfile))
    ref_fv = O.leg_forward(imgs, w, CFG, np.float64)
    ov, yaw = inf.infer_one("a/000002.bin", "b/000001.bin")       # l = name2 (frame 1), r = name1 (frame 2)
    o_ov, o_yaw, _, _ = O.heads_forward(ref_fv[[1]], ref_fv[[2]], w)
    assert abs(ov[0] - o_ov[0]) < 1e-4 and yaw[0] == o_yaw[0]
    for i in range(4):
        res = inf.infer_multiple(i, list(range(i)))
    o_ov, o_yaw, _, _ = O.heads_forward(ref_fv[[0, 1, 2]], ref_fv[[3, 3, 3]], w)
    assert np.max(np.abs(res[0] - o_ov)) < 1e-4 and np.array_equal(res[1], o_yaw)
    # the cache is list-like and lives on the device: indexing copies one volume to the host
    fvs = inf.feature_volumes
    assert len(fvs) == 4 and fvs[3].shape == (1, 360, 128) and fvs[-1].dtype == np.float32
    assert np.max(np.abs(fvs[2][0] - ref_fv[2, 0])) <= 2e-5 * np.max(ref_fv)
    assert np.array(fvs).shape == (4, 1, 360, 128) and len(list(iter(fvs))) == 4
    # optional extension key: everything on the fp32 matrix cores
    inf32 = Infer(_config(tmp_path / "data", wfile, precision="f32", model=dict(CFG, inputShape=[64, 900])))
    assert inf32.engine.leg_precision == "f32" and inf32.engine.head_precision == "f32"
    ov32, yaw32 = inf32.infer_one("a/000002.bin", "b/000001.bin")
    assert abs(ov32[0] - ov[0]) < 2e-5 and yaw32[0] == yaw[0]
    with pytest.raises(Exception, match="precision"):
        Infer(_config(tmp_path / "data", wfile, precision="int8", model=dict(CFG, inputShape=[64, 900])))
    with pytest.raises(Exception, match="weight file not found"):
        Infer(_config(tmp_path / "data", str(tmp_path / "nope.weight"), model=dict(CFG, inputShape=[64, 900])))


def test_evaluation_run_on_the_hip_path(tmp_path, fixture_npz):
    """testing.py's run: GT pairs from an npz in the demo4 layout -> predictions by the real Infer -> error statistics and
    validation_results.npz; predictions checked against the oracle wired like testing.py:236-243 (img1 -> left, img2 -> right)."""
    from overlapnet_amd import evaluate as E
    from overlapnet_amd.infer import Infer
    imgs = _write_sequence(tmp_path / "data", fixture_npz, 6)
    w = S.make_test_weights(4, seed=0)
    rng = np.random.default_rng(3)
    pairs = np.array([[0, 1], [2, 0], [3, 3], [5, 2], [1, 4], [4, 4], [0, 5]])
    gt = np.zeros((len(pairs), 4))
    gt[:, :2] = pairs
    gt[:, 2] = rng.uniform(0.0, 1.0, len(pairs))
    gt[:, 3] = rng.integers(0, 360, len(pairs))
    gt[2, 2] = gt[5, 2] = 0.95                     # self pairs: overlap above the 0.7 yaw threshold
    gt[2, 3] = gt[5, 3] = 180
    seq = np.full((len(pairs), 2), "07", dtype=object)
    npz = str(tmp_path / "ground_truth.npz")
    np.savez(npz, overlaps=gt, seq=seq)
    inf = Infer(_config(tmp_path / "data"), weights=w)
    stats = E.run_test(inf, [npz], out_dir=str(tmp_path / "out"))
    res = np.load(tmp_path / "out" / "validation_results.npz")
    m = res[res.files[0]]
    assert m.shape == (len(pairs), 4) and np.array_equal(m[:, :2], pairs.astype(float))
    fv = O.leg_forward(imgs, w, CFG, np.float64)
    o_ov, o_yaw, _, corr = O.heads_forward(fv[pairs[:, 0]], fv[pairs[:, 1]], w)
    assert np.max(np.abs(m[:, 2] - o_ov)) < 1e-4
    assert np.array_equal(m[:, 3].astype(int), np.argmax(corr, axis=1))           # the stored value is the argmax bin (testing.py:343)
    want = E.error_statistics(o_ov, gt[:, 2], np.argmax(corr, axis=1), gt[:, 3].astype(np.int64))
    assert stats["n"] == len(pairs) and stats["yaw_n"] == want["yaw_n"] >= 2
    for k in ("overlap_mae", "overlap_rms", "overlap_max"):
        assert abs(stats[k] - want[k]) < 1e-4
    assert stats["yaw_mean_err_deg"] == want["yaw_mean_err_deg"] and stats["yaw_max_err_deg"] == want["yaw_max_err_deg"]
    assert stats["yaw_max_err_deg"] == 0              # self pairs -> bin 180 exactly


def test_c_abi_collective_one_rank():
    """ovn_comm_* / ovn_gather_scores (the collective for consumers without torch.distributed) with world_size 1 on the GPU:
    RCCL is loaded lazily, the gather is a send-to-self inside one group call.  (world_size 2 needs two GPUs; the shard / offset
    arithmetic it shares with the torch path is covered by the gloo tests.)"""
    import ctypes as C
    from overlapnet_amd import _lib
    from overlapnet_amd.engine import OvnEngine
    eng = OvnEngine(64, 900, 4)
    lib, h = eng.lib, eng._h
    ident = (C.c_ubyte * 128)()
    _lib.check(lib.ovn_comm_unique_id(ident), "ovn_comm_unique_id")
    assert any(ident)
    _lib.check(lib.ovn_comm_init(h, 0, 1, ident), "ovn_comm_init")
    assert lib.ovn_comm_init(h, 0, 1, ident) != 0            # one communicator per context
    n = 1000
    ov = torch.rand(n, device="cuda")
    yw = torch.randint(-179, 181, (n,), dtype=torch.int32, device="cuda")
    ov_all = torch.zeros(n, device="cuda")
    yw_all = torch.zeros(n, dtype=torch.int32, device="cuda")
    counts = (C.c_int64 * 1)(n)
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.ovn_gather_scores(h, ov.data_ptr(), yw.data_ptr(), counts, 0, ov_all.data_ptr(), yw_all.data_ptr(), stream),
               "ovn_gather_scores")
    torch.cuda.synchronize()
    assert torch.equal(ov_all, ov) and torch.equal(yw_all, yw)
    bad = (C.c_int64 * 1)(-1)
    assert lib.ovn_gather_scores(h, ov.data_ptr(), yw.data_ptr(), bad, 0, ov_all.data_ptr(), yw_all.data_ptr(), stream) != 0
    _lib.check(lib.ovn_comm_destroy(h), "ovn_comm_destroy")
    assert lib.ovn_gather_scores(h, ov.data_ptr(), yw.data_ptr(), counts, 0, ov_all.data_ptr(), yw_all.data_ptr(), stream) != 0
    eng.close()


def test_streaming_lookahead_changes_nothing(tmp_path, fixture_npz):
    """`infer_multiple(i, ...)` reads frame i + 1's files and runs its leg on a second context / stream in the shadow of frame i's head
    kernels (host plumbing of the demo3 loop, demo3_lcd.py:88-123).  A run that hits the look-ahead, one with it disabled, and one
    that then asks for something else must return the same bits; a guess that names a missing file must not surface as an error."""
    from overlapnet_amd.infer import Infer
    root = tmp_path / "data"
    _write_sequence(str(root), fixture_npz, 6)
    w = S.make_test_weights(4, seed=0)

    def run(lookahead):
        inf = Infer(_config(root), weights=w)
        if not lookahead:
            inf._start_ahead = lambda i: None
        out, hits = [], 0
        for i in range(6):
            hits += int(inf._ahead_fv == "%06d" % i)
            out.append(inf.infer_multiple(i, list(range(i))))
        return out, hits, inf

    a, hits_a, inf_a = run(True)     # frames 2 .. 5 come from the side stream (frame 0's call has no heads to hide behind; frame 6 does
    b, hits_b, _ = run(False)        # not exist: that guess fails silently)
    assert hits_a == 4 and hits_b == 0 and inf_a._ahead_fv is None
    for i in range(1, 6):
        assert np.array_equal(a[i][0], b[i][0]) and np.array_equal(a[i][1], b[i][1])
    fa, fb = inf_a.feature_volumes, run(False)[2].feature_volumes
    assert torch.equal(fa.device_features, fb.device_features) and torch.equal(fa.device_spectra, fb.device_spectra)
    used = 360 * 128 + 24 * 128 + 4          # words | linear term | {max, min, 0, 0}; the rest of a row is alignment padding, never written
    assert torch.equal(fa.device_delta_cache[:, :used], fb.device_delta_cache[:, :used])
    # a guess that is wrong: frame 1's call starts frame 2, but the next requests are something else
    inf = Infer(_config(root), weights=w)
    inf.infer_multiple(0, [])
    inf.infer_multiple(1, [0])
    assert inf._ahead_fv == "000002"
    ref = Infer(_config(root), weights=w)
    ref._start_ahead = lambda i: None
    assert np.array_equal(inf.create_feature_volumes(["000004"]), ref.create_feature_volumes(["000004"]))
    inf.feature_volumes = []                           # the reference's way to reset the cache; the pending frame 2 is dropped
    ref.feature_volumes = []
    for i in (0, 1, 2):
        ra, rb = inf.infer_multiple(i, list(range(i))), ref.infer_multiple(i, list(range(i)))
        assert (ra is None and rb is None) or (np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1]))
    got = inf.infer_best_match(3, [0, 1, 2], overlap_thres=0.0)      # frame 3 comes from the side stream here too
    assert got == ref.infer_best_match(3, [0, 1, 2], overlap_thres=0.0)


def test_lookahead_is_dropped_when_the_next_frame_is_rewritten(tmp_path, fixture_npz):
    """VERDICT r3 item 9 / ADVICE r3: the speculative leg of frame i + 1 is keyed by (dataset path, sequence, name, mtime + size of
    every cue file).  A frame rewritten between the two calls (live preprocessing), a changed `seq`, or `stream_ahead: False` must give
    the features of the files as they are when the frame is asked for."""
    import time
    from overlapnet_amd.infer import Infer
    root = tmp_path / "data"
    _write_sequence(str(root), fixture_npz, 4)
    w = S.make_test_weights(4, seed=0)
    inf = Infer(_config(root), weights=w)
    inf.infer_multiple(0, [])
    inf.infer_multiple(1, [0])
    assert inf._ahead_fv == "000002"                       # frame 2's leg ran on the side stream, from the OLD files
    old = np.load(root / "07" / "depth" / "000002.npy")
    new = np.roll(old, 111, axis=1)
    time.sleep(0.01)
    np.save(root / "07" / "depth" / "000002.npy", new)     # same size, new mtime, other content
    np.save(root / "07" / "normal" / "000002.npy", np.roll(np.load(root / "07" / "normal" / "000002.npy"), 111, axis=1))
    got = inf.infer_multiple(2, [0, 1])
    ref = Infer(dict(_config(root), stream_ahead=False), weights=w)
    for i in range(3):
        want = ref.infer_multiple(i, list(range(i)))
    assert ref._qa is None and ref._ahead_fv is None       # opt-out: no second context was ever built
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert torch.equal(inf.feature_volumes.device_features, ref.feature_volumes.device_features)
    # the look-ahead itself still works afterwards (frame 3 untouched)
    assert inf._ahead_fv == "000003"
    a, b = inf.infer_multiple(3, [0, 1, 2]), ref.infer_multiple(3, [0, 1, 2])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    inf.close()
    ref.close()


def test_infer_from_raw_scans(tmp_path, fixture_npz):
    """`config['scan_folder']` (VERDICT r3 "missing" 4, BASELINE configs[4]): `Infer` reads the raw .bin scans and projects them on the
    GPU -- demo1 + demo2/demo3 in one object.  Same bits as the .npy route on the files demo1's drivers write from the same scans,
    look-ahead included, and within tolerance of the fp64 oracle that starts from the raw clouds (own projection + normals)."""
    from overlapnet_amd import preprocess as P
    from overlapnet_amd.infer import Infer
    scans = tmp_path / "scans"
    os.makedirs(scans)
    n = 5
    clouds = [S.transformed_cloud(fixture_npz, i) for i in range(n)]
    for i, c in enumerate(clouds):
        c.tofile(scans / ("%06d.bin" % i))
    dst = tmp_path / "data" / "07"
    os.makedirs(dst)
    P.gen_depth_data(str(scans), str(dst))
    P.gen_normal_data(str(scans), str(dst))
    w = S.make_test_weights(4, seed=0)
    a = Infer(_config(tmp_path / "data", scan_folder=str(scans)), weights=w)
    b = Infer(_config(tmp_path / "data"), weights=w)
    for i in range(n):
        ra, rb = a.infer_multiple(i, list(range(i))), b.infer_multiple(i, list(range(i)))
        assert (ra is None and rb is None) or (np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1])), i
    assert torch.equal(a.feature_volumes.device_features, b.feature_volumes.device_features)
    assert a._qa is not None                                              # frames 2.. came from the look-ahead's projection + leg
    oa, ya = a.infer_one(str(scans / "000001.bin"), str(scans / "000003.bin"))
    ob, yb = b.infer_one(str(scans / "000001.bin"), str(scans / "000003.bin"))
    assert np.array_equal(oa, ob) and np.array_equal(ya, yb)
    # oracle from the raw clouds: l = frame 3 (second argument), r = frame 1 (infer.py:140,150-152)
    imgs = []
    for i in (3, 1):
        rng, vtx, _, _ = O.range_projection(clouds[i])
        imgs.append(S.stack(rng, O.gen_normal_map(rng, vtx), None, (True, True, False)))
    o_ov, o_yaw, _, _, _ = O.infer_pairs(np.stack(imgs), np.array([[0, 1]]), w, S.REFERENCE_MODEL_CFG, np.float64)
    assert abs(float(oa[0]) - float(o_ov[0])) <= 1e-4 and int(ya[0]) == int(o_yaw[0])
    with pytest.raises(Exception, match="Could not read scan file"):
        a.infer_multiple_vs_multiple(["missing.bin"], [0], [0])
    a.close()
    b.close()


def test_reference_demos_replayed_on_the_drop_in(tmp_path, fixture_npz):
    """VERDICT r3 "missing" 5: tests/golden/demo_transcript.json is what the reference's OWN demo2_infer.py / demo3_lcd.py (imported
    unmodified, a recorder in place of `infer`) do with the `Infer` object: constructor config, every call with argument values and
    types, every attribute read.  The drop-in must take exactly those calls and return objects the scripts' next lines work on
    (demo2_infer.py:19-30,45; demo3_lcd.py:28,85-123)."""
    import json
    from overlapnet_amd.infer import Infer
    t = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "demo_transcript.json")))
    w = S.make_test_weights(4, seed=0)
    data = tmp_path / "data"

    def write(seq, n):
        for sub in ("depth", "normal"):
            os.makedirs(data / seq / sub, exist_ok=True)
        for i in range(n):
            s, shift = i % 2, (37 * i) % 900
            np.save(data / seq / "depth" / ("%06d.npy" % i), np.roll(fixture_npz["range_%d" % s], shift, axis=1))
            np.save(data / seq / "normal" / ("%06d.npy" % i), np.roll(fixture_npz["normal_%d" % s], shift, axis=1))

    # ---- demo2 ----
    ev = t["demo2"]
    assert [e["event"] for e in ev] == ["Infer", "infer_one", "getattr", "getattr", "getattr"]
    cfg = ev[0]["config"]
    assert cfg["data_root_folder"] == "data/" and cfg["pretrained_weightsfilename"] == "data/model_geo.weight"
    cfg["data_root_folder"] = str(data) + "/"
    cfg["pretrained_weightsfilename"] = ""            # the authors' weight file is not in the tree (README.md:120): seeded weights
    write(cfg["infer_seqs"], 2)
    infer = Infer(cfg, weights=w)
    overlap, yaw = infer.infer_one(*ev[1]["args"])    # ('data/scans/000001.bin', 'data/scans/000000.bin'): only the names are used
    assert [e["name"] for e in ev[2:]] == ["datasetpath", "seq", "filenames"]
    folder = os.path.join(infer.datasetpath, infer.seq, "depth")                          # demo2_infer.py:25
    depth_data = [np.load(os.path.join(folder, filename + ".npy")) for filename in infer.filenames]   # :27-29
    assert len(depth_data) == 2 and list(infer.filenames) == ["000000", "000001"]
    title = "Overlap: " + str(overlap) + "  Yaw: " + str(yaw)                              # :45
    assert overlap.shape == (1,) and overlap.dtype == np.float32 and yaw.shape == (1,) and title.startswith("Overlap: [0.")
    infer.close()

    # ---- demo3: every infer_multiple call of a 259-frame run, in order, with the recorded argument types ----
    ev = t["demo3"]
    assert ev[0]["event"] == "Infer"
    cfg = ev[0]["config"]
    cfg["data_root_folder"] = str(data) + "/"
    cfg["pretrained_weightsfilename"] = ""
    calls = [e for e in ev if e["event"] == "infer_multiple"]
    assert len(calls) == t["demo3_frames"] == 259
    write(cfg["infer_seqs"], len(calls))
    infer = Infer(cfg, weights=w)
    checked = closures = 0
    for c in calls:
        assert c["cur_type"] == "int"
        refs = np.asarray(c["refs"], dtype=np.int64) if c["refs_type"].startswith("ndarray") else list(c["refs"])
        r = infer.infer_multiple(c["cur"], refs)
        if len(refs) == 0:
            assert r is None                                                               # demo3_lcd.py:89,122
            continue
        overlaps, _ = r                                                                    # :118
        assert overlaps.dtype == np.float32 and overlaps.shape == ((len(refs),) if len(refs) > 1 else ())
        assert _.shape == (len(refs),) and _.dtype == np.int64
        if np.max(overlaps) > 0.3:                                                         # :119-120
            k = refs[np.argmax(overlaps)]
            assert k in refs
            closures += 1
        checked += 1
    assert checked == 83 and len(infer.feature_volumes) == 259
    infer.close()
