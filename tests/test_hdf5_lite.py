"""The built-in HDF5 reader against fixtures written by real h5py 3.3 / HDF5 1.10.6
(tests/golden/make_hdf5_golden.py; expected values stored as .npz so neither h5py nor the script is needed)."""
import os

import numpy as np
import pytest

from overlapnet_amd import hdf5_lite as H
from overlapnet_amd import weights as W

G = os.path.join(os.path.dirname(__file__), "golden")


def test_keras_layout_by_name():
    exp = np.load(os.path.join(G, "keras_layout_small_expected.npz"))
    w = W.load_weights_file(os.path.join(G, "keras_layout_small.weight"))
    assert sorted(w) == sorted(exp.files) and len(w) == 30      # 15 layers x (kernel, bias); optimizer ignored
    for k in exp.files:
        assert w[k].dtype == np.float32 and np.array_equal(w[k], exp[k]), k


def test_keras_layout_structure():
    with H.File(os.path.join(G, "keras_layout_small.weight")) as f:
        assert f.keys() == ["model_weights", "optimizer_weights"]
        assert f.attrs["keras_version"] == b"2.1.5" and f.attrs["backend"] == b"tensorflow"
        assert f.attrs["model_config"].startswith(b'{"class_name": "Model"')
        mw = f["model_weights"]
        names = [n.decode() for n in mw.attrs["layer_names"]]
        assert len(names) == 27 and sorted(names) == mw.keys()
        assert len(mw["input_1"].attrs["weight_names"]) == 0 and len(mw["input_1"]) == 0
        assert list(mw["c_conv2"].attrs["weight_names"]) == [b"c_conv2/kernel:0", b"c_conv2/bias:0"]
        d = mw["c_conv2/c_conv2/kernel:0"]
        assert d.shape == (15, 1, 4, 8) and d.dtype == np.float32 and d.name == "/model_weights/c_conv2/c_conv2/kernel:0"
        assert "c_conv2/c_conv2/bias:0" in mw and "c_conv2/nope" not in mw
        assert f["/optimizer_weights/training/Adagrad/Variable:0"].shape == (3, 5)
        seen = []
        mw["s_conv1"].visititems(lambda n, o: seen.append(n))
        assert seen == ["s_conv1", "s_conv1/bias:0", "s_conv1/kernel:0"]


def test_feature_coverage():
    exp = np.load(os.path.join(G, "hdf5_features_expected.npz"))
    with H.File(os.path.join(G, "hdf5_features.h5")) as f:
        for k in exp.files:
            d = f[k.replace("|", "/")]
            v = d[()]
            assert d.shape == exp[k].shape and v.dtype == exp[k].dtype and np.array_equal(v, exp[k]), k
        a = f.attrs
        assert a["vlen_str"].decode("utf8") == "variable length ä"
        assert list(a["vlen_list"]) == [b"one", b"two", b"three"]
        assert a["fixed_bytes"] == b"fixed" and list(a["fixed_array"]) == [b"ab", b"cdef", b"g"]
        assert np.array_equal(a["f64_vec"], [1.5, -2.5, 1e300]) and a["i8"] == -3 and a["empty"] is None
        many = f["many"]                              # 70 links: several symbol-table nodes below the B-tree
        assert len(many) == 70 and many.keys() == ["member_%03d" % i for i in range(70)]
        assert all(many["member_%03d" % i].attrs["index"] == i for i in range(70))
        big = f["many_attrs"].attrs                   # object-header continuation blocks
        assert len(big) == 40 and all(np.array_equal(big["attr_%02d" % i], np.arange(i, i + 4)) for i in range(40))


def test_libver_latest_file():
    exp = np.load(os.path.join(G, "hdf5_latest_expected.npz"))
    w = W.load_weights_file(os.path.join(G, "hdf5_latest.h5"))
    assert sorted(w) == sorted(exp.files)
    for k in exp.files:
        assert np.array_equal(w[k], exp[k]), k


def test_errors_are_loud(tmp_path):
    src = open(os.path.join(G, "keras_layout_small.weight"), "rb").read()
    p = tmp_path / "cut.weight"
    p.write_bytes(src[: len(src) // 3])               # truncated download
    with pytest.raises(H.Hdf5Error, match="truncated"):
        W.load_weights_file(str(p))
    q = tmp_path / "junk.weight"
    q.write_bytes(b"not a weight file at all")
    with pytest.raises(Exception, match="unrecognised"):
        W.load_weights_file(str(q))
    with pytest.raises(H.Hdf5Error, match="read-only"):
        H.File(os.path.join(G, "hdf5_features.h5"), "w")
    with H.File(os.path.join(G, "hdf5_features.h5")) as f:
        with pytest.raises(KeyError):
            f["missing/thing"]


def test_full_shape_check_on_hdf5(tmp_path):
    # the small fixture has the layer names of the real network but reduced shapes: check_weights must say so
    w = W.load_weights_file(os.path.join(G, "keras_layout_small.weight"))
    with pytest.raises(ValueError, match="s_conv2/kernel"):
        W.check_weights(w, 4, {"additional_unsymmetric_layer3a": True, "strides_layer1": [2, 2]})


CONDA_PY = "/opt/conda/bin/python3.9"   # has h5py in the build container; absent on the GPU box -> skipped there


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="no interpreter with h5py on this machine")
def test_full_size_model_file_written_by_h5py(tmp_path):
    import subprocess

    path = str(tmp_path / "model_geo_shaped.weight")
    subprocess.run([CONDA_PY, os.path.join(G, "make_hdf5_golden.py"), path, "--full-only"], check=True,
                   capture_output=True, timeout=300)
    exp = np.load(path + ".expected.npz")
    w = W.load_weights_file(path)
    W.check_weights(w, 4, {"additional_unsymmetric_layer3a": True, "strides_layer1": [2, 2]})
    for k in exp.files:
        assert np.array_equal(w[k], exp[k]), k


def test_corrupted_files_fail_with_parser_errors_only(tmp_path):
    """Random byte damage and truncation of the Keras-layout fixture: every failure is an Hdf5Error / KeyError / the
    loader's own Exception -- no AttributeError, UnicodeDecodeError, IndexError, struct.error, recursion or hang."""
    import random
    src = open(os.path.join(G, "keras_layout_small.weight"), "rb").read()
    rnd = random.Random(7)
    p = tmp_path / "f.weight"
    outcomes = {"ok": 0, "error": 0}
    for it in range(150):
        b = bytearray(src)
        if it % 3 == 0:
            for _ in range(rnd.randint(1, 8)):
                b[rnd.randrange(len(b))] = rnd.randrange(256)
        elif it % 3 == 1:
            b = b[: rnd.randrange(64, len(b))]
        else:
            pos = rnd.randrange(0, 6000)
            for k in range(8):
                b[(pos + k) % len(b)] = rnd.choice([0, 255, rnd.randrange(256)])
        p.write_bytes(bytes(b))
        try:
            W.load_weights_file(str(p))
            outcomes["ok"] += 1
        except (H.Hdf5Error, KeyError) as e:
            outcomes["error"] += 1
        except Exception as e:                      # the loader's own messages only
            assert type(e) is Exception, (type(e), e)
            outcomes["error"] += 1
    assert outcomes["error"] > 30 and outcomes["ok"] > 0
