"""Generates the HDF5 fixtures that pin `overlapnet_amd/hdf5_lite.py` (run once, outputs committed).

Needs h5py, which the framework's own interpreter does not have; in the build container an Anaconda
interpreter does:   /opt/conda/bin/python3.9 tests/golden/make_hdf5_golden.py
(h5py 3.3.0 / HDF5 1.10.6 / numpy 1.26 there).  The expected values are stored next to the files as .npz so
that the tests need neither h5py nor this script.

keras_layout_small.weight reproduces, with small tensors, exactly what Keras 2.1.x `model.save` writes
(reference: src/two_heads/training.py:349; read back by name in src/two_heads/infer.py:117-120):
  /            attrs keras_version, backend, model_config, training_config      (fixed-length byte strings)
  /model_weights                attrs layer_names (array of S), backend, keras_version
  /model_weights/<layer>        attrs weight_names (array of S, empty for weight-less layers)
  /model_weights/<layer>/<layer>/kernel:0, bias:0     contiguous little-endian float32
  /optimizer_weights/...        ignored by the loader
"""
import json
import os
import sys

import h5py
import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(20260924)


def keras_like(path, shapes, weightless, optimizer=True):
    expected = {}
    with h5py.File(path, "w") as f:
        f.attrs["keras_version"] = "2.1.5".encode("utf8")
        f.attrs["backend"] = "tensorflow".encode("utf8")
        f.attrs["model_config"] = json.dumps({"class_name": "Model", "config": {"name": "model_1", "layers": [
            {"name": n} for n in list(weightless) + [k for k in shapes]]}}).encode("utf8")
        f.attrs["training_config"] = json.dumps({"optimizer_config": {"class_name": "Adagrad"}, "loss": "mse"}).encode("utf8")
        g = f.create_group("model_weights")
        # Keras keeps model.layers order: inputs, shared leg layers, head layers interleaved with weight-less ones
        order = []
        wl = list(weightless)
        for i, k in enumerate(shapes):
            if i % 3 == 0 and wl:
                order.append(wl.pop(0))
            order.append(k)
        order += wl
        g.attrs["layer_names"] = np.array([n.encode("utf8") for n in order])  # h5py 2.x: list of bytes -> fixed-length S array
        g.attrs["backend"] = "tensorflow".encode("utf8")
        g.attrs["keras_version"] = "2.1.5".encode("utf8")
        for n in order:
            lg = g.create_group(n)
            if n in shapes:
                names = [("%s/kernel:0" % n).encode("utf8"), ("%s/bias:0" % n).encode("utf8")]
                lg.attrs["weight_names"] = np.array(names)
                for nm, shp in zip(names, (shapes[n], (shapes[n][-1],))):
                    val = rng.standard_normal(shp).astype(np.float32)
                    d = lg.create_dataset(nm, val.shape, dtype=val.dtype)
                    d[:] = val
                    expected[n + "/" + nm.decode().split("/")[-1].split(":")[0]] = val
            else:
                lg.attrs["weight_names"] = []
        if optimizer:
            og = f.create_group("optimizer_weights")
            og.attrs["weight_names"] = np.array([b"training/Adagrad/Variable:0"])
            og.create_dataset("training/Adagrad/Variable:0", data=rng.standard_normal((3, 5)).astype(np.float32))
    return expected


def write_fixtures():
    small_shapes = {
        "s_conv1": (5, 15, 4, 16), "s_conv2": (3, 15, 16, 8), "s_conv3": (3, 5, 8, 8), "s_conv3a": (3, 4, 8, 8),
        "s_conv4": (2, 3, 8, 8), "s_conv5": (1, 3, 8, 8), "s_conv6": (1, 3, 8, 8), "s_conv7": (1, 3, 8, 8),
        "s_conv8": (1, 3, 8, 8), "s_conv9": (1, 3, 8, 8), "s_conv10": (1, 3, 8, 8),
        "c_conv1": (1, 15, 8, 4), "c_conv2": (15, 1, 4, 8), "c_conv3": (3, 3, 8, 16), "overlap_output": (400, 1),
    }
    weightless = ["input_1", "input_2", "reshape_1", "reshape_2", "lambda_1", "lambda_2", "lambda_3", "lambda_4",
                  "flatten_1", "range_padding2d_1", "normalized_correlation2d_1", "orientation_output"]
    exp = keras_like(os.path.join(here, "keras_layout_small.weight"), small_shapes, weightless)
    np.savez(os.path.join(here, "keras_layout_small_expected.npz"), **exp)

    # ---- feature coverage beyond the Keras layout --------------------------------------------------------------
    exp2 = {}
    with h5py.File(os.path.join(here, "hdf5_features.h5"), "w") as f:
        a = rng.standard_normal((37, 29)).astype(np.float32)
        f.create_dataset("chunked_gzip_shuffle", data=a, chunks=(8, 16), compression="gzip", shuffle=True, fletcher32=True)
        exp2["chunked_gzip_shuffle"] = a
        b = rng.integers(-1000, 1000, size=(5, 6, 7)).astype(np.int32)
        f.create_dataset("chunked_plain", data=b, chunks=(2, 3, 7))
        exp2["chunked_plain"] = b
        c = rng.standard_normal((4, 3))
        f.create_dataset("be_float64", data=c, dtype=">f8")
        exp2["be_float64"] = c
        f.create_dataset("scalar_f32", data=np.float32(3.25))
        exp2["scalar_f32"] = np.float32(3.25)
        d = rng.integers(0, 255, size=(11,)).astype(np.uint8)
        dcpl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
        dcpl.set_layout(h5py.h5d.COMPACT)
        sid = h5py.h5s.create_simple(d.shape)
        did = h5py.h5d.create(f.id, b"compact_u8", h5py.h5t.NATIVE_UINT8, sid, dcpl=dcpl)
        did.write(h5py.h5s.ALL, h5py.h5s.ALL, d)
        exp2["compact_u8"] = d
        f.create_dataset("never_written", shape=(3, 2), dtype="<f4")
        exp2["never_written"] = np.zeros((3, 2), np.float32)
        f.create_dataset("i64", data=np.arange(-5, 5, dtype=np.int64))
        exp2["i64"] = np.arange(-5, 5, dtype=np.int64)
        f.create_dataset("f16", data=np.linspace(-2, 2, 9).astype(np.float16))
        exp2["f16"] = np.linspace(-2, 2, 9).astype(np.float16)
        g = f.create_group("many")
        for i in range(70):  # > 2 * leaf K entries: several SNOD nodes below one B-tree node
            sg = g.create_group("member_%03d" % i)
            sg.attrs["index"] = np.int32(i)
        deep = f.create_group("a/b/c/d")
        deep.create_dataset("leaf", data=np.arange(6, dtype=np.float32).reshape(2, 3))
        exp2["a/b/c/d/leaf"] = np.arange(6, dtype=np.float32).reshape(2, 3)
        f.attrs["vlen_str"] = "variable length ä"          # h5py 3 str -> variable-length UTF-8
        f.attrs["vlen_list"] = np.array(["one", "two", "three"], dtype=h5py.string_dtype())
        f.attrs["fixed_bytes"] = np.bytes_(b"fixed")
        f.attrs["fixed_array"] = np.array([b"ab", b"cdef", b"g"])
        f.attrs["f64_vec"] = np.array([1.5, -2.5, 1e300])
        f.attrs["i8"] = np.int8(-3)
        f.attrs["empty"] = h5py.Empty("f4")
        big = f.create_group("many_attrs")          # object header continuation blocks
        for i in range(40):
            big.attrs["attr_%02d" % i] = np.arange(i, i + 4, dtype=np.float32)
    np.savez(os.path.join(here, "hdf5_features_expected.npz"), **{k.replace("/", "|"): v for k, v in exp2.items()})

    # ---- a file written with libver='latest' (superblock v3, version-2 object headers, compact links) ------------
    exp3 = {}
    with h5py.File(os.path.join(here, "hdf5_latest.h5"), "w", libver="latest") as f:
        g = f.create_group("model_weights")
        g.attrs["layer_names"] = [b"s_conv1", b"overlap_output"]
        for n, shp in (("s_conv1", (5, 15, 4, 16)), ("overlap_output", (40, 1))):
            lg = g.create_group(n)
            lg.attrs["weight_names"] = [("%s/kernel:0" % n).encode(), ("%s/bias:0" % n).encode()]
            for leaf, s in (("kernel:0", shp), ("bias:0", (shp[-1],))):
                v = rng.standard_normal(s).astype(np.float32)
                lg.create_dataset("%s/%s" % (n, leaf), data=v)
                exp3["%s/%s" % (n, leaf.split(":")[0])] = v
    np.savez(os.path.join(here, "hdf5_latest_expected.npz"), **exp3)


weightless = ["input_1", "input_2", "reshape_1", "reshape_2", "lambda_1", "lambda_2", "lambda_3", "lambda_4",
              "flatten_1", "range_padding2d_1", "normalized_correlation2d_1", "orientation_output"]
if "--full-only" not in sys.argv:
    write_fixtures()

if len(sys.argv) > 1:  # optional: a full-size model_geo-shaped file for a local timing / equality check (not committed)
    full = {
        "s_conv1": (5, 15, 4, 16), "s_conv2": (3, 15, 16, 32), "s_conv3": (3, 15, 32, 64), "s_conv3a": (3, 12, 64, 64),
        "s_conv4": (2, 9, 64, 128), "s_conv5": (1, 9, 128, 128), "s_conv6": (1, 9, 128, 128), "s_conv7": (1, 9, 128, 128),
        "s_conv8": (1, 7, 128, 128), "s_conv9": (1, 5, 128, 128), "s_conv10": (1, 3, 128, 128),
        "c_conv1": (1, 15, 128, 64), "c_conv2": (15, 1, 64, 128), "c_conv3": (3, 3, 128, 256), "overlap_output": (123904, 1),
    }
    e = keras_like(sys.argv[1], full, weightless)
    np.savez(sys.argv[1] + ".expected.npz", **e)
print("ok")
