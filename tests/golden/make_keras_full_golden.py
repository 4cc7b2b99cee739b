"""Full-size (C = 4) Keras-layout weight file for the GPU test of `Infer(config)` with `pretrained_weightsfilename`
(reference: infer.py:117-120 loads `model_geo.weight`, written by training.py:349) -> tests/golden/keras_layout_full_c4.weight.

Needs h5py:   /opt/conda/bin/python3.9 tests/golden/make_keras_full_golden.py      (h5py 3.3.0 in the build container)

Same group / attribute structure as Keras 2.1.x `model.save` (see make_hdf5_golden.py) with the real layer shapes.  The 1.77 M
weights are the seeded test weights rounded to multiples of 2^-9 (about 40 distinct values per layer) and stored chunked +
shuffle + gzip, so that the committed file is ~1 MB instead of 7 MB; the dense kernel keeps its 5x gain and bias.  The test
reads the file with the framework's own parser on both sides (HIP path and oracle), so the rounding is immaterial."""
import json
import os
import sys

import h5py
import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from tools import synthetic as S  # noqa: E402

w = S.make_test_weights(4, seed=3)
q = {k: (np.round(v * 512.0) / 512.0).astype(np.float32) for k, v in w.items()}
layers = [k.split("/")[0] for k in w if k.endswith("/kernel")]
weightless = ["input_1", "input_2", "reshape_1", "reshape_2", "lambda_1", "lambda_2", "lambda_3", "lambda_4",
              "flatten_1", "range_padding2d_1", "normalized_correlation2d_1", "orientation_output"]
order = []
wl = list(weightless)
for i, k in enumerate(layers):
    if i % 3 == 0 and wl:
        order.append(wl.pop(0))
    order.append(k)
order += wl
path = os.path.join(here, "keras_layout_full_c4.weight")
with h5py.File(path, "w") as f:
    f.attrs["keras_version"] = "2.1.5".encode("utf8")
    f.attrs["backend"] = "tensorflow".encode("utf8")
    f.attrs["model_config"] = json.dumps({"class_name": "Model", "config": {"name": "model_1", "layers": [{"name": n} for n in order]}}).encode("utf8")
    g = f.create_group("model_weights")
    g.attrs["layer_names"] = np.array([n.encode("utf8") for n in order])
    g.attrs["backend"] = "tensorflow".encode("utf8")
    g.attrs["keras_version"] = "2.1.5".encode("utf8")
    for n in order:
        lg = g.create_group(n)
        if n in layers:
            names = [("%s/kernel:0" % n).encode("utf8"), ("%s/bias:0" % n).encode("utf8")]
            lg.attrs["weight_names"] = np.array(names)
            for nm, key in zip(names, (n + "/kernel", n + "/bias")):
                val = q[key]
                if val.size >= 1024:
                    lg.create_dataset(nm, data=val, chunks=True, compression="gzip", compression_opts=9, shuffle=True)
                else:
                    lg.create_dataset(nm, data=val)
        else:
            lg.attrs["weight_names"] = []
np.savez_compressed(os.path.join(here, "keras_layout_full_c4_checksums.npz"),
                    **{k: np.array([float(v.astype(np.float64).sum()), float(np.abs(v).astype(np.float64).sum())]) for k, v in q.items()})
print("wrote", path, os.path.getsize(path), "bytes")
