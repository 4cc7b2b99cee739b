"""Layer table and weight containers for the OverlapNet inference path.

Mirrors the layer names / kernel shapes that the reference builds with Keras
(`src/two_heads/generateNet.py:161-214` for the leg, `:96-114` for the Delta head), so that a
weight file keyed by the reference's layer names loads by name exactly as
`Infer.__init__` does (`src/two_heads/infer.py:117-120`, `load_weights(..., by_name=True)`).

Kernel layout is Keras' `(kh, kw, cin, cout)`; Dense kernel is `(in, out)`.

Native container: `.npz` with keys `<layer>/kernel` and `<layer>/bias`.
A Keras HDF5 file (`model_geo.weight`, written by `training.py:349`) is read directly through the
dependency-free parser in `hdf5_lite.py` (h5py is not in the ROCm image).
"""
from __future__ import annotations

import os
from typing import Dict, List, NamedTuple, Sequence, Tuple

import numpy as np


class ConvSpec(NamedTuple):
    name: str
    kh: int
    kw: int
    cin: int
    cout: int
    sh: int
    sw: int
    relu: bool


def leg_layers(in_channels: int, model_cfg: dict | None = None) -> List[ConvSpec]:
    """The `360OutputkLegs` trunk (reference `generateNet.py:119-219`; the `...Fixed` variant
    `:222-324` is the same topology with `trainable=False`, identical at inference).

    Config keys honoured (same defaults as `generateNet.py:143-146`):
      strides_layer1 (default (2,2)), additional_unsymmetric_layer3a (default False).
    """
    cfg = model_cfg or {}
    s1 = tuple(cfg.get("strides_layer1", (2, 2)))
    if len(s1) != 2:
        raise ValueError("strides_layer1 must have two entries")
    layers = [
        ConvSpec("s_conv1", 5, 15, in_channels, 16, int(s1[0]), int(s1[1]), True),
        ConvSpec("s_conv2", 3, 15, 16, 32, 2, 1, True),
        ConvSpec("s_conv3", 3, 15, 32, 64, 2, 1, True),
    ]
    if cfg.get("additional_unsymmetric_layer3a", False):
        layers.append(ConvSpec("s_conv3a", 3, 12, 64, 64, 2, 1, True))
    layers += [
        ConvSpec("s_conv4", 2, 9, 64, 128, 2, 1, True),
        ConvSpec("s_conv5", 1, 9, 128, 128, 1, 1, True),
        ConvSpec("s_conv6", 1, 9, 128, 128, 1, 1, True),
        ConvSpec("s_conv7", 1, 9, 128, 128, 1, 1, True),
        ConvSpec("s_conv8", 1, 7, 128, 128, 1, 1, True),
        ConvSpec("s_conv9", 1, 5, 128, 128, 1, 1, True),
        ConvSpec("s_conv10", 1, 3, 128, 128, 1, 1, True),
    ]
    return layers


def leg_output_shape(h: int, w: int, layers: Sequence[ConvSpec]) -> Tuple[int, int, int]:
    """'valid' convolution shape chain (Keras default padding, generateNet.py:161)."""
    c = layers[0].cin
    for l in layers:
        if h < l.kh or w < l.kw:
            raise ValueError("input too small for layer %s" % l.name)
        h = (h - l.kh) // l.sh + 1
        w = (w - l.kw) // l.sw + 1
        c = l.cout
    return h, w, c


def head_layers(feat_channels: int = 128, conv1size: int = 15) -> List[ConvSpec]:
    """Delta head convolutions (reference `generateNet.py:96-110`). c_conv1 is *linear*."""
    s = int(conv1size)
    return [
        ConvSpec("c_conv1", 1, s, feat_channels, 64, 1, s, False),
        ConvSpec("c_conv2", s, 1, 64, 128, s, 1, True),
        ConvSpec("c_conv3", 3, 3, 128, 256, 1, 1, True),
    ]


def dense_in_features(feat_w: int = 360, conv1size: int = 15) -> int:
    g = feat_w // conv1size
    return (g - 2) * (g - 2) * 256


def expected_shapes(in_channels: int, model_cfg: dict | None = None, feat_w: int = 360) -> Dict[str, Tuple[int, ...]]:
    cfg = model_cfg or {}
    s = int(cfg.get("conv1NetworkHead_conv1size", 15))
    shapes: Dict[str, Tuple[int, ...]] = {}
    for l in leg_layers(in_channels, cfg) + head_layers(128, s):
        shapes[l.name + "/kernel"] = (l.kh, l.kw, l.cin, l.cout)
        shapes[l.name + "/bias"] = (l.cout,)
    shapes["overlap_output/kernel"] = (dense_in_features(feat_w, s), 1)
    shapes["overlap_output/bias"] = (1,)
    return shapes


def _glorot_uniform(rng: np.random.Generator, shape: Tuple[int, ...]) -> np.ndarray:
    """Keras' default kernel initialiser (glorot_uniform): U(-l, l), l = sqrt(6/(fan_in+fan_out))."""
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def keras_default_init(in_channels: int, model_cfg: dict | None = None, seed: int = 0) -> Dict[str, np.ndarray]:
    """What the reference is left with when `pretrained_weightsfilename` is empty
    (`infer.py:121-122`): Keras default init = Glorot-uniform kernels, zero biases."""
    rng = np.random.default_rng(seed)
    out = {}
    for k, shp in expected_shapes(in_channels, model_cfg).items():
        out[k] = _glorot_uniform(rng, shp) if k.endswith("/kernel") else np.zeros(shp, np.float32)
    return out


def synthetic_weights(in_channels: int, model_cfg: dict | None = None, seed: int = 0,
                      kernel_gain: float = 1.0, bias_scale: float = 0.05,
                      gains: Dict[str, float] | None = None) -> Dict[str, np.ndarray]:
    """Seeded synthetic weights for parity tests and the benchmark (no trained weights ship with the
    reference: `.gitignore:9`). Glorot-uniform kernels scaled by `kernel_gain` (per-layer overrides
    in `gains`), and small NON-zero biases so that the bias path of every kernel is exercised."""
    rng = np.random.default_rng(seed)
    out = {}
    g = gains or {}
    for k, shp in expected_shapes(in_channels, model_cfg).items():
        layer = k.split("/")[0]
        if k.endswith("/kernel"):
            out[k] = (_glorot_uniform(rng, shp) * np.float32(g.get(layer, kernel_gain))).astype(np.float32)
        else:
            out[k] = rng.uniform(-bias_scale, bias_scale, size=shp).astype(np.float32)
    return out


def check_weights(weights: Dict[str, np.ndarray], in_channels: int, model_cfg: dict | None = None) -> None:
    exp = expected_shapes(in_channels, model_cfg)
    for k, shp in exp.items():
        if k not in weights:
            raise KeyError("weight '%s' missing" % k)
        if tuple(weights[k].shape) != shp:
            raise ValueError("weight '%s' has shape %s, expected %s" % (k, tuple(weights[k].shape), shp))


def save_npz(path: str, weights: Dict[str, np.ndarray]) -> None:
    np.savez(path, **{k: np.asarray(v, np.float32) for k, v in weights.items()})


def load_npz(path: str) -> Dict[str, np.ndarray]:
    with np.load(path) as z:
        return {k: np.asarray(z[k], np.float32) for k in z.files}


def load_keras_hdf5(path: str) -> Dict[str, np.ndarray]:
    """Read a Keras 2.1.x full-model (`model.save`, reference `training.py:349`) or weights-only HDF5 file by
    layer name, the way `load_weights(..., by_name=True)` does (`infer.py:119-120`).

    Layout: `[model_weights/]<layer>/<weight name>` with the layer's `weight_names` attribute listing e.g.
    `s_conv1/kernel:0`, `s_conv1/bias:0` (so datasets sit at `<layer>/<layer>/kernel:0`).  Parsed by the
    built-in reader `hdf5_lite` (no h5py needed); layers without weights and the `optimizer_weights`
    group are skipped."""
    from . import hdf5_lite

    out: Dict[str, np.ndarray] = {}
    with hdf5_lite.File(path) as f:
        root = f["model_weights"] if "model_weights" in f else f
        names = root.attrs.get("layer_names")
        layers = [n.decode("utf8", "replace") if isinstance(n, bytes) else str(n) for n in np.asarray(names).ravel()] \
            if names is not None else root.keys()
        for layer in layers:
            if layer not in root:
                raise Exception("weight file '%s': layer '%s' listed in layer_names has no group" % (path, layer))
            grp = root[layer]
            wnames = grp.attrs.get("weight_names")
            if wnames is None:  # no attribute: walk the group
                found: List[str] = []
                grp.visititems(lambda name, obj: found.append(name) if isinstance(obj, hdf5_lite.Dataset) else None)
                wnames = found
            for wn in np.asarray(wnames).ravel():
                wn = wn.decode("utf8", "replace") if isinstance(wn, bytes) else str(wn)
                leaf = wn.split("/")[-1].split(":")[0]
                if leaf in ("kernel", "bias"):
                    out["%s/%s" % (layer, leaf)] = np.ascontiguousarray(grp[wn][()], dtype=np.float32)
    if not out:
        raise Exception("weight file '%s' holds no kernel/bias datasets" % path)
    return out


def load_weights_file(path: str) -> Dict[str, np.ndarray]:
    if not os.path.isfile(path):
        raise Exception("weight file not found: %s" % path)
    with open(path, "rb") as f:
        magic = f.read(8)
    if magic[:4] == b"PK\x03\x04":  # zip container = npz
        return load_npz(path)
    if magic == b"\x89HDF\r\n\x1a\n":
        return load_keras_hdf5(path)
    raise Exception("unrecognised weight file format: %s" % path)


if __name__ == "__main__":  # tiny converter CLI: HDF5 -> npz
    import sys

    if len(sys.argv) != 3:
        print("usage: python -m overlapnet_amd.weights model_geo.weight model_geo.npz")
        sys.exit(2)
    save_npz(sys.argv[2], load_weights_file(sys.argv[1]))
    print("wrote", sys.argv[2])
