"""Seeded synthetic weights and inputs for parity tests, the smoke test and the benchmark (test / bench infrastructure: not part
of the product package -- it reads the reference-generated fixtures under tests/golden/).

The reference ships neither trained weights (`.gitignore:9`) nor KITTI sequences (`.gitignore:8`);
what it does ship are two scans (`data/scans/00000{0,1}.bin`) whose reference-generated range /
normal / intensity images are stored in `tests/golden/kitti_preprocess.npz`.  Everything here is
derived from those two scans and fixed seeds, so the GPU box can rebuild identical inputs.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import numpy as np

from overlapnet_amd import weights as W

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_PREPROCESS = os.path.join(REPO_ROOT, "tests", "golden", "kitti_preprocess.npz")

# network.yml model section of the reference (config/network.yml:64-82)
REFERENCE_MODEL_CFG = {
    "legsType": "360OutputkLegs",
    "overlap_head": "DeltaLayerConv1NetworkHead",
    "orientation_head": "CorrelationHead",
    "inputShape": [64, 900],
    "leg_output_width": 360,
    "strides_layer1": [2, 2],
    "additional_unsymmetric_layer3a": True,
}

# Gains found with the fp64 oracle so that leg features are O(1) (max ~12) and overlap logits do not collapse to
# sigmoid(0) (Glorot weights alone give logit ~ -0.04 +- 0.005).  With these gains the logits of the candidate pool
# sit in about [-0.8, 0.05]: a narrow range -- `make_trained_like_weights` below is the wide-range set.
_LEG_GAIN = 1.34
_DENSE_GAIN = 5.0
_DENSE_BIAS = {1: 1.1, 4: 4.6, 5: 1.7}


def channels_of(use_depth: bool, use_normals: bool, use_intensity: bool) -> int:
    return int(use_depth) + 3 * int(use_normals) + int(use_intensity)


def flags_of(channels: int) -> Tuple[bool, bool, bool]:
    """(use_depth, use_normals, use_intensity) for the channel counts the reference configs produce."""
    table = {1: (True, False, False), 4: (True, True, False), 5: (True, True, True), 3: (False, True, False),
             2: (True, False, True)}
    if channels not in table:
        raise ValueError("no use_* flag combination gives %d channels" % channels)
    return table[channels]


def make_test_weights(channels: int = 4, seed: int = 0, model_cfg: Optional[dict] = None) -> Dict[str, np.ndarray]:
    """Non-degenerate seeded weights (Glorot-uniform scaled per layer, non-zero biases)."""
    cfg = model_cfg or REFERENCE_MODEL_CFG
    gains = {l.name: _LEG_GAIN for l in W.leg_layers(channels, cfg)}
    gains["overlap_output"] = _DENSE_GAIN
    w = W.synthetic_weights(channels, cfg, seed=seed, kernel_gain=1.0, bias_scale=0.05, gains=gains)
    w["overlap_output/bias"] = np.array([_DENSE_BIAS.get(channels, 2.0)], np.float32)
    return w



def make_trained_like_weights(channels: int = 4, seed: int = 7, model_cfg: Optional[dict] = None) -> Dict[str, np.ndarray]:
    """Second seeded weight set with the dynamic range of a trained network rather than of an initialiser:
    leg kernels at 1.6 x Glorot (features up to ~50 on un-normalised depth images in metres), O(1) biases
    (uniform +-0.5) in every layer, Dense kernel at 10 x Glorot, Dense bias +6 -- overlap logits of the candidate
    pool then span about [-13, 7] (both sigmoid tails and the 0.3 loop-closure threshold region are populated)."""
    cfg = model_cfg or REFERENCE_MODEL_CFG
    gains = {l.name: 1.6 for l in W.leg_layers(channels, cfg)}
    gains.update({"c_conv1": 0.8, "c_conv2": 0.8, "c_conv3": 0.8, "overlap_output": 10.0})
    w = W.synthetic_weights(channels, cfg, seed=seed, kernel_gain=1.0, bias_scale=0.5, gains=gains)
    w["overlap_output/bias"] = np.array([6.0], np.float32)
    return w


WEIGHT_SETS = {"glorot": make_test_weights, "trained_like": make_trained_like_weights}


def load_fixture_images() -> Dict[str, np.ndarray]:
    """Reference-generated preprocessing outputs of the two shipped scans."""
    with np.load(GOLDEN_PREPROCESS) as z:
        return {k: z[k] for k in z.files}


def stack(depth, normal, intensity, flags: Tuple[bool, bool, bool]) -> np.ndarray:
    """Leg input in the reference's channel order depth | normals | intensity
    (ImagePairOverlapOrientationSequence.py:143-207)."""
    parts = []
    if flags[0]:
        parts.append(depth[..., None])
    if flags[1]:
        parts.append(normal)
    if flags[2]:
        parts.append(intensity[..., None])
    return np.concatenate(parts, axis=-1).astype(np.float32)


def candidate_images(n: int, channels: int = 4, seed: int = 1234, noise_m: float = 0.02,
                     fixture: Optional[Dict[str, np.ndarray]] = None) -> np.ndarray:
    """(n, 64, 900, C) synthetic scans per SURVEY.md section 8d: candidate i = fixture scan (i mod 2)
    circularly shifted by (37*i) mod 900 columns, N(0, noise_m) depth noise on valid pixels, invalid
    pixels stay -1; normals / intensity are shifted identically."""
    fx = fixture or load_fixture_images()
    flags = flags_of(channels)
    rng = np.random.default_rng(seed)
    out = np.empty((n, 64, 900, channels), np.float32)
    for i in range(n):
        s = i % 2
        shift = (37 * i) % 900
        d = np.roll(fx["range_%d" % s], shift, axis=1).copy()
        nm = np.roll(fx["normal_%d" % s], shift, axis=1)
        it = np.roll(fx["intensity_%d" % s], shift, axis=1)
        valid = d > 0
        if noise_m > 0:
            noise = rng.normal(0.0, noise_m, size=d.shape).astype(np.float32)
            d = np.where(valid, np.maximum(d + noise, np.float32(1e-3)), d).astype(np.float32)
        out[i] = stack(d, nm, it, flags)
    return out


def sweep_pool_images(pool: int, channels: int = 4, rank: int = 0, fixture: Optional[Dict[str, np.ndarray]] = None,
                      chunk: int = 128):
    """The candidate pool of the benchmark's 1-vs-`pool` sweep (BASELINE.json configs[1]) as a generator of
    (start, images) chunks: `candidate_images` per chunk with distinct seeds and one more column roll per chunk and
    rank, so that no two candidates of a pool (or of two ranks' pools) are the same scan.  bench.py, the parity test
    at scale and the script that produced its golden oracle outputs all draw the pool from here."""
    fx = fixture or load_fixture_images()
    for s in range(0, pool, chunk):
        n = min(chunk, pool - s)
        imgs = candidate_images(n, channels, seed=1234 + 7919 * rank + s, fixture=fx)
        yield s, np.ascontiguousarray(np.roll(imgs, (s * 37 + rank * 11) % 900, axis=2))


def sweep_query_image(channels: int = 4, fixture: Optional[Dict[str, np.ndarray]] = None) -> np.ndarray:
    """(1, 64, 900, C): the query scan of the benchmark sweep = the first shipped scan, unshifted."""
    fx = fixture or load_fixture_images()
    return stack(fx["range_0"], fx["normal_0"], fx["intensity_0"], flags_of(channels))[None]


# ---- transformed copies of the two shipped scans (tests/golden/preprocess_transformed.npz, bench.py's raw clouds) ----
N_TRANSFORMED = 24


def z_rotated(points: np.ndarray, cols: int) -> np.ndarray:
    """The scan rotated about z by `cols` image columns (2 pi cols / 900), float32, one rounded multiply / subtract / add per
    element exactly as bench.py does it on the device with torch (`c * x - s * y`, `s * x + c * y`)."""
    th = 2.0 * np.pi * cols / 900.0
    c, s = np.float32(np.cos(th)), np.float32(np.sin(th))
    q = np.array(points, dtype=np.float32, copy=True)
    x, y = points[:, 0].astype(np.float32), points[:, 1].astype(np.float32)
    q[:, 0] = c * x - s * y
    q[:, 1] = s * x + c * y
    return q


def transformed_cloud(fixture: Dict[str, np.ndarray], i: int) -> np.ndarray:
    """Cloud i (0 .. N_TRANSFORMED-1) of the preprocessing parity set: 0-11 = bench.py's raw clouds (scan i mod 2 rotated about z
    by 37 i mod 900 columns); 12-15 translations; 16-19 pitch / roll tilts; 20-23 rotation + tilt + translation.  Everything is
    evaluated elementwise (float64 products summed left to right, rounded once to float32; the z rotations in float32 as the
    bench does) so that every host produces the same bits."""
    base = fixture["points_%d" % (i % 2)]
    if i < 12:
        return z_rotated(base, (37 * i) % 900)
    k = i - 12
    shifts = [(1.5, -0.7, 0.2), (-2.25, 0.4, -0.15), (0.3, 3.1, 0.05), (-0.8, -1.9, 0.3)]
    tilts = [(2.0, 0.0), (-2.0, 0.0), (0.0, 3.0), (1.0, -3.0)]      # (pitch about y, roll about x) in degrees
    yaw_cols = 0
    t = (0.0, 0.0, 0.0)
    pitch = roll = 0.0
    if k < 4:
        t = shifts[k]
    elif k < 8:
        pitch, roll = tilts[k - 4]
    else:
        t = shifts[k - 8]
        pitch, roll = tilts[(k - 8 + 1) % 4]
        yaw_cols = 113 * (k - 7)
    p = base[:, :3].astype(np.float64)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    a = 2.0 * np.pi * yaw_cols / 900.0
    x, y = np.cos(a) * x - np.sin(a) * y, np.sin(a) * x + np.cos(a) * y
    b = np.deg2rad(pitch)
    x, z = np.cos(b) * x + np.sin(b) * z, -np.sin(b) * x + np.cos(b) * z
    g = np.deg2rad(roll)
    y, z = np.cos(g) * y - np.sin(g) * z, np.sin(g) * y + np.cos(g) * z
    q = np.array(base, dtype=np.float32, copy=True)
    q[:, 0] = (x + t[0]).astype(np.float32)
    q[:, 1] = (y + t[1]).astype(np.float32)
    q[:, 2] = (z + t[2]).astype(np.float32)
    return q


def fullstack_cloud(fixture: Dict[str, np.ndarray], i: int) -> np.ndarray:
    """Raw cloud i of bench.py's `fullstack` step (BASELINE configs[4] emulated): shipped scan (i mod 2) rotated about
    z by (37 i mod 900) image columns.  Clouds 0-11 are clouds 0-11 of the preprocessing parity set (`transformed_cloud`)."""
    return z_rotated(fixture["points_%d" % (i % 2)], (37 * i) % 900)


# other projection geometries of the preprocessing parity set: (cloud index, extra pitch about y in degrees, proj_H, proj_W, fov_up,
# fov_down, max_range)
GEOMETRY_CASES = [
    (3, 0.0, 32, 1800, 10.0, -30.0, 80),
    (17, 35.0, 64, 900, 60.0, -60.0, 50),   # steep: the cloud pitched by 35 degrees, |sin(pitch)| >= 0.5 for a third of the points
    (0, 0.0, 128, 2048, 3.0, -25.0, 120),
    (20, 0.0, 64, 1024, 2.0, -24.8, 50),
]


def geometry_cloud(fixture: Dict[str, np.ndarray], k: int) -> np.ndarray:
    """Input cloud of GEOMETRY_CASES[k]: a transformed cloud, optionally pitched further (float64 elementwise, rounded once)."""
    ci, pitch = GEOMETRY_CASES[k][0], GEOMETRY_CASES[k][1]
    q = transformed_cloud(fixture, ci)
    if pitch:
        b = np.deg2rad(pitch)
        x, z = q[:, 0].astype(np.float64), q[:, 2].astype(np.float64)
        q = q.copy()
        q[:, 0] = (np.cos(b) * x + np.sin(b) * z).astype(np.float32)
        q[:, 2] = (-np.sin(b) * x + np.cos(b) * z).astype(np.float32)
    return q
