"""Seeded synthetic weights and inputs for parity tests, the smoke test and the benchmark.

The reference ships neither trained weights (`.gitignore:9`) nor KITTI sequences (`.gitignore:8`);
what it does ship are two scans (`data/scans/00000{0,1}.bin`) whose reference-generated range /
normal / intensity images are stored in `tests/golden/kitti_preprocess.npz`.  Everything here is
derived from those two scans and fixed seeds, so the GPU box can rebuild identical inputs.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import numpy as np

from . import weights as W

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

# Gains found with the fp64 oracle so that leg features are O(1) and overlap logits are spread over
# roughly [-3, 3] instead of collapsing to sigmoid(0) (Glorot weights alone give logit ~ -0.04 +- 0.005).
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
