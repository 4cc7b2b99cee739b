"""CPU ORACLE for the OverlapNet inference hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.
The product (`overlapnet_amd/`) never imports it and fails loudly without its HIP library.

It is a from-scratch restatement (NumPy + PyTorch-CPU) of what the reference computes on this path;
every function cites the reference lines it follows (paths relative to the reference repo root).

PARITY STATUS
  * Preprocessing (`range_projection`, `gen_normal_map`): PINNED.  `tests/golden/kitti_preprocess.npz`
    was produced by running the reference's own `src/utils/utils.py` in the build container and was
    asserted equal to the `.npy` outputs the reference ships (`data/preprocess_data_demo/...`);
    `tests/golden/preprocess_transformed.npz` holds the reference's outputs for 24 rotated / translated / tilted copies of the
    two scans; `tests/test_oracle_preprocess.py` checks this module against both bit-for-bit.  The float32 `arctan2` /
    `arcsin` of `utils.py:86-87` live in NumPy -> Intel SVML on the generating machine: restated in oracle/svml_f32.c and
    pinned against NumPy itself (tests/test_oracle_svml.py).
  * Neural path (leg, Delta head, correlation head): PARITY UNPINNED.  The arithmetic lives in
    TensorFlow/Keras (requirements.txt:4-5: tensorflow-gpu==2.5.2, keras==2.1.5), neither of which is
    installable here, the pretrained `model_geo.weight` is not in the tree (.gitignore:9), and the
    reference has no test that pins a numeric result at this boundary.  The restatement is anchored on
    the reference's layer definitions, on the two known-answer vectors its sources carry
    (`RangePadding2D.py:5`, the ramp demo `NormalizedCorrelation2D.py:112-144`) and on a second,
    literal formulation of each op kept in this file (`*_literal`) that the fast one is tested against.

Conventions honoured (Keras 2.1 / TF defaults): channels_last, padding='valid', use_bias=True,
cross-correlation (no kernel flip), kernel layout (kh,kw,cin,cout), Flatten row-major over (H,W,C).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

try:  # torch is only the conv engine of the oracle; numpy literal forms below do not need it
    import torch
    import torch.nn.functional as F
except Exception:  # pragma: no cover
    torch = None
    F = None

F32 = np.float32
F64 = np.float64

# --------------------------------------------------------------------------------------------------
# Preprocessing
# --------------------------------------------------------------------------------------------------


def range_projection(points: np.ndarray, fov_up: float = 3.0, fov_down: float = -25.0, proj_H: int = 64,
                     proj_W: int = 900, max_range: float = 50.0, trig64: bool = False, trig: Optional[str] = None):
    """Spherical projection, reference `src/utils/utils.py:59-134`.

    depth = ||xyz||_2 in float32 (`:75`), keep 0 < depth < max_range (`:76-77`), yaw = -atan2(y,x),
    pitch = asin(z/depth) (`:86-87`), proj_x = 0.5(yaw/pi+1)W, proj_y = (1-(pitch+|fov_down|)/fov)H
    (`:90-95`), floor + clamp (`:98-104`).  The reference then sorts by decreasing depth and scatters
    (`:107-132`) so the NEAREST point wins a pixel; here that is restated as a per-pixel minimum over the
    key (depth bits, kept-point index) -- the formulation the HIP kernel uses with a 64-bit atomicMin.
    Equal-depth ties are undefined in the reference (unstable argsort); lowest index wins here.
    Returns (range (H,W) f32, vertex (H,W,4) f32, intensity (H,W) f32, idx (H,W) i32); empty = -1.

    `trig` selects how the two float32 angle functions of `:86-87` are evaluated:
      'svml'  (default) the restatement of the SVML kernels NumPy's float32 arctan2 / arcsin run on AVX512_SKX x86-64 CPUs
              (oracle/svml_f32.c: same bits as NumPy on the machine the golden vectors come from, on any host; what the HIP
              kernel computes since round 4);
      'numpy' this host's NumPy float32 functions, literally what the reference calls (CPU-dependent: SVML, or libm elsewhere);
      'f64'   float64 functions rounded once to float32 (the HIP kernel of rounds 1-3; differs from the reference on a few
              pixels per million points).  `trig64=True` is the old spelling of 'f64'.
    """
    pts = np.ascontiguousarray(points, dtype=F32).reshape(-1, 4)
    up = fov_up / 180.0 * np.pi
    down = fov_down / 180.0 * np.pi
    fov = abs(down) + abs(up)
    x, y, z, inten = pts[:, 0], pts[:, 1], pts[:, 2], pts[:, 3]
    depth = np.sqrt((x * x + y * y) + z * z)  # float32, same association as np.linalg.norm(axis=1)
    keep = (depth > 0) & (depth < max_range)
    x, y, z, inten, depth = x[keep], y[keep], z[keep], inten[keep], depth[keep]
    trig = trig or ("f64" if trig64 else "svml")
    if trig == "f64":
        yaw = (-np.arctan2(y.astype(F64), x.astype(F64))).astype(F32)
        pitch = np.arcsin((z / depth).astype(F64)).astype(F32)
    elif trig == "numpy":
        yaw = -np.arctan2(y, x)
        pitch = np.arcsin(z / depth)
    elif trig == "svml":
        from oracle import build_oracle
        yaw = -build_oracle.svml_arctan2(y, x)
        pitch = build_oracle.svml_arcsin(z / depth)
    else:
        raise ValueError("trig must be 'svml', 'numpy' or 'f64'")
    px = F32(0.5) * (yaw / F32(np.pi) + F32(1.0))
    py = F32(1.0) - (pitch + F32(abs(down))) / F32(fov)
    px = px * F32(proj_W)
    py = py * F32(proj_H)
    px = np.maximum(0, np.minimum(proj_W - 1, np.floor(px))).astype(np.int64)
    py = np.maximum(0, np.minimum(proj_H - 1, np.floor(py))).astype(np.int64)

    n = depth.shape[0]
    key = (depth.view(np.uint32).astype(np.uint64) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
    empty = np.iinfo(np.uint64).max
    best = np.full(proj_H * proj_W, empty, dtype=np.uint64)
    np.minimum.at(best, py * proj_W + px, key)
    valid = best != empty
    win = (best[valid] & np.uint64(0xFFFFFFFF)).astype(np.int64)

    rng = np.full(proj_H * proj_W, -1, F32)
    vtx = np.full((proj_H * proj_W, 4), -1, F32)
    itn = np.full(proj_H * proj_W, -1, F32)
    idx = np.full(proj_H * proj_W, -1, np.int32)
    rng[valid] = depth[win]
    itn[valid] = inten[win]
    idx[valid] = win
    vtx[valid, 0] = x[win]
    vtx[valid, 1] = y[win]
    vtx[valid, 2] = z[win]
    vtx[valid, 3] = 1
    return (rng.reshape(proj_H, proj_W), vtx.reshape(proj_H, proj_W, 4), itn.reshape(proj_H, proj_W),
            idx.reshape(proj_H, proj_W))


def _norm3_like_numpy(a: np.ndarray) -> np.ndarray:
    """`np.linalg.norm` of a float32 3-vector as the reference evaluates it (`utils.py:165-166,169`):
    sqrt(x.dot(x)) where OpenBLAS' sdot rounds each product to float32 and accumulates the <32-element
    tail in a double, returning a float32.  Found by matching the shipped normal fixtures bit-for-bit."""
    x, y, z = a[..., 0], a[..., 1], a[..., 2]
    s = ((x * x).astype(F64) + (y * y).astype(F64)) + (z * z).astype(F64)
    return np.sqrt(s.astype(F32))


def gen_normal_map(current_range: np.ndarray, current_vertex: np.ndarray, proj_H: int = 64,
                   proj_W: int = 900) -> np.ndarray:
    """Normal image, reference `src/utils/utils.py:137-186` (vectorised restatement of its double loop).

    For y < H-1: p = vertex(y,x), u = vertex(y, x+1 wrapped) (`:155`, `wrap` `:178-186`), v = vertex(y+1,x);
    needs range > 0 at all three (`:153,157,162`); n = normalize(cross(normalize(v-p), normalize(u-p)))
    (`:165-173`), kept only if |w| > 0 (NaNs from zero-length edges fail that test, `:170`).
    Last row and every rejected pixel stay -1 (`:146,150`).
    """
    rng = np.asarray(current_range, F32)
    p = np.asarray(current_vertex, F32)[:, :, :3]
    u = np.roll(p, -1, axis=1)
    ud = np.roll(rng, -1, axis=1)
    v = np.zeros_like(p)
    v[:-1] = p[1:]
    vd = np.full_like(rng, -1)
    vd[:-1] = rng[1:]
    ok = (rng > 0) & (ud > 0) & (vd > 0)
    ok[-1, :] = False
    out = np.full((proj_H, proj_W, 3), -1, F32)
    with np.errstate(all="ignore"):
        du = u - p
        dv = v - p
        un = du / _norm3_like_numpy(du)[..., None]
        vn = dv / _norm3_like_numpy(dv)[..., None]
        a, b = vn, un  # np.cross(v_norm, u_norm), utils.py:168
        w = np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                      a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                      a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)
        nw = _norm3_like_numpy(w)
        good = ok & (nw > 0)
        out[good] = (w / nw[..., None])[good]
    return out


def gen_normal_map_literal(current_range: np.ndarray, current_vertex: np.ndarray, proj_H: int = 64,
                           proj_W: int = 900) -> np.ndarray:
    """The normal image the way the reference computes it (`src/utils/utils.py:137-186`): one pixel at a time in a Python double
    loop over columns and rows 0 .. H-2, NumPy calls on 3-vectors inside (`np.linalg.norm`, `np.cross`).  Cross-checks the
    vectorised `gen_normal_map` (same values: `tests/test_oracle_preprocess.py`) and is what `bench.py` times as the CPU baseline of the
    preprocessing stage "as shipped" (single thread, ~1 s per scan)."""
    rng = np.asarray(current_range, F32)
    vtx = np.asarray(current_vertex, F32)
    out = np.full((proj_H, proj_W, 3), -1, F32)
    for col in range(proj_W):
        right = col + 1 if col + 1 < proj_W else col + 1 - proj_W          # `wrap`, utils.py:178-186
        for row in range(proj_H - 1):                                      # the last row keeps -1 (utils.py:150)
            if not rng[row, col] > 0:
                continue
            if not rng[row, right] > 0 or not rng[row + 1, col] > 0:
                continue
            here = vtx[row, col, :3]
            to_right = vtx[row, right, :3] - here
            to_lower = vtx[row + 1, col, :3] - here
            with np.errstate(all="ignore"):
                u_n = to_right / np.linalg.norm(to_right)
                v_n = to_lower / np.linalg.norm(to_lower)
                w = np.cross(v_n, u_n)                                     # utils.py:168
                length = np.linalg.norm(w)
            if length > 0:                                                 # NaN fails the test (utils.py:170)
                out[row, col] = w / length
    return out


def stack_channels(depth: Optional[np.ndarray], normal: Optional[np.ndarray],
                   intensity: Optional[np.ndarray], probs: Optional[np.ndarray] = None) -> np.ndarray:
    """Channel stacking of one leg input, reference
    `src/two_heads/ImagePairOverlapOrientationSequence.py:143-207`:
    order depth(1) -> normals(3) -> class probabilities(20|3) -> intensity(1); raw values, no
    normalisation.  Returns (H, W, C) float32."""
    parts = []
    if depth is not None:
        parts.append(np.asarray(depth, F32)[..., None])
    if normal is not None:
        parts.append(np.asarray(normal, F32))
    if probs is not None:
        parts.append(np.asarray(probs, F32))
    if intensity is not None:
        parts.append(np.asarray(intensity, F32)[..., None])
    return np.concatenate(parts, axis=-1)


# --------------------------------------------------------------------------------------------------
# Network -- fast forms (PyTorch-CPU as the conv engine)
# --------------------------------------------------------------------------------------------------

# (name, (stride_h, stride_w), relu) -- kernel sizes come from the weight shapes themselves.
_LEG_TAIL = [("s_conv4", (2, 1)), ("s_conv5", (1, 1)), ("s_conv6", (1, 1)), ("s_conv7", (1, 1)),
             ("s_conv8", (1, 1)), ("s_conv9", (1, 1)), ("s_conv10", (1, 1))]


def _leg_plan(model_cfg: Optional[dict]) -> List[Tuple[str, Tuple[int, int]]]:
    """Layer order/strides of `generate360OutputkLegs`, reference `generateNet.py:161-214`:
    s_conv1 stride = config strides_layer1 (default (2,2), `:143-144`), s_conv2/3 stride (2,1),
    optional s_conv3a stride (2,1) when additional_unsymmetric_layer3a (`:145-146,178-182`),
    s_conv4 stride (2,1), s_conv5..10 stride 1.  Every layer: valid padding, bias, ReLU."""
    cfg = model_cfg or {}
    s1 = tuple(int(v) for v in cfg.get("strides_layer1", (2, 2)))
    plan = [("s_conv1", s1), ("s_conv2", (2, 1)), ("s_conv3", (2, 1))]
    if cfg.get("additional_unsymmetric_layer3a", False):
        plan.append(("s_conv3a", (2, 1)))
    return plan + _LEG_TAIL


def _t(a, dtype):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype)


def _conv_valid(x_nchw, kernel_hwio: np.ndarray, bias: np.ndarray, stride, relu: bool, dtype):
    w = _t(kernel_hwio, dtype).permute(3, 2, 0, 1).contiguous()  # (kh,kw,cin,cout) -> (cout,cin,kh,kw)
    y = F.conv2d(x_nchw, w, _t(bias, dtype), stride=stride, padding=0)
    return torch.relu(y) if relu else y


def leg_forward(images_nhwc: np.ndarray, weights: Dict[str, np.ndarray], model_cfg: Optional[dict] = None,
                dtype=F64) -> np.ndarray:
    """Shared conv leg, reference `generateNet.py:119-219`; input (n,64,900,C) -> (n,1,360,128)."""
    tdt = torch.float64 if dtype == F64 else torch.float32
    x = _t(images_nhwc, tdt).permute(0, 3, 1, 2).contiguous()
    for name, stride in _leg_plan(model_cfg):
        x = _conv_valid(x, weights[name + "/kernel"], weights[name + "/bias"], stride, True, tdt)
    return x.permute(0, 2, 3, 1).contiguous().numpy()


def delta_head_forward(feat_l: np.ndarray, feat_r: np.ndarray, weights: Dict[str, np.ndarray],
                       conv1size: int = 15, dtype=F64, return_intermediates: bool = False):
    """Delta (overlap) head, reference `generateNet.py:15-61` (DeltaLayer) and `:64-116`.

    feat_l, feat_r: (n, 1, 360, 128).  diff[b,i,j,c] = |l[b,i,c] - r[b,j,c]| (`:48-59`), then
    c_conv1 (1 x s, stride (1,s), LINEAR, `:96-100`), c_conv2 (s x 1, stride (s,1), ReLU, `:102-106`),
    c_conv3 (3x3, ReLU, `:108-110`), Flatten over (H,W,C) (`:112`), Dense(1) sigmoid (`:114`).
    The diff tensor is materialised per sample exactly as the reference graph does.
    Returns (overlap (n,), logit (n,)) [+ dict of intermediates for the last sample].
    """
    tdt = torch.float64 if dtype == F64 else torch.float32
    s = int(conv1size)
    n = feat_l.shape[0]
    wd = _t(weights["overlap_output/kernel"], tdt).reshape(-1)
    bd = _t(weights["overlap_output/bias"], tdt).reshape(())
    logits = np.zeros(n, dtype)
    inter = {}
    for b in range(n):
        l = _t(feat_l[b].reshape(-1, feat_l.shape[-1]), tdt)  # (360,128), row-major reshape (`:48-49`)
        r = _t(feat_r[b].reshape(-1, feat_r.shape[-1]), tdt)
        diff = (l[:, None, :] - r[None, :, :]).abs()  # (i, j, c)
        x = diff.permute(2, 0, 1).unsqueeze(0)  # NCHW: H = i (left), W = j (right)
        o1 = _conv_valid(x, weights["c_conv1/kernel"], weights["c_conv1/bias"], (1, s), False, tdt)
        o2 = _conv_valid(o1, weights["c_conv2/kernel"], weights["c_conv2/bias"], (s, 1), True, tdt)
        o3 = _conv_valid(o2, weights["c_conv3/kernel"], weights["c_conv3/bias"], (1, 1), True, tdt)
        flat = o3[0].permute(1, 2, 0).reshape(-1)  # (H,W,C) row-major, Keras Flatten
        logits[b] = float((flat * wd).sum() + bd)
        if return_intermediates and b == n - 1:
            inter = {"o1": o1[0].permute(1, 2, 0).numpy(), "o2": o2[0].permute(1, 2, 0).numpy(),
                     "o3": o3[0].permute(1, 2, 0).numpy()}
    with np.errstate(over="ignore"):   # exp(+large) = inf -> sigmoid 0, as Keras' sigmoid saturates
        overlap = 1.0 / (1.0 + np.exp(-logits.astype(F64)))
    overlap = overlap.astype(dtype)
    if return_intermediates:
        return overlap, logits, inter
    return overlap, logits


def correlation_head_forward(feat_l: np.ndarray, feat_r: np.ndarray, dtype=F64) -> np.ndarray:
    """Correlation (yaw) head with normalize='none', reference `generateNet.py:327-354`,
    `NormalizedCorrelation2D.py:43-109`, `RangePadding2D.py:31-38`.

    pad_l[m] = l[(m + W/2) mod W], m in [0, 2W-1)  (padding = W//2, `NormalizedCorrelation2D.py:77`,
    concat [x[p:], x, x[:p-1]] `RangePadding2D.py:34`); per sample valid conv of pad_l with r as the
    kernel (`NormalizedCorrelation2D.py:100-105`):
        corr[k] = sum_{j<W} sum_c l[(k + j + W/2) mod W, c] * r[j, c],   k in [0, W).
    Evaluated here through the Gram matrix G = l r^T and its wrapped diagonals.  Returns (n, W)."""
    n, _, W, C = feat_l.shape
    out = np.zeros((n, W), dtype)
    jj = np.arange(W)
    for b in range(n):
        G = feat_l[b, 0].astype(dtype) @ feat_r[b, 0].astype(dtype).T  # G[i, j]
        for k in range(W):
            out[b, k] = G[(k + jj + W // 2) % W, jj].sum()
    return out


def yaw_from_orientation(orientation: np.ndarray) -> np.ndarray:
    """`yaw = 180 - argmax` (first maximum wins), reference `infer.py:158,198,233`."""
    return 180 - np.argmax(orientation, axis=1)


def heads_forward(feat_l, feat_r, weights, conv1size=15, dtype=F64):
    """Both heads on (n,1,360,128) pairs -> (overlap (n,), yaw (n,) int64, logit (n,), corr (n,360))."""
    overlap, logit = delta_head_forward(feat_l, feat_r, weights, conv1size, dtype)
    corr = correlation_head_forward(feat_l, feat_r, dtype)
    return overlap, yaw_from_orientation(corr), logit, corr


# --------------------------------------------------------------------------------------------------
# Literal forms (NumPy only) -- second opinion for the fast forms above, used on small shapes
# --------------------------------------------------------------------------------------------------


def conv2d_valid_literal(x_hwc: np.ndarray, kernel_hwio: np.ndarray, bias: np.ndarray, stride=(1, 1),
                         relu: bool = True) -> np.ndarray:
    """Direct 'valid' cross-correlation, channels_last, float64 (Keras Conv2D semantics)."""
    x = np.asarray(x_hwc, F64)
    k = np.asarray(kernel_hwio, F64)
    kh, kw, cin, cout = k.shape
    sh, sw = stride
    oh = (x.shape[0] - kh) // sh + 1
    ow = (x.shape[1] - kw) // sw + 1
    out = np.zeros((oh, ow, cout), F64)
    kmat = k.reshape(kh * kw * cin, cout)
    for y in range(oh):
        for xx in range(ow):
            patch = x[y * sh:y * sh + kh, xx * sw:xx * sw + kw, :].reshape(-1)
            out[y, xx] = patch @ kmat
    out += np.asarray(bias, F64)
    return np.maximum(out, 0) if relu else out


def delta_layer_literal(l_hwc: np.ndarray, r_hwc: np.ndarray) -> np.ndarray:
    """DeltaLayer exactly as the reference builds it (`generateNet.py:45-59`): reshape to (wh,1,c) and
    (1,wh,c), tile both to (wh,wh,c), abs(difference)."""
    w, h, c = l_hwc.shape
    rl = np.asarray(l_hwc, F64).reshape(w * h, 1, c)
    rr = np.asarray(r_hwc, F64).reshape(1, w * h, c)
    tl = np.tile(rl, (1, w * h, 1))
    tr = np.tile(rr, (w * h, 1, 1))
    return np.abs(tl - tr)


def range_padding_literal(x_bhwc: np.ndarray, padding: int) -> np.ndarray:
    """`RangePadding2D.call`, reference `RangePadding2D.py:34`."""
    return np.concatenate([x_bhwc[:, :, padding:, :], x_bhwc, x_bhwc[:, :, :padding - 1, :]], axis=2)


def correlation_literal(l_bhwc: np.ndarray, r_bhwc: np.ndarray, normalize: str = "none") -> np.ndarray:
    """`NormalizedCorrelation2D.call`, reference `NormalizedCorrelation2D.py:43-109`, as a sliding-window
    'valid' convolution of the range-padded left features with the right features as the kernel.
    'euclidean' = l2-normalise along the width axis (`:56-58`) -- only used by the reference's demo."""
    l = np.asarray(l_bhwc, F64)
    r = np.asarray(r_bhwc, F64)
    if normalize == "euclidean":
        l = l / np.sqrt(np.maximum((l * l).sum(axis=2, keepdims=True), 1e-12))
        r = r / np.sqrt(np.maximum((r * r).sum(axis=2, keepdims=True), 1e-12))
    elif normalize != "none":
        raise ValueError(normalize)
    B, H, W, C = l.shape
    pad = range_padding_literal(l, W // 2)  # (B,H,2W-1,C)
    out = np.zeros((B, H, W, 1), F64)
    for b in range(B):
        for k in range(W):
            out[b, 0, k, 0] = (pad[b, :, k:k + W, :] * r[b]).sum()  # H == kernel height -> one output row
    return out


# --------------------------------------------------------------------------------------------------
# End-to-end helper structured like the reference's Infer (leg model, then head model, batch 16)
# --------------------------------------------------------------------------------------------------


def infer_pairs(images_nhwc: np.ndarray, pairs: np.ndarray, weights, model_cfg=None, dtype=F64,
                batch_size: int = 16):
    """leg over all scans, then heads over `pairs` with l = fv[pairs[:,0]], r = fv[pairs[:,1]]
    (reference `ImagePairOverlapSequenceFeatureVolume.py:43-47`), batched like `infer.py:155-158`."""
    fv = []
    for s in range(0, images_nhwc.shape[0], batch_size):
        fv.append(leg_forward(images_nhwc[s:s + batch_size], weights, model_cfg, dtype))
    fv = np.concatenate(fv, axis=0)
    s = int((model_cfg or {}).get("conv1NetworkHead_conv1size", 15))
    ov, yw, lg, cr = [], [], [], []
    for b in range(0, len(pairs), batch_size):
        p = np.asarray(pairs[b:b + batch_size])
        o, y, g, c = heads_forward(fv[p[:, 0]], fv[p[:, 1]], weights, s, dtype)
        ov.append(o); yw.append(y); lg.append(g); cr.append(c)
    return (np.concatenate(ov), np.concatenate(yw), np.concatenate(lg), np.concatenate(cr), fv)


# ------------------------------------------------------------------------------------------------------------
# ground-truth overlap / yaw -- PINNED by tests/golden/gt_overlap_yaw.npz, which was produced by
# running the reference's own src/utils/com_overlap_yaw.py in the build container (tests/golden/make_gt_golden.py)
# ------------------------------------------------------------------------------------------------------------
def range_image_f64(points_xyz1: np.ndarray, fov_up: float = 3.0, fov_down: float = -25.0, proj_H: int = 64,
                    proj_W: int = 900, max_range: float = 50.0) -> np.ndarray:
    """The range image `range_projection` (utils.py:59-134) yields for FLOAT64 homogeneous points, which is what
    com_overlap_yaw.py:30,38-42 feeds it (`load_vertex`, utils.py:217-230, builds float64 arrays): all arithmetic in
    float64, the winning depth rounded to float32 when stored (`proj_range` is a float32 image, utils.py:120-121).
    Nearest point wins = per-pixel minimum of the depth."""
    p = np.asarray(points_xyz1, F64)
    up = fov_up / 180.0 * np.pi
    down = fov_down / 180.0 * np.pi
    fov = abs(down) + abs(up)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    depth = np.sqrt((x * x + y * y) + z * z)
    keep = (depth > 0) & (depth < max_range)
    x, y, z, depth = x[keep], y[keep], z[keep], depth[keep]
    yaw = -np.arctan2(y, x)
    pitch = np.arcsin(z / depth)
    px = 0.5 * (yaw / np.pi + 1.0)
    py = 1.0 - (pitch + abs(down)) / fov
    px = px * proj_W
    py = py * proj_H
    px = np.maximum(0, np.minimum(proj_W - 1, np.floor(px))).astype(np.int64)
    py = np.maximum(0, np.minimum(proj_H - 1, np.floor(py))).astype(np.int64)
    best = np.full(proj_H * proj_W, np.inf, F64)
    np.minimum.at(best, py * proj_W + px, depth)
    out = np.where(np.isfinite(best), best, -1.0).astype(F32)
    return out.reshape(proj_H, proj_W)


def yaw_bin_from_poses(current_pose: np.ndarray, reference_pose: np.ndarray, leg_output_width: int = 360) -> int:
    """com_overlap_yaw.py:49-55 with euler_angles_from_rotation_matrix (utils.py:186-214), Python operator
    precedence included: `int(-(yaw / pi) * W // 2 + W // 2)` floors (-(yaw/pi) * W) / 2 BEFORE adding W // 2."""
    import math
    R = np.linalg.inv(current_pose).dot(reference_pose)[:3, :3]

    def isclose(a, b, rtol=1.e-5, atol=1.e-8):
        return abs(a - b) <= atol + rtol * abs(b)

    phi = 0.0
    if isclose(R[2, 0], -1.0) or isclose(R[2, 0], 1.0):
        pass  # gimbal lock: the reference leaves phi (yaw) at 0
    else:
        theta = -math.asin(R[2, 0])
        cos_theta = math.cos(theta)
        phi = math.atan2(R[1, 0] / cos_theta, R[0, 0] / cos_theta)
    return int(-(phi / np.pi) * leg_output_width // 2 + leg_output_width // 2)


def com_overlap_yaw(scans_xyz: Sequence[np.ndarray], poses: np.ndarray, frame_idx: int, leg_output_width: int = 360
                    ) -> np.ndarray:
    """Ground-truth mapping of one frame against all scans (com_overlap_yaw.py:10-68): rows
    [frame_idx, reference_idx, overlap, yaw_bin].  `scans_xyz[i]` = (N_i, >=3) points of scan i."""
    def homog(a):
        a = np.asarray(a)
        h = np.ones((a.shape[0], 4), F64)
        h[:, :3] = a[:, :3]
        return h

    cur = range_image_f64(homog(scans_xyz[frame_idx]))
    valid_num = int(np.count_nonzero(cur > 0))
    cur_pose = poses[frame_idx]
    n = len(scans_xyz)
    out = np.zeros((n, 4))
    out[:, 0] = frame_idx
    out[:, 1] = np.arange(n)
    inv_cur = np.linalg.inv(cur_pose)
    for r in range(n):
        world = poses[r].dot(homog(scans_xyz[r]).T).T
        ref = range_image_f64(inv_cur.dot(world.T).T)
        sel = ref > 0
        out[r, 2] = np.count_nonzero(np.abs(ref[sel] - cur[sel]) < 1) / valid_num
        out[r, 3] = yaw_bin_from_poses(cur_pose, poses[r], leg_output_width)
    return out
