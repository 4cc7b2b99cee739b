#!/usr/bin/env python3
"""Generate tests/golden/preprocess_transformed.npz by RUNNING the reference's own `range_projection` + `gen_normal_map`
(src/utils/utils.py:59-186, imported unmodified) on 24 transformed copies of the two shipped scans
(tools/synthetic.transformed_cloud: the 12 z-rotated clouds bench.py's fullstack leg feeds, 4 translations, 4 pitch / roll
tilts, 4 combinations).  Build container only (needs /root/reference):

    python tests/golden/make_preprocess_transformed_golden.py

Stored per cloud i (the clouds themselves are regenerated from kitti_preprocess.npz, deterministic float arithmetic):
  idx_i        proj_idx (64,900) i32 -- the winning point of every pixel; range / vertex / intensity follow from it
  sha_range_i, sha_intensity_i, sha_normal_i, sha_cloud_i   SHA-256 of the reference's images (and of the input cloud)
  n_kept_i     points that pass the range filter
plus `numpy_version`, `cpu_dispatch` (the float32 arctan2 / arcsin of utils.py:86-87 are NumPy's; on this machine: SVML).
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("OVERLAPNET_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REF, "src", "utils"))
import utils as ref_utils  # noqa: E402  (the reference module, not ours)
from tools import synthetic as S  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


fx = S.load_fixture_images()
out = {"numpy_version": np.__version__}
from numpy._core._multiarray_umath import __cpu_features__ as feats  # noqa: E402
out["cpu_dispatch"] = "AVX512_SKX" if feats.get("AVX512_SKX") else "baseline"
for i in range(S.N_TRANSFORMED):
    pts = S.transformed_cloud(fx, i)
    rng, vtx, inten, idx = ref_utils.range_projection(pts)
    nrm = ref_utils.gen_normal_map(rng, vtx)
    out["idx_%d" % i] = idx.astype(np.int32)
    out["sha_range_%d" % i] = sha(rng.astype(np.float32))
    out["sha_intensity_%d" % i] = sha(inten.astype(np.float32))
    out["sha_normal_%d" % i] = sha(nrm.astype(np.float32))
    out["sha_cloud_%d" % i] = sha(pts)
    d = np.linalg.norm(pts[:, :3], 2, axis=1)
    out["n_kept_%d" % i] = int(((d > 0) & (d < 50)).sum())
    print(i, "valid range px %.4f" % np.mean(rng > 0), "valid normal px %.4f" % np.mean(nrm[..., 0] != -1), flush=True)
# other image geometries / fields of view (the reference's keyword arguments, utils.py:59)
for k, (ci, _, H, W, up, down, mr) in enumerate(S.GEOMETRY_CASES):
    pts = S.geometry_cloud(fx, k)
    rng, vtx, inten, idx = ref_utils.range_projection(pts, fov_up=up, fov_down=down, proj_H=H, proj_W=W, max_range=mr)
    nrm = ref_utils.gen_normal_map(rng, vtx, proj_H=H, proj_W=W)
    out["geo_idx_%d" % k] = idx.astype(np.int32)
    out["geo_sha_range_%d" % k] = sha(rng.astype(np.float32))
    out["geo_sha_intensity_%d" % k] = sha(inten.astype(np.float32))
    out["geo_sha_normal_%d" % k] = sha(nrm.astype(np.float32))
    out["geo_sha_cloud_%d" % k] = sha(pts)
    print("geometry", k, (H, W, up, down, mr), "valid range px %.4f" % np.mean(rng > 0), "valid normal px %.4f" % np.mean(nrm[..., 0] != -1),
          "steep points %.3f" % np.mean(np.abs(pts[:, 2]) / np.maximum(np.linalg.norm(pts[:, :3], axis=1), 1e-9) >= 0.5), flush=True)
dst = os.path.join(HERE, "preprocess_transformed.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, os.path.getsize(dst), "bytes")
