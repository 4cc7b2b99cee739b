"""Preprocessing front-end on MI355X: same function names/arguments/returns as the reference's
`src/utils/utils.py` (range_projection :59, gen_normal_map :137, load_files :233) and its drivers
`gen_depth_data.py:10`, `gen_normal_data.py:10`, `gen_intensity_data.py:10`, backed by the HIP scatter
kernel (`ovn_project`, csrc/projection.hip).  Results come back as host NumPy arrays like the
reference's; `project_scans` keeps everything on the device for the full-stack path
(.bin -> projection -> leg) that the reference never composes (demo1 writes .npy, Infer reads them).
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .engine import OvnEngine

_engine: Optional[OvnEngine] = None


def _get_engine() -> OvnEngine:
    global _engine
    if _engine is None:
        _engine = OvnEngine(64, 900, 4)
    return _engine


def load_files(folder):
    """ Load all files in a folder and sort (utils.py:233-239). """
    file_paths = [os.path.join(dp, f) for dp, dn, fn in os.walk(os.path.expanduser(folder)) for f in fn]
    file_paths.sort()
    return file_paths


def project_scans(scans: Sequence[np.ndarray], engine: Optional[OvnEngine] = None, proj_H=64, proj_W=900,
                  fov_up=3.0, fov_down=-25.0, max_range=50, want=("range", "normal", "intensity"),
                  stacked_flags: Optional[Tuple[bool, bool, bool]] = None):
    """Project a list of (N_i,4) float32 scans in ONE batched launch sequence; returns device tensors."""
    eng = engine or _get_engine()
    counts = [int(np.asarray(s).reshape(-1, 4).shape[0]) for s in scans]
    offsets = np.zeros(len(scans) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(counts)
    if offsets[-1] > 0:
        flat = np.concatenate([np.asarray(s, np.float32).reshape(-1, 4) for s in scans], axis=0)
    else:
        flat = np.zeros((0, 4), np.float32)
    pts = torch.from_numpy(np.ascontiguousarray(flat)).to(eng.device)
    if pts.numel() == 0:
        pts = torch.zeros((1, 4), dtype=torch.float32, device=eng.device)
    off = torch.from_numpy(offsets).to(eng.device)
    return eng.project(pts, off, max(counts) if counts else 0, proj_H, proj_W, fov_up, fov_down, max_range,
                       want=want, stacked_flags=stacked_flags)


def range_projection(current_vertex, fov_up=3.0, fov_down=-25.0, proj_H=64, proj_W=900, max_range=50):
    """ Spherical projection of one point cloud (utils.py:59-134).
        Returns proj_range (H,W), proj_vertex (H,W,4), proj_intensity (H,W), proj_idx (H,W) int32. """
    r = project_scans([current_vertex], proj_H=proj_H, proj_W=proj_W, fov_up=fov_up, fov_down=fov_down,
                      max_range=max_range, want=("range", "vertex", "intensity", "idx"))
    return (r["range"][0].cpu().numpy(), r["vertex"][0].cpu().numpy(), r["intensity"][0].cpu().numpy(),
            r["idx"][0].cpu().numpy())


def gen_normal_map(current_range, current_vertex, proj_H=64, proj_W=900):
    """ Normal image from a range image and its vertex map (utils.py:137-175). """
    eng = _get_engine()
    rng = torch.from_numpy(np.ascontiguousarray(current_range, np.float32).reshape(1, proj_H, proj_W)).to(eng.device)
    vtx = torch.from_numpy(np.ascontiguousarray(current_vertex, np.float32).reshape(1, proj_H, proj_W, 4)).to(eng.device)
    return eng.normals(rng, vtx)[0].cpu().numpy()


def _gen_data(scan_folder, dst_folder, sub, key, normalize=False):
    dst = os.path.join(dst_folder, sub)
    try:
        os.stat(dst)
        print('generating %s data in: ' % sub, dst)
    except OSError:
        print('creating new %s folder: ' % sub, dst)
        os.mkdir(dst)
    scan_paths = load_files(scan_folder)
    outs: List[np.ndarray] = []
    bs = 64
    for s in range(0, len(scan_paths), bs):
        scans = [np.fromfile(p, dtype=np.float32).reshape((-1, 4)) for p in scan_paths[s:s + bs]]
        r = project_scans(scans, want=(key,))
        imgs = r[key].cpu().numpy()
        for k in range(imgs.shape[0]):
            img = imgs[k]
            if normalize:
                img = img / np.max(img)
            dst_path = os.path.join(dst, str(s + k).zfill(6))   # enumeration index, gen_depth_data.py:41
            np.save(dst_path, img)
            outs.append(img)
            print('finished generating %s data at: ' % sub, dst_path)
    return outs


def gen_depth_data(scan_folder, dst_folder, normalize=False):
    """ (64,900) range images -> dst_folder/depth/%06d.npy (gen_depth_data.py:10-48). """
    return _gen_data(scan_folder, dst_folder, 'depth', 'range', normalize)


def gen_normal_data(scan_folder, dst_folder, normalize=False):
    """ (64,900,3) normal images -> dst_folder/normal/%06d.npy (gen_normal_data.py:10-46). """
    return _gen_data(scan_folder, dst_folder, 'normal', 'normal', False)


def gen_intensity_data(scan_folder, dst_folder, normalize=False):
    """ (64,900) intensity images -> dst_folder/intensity/%06d.npy (gen_intensity_data.py:10-43). """
    return _gen_data(scan_folder, dst_folder, 'intensity', 'intensity', normalize)


def gen_semantic_data(semantic_folder, scan_folder, dst_folder, proj_H=64, proj_W=900):
    """ (64,900,20) projected class probabilities -> dst_folder/semantic/<scan name>.npy (gen_semantic_data.py:11-57).
        Raw inputs: per-point float32 (N,20) probability files, one per scan, in sorted order.  The correspondences are
        the projection's `proj_idx` with max_range = inf (gen_semantic_data.py:39); like the reference, that index (taken
        after the depth > 0 filter) addresses the unfiltered probability array. """
    dst = os.path.join(dst_folder, 'semantic')
    try:
        os.stat(dst)
        print('generating semantic data in: ', dst)
    except OSError:
        print('creating new semantic folder: ', dst)
        os.mkdir(dst)
    prob_paths = load_files(semantic_folder)
    scan_paths = load_files(scan_folder)
    semantics = []
    bs = 64
    for s in range(0, len(prob_paths), bs):
        scans = [np.fromfile(p, dtype=np.float32).reshape((-1, 4)) for p in scan_paths[s:s + bs][:len(prob_paths) - s]]
        idx = project_scans(scans, proj_H=proj_H, proj_W=proj_W, max_range=np.inf, want=("idx",))["idx"].cpu().numpy()
        for k in range(idx.shape[0]):
            probs = np.fromfile(prob_paths[s + k], dtype=np.float32).reshape((-1, 20))
            proj_idx = idx[k]
            proj_prob = np.full((proj_H, proj_W, 20), -1, dtype=np.float32)
            proj_prob[proj_idx >= 0] = probs[proj_idx[proj_idx >= 0]]
            base_name = os.path.basename(scan_paths[s + k]).replace('.bin', '')
            dst_path = os.path.join(dst, base_name)
            np.save(dst_path, proj_prob)
            semantics.append(proj_prob)
            print('finished generating semantic data at: ', dst_path)
    return semantics
