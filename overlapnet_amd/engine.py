"""Thin host wrapper around one libovn_hip context (one per GPU per process).

PyTorch-ROCm is used only as the owner of device memory and for the current HIP stream; every
computation is a C-ABI call into the hand-written HIP kernels.  Nothing here falls back to torch
math or to the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from . import weights as W

FEAT_W = 360
FEAT_C = 128


class _NoContext(object):
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_CONTEXT = _NoContext()
_MATCH_PENDING = 0x7FFFFFF0      # word 3 of a best-match record that the kernel has not written yet (it writes 0 or 1)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else C.c_void_p(t.data_ptr())


def _require_gpu() -> None:
    if not torch.cuda.is_available():
        raise _lib.OvnError("no HIP device visible: overlapnet_amd runs on MI355X (gfx950) only, there is no CPU path")


class OvnEngine:
    """Owns the native context + device copies of nothing but the library's own re-tiled weights."""

    def __init__(self, in_h: int = 64, in_w: int = 900, in_c: int = 4, device: Optional[int] = None):
        _require_gpu()
        self.lib = _lib.load()
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        self.in_h, self.in_w, self.in_c = int(in_h), int(in_w), int(in_c)
        h = C.c_void_p()
        with self._dev():
            _lib.check(self.lib.ovn_create(self.device_index, self.in_h, self.in_w, self.in_c, C.byref(h)), "ovn_create")
        self._h = h
        self.feat_w = 0
        self._leg_ready = False
        self._head_ready = False
        self.head_precision = "f16x3"
        self.leg_precision = "f16x3"
        self.projection_trig = "numpy_avx512"
        self.head_compaction = True
        self.conv1size = 15
        self.check_device_indices = False    # opt-in range check of pair-index tensors that already live on the device (_idx)

    # -- lifetime -----------------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            with self._dev():   # the C side restores the caller's device too; belt and braces
                self.lib.ovn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self):
        """Context that makes this engine's GPU torch's current device -- a no-op when it already is (the library selects its context's
        device itself; what needs the torch side are allocations and streams, and switching costs microseconds per call on the chain
        of a single query)."""
        return _NO_CONTEXT if torch.cuda.current_device() == self.device_index else torch.cuda.device(self.device)

    # -- weights ------------------------------------------------------------------------------------
    def load_weights(self, weights: Dict[str, np.ndarray], model_cfg: Optional[dict] = None) -> None:
        """Register leg + head weights given by Keras layer name (reference infer.py:117-120)."""
        cfg = model_cfg or {}
        W.check_weights(weights, self.in_c, cfg)
        self.conv1size = int(cfg.get("conv1NetworkHead_conv1size", 15))     # generateNet.py:88-89
        with self._dev():
            st = self._stream()
            for l in W.leg_layers(self.in_c, cfg):
                k = torch.from_numpy(np.ascontiguousarray(weights[l.name + "/kernel"], np.float32)).to(self.device)
                b = torch.from_numpy(np.ascontiguousarray(weights[l.name + "/bias"], np.float32)).to(self.device)
                _lib.check(self.lib.ovn_add_leg_layer(self._h, l.name.encode(), _ptr(k), _ptr(b), l.kh, l.kw, l.cin,
                                                      l.cout, l.sh, l.sw, st), "ovn_add_leg_layer(%s)" % l.name)
            fw = C.c_int(0)
            _lib.check(self.lib.ovn_finalize(self._h, C.byref(fw)), "ovn_finalize")
            self.feat_w = fw.value
            self._leg_ready = True
            if self.conv1size != 15:   # any other value: the library's general fp32 Delta path (no Delta cache)
                _lib.check(self.lib.ovn_set_head_geometry(self._h, self.conv1size), "ovn_set_head_geometry")
            names = ["c_conv1", "c_conv2", "c_conv3", "overlap_output"]
            ts = []
            for n in names:
                ts.append(torch.from_numpy(np.ascontiguousarray(weights[n + "/kernel"], np.float32)).to(self.device))
                ts.append(torch.from_numpy(np.ascontiguousarray(weights[n + "/bias"], np.float32)).to(self.device))
            _lib.check(self.lib.ovn_set_head_weights(self._h, *[_ptr(t) for t in ts], st), "ovn_set_head_weights")
            self._head_ready = True
            torch.cuda.synchronize(self.device)

    # -- leg ----------------------------------------------------------------------------------------
    def leg(self, images: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """images (n, H, W, C) float32 on this device -> feature volumes (n, 360, 128)."""
        if not self._leg_ready:
            raise _lib.OvnError("leg weights not loaded")
        if images.device != self.device or images.dtype != torch.float32 or not images.is_contiguous():
            raise _lib.OvnError("leg input must be a contiguous float32 tensor on %s" % self.device)
        if tuple(images.shape[1:]) != (self.in_h, self.in_w, self.in_c):
            raise _lib.OvnError("leg input shape %s, expected (n,%d,%d,%d)" % (tuple(images.shape), self.in_h, self.in_w, self.in_c))
        n = images.shape[0]
        if out is None:
            out = torch.empty((n, self.feat_w, FEAT_C), dtype=torch.float32, device=self.device)
        with self._dev():
            _lib.check(self.lib.ovn_leg(self._h, _ptr(images), n, _ptr(out), self._stream()), "ovn_leg")
        return out

    # -- heads --------------------------------------------------------------------------------------
    def _check_feats(self, t: torch.Tensor, what: str) -> None:
        if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous():
            raise _lib.OvnError("%s must be a contiguous float32 tensor on %s" % (what, self.device))
        if t.numel() % (FEAT_W * FEAT_C) != 0:
            raise _lib.OvnError("%s is not a stack of 360x128 feature volumes" % what)

    def _idx(self, idx, n: Optional[int], bound: Optional[int] = None, what: str = "pair index") -> Optional[torch.Tensor]:
        """Index list -> int32 device tensor.  Host lists / arrays are range-checked ON THE HOST before the upload (no device
        synchronisation).  A tensor that already lives on the device is TRUSTED (the caller built it; the kernels read
        feats[idx[p]] without a bound check, an out-of-range entry reads foreign memory) unless `self.check_device_indices` is set,
        which range-checks it on the device at the price of one synchronisation per call."""
        if idx is None:
            return None
        if isinstance(idx, torch.Tensor) and idx.device == self.device:
            if idx.dtype not in (torch.int32, torch.int64):
                raise IndexError("%s tensor must be int32 or int64, not %s" % (what, idx.dtype))
            if self.check_device_indices and bound is not None and idx.numel():
                lo, hi = int(idx.min()), int(idx.max())          # synchronises: opt-in (OvnEngine.check_device_indices)
                if lo < 0 or hi >= bound:
                    raise IndexError("%s out of range: [%d, %d] not within [0, %d)" % (what, lo, hi, bound))
            t = idx.to(torch.int32).contiguous()
        else:
            a = np.ascontiguousarray(idx.cpu().numpy() if isinstance(idx, torch.Tensor) else idx).reshape(-1)
            if a.size and not np.issubdtype(a.dtype, np.integer):
                if not np.all(a == np.floor(a)):
                    raise IndexError("%s list holds non-integer values" % what)
            a = a.astype(np.int64)
            if bound is not None and a.size and (int(a.min()) < 0 or int(a.max()) >= bound):
                raise IndexError("%s out of range" % what)
            t = torch.from_numpy(a.astype(np.int32)).to(self.device)
        if n is not None and t.numel() != n:
            raise _lib.OvnError("index list has %d entries, expected %d" % (t.numel(), n))
        return t

    def _pairs(self, nl: int, nr: int, lidx, ridx, n: Optional[int]):
        """Shared argument checking of every head entry point: (lidx tensor | None, ridx tensor | None, n)."""
        li = self._idx(lidx, None, nl, "left pair index")
        if n is None:
            n = li.numel() if li is not None else nl
        n = int(n)
        if li is not None and li.numel() != n:
            raise _lib.OvnError("lidx has %d entries, expected %d" % (li.numel(), n))
        if li is None and n > nl:
            raise _lib.OvnError("n=%d pairs but only %d left feature volumes" % (n, nl))
        ri = self._idx(ridx, n, nr, "right pair index")
        if ri is None and n > 0 and nr < 1:
            raise _lib.OvnError("the right-hand side holds no feature volume")
        return li, ri, n

    def heads(self, feats_l: torch.Tensor, feats_r: torch.Tensor, lidx=None, ridx=None, n: Optional[int] = None,
              want_logit: bool = False, want_corr: bool = False, spec_l: Optional[torch.Tensor] = None,
              spec_r: Optional[torch.Tensor] = None, dcache_l: Optional[torch.Tensor] = None):
        """Both heads on n pairs: pair p = (l = feats_l[lidx[p]], r = feats_r[ridx[p]]).
        lidx None -> p, ridx None -> 0 (1-vs-N: feats_r holds the single query).
        spec_l / spec_r: cached spectra (`spectrum`) -> the HBM-bound spectral yaw head; dcache_l: the left pool's Delta cache rows
        (`delta_cache`), used by 1-vs-N sweeps -- same results with or without it.
        Returns dict of device tensors: overlap (n) f32, yaw (n) i32 [, logit (n), corr (n,360)]."""
        if not self._head_ready:
            raise _lib.OvnError("head weights not loaded")
        self._check_feats(feats_l, "feats_l")
        self._check_feats(feats_r, "feats_r")
        nl = feats_l.numel() // (FEAT_W * FEAT_C)
        nr = feats_r.numel() // (FEAT_W * FEAT_C)
        li, ri, n = self._pairs(nl, nr, lidx, ridx, n)
        overlap = torch.empty(n, dtype=torch.float32, device=self.device)
        logit = torch.empty(n, dtype=torch.float32, device=self.device) if want_logit else None
        if spec_l is not None or spec_r is not None:
            # cached spectra given: Delta head on the features, correlation head in its HBM-bound spectral form
            if spec_l is None or spec_r is None:
                raise _lib.OvnError("spec_l and spec_r must be given together")
            for t, what in ((spec_l, "spec_l"), (spec_r, "spec_r")):
                if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous():
                    raise _lib.OvnError("%s must be a contiguous float32 tensor on %s" % (what, self.device))
            if spec_l.numel() != nl * FEAT_C * self.SPEC_W or spec_r.numel() != nr * FEAT_C * self.SPEC_W:
                raise _lib.OvnError("spec_l / spec_r must hold one 128x368 spectrum per feature volume")
            if dcache_l is not None:
                if dcache_l.device != self.device or dcache_l.dtype != torch.float32 or not dcache_l.is_contiguous():
                    raise _lib.OvnError("dcache_l must be a contiguous float32 tensor on %s" % self.device)
                if dcache_l.numel() != nl * self.DELTA_CACHE_ELEMS:
                    raise _lib.OvnError("dcache_l must hold one Delta cache row per left feature volume")
            yaw = torch.empty(n, dtype=torch.int32, device=self.device)
            corr = torch.empty((n, FEAT_W), dtype=torch.float32, device=self.device) if want_corr else None
            with self._dev():
                _lib.check(self.lib.ovn_heads_spectral(self._h, _ptr(feats_l), _ptr(spec_l), _ptr(dcache_l), _ptr(li), _ptr(feats_r),
                                                       _ptr(spec_r), _ptr(ri), n, _ptr(overlap), _ptr(yaw), _ptr(logit), _ptr(corr),
                                                       self._stream()), "ovn_heads_spectral")
            out = {"overlap": overlap, "yaw": yaw}
            if want_logit:
                out["logit"] = logit
            if want_corr:
                out["corr"] = corr
            return out
        yaw = torch.empty(n, dtype=torch.int32, device=self.device)
        corr = torch.empty((n, FEAT_W), dtype=torch.float32, device=self.device) if want_corr else None
        with self._dev():
            _lib.check(self.lib.ovn_heads(self._h, _ptr(feats_l), _ptr(li), _ptr(feats_r), _ptr(ri), n, _ptr(overlap),
                                          _ptr(yaw), _ptr(logit), _ptr(corr), self._stream()), "ovn_heads")
        out = {"overlap": overlap, "yaw": yaw}
        if want_logit:
            out["logit"] = logit
        if want_corr:
            out["corr"] = corr
        return out

    def corr_head(self, feats_l: torch.Tensor, feats_r: torch.Tensor, lidx=None, ridx=None, n: Optional[int] = None,
                  want_corr: bool = False):
        self._check_feats(feats_l, "feats_l")
        self._check_feats(feats_r, "feats_r")
        li, ri, n = self._pairs(feats_l.numel() // (FEAT_W * FEAT_C), feats_r.numel() // (FEAT_W * FEAT_C), lidx, ridx, n)
        yaw = torch.empty(n, dtype=torch.int32, device=self.device)
        corr = torch.empty((n, FEAT_W), dtype=torch.float32, device=self.device) if want_corr else None
        with self._dev():
            _lib.check(self.lib.ovn_corr_head(self._h, _ptr(feats_l), _ptr(li), _ptr(feats_r), _ptr(ri), n, _ptr(yaw),
                                              _ptr(corr), self._stream()), "ovn_corr_head")
        return {"yaw": yaw, "corr": corr} if want_corr else {"yaw": yaw}

    SPEC_W = 368
    DELTA_CACHE_ELEMS = 49216      # floats per Delta cache row (include/ovn_hip.h: OVN_DELTA_CACHE_ELEMS)

    @property
    def has_delta_cache(self) -> bool:
        return self.conv1size == 15

    def delta_cache(self, feats: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """feature volumes (n,360,128) -> Delta cache rows (n, 49216): the candidate-side half of the Delta head's preparation
        (packed words, TT + b2, value range), cached next to the volume like its spectrum."""
        if not self._head_ready:
            raise _lib.OvnError("head weights not loaded")
        if not self.has_delta_cache:
            raise _lib.OvnError("the Delta cache exists for conv1NetworkHead_conv1size=15 only")
        self._check_feats(feats, "feats")
        n = feats.numel() // (FEAT_W * FEAT_C)
        if out is None:
            out = torch.empty((n, self.DELTA_CACHE_ELEMS), dtype=torch.float32, device=self.device)
        with self._dev():
            _lib.check(self.lib.ovn_delta_cache(self._h, _ptr(feats), n, _ptr(out), self._stream()), "ovn_delta_cache")
        return out

    def spectrum(self, feats: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """feature volumes (n,360,128) -> cached spectra (n,128,368) for the spectral correlation head."""
        self._check_feats(feats, "feats")
        n = feats.numel() // (FEAT_W * FEAT_C)
        if out is None:
            out = torch.empty((n, FEAT_C, self.SPEC_W), dtype=torch.float32, device=self.device)
        with self._dev():
            _lib.check(self.lib.ovn_spectrum(self._h, _ptr(feats), n, _ptr(out), self._stream()), "ovn_spectrum")
        return out

    def corr_head_spectral(self, spec_l: torch.Tensor, spec_r: torch.Tensor, lidx=None, ridx=None,
                           n: Optional[int] = None, want_corr: bool = False):
        """Correlation head on cached spectra: dict(yaw (n) i32 [, corr (n,360)])."""
        for t, what in ((spec_l, "spec_l"), (spec_r, "spec_r")):
            if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous():
                raise _lib.OvnError("%s must be a contiguous float32 tensor on %s" % (what, self.device))
            if t.numel() % (FEAT_C * self.SPEC_W) != 0:
                raise _lib.OvnError("%s is not a stack of 128x368 spectra" % what)
        li, ri, n = self._pairs(spec_l.numel() // (FEAT_C * self.SPEC_W), spec_r.numel() // (FEAT_C * self.SPEC_W), lidx, ridx, n)
        yaw = torch.empty(n, dtype=torch.int32, device=self.device)
        corr = torch.empty((n, FEAT_W), dtype=torch.float32, device=self.device) if want_corr else None
        with self._dev():
            _lib.check(self.lib.ovn_corr_head_spectral(self._h, _ptr(spec_l), _ptr(li), _ptr(spec_r), _ptr(ri), n,
                                                       _ptr(yaw), _ptr(corr), self._stream()), "ovn_corr_head_spectral")
        return {"yaw": yaw, "corr": corr} if want_corr else {"yaw": yaw}

    # -- ground-truth labels -------------------------------------------------------------------------
    def gt_range_images(self, points: torch.Tensor, offsets: torch.Tensor, max_points: int,
                        ref_poses: Optional[torch.Tensor] = None, inv_cur_pose: Optional[torch.Tensor] = None,
                        proj_h: int = 64, proj_w: int = 900, fov_up: float = 3.0, fov_down: float = -25.0,
                        max_range: float = 50.0) -> torch.Tensor:
        """Float64 range projection of scans moved by inv_cur_pose . ref_poses[s] (com_overlap_yaw.py:37-40):
        (n, H, W) f32 device tensor, -1 = empty.  points (total,4) f32, offsets (n+1) i64, poses float64 device tensors."""
        n = int(offsets.numel()) - 1
        for t, what, dt in ((points, "points", torch.float32), (offsets, "offsets", torch.int64),
                            (ref_poses, "ref_poses", torch.float64), (inv_cur_pose, "inv_cur_pose", torch.float64)):
            if t is not None and (t.device != self.device or t.dtype != dt or not t.is_contiguous()):
                raise _lib.OvnError("%s must be a contiguous %s tensor on %s" % (what, dt, self.device))
        if ref_poses is not None and ref_poses.numel() != 16 * n:
            raise _lib.OvnError("ref_poses must hold %d 4x4 matrices" % n)
        if inv_cur_pose is not None and inv_cur_pose.numel() != 16:
            raise _lib.OvnError("inv_cur_pose must be one 4x4 matrix")
        out = torch.empty((n, proj_h, proj_w), dtype=torch.float32, device=self.device)
        with self._dev():
            _lib.check(self.lib.ovn_gt_range_images(self._h, _ptr(points), _ptr(offsets), n, int(max_points), _ptr(ref_poses),
                                                    _ptr(inv_cur_pose), proj_h, proj_w, float(fov_up), float(fov_down),
                                                    float(max_range), _ptr(out), self._stream()), "ovn_gt_range_images")
        return out

    def gt_overlap_counts(self, ref_ranges: torch.Tensor, cur_range: torch.Tensor) -> torch.Tensor:
        """(n+1) int32: per reference scan the pixels with |ref - cur| < 1 (ref > 0); last entry = #{cur > 0}."""
        n, h, w = ref_ranges.shape
        for t, what in ((ref_ranges, "ref_ranges"), (cur_range, "cur_range")):
            if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous():
                raise _lib.OvnError("%s must be a contiguous float32 tensor on %s" % (what, self.device))
        if tuple(cur_range.shape[-2:]) != (h, w) or cur_range.numel() != h * w:
            raise _lib.OvnError("cur_range must be one %dx%d image" % (h, w))
        counts = torch.empty(n + 1, dtype=torch.int32, device=self.device)
        with self._dev():
            _lib.check(self.lib.ovn_gt_overlap_counts(self._h, _ptr(ref_ranges), _ptr(cur_range), n, h, w, _ptr(counts),
                                                      self._stream()), "ovn_gt_overlap_counts")
        return counts

    # -- loop-closure decision -----------------------------------------------------------------------
    def best_match(self, overlap: torch.Tensor, yaw: Optional[torch.Tensor] = None, threshold: float = 0.3,
                   ids: Optional[torch.Tensor] = None, index_offset: int = 0, host: bool = False) -> torch.Tensor:
        """On-device `argmax overlap, > threshold` of demo3 (demo3_lcd.py:117-120).  Returns a 4 x int32 device
        record {candidate id, float bits of overlap, yaw, found}; decode with `decode_match`.  host=True: the kernel writes the
        record straight into pinned host memory (device-visible at the same address) and the call waits for the stream -- the
        caller reads the decision without a device-to-host copy (one blit kernel and its hand-off less per query); the record comes back
        as a NumPy int32 array."""
        n = int(overlap.numel())
        for t, what, dt in ((overlap, "overlap", torch.float32), (yaw, "yaw", torch.int32), (ids, "ids", torch.int32)):
            if t is None:
                continue
            if t.device != self.device or t.dtype != dt or not t.is_contiguous() or t.numel() != n:
                raise _lib.OvnError("%s must be a contiguous %s tensor of %d elements on %s" % (what, dt, n, self.device))
        if host:
            if getattr(self, "_match_host", None) is None:
                self._match_host = torch.empty(4, dtype=torch.int32).pin_memory()
                self._match_host_np = self._match_host.numpy()          # the same memory
            out, view = self._match_host, self._match_host_np
            view[3] = _MATCH_PENDING                                      # the kernel's one 16-byte store replaces it (0 or 1)
        else:
            out = torch.empty(4, dtype=torch.int32, device=self.device)
        with self._dev():
            _lib.check(self.lib.ovn_best_match(self._h, _ptr(overlap), _ptr(yaw), _ptr(ids), n, float(threshold),
                                               int(index_offset), _ptr(out), self._stream()), "ovn_best_match")
            if host:
                # poll the record instead of waiting for the stream: the wake-up of a stream wait costs more than the kernel
                spins = 0
                while view[3] == _MATCH_PENDING:
                    spins += 1
                    if spins > 200000:                                    # (~20 ms) something is wrong: let the stream say what
                        torch.cuda.current_stream(self.device).synchronize()
                        if view[3] == _MATCH_PENDING:
                            raise _lib.OvnError("best_match(host=True): the decision record never arrived")
                return view.copy()            # (a NumPy record: `decode_match` takes it as it is)
        return out

    # -- preprocessing ------------------------------------------------------------------------------
    def project(self, points: torch.Tensor, offsets: torch.Tensor, max_points: int, proj_h: int = 64,
                proj_w: int = 900, fov_up: float = 3.0, fov_down: float = -25.0, max_range: float = 50.0,
                want: Sequence[str] = ("range", "normal", "intensity"), stacked_flags: Optional[Tuple[bool, bool, bool]] = None,
                stacked_out: Optional[torch.Tensor] = None):
        """Batch spherical projection.  points: (total,4) f32 device tensor of concatenated scans,
        offsets: (n_scans+1) int64 device tensor (absolute positions in `points`: a slice of a longer offsets tensor projects
        that range of scans).  `want` selects outputs among range, vertex, intensity, idx, normal;
        stacked_flags=(use_depth,use_normals,use_intensity) additionally assembles the (n,H,W,C) leg input (into `stacked_out`
        when given).  Returns a dict of device tensors."""
        if points.device != self.device or points.dtype != torch.float32 or not points.is_contiguous():
            raise _lib.OvnError("points must be a contiguous float32 tensor on %s" % self.device)
        if offsets.device != self.device or offsets.dtype != torch.int64:
            raise _lib.OvnError("offsets must be an int64 tensor on %s" % self.device)
        n = offsets.numel() - 1
        dev = self.device
        out = {}
        mk = lambda *shape, dt=torch.float32: torch.empty(shape, dtype=dt, device=dev)
        rng = mk(n, proj_h, proj_w) if "range" in want else None
        vtx = mk(n, proj_h, proj_w, 4) if "vertex" in want else None
        itn = mk(n, proj_h, proj_w) if "intensity" in want else None
        idx = mk(n, proj_h, proj_w, dt=torch.int32) if "idx" in want else None
        nrm = mk(n, proj_h, proj_w, 3) if "normal" in want else None
        stk = None
        ud = un = ui = 0
        if stacked_flags is not None:
            ud, un, ui = (int(bool(v)) for v in stacked_flags)
            if stacked_out is not None:
                if (stacked_out.device != self.device or stacked_out.dtype != torch.float32 or not stacked_out.is_contiguous()
                        or tuple(stacked_out.shape) != (n, proj_h, proj_w, ud + 3 * un + ui)):
                    raise _lib.OvnError("stacked_out must be a contiguous float32 (%d,%d,%d,%d) tensor on %s"
                                        % (n, proj_h, proj_w, ud + 3 * un + ui, self.device))
                stk = stacked_out
            else:
                stk = mk(n, proj_h, proj_w, ud + 3 * un + ui)
        with self._dev():
            _lib.check(self.lib.ovn_project(self._h, _ptr(points), _ptr(offsets), n, int(max_points), proj_h, proj_w,
                                            float(fov_up), float(fov_down), float(max_range), _ptr(rng), _ptr(vtx),
                                            _ptr(itn), _ptr(idx), _ptr(nrm), _ptr(stk), ud, un, ui, self._stream()),
                       "ovn_project")
        for k, v in (("range", rng), ("vertex", vtx), ("intensity", itn), ("idx", idx), ("normal", nrm), ("stacked", stk)):
            if v is not None:
                out[k] = v
        return out

    def normals(self, rng: torch.Tensor, vtx: torch.Tensor) -> torch.Tensor:
        """range (n,H,W) + vertex (n,H,W,4) device tensors -> normal map (n,H,W,3)."""
        for t in (rng, vtx):
            if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous():
                raise _lib.OvnError("normals(): inputs must be contiguous float32 tensors on %s" % self.device)
        n, h, w = rng.shape
        if tuple(vtx.shape) != (n, h, w, 4):
            raise _lib.OvnError("normals(): vertex shape %s does not match range %s" % (tuple(vtx.shape), tuple(rng.shape)))
        out = torch.empty((n, h, w, 3), dtype=torch.float32, device=self.device)
        with self._dev():
            _lib.check(self.lib.ovn_normals(self._h, _ptr(rng), _ptr(vtx), n, h, w, _ptr(out), self._stream()),
                       "ovn_normals")
        return out

    def projection_angles(self, points: torch.Tensor, proj_h: int = 64, proj_w: int = 900, fov_up: float = 3.0,
                          fov_down: float = -25.0, max_range: float = 50.0):
        """(yaw, pitch, pixel) of every point of an (n, 4) float32 device tensor as the projection kernel evaluates them
        (utils.py:75-104; pixel = -1 for points the range filter drops) -- validation entry, include/ovn_hip.h."""
        if points.device != self.device or points.dtype != torch.float32 or not points.is_contiguous() or points.dim() != 2 \
                or points.shape[1] != 4:
            raise _lib.OvnError("projection_angles(): points must be a contiguous (n, 4) float32 tensor on %s" % self.device)
        n = points.shape[0]
        yaw = torch.empty(n, dtype=torch.float32, device=self.device)
        pitch = torch.empty(n, dtype=torch.float32, device=self.device)
        pix = torch.empty(n, dtype=torch.int32, device=self.device)
        with self._dev():
            _lib.check(self.lib.ovn_projection_angles(self._h, _ptr(points), n, proj_h, proj_w, float(fov_up), float(fov_down),
                                                      float(max_range), _ptr(yaw), _ptr(pitch), _ptr(pix), self._stream()),
                       "ovn_projection_angles")
        return yaw, pitch, pix

    def debug_head_activations(self, n: int):
        """(o2 (n,24,24,128), o3 (n,22,22,256)) left in scratch by the last heads() call -- test hook."""
        o2 = torch.empty((n, 24, 24, 128), dtype=torch.float32, device=self.device)
        o3 = torch.empty((n, 22, 22, 256), dtype=torch.float32, device=self.device)
        with self._dev():
            _lib.check(self.lib.ovn_debug_head_activations(self._h, n, _ptr(o2), _ptr(o3), self._stream()),
                       "ovn_debug_head_activations")
        return o2, o3

    def set_head_pipeline(self, chunk_pairs: int = 1024, sub_chunk_pairs: int = 0, streams: int = 1, yaw_on_side_stream: bool = False):
        """Launch structure of the head calls (include/ovn_hip.h: ovn_set_head_pipeline); results do not depend on it."""
        _lib.check(self.lib.ovn_set_head_pipeline(self._h, int(chunk_pairs), int(sub_chunk_pairs), int(streams),
                                                  int(bool(yaw_on_side_stream))), "ovn_set_head_pipeline")

    def head_pipeline(self):
        """(chunk_pairs, sub_chunk_pairs, streams, yaw_on_side_stream) currently in effect."""
        a, b, c, d = C.c_int64(), C.c_int64(), C.c_int(), C.c_int()
        _lib.check(self.lib.ovn_get_head_pipeline(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "ovn_get_head_pipeline")
        return int(a.value), int(b.value), int(c.value), bool(d.value)

    def set_head_precision(self, mode: str) -> None:
        """Arithmetic of the Delta-head contractions (fp32 storage and accumulation in both modes):
        'f16x3' (default) = scaled 3-term fp16 split on the fp16 matrix cores (22 significand bits per operand: the error of an
        fp32 evaluation), 'f32' = fp32 matrix cores (bit-for-bit an fp32 FMA chain, 1/16 of the rate)."""
        table = {"f32": 0, "f16x3": 1}
        if mode not in table:
            raise ValueError("head precision must be one of %s" % sorted(table))
        _lib.check(self.lib.ovn_set_head_precision(self._h, table[mode]), "ovn_set_head_precision")
        self.head_precision = mode

    def set_head_compaction(self, on: bool) -> None:
        """1-vs-N sweeps drop the query's dead feature channels (zero in all 360 columns) from the Delta head's contraction (default on;
        exact -- include/ovn_hip.h: ovn_set_head_compaction).  Off: every pair walks all 128 channels, as indexed pairs always do."""
        _lib.check(self.lib.ovn_set_head_compaction(self._h, int(bool(on))), "ovn_set_head_compaction")
        self.head_compaction = bool(on)

    def head_walk_stats(self) -> dict:
        """The K walk the Delta head's contraction took in the most recent 1-vs-N sweep (`ovn_head_walk_stats`): per column-group pair
        the slices of 32 channels its live channels need (`slices_per_group_pair`, 12 entries), the query's live channels, and
        `k_walk_frac` = the MFMAs the kernel issued / those of the 128-channel walk: wave w of the contraction kernel holds column
        groups 3 w .. 3 w + 2 and skips a slice when none of them walks it; a last slice of <= 16 live channels is packed tap-major
        into `packed_last_slice_steps` steps instead of 15 (csrc/delta_head_f16x3.hip)."""
        out = (C.c_int32 * 16)()
        _lib.check(self.lib.ovn_head_walk_stats(self._h, out, self._stream()), "ovn_head_walk_stats")
        spp = [int(out[2 + p]) for p in range(12)]
        per_wave = [max(spp[(3 * w) // 2], spp[(3 * w + 2) // 2]) for w in range(8)]
        nsm, tail = int(out[0]), int(out[15])

        def steps(ns):      # MFMA steps of a walk of ns slices: 15 each, the packed last slice (if it is among them) `tail`
            return 15 * ns - ((15 - tail) if (tail and ns == nsm) else 0)
        return {"max_slices": nsm, "live_channels": int(out[1]), "slices_per_group_pair": spp, "packed_last_slice_steps": tail,
                "compacted": bool(out[14]), "k_walk_frac": sum(steps(n) for n in per_wave) / (8 * 60.0),
                "k_walk_frac_time": steps(nsm) / 60.0}

    def set_projection_trig(self, mode: str) -> None:
        """Which float32 `np.arctan2` / `np.arcsin` (utils.py:86-87) `project` reproduces: 'numpy_avx512' (default: NumPy >= 1.22 on an
        AVX512_SKX x86-64 host -- Intel SVML, bit for bit; the machine the reference's shipped .npy files were made on) or 'rounded'
        (the correctly rounded float32 results: NumPy on hosts whose float32 loops call a correctly rounded libm)."""
        table = {"numpy_avx512": 0, "rounded": 1}
        if mode not in table:
            raise ValueError("projection trig must be one of %s" % sorted(table))
        _lib.check(self.lib.ovn_set_projection_trig(self._h, table[mode]), "ovn_set_projection_trig")
        self.projection_trig = mode

    def set_leg_precision(self, mode: str) -> None:
        """Arithmetic of the leg convolutions: 'f16x3' (default, as above) or 'f32' (fp32 matrix cores)."""
        table = {"f32": 0, "f16x3": 1}
        if mode not in table:
            raise ValueError("leg precision must be one of %s" % sorted(table))
        _lib.check(self.lib.ovn_set_leg_precision(self._h, table[mode]), "ovn_set_leg_precision")
        self.leg_precision = mode

    PROFILE_KINDS = ("leg_conv", "corr_head", "delta_c12", "c_conv3", "dense_sigmoid", "projection", "spectrum",
                     "corr_spectral", "delta_prep", "delta_c2")

    def profile_begin(self) -> None:
        _lib.check(self.lib.ovn_profile_begin(self._h), "ovn_profile_begin")

    def profile_end(self):
        """{kind: (total_ms, launches)} measured with HIP events on the launch stream."""
        ms = (C.c_double * len(self.PROFILE_KINDS))()
        cnt = (C.c_int64 * len(self.PROFILE_KINDS))()
        with self._dev():
            _lib.check(self.lib.ovn_profile_end(self._h, ms, cnt), "ovn_profile_end")
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(self.PROFILE_KINDS)}

    def selftest(self) -> None:
        with self._dev():
            _lib.check(self.lib.ovn_selftest(self._h), "ovn_selftest")

    def workspace_bytes(self) -> int:
        return int(self.lib.ovn_workspace_bytes(self._h))


class QueryAhead:
    """Leg + spectrum of the NEXT query scan on a second library context and HIP stream, beside the head kernels that the caller's
    stream is running for the CURRENT query (a recorded sequence, or a live one whose next scan has arrived: the 1-vs-N sweep of
    `Infer.infer_multiple`, infer.py:162-203, spends 0.15 ms of its ~5.6 ms per query in a single-scan leg whose five kernels are a
    handful of workgroups deep in their own latency and leave the GPU idle).  A context owns its scratch, hence the second context;
    features and spectra are double-buffered: the pair handed out by take() may be read by work enqueued on the current stream
    up to the NEXT submit() / take() call (that call records, on the current stream, the event the slot's next overwrite waits
    for -- whatever `wait_current` says); work enqueued later must read its own copy.

        qa = QueryAhead(engine, weights, model_cfg)
        qa.submit(image_0)
        for k in range(n):
            if k + 1 < n: qa.submit(image_{k+1})          # enqueued on the side stream, returns at once
            fv, spec = qa.take()                          # the current stream now waits for query k's features (not the host)
            engine.heads(cands, fv, spec_l=cand_spec, spec_r=spec, dcache_l=cand_dc)

    Same kernels, same bits as `engine.leg` / `engine.spectrum` on the caller's stream (tests/test_gpu_parity.py)."""

    def __init__(self, engine: OvnEngine, weights: Dict[str, np.ndarray], model_cfg: Optional[dict] = None,
                 with_delta_cache: bool = False):
        """with_delta_cache: also compute the query's Delta cache row (`take_all`), for callers that cache every query as a future
        candidate (Infer.infer_multiple)."""
        self.main = engine
        self.side = OvnEngine(engine.in_h, engine.in_w, engine.in_c, device=engine.device_index)
        self.side.load_weights(weights, model_cfg)
        self.side.set_leg_precision(engine.leg_precision)
        self.side.set_head_precision(engine.head_precision)       # the spectrum kernel follows the head arithmetic
        self.side.set_projection_trig(engine.projection_trig)     # (the look-ahead of Infer projects raw scans in this context)
        dev = engine.device
        with torch.cuda.device(dev):
            self.stream = torch.cuda.Stream(device=dev)
            self._fv = [torch.empty((1, FEAT_W, FEAT_C), dtype=torch.float32, device=dev) for _ in range(2)]
            self._spec = [torch.empty((1, FEAT_C, engine.SPEC_W), dtype=torch.float32, device=dev) for _ in range(2)]
            self._dc = ([torch.empty((1, engine.DELTA_CACHE_ELEMS), dtype=torch.float32, device=dev) for _ in range(2)]
                        if with_delta_cache and engine.has_delta_cache else None)
            self._ready = [torch.cuda.Event(), torch.cuda.Event()]
            self._released = [None, None]      # recorded on the consumer's stream behind the last reader of a slot's contents
            self._has_dc = [False, False]      # did the slot's last submit compute the Delta cache row?
        self._submitted = 0
        self._taken = 0
        self._last = None                      # slot handed out by the latest take(), not yet released

    def _release_last(self) -> None:
        """The consumer has enqueued its readers of the slot last handed out by now (contract above): an event behind them on the
        current stream is what the next overwrite of that slot waits for."""
        if self._last is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.main.device))
            self._released[self._last] = ev
            self._last = None

    def submit(self, image: torch.Tensor, wait_current: bool = True, with_delta: bool = True) -> None:
        """Enqueue leg + spectrum of `image` (1, in_h, in_w, in_c) on the side stream; at most two queries may be in flight.
        `wait_current=False`: the image was produced on `self.stream` itself (e.g. its host-to-device copy was issued there), so the
        side stream need not wait for the work already enqueued on the caller's stream -- except for the readers of the slot it is
        about to overwrite, which it always waits for.  `with_delta=False`: skip this query's Delta cache row even if the object was
        built `with_delta_cache` (a sharded `Infer` whose rank does not own the frame); `take_all` then returns None for it."""
        if self._submitted - self._taken >= 2:
            raise _lib.OvnError("QueryAhead.submit: two queries are already in flight, take() one first")
        self._release_last()
        slot = self._submitted & 1
        if self.side.leg_precision != self.main.leg_precision:       # follow the main engine's arithmetic (same bits as engine.leg)
            self.side.set_leg_precision(self.main.leg_precision)
        if self.side.head_precision != self.main.head_precision:
            self.side.set_head_precision(self.main.head_precision)
        if self.side.projection_trig != self.main.projection_trig:
            self.side.set_projection_trig(self.main.projection_trig)
        if wait_current:
            self.stream.wait_stream(torch.cuda.current_stream(self.main.device))   # the image belongs to the caller's stream
        if self._released[slot] is not None:
            self.stream.wait_event(self._released[slot])     # the heads that read this slot two queries ago are done with it
        with torch.cuda.stream(self.stream):
            self.side.leg(image, out=self._fv[slot])
            self.side.spectrum(self._fv[slot], out=self._spec[slot])
            self._has_dc[slot] = self._dc is not None and bool(with_delta)
            if self._has_dc[slot]:
                self.side.delta_cache(self._fv[slot], out=self._dc[slot])
            self._ready[slot].record(self.stream)
        image.record_stream(self.stream)
        self._submitted += 1

    def take(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(feature volume (1, 360, 128), spectrum) of the oldest submitted query; the CURRENT stream waits for them, the host does
        not.  Readers of the pair must be enqueued on the current stream before the next submit() / take() call (class docstring)."""
        if self._taken >= self._submitted:
            raise _lib.OvnError("QueryAhead.take: nothing submitted")
        self._release_last()
        slot = self._taken & 1
        cur = torch.cuda.current_stream(self.main.device)
        cur.wait_event(self._ready[slot])
        self._taken += 1
        self._last = slot
        self._last_taken = slot
        return self._fv[slot], self._spec[slot]

    def take_all(self):
        """take() plus the Delta cache row (None unless built with `with_delta_cache` on a head geometry that has one)."""
        fv, spec = self.take()
        return fv, spec, (self._dc[self._last_taken] if self._has_dc[self._last_taken] else None)

    def close(self) -> None:
        self.stream.synchronize()
        self.side.close()


def decode_match(record) -> Optional[Tuple[int, float, int]]:
    """(candidate id, overlap, yaw) from a best-match record, or None when nothing exceeded the threshold."""
    import numpy as np

    r = np.asarray(record.cpu() if hasattr(record, "cpu") else record, dtype=np.int32).reshape(4)
    if r[3] == 0 or r[0] < 0:
        return None
    return int(r[0]), float(r[1:2].view(np.float32)[0]), int(r[2])
