#!/usr/bin/env python3
"""Correlation (yaw) head alone: direct form vs spectral form, 1-vs-N sweep over cached candidates.
Reports pairs/s and the algorithmic HBM rate (184,328 B per pair; the spectral form
actually streams 188,416 B of spectrum per pair) against the 8 TB/s HBM3E peak.
    python tools/bench_corr.py [--n 16384] [--iters 20]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from overlapnet_amd.engine import OvnEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=16384)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
torch.cuda.set_device(0)
eng = OvnEngine(64, 900, 4)
g = torch.Generator(device="cuda").manual_seed(1234)
feats = torch.relu(torch.randn((a.n, 360, 128), device="cuda", generator=g) + 0.1).contiguous()
query = feats[7:8].contiguous()
spec = eng.spectrum(feats)
qspec = eng.spectrum(query)
out = {}
for name, fn in (("direct", lambda: eng.corr_head(feats, query)), ("spectral", lambda: eng.corr_head_spectral(spec, qspec))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    eng.profile_begin()
    for _ in range(a.iters):
        r = fn()
    torch.cuda.synchronize()
    prof = eng.profile_end()
    key = "corr_head" if name == "direct" else "corr_spectral"
    ms = prof[key][0] / prof[key][1]
    pairs_s = a.n / (ms * 1e-3)
    out[name] = {"ms_per_sweep": ms, "pairs_per_s": pairs_s, "algorithmic_GBps": pairs_s * 184328 / 1e9,
                 "frac_of_8TBps": pairs_s * 184328 / 8e12, "yaw_head": r["yaw"][:4].tolist()}
eng.profile_begin()
for _ in range(5):
    eng.spectrum(feats, out=spec)
torch.cuda.synchronize()
p = eng.profile_end()["spectrum"]
out["spectrum"] = {"ms_per_%d_scans" % a.n: p[0] / p[1], "scans_per_s": a.n / (p[0] / p[1] * 1e-3)}
yd = eng.corr_head(feats, query)["yaw"]
ys = eng.corr_head_spectral(spec, qspec)["yaw"]
mism = torch.nonzero(yd != ys).flatten().cpu().numpy()
out["yaw_mismatches_direct_vs_spectral"] = int(mism.size)
if mism.size:  # adjudicate with the fp64 oracle: are these genuine near-ties?
    from oracle import overlapnet_oracle as O
    idx = mism[:16]
    fl = feats[idx].cpu().numpy().reshape(-1, 1, 360, 128).astype(np.float64)
    fr = np.repeat(query.cpu().numpy().reshape(1, 1, 360, 128).astype(np.float64), len(idx), axis=0)
    corr = O.correlation_head_forward(fl, fr)
    srt = np.sort(corr, axis=1)
    out["mismatch_top2_rel_gap_max"] = float(np.max((srt[:, -1] - srt[:, -2]) / np.abs(srt[:, -1])))
    oy = O.yaw_from_orientation(corr)
    out["mismatch_direct_matches_oracle"] = int(np.sum(yd[idx].cpu().numpy() == oy))
    out["mismatch_spectral_matches_oracle"] = int(np.sum(ys[idx].cpu().numpy() == oy))
out["n"] = a.n
print(json.dumps(out))
