"""Latency of one loop-closure query against N cached candidates (the demo3 use: N is tens to hundreds, not 1024):
query leg + spectrum + both heads + on-device decision, one record back to the host.   python tools/bench_latency.py"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synthetic as S
from overlapnet_amd.engine import OvnEngine, decode_match

C = 4
eng = OvnEngine(64, 900, C)
eng.load_weights(S.make_test_weights(C, 0), S.REFERENCE_MODEL_CFG)
imgs = torch.from_numpy(S.candidate_images(64, C, seed=5)).cuda()
fv = eng.leg(imgs)
pool = fv.repeat(16, 1, 1).contiguous()              # 1024 candidates
spec = eng.spectrum(pool)
query = imgs[:1].contiguous()
out = {}
for n in (1, 4, 16, 64, 100, 256, 1024):
    def step():
        q = eng.leg(query)
        qs = eng.spectrum(q)
        r = eng.heads(pool[:n], q, spec_l=spec[:n], spec_r=qs)
        return eng.best_match(r["overlap"], r["yaw"], 0.3)
    for _ in range(3):
        rec = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 20
    for _ in range(iters):
        rec = step()
        decode_match(rec)                               # the host reads the decision every query
    torch.cuda.synchronize()
    out[n] = round((time.perf_counter() - t0) / iters * 1e3, 3)
print(json.dumps({"ms_per_query_by_candidates": out}))
