"""Warm 1-vs-1024 head sweep with and without the dead-channel compaction, for both seeded weight sets (how many channels of the query
are alive, per column-group pair, decides what the compaction buys).    python tools/experiments/compaction_time.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import synthetic as S
from overlapnet_amd.engine import OvnEngine
torch.cuda.set_device(0)
fx = S.load_fixture_images()
for wname, wf in S.WEIGHT_SETS.items():
    eng = OvnEngine(64, 900, 4)
    eng.load_weights(wf(4), S.REFERENCE_MODEL_CFG)
    cands = torch.empty((1024, 360, 128), device="cuda")
    for s, imgs in S.sweep_pool_images(1024, 4, 0, fx):
        eng.leg(torch.from_numpy(imgs).cuda(), out=cands[s:s + imgs.shape[0]])
    q = eng.leg(torch.from_numpy(S.sweep_query_image(4, fx)).cuda())
    spec, qs, dc = eng.spectrum(cands), eng.spectrum(q), eng.delta_cache(cands)
    alive = (q[0] != 0)
    per_pair = [int(alive[30 * p:30 * p + 30].any(dim=0).sum()) for p in range(12)]
    res = {}
    for on in (True, False, True, False):
        eng.set_head_compaction(on)
        for _ in range(3): r = eng.heads(cands, q, spec_l=spec, spec_r=qs, dcache_l=dc)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): r = eng.heads(cands, q, spec_l=spec, spec_r=qs, dcache_l=dc)
        torch.cuda.synchronize()
        res.setdefault(on, []).append(1e3 * (time.perf_counter() - t0) / 10)
        res[("ov", on)] = r["overlap"].cpu().numpy()
    print("%-13s live channels %d (per column-group pair: %s)  heads ms per 1024 pairs: compaction on %s | off %s | max |d overlap| on vs off %.2e"
          % (wname, int(alive.any(dim=0).sum()), per_pair, ["%.3f" % t for t in res[True]], ["%.3f" % t for t in res[False]],
             float(np.max(np.abs(res[("ov", True)] - res[("ov", False)])))))
    eng.close()
