"""Does the query leg of step k + 1 hide beside the head kernels of step k?  Two library contexts (each owns its scratch), two
streams, double-buffered query features / spectra:   python tools/experiments/query_ahead.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import synthetic as S
from overlapnet_amd.engine import OvnEngine
torch.cuda.set_device(0)
w = S.make_test_weights(4, 0)
A = OvnEngine(64, 900, 4); A.load_weights(w, S.REFERENCE_MODEL_CFG)
B = OvnEngine(64, 900, 4); B.load_weights(w, S.REFERENCE_MODEL_CFG)
imgs = torch.from_numpy(S.candidate_images(64, 4, seed=5)).cuda()
cands = A.leg(imgs).repeat(16, 1, 1).contiguous()
spec, dc = A.spectrum(cands), A.delta_cache(cands)
query = imgs[:1].contiguous()
qfv = [torch.empty((1, 360, 128), device="cuda") for _ in range(2)]
qsp = [torch.empty((1, 128, A.SPEC_W), device="cuda") for _ in range(2)]
side = torch.cuda.Stream()
K = 20

def serial():
    for k in range(K):
        A.leg(query, out=qfv[0]); A.spectrum(qfv[0], out=qsp[0])
        r = A.heads(cands, qfv[0], spec_l=spec, spec_r=qsp[0], dcache_l=dc)
    return r

def piped():
    main = torch.cuda.current_stream()
    ready = [torch.cuda.Event(), torch.cuda.Event()]
    used = [torch.cuda.Event(), torch.cuda.Event()]
    def leg_on_side(k):
        b = k & 1
        with torch.cuda.stream(side):
            if k >= 2: side.wait_event(used[b])
            B.leg(query, out=qfv[b]); B.spectrum(qfv[b], out=qsp[b]); ready[b].record(side)
    side.wait_stream(main)
    leg_on_side(0)
    for k in range(K):
        b = k & 1
        if k + 1 < K: leg_on_side(k + 1)
        main.wait_event(ready[b])
        r = A.heads(cands, qfv[b], spec_l=spec, spec_r=qsp[b], dcache_l=dc)
        used[b].record(main)
    return r

for name, fn in (("serial", serial), ("piped", piped), ("serial", serial), ("piped", piped)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("%s: %.3f ms per step = %.1f k pairs/s  (overlap sum %.6f)" % (name, 1e3 * dt, 1.024 / dt, float(r["overlap"].double().sum())))
