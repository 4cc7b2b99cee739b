"""One query against N cached candidates, repeated; run under rocprofv3 --kernel-trace for the per-kernel breakdown:
    python tools/experiments/latency_one.py [N]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import overlapnet_amd._lib as L
if os.environ.get('OVN_LIB'):
    L.LIB_PATH = os.path.abspath(os.environ['OVN_LIB'])
from tools import synthetic as S
from overlapnet_amd.engine import OvnEngine, decode_match
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.cuda.set_device(0)
eng = OvnEngine(64, 900, 4)
eng.load_weights(S.make_test_weights(4, 0), S.REFERENCE_MODEL_CFG)
imgs = torch.from_numpy(S.candidate_images(64, 4, seed=5)).cuda()
pool = eng.leg(imgs).repeat((N + 63) // 64, 1, 1)[:N].contiguous()
spec, dc = eng.spectrum(pool), eng.delta_cache(pool)
query = imgs[:1].contiguous()
def step():
    q = eng.leg(query)
    qs = eng.spectrum(q)
    r = eng.heads(pool, q, spec_l=spec, spec_r=qs, dcache_l=dc)
    return decode_match(eng.best_match(r["overlap"], r["yaw"], 0.3, host=True))
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize()
print(os.path.basename(L.LIB_PATH), "N=%d: %.1f us per query" % (N, 1e6 * (time.perf_counter() - t0) / 50))
