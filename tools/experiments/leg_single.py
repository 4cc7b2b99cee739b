"""Per-kernel time of the leg for ONE scan and for 8 scans (query latency path); run under rocprofv3 --kernel-trace:
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_leg1 -o leg1 -- python tools/experiments/leg_single.py [n]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import synthetic as S
from overlapnet_amd.engine import OvnEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.cuda.set_device(0)
eng = OvnEngine(64, 900, 4)
eng.load_weights(S.make_test_weights(4, seed=0), S.REFERENCE_MODEL_CFG)
imgs = torch.from_numpy(S.candidate_images(max(n, 2), 4, seed=5)).cuda()[:n].contiguous()
out = torch.empty((n, 360, 128), device="cuda")
for _ in range(5): eng.leg(imgs, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): eng.leg(imgs, out=out)
torch.cuda.synchronize()
print("leg of %d scan(s): %.1f us per call" % (n, 1e6 * (time.perf_counter() - t0) / 50))
