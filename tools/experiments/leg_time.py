"""Batched-leg timing with an alternative build of the library (A/B of kernel edits in one gpurun call):
    python tools/experiments/leg_time.py [path/to/libovn_hip_variant.so]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import overlapnet_amd._lib as L
if len(sys.argv) > 1:
    L.LIB_PATH = os.path.abspath(sys.argv[1])
from tools import synthetic as S
from overlapnet_amd.engine import OvnEngine
torch.cuda.set_device(0)
eng = OvnEngine(64, 900, 4)
eng.load_weights(S.make_test_weights(4, seed=0), S.REFERENCE_MODEL_CFG)
base = torch.from_numpy(S.candidate_images(64, 4, seed=5)).cuda()
imgs = base[torch.arange(1025) % 64].contiguous()
out = torch.empty((1025, 360, 128), device="cuda")
for _ in range(2): eng.leg(imgs, out=out)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); eng.leg(imgs, out=out); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print(os.path.basename(L.LIB_PATH), "leg ms per 1025 scans: min %.3f med %.3f" % (min(ts), sorted(ts)[2]), "checksum %.6e" % float(out.double().sum()))
