"""Projection timing (1025 clouds -> stacked leg input), HIP events around ovn_project:  python tools/experiments/proj_time.py"""
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
fx = S.load_fixture_images()
base = [torch.from_numpy(fx["points_%d" % i]).cuda() for i in range(2)]
pts, offs = [], [0]
for i in range(1025):
    b = base[i % 2]
    th = 2.0 * np.pi * ((i * 37) % 900) / 900.0
    q = b.clone()
    q[:, 0] = float(np.cos(th)) * b[:, 0] - float(np.sin(th)) * b[:, 1]
    q[:, 1] = float(np.sin(th)) * b[:, 0] + float(np.cos(th)) * b[:, 1]
    pts.append(q); offs.append(offs[-1] + q.shape[0])
P = torch.cat(pts).contiguous(); O = torch.tensor(offs, dtype=torch.int64, device="cuda"); mp = max(p.shape[0] for p in base)
for _ in range(3): r = eng.project(P, O, mp, want=(), stacked_flags=(True, True, False))
torch.cuda.synchronize()
ts = []
for _ in range(7):
    t0 = time.perf_counter(); r = eng.project(P, O, mp, want=(), stacked_flags=(True, True, False)); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print(os.path.basename(L.LIB_PATH), "OVN_PROJ_MODE=%s projection of 1025 clouds: min %.3f med %.3f ms, checksum %.6e" % (os.environ.get("OVN_PROJ_MODE", "0"), min(ts), sorted(ts)[3], float(r["stacked"].double().sum())))
