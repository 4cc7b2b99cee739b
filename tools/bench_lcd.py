"""BASELINE config 3 emulated (KITTI 07 is not in the tree): a 1101-frame sequence, every frame's scan goes through the leg and
is compared with ALL previous frames (ungated, 605,550 pairs), decision on the device, one record per frame back to the
host -- the streaming use of demo3_lcd.py with the feature / spectrum caches resident in HBM.   python tools/bench_lcd.py"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from overlapnet_amd import synthetic as S
from overlapnet_amd.engine import OvnEngine, decode_match

C, F = 4, 1101
eng = OvnEngine(64, 900, C)
eng.load_weights(S.make_test_weights(C, 0), S.REFERENCE_MODEL_CFG)
imgs = torch.from_numpy(S.candidate_images(128, C, seed=3)).cuda()      # 128 distinct synthetic scans, reused cyclically
feats = torch.empty((F, 360, 128), dtype=torch.float32, device="cuda")
specs = torch.empty((F, 128, eng.SPEC_W), dtype=torch.float32, device="cuda")


def run():
    found = 0
    for i in range(F):
        q = imgs[i % 128:i % 128 + 1]
        eng.leg(q, out=feats[i:i + 1])
        eng.spectrum(feats[i:i + 1], out=specs[i:i + 1])
        if i == 0:
            continue
        r = eng.heads(feats[:i], feats[i:i + 1], spec_l=specs[:i], spec_r=specs[i:i + 1])
        m = decode_match(eng.best_match(r["overlap"], r["yaw"], 0.3))
        found += m is not None
    return found


run()
torch.cuda.synchronize()
t0 = time.perf_counter()
found = run()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
pairs = F * (F - 1) // 2
print(json.dumps({"frames": F, "pairs": pairs, "seconds": round(dt, 3), "frames_per_s": round(F / dt, 1),
                  "pairs_per_s": round(pairs / dt), "loop_closures_reported": int(found)}))
