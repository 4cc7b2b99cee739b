#!/usr/bin/env python3
"""Timing-only ablations of the f16x3 Delta kernel (library built with `make -C overlapnet_amd/csrc ABLATE=1`).
    python tools/delta_ablate.py [abl ...]       # see the ABL comment in csrc/delta_head_f16x3.hip
Prints ms per 1024 pairs of the delta_c12 scope minus the prepare kernels (both from HIP events)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synthetic as S
from overlapnet_amd.engine import OvnEngine
torch.cuda.set_device(0)
eng = OvnEngine(64, 900, 4)
eng.load_weights(S.make_test_weights(4, seed=0), S.REFERENCE_MODEL_CFG)
g = torch.Generator(device="cuda").manual_seed(1)
feats = torch.relu(torch.randn((1024, 360, 128), device="cuda", generator=g) + 0.1).contiguous()
q = feats[3:4].contiguous()
res = {}
for abl in [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 12, 16, 32, 64, 67, 79, 95, 111, 48]:
    os.environ["OVN_DELTA_ABL"] = str(abl)
    for _ in range(2):
        eng.heads(feats, q)
    torch.cuda.synchronize()
    eng.profile_begin()
    for _ in range(5):
        eng.heads(feats, q)
    torch.cuda.synchronize()
    p = eng.profile_end()
    res[abl] = {"delta_c12_ms": p["delta_c12"][0] / p["delta_c12"][1], "delta_prep_ms": p["delta_prep"][0] / max(p["delta_prep"][1], 1)}
    print(abl, res[abl], flush=True)
print(json.dumps(res))
