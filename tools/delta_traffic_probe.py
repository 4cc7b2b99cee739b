#!/usr/bin/env python3
"""Does the Delta kernel's time depend on the candidate stream?  Same 1024 pairs, (a) 1024 distinct candidates,
(b) every pair reads candidate 0 (the L stream becomes L2-resident)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synthetic as S
from overlapnet_amd.engine import OvnEngine
torch.cuda.set_device(0)
eng = OvnEngine(64, 900, 4); eng.load_weights(S.make_test_weights(4, 0), S.REFERENCE_MODEL_CFG)
g = torch.Generator(device="cuda").manual_seed(1)
feats = torch.relu(torch.randn((1024, 360, 128), device="cuda", generator=g) + 0.1).contiguous()
q = feats[5:6].contiguous()
out = {}
for name, kw in (("distinct", dict()), ("same_candidate", dict(lidx=np.zeros(1024, np.int64), ridx=np.zeros(1024, np.int64)))):
    fr = q if name == "distinct" else feats
    for _ in range(2): eng.heads(feats, fr if name == "distinct" else feats[5:6].contiguous().expand(1,360,128).contiguous(), **({} if name=="distinct" else dict(lidx=np.zeros(1024,np.int64), ridx=np.zeros(1024,np.int64))))
    torch.cuda.synchronize(); eng.profile_begin()
    for _ in range(5): eng.heads(feats, fr if name == "distinct" else feats[5:6].contiguous(), **({} if name=="distinct" else dict(lidx=np.zeros(1024,np.int64), ridx=np.zeros(1024,np.int64))))
    torch.cuda.synchronize(); p = eng.profile_end()
    out[name] = {k: round(v[0] / v[1], 3) for k, v in p.items() if v[1]}
print(json.dumps(out))
