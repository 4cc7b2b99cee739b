#!/usr/bin/env python3
"""Same-run A/B of the two f16x3 Delta paths (split: c_conv1 contraction + c_conv2 GEMM; fused: OVN_DELTA_FUSED=1).
    python tools/delta_ab.py [n_pairs]
Prints per-launch milliseconds of the prepare / main / c_conv2 scopes (HIP events) and the largest |difference| of the logits."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synthetic as S
from overlapnet_amd.engine import OvnEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.cuda.set_device(0)
g = torch.Generator(device="cuda").manual_seed(1)
feats = torch.relu(torch.randn((n, 360, 128), device="cuda", generator=g) + 0.1).contiguous()
q = feats[3:4].contiguous()
out = {}
res = {}
for name, env in (("split", "0"), ("fused", "1"), ("split2", "0")):
    os.environ["OVN_DELTA_FUSED"] = env
    eng = OvnEngine(64, 900, 4)
    eng.load_weights(S.make_trained_like_weights(4, seed=0), S.REFERENCE_MODEL_CFG)
    spec, qs = eng.spectrum(feats), eng.spectrum(q)
    for _ in range(2):
        r = eng.heads(feats, q, spec_l=spec, spec_r=qs, want_logit=True)
    torch.cuda.synchronize()
    eng.profile_begin()
    for _ in range(5):
        r = eng.heads(feats, q, spec_l=spec, spec_r=qs, want_logit=True)
    torch.cuda.synchronize()
    p = eng.profile_end()
    out[name] = r["logit"].cpu().numpy()
    res[name] = {k: round(p[k][0] / max(p[k][1], 1), 4) for k in ("delta_prep", "delta_c12", "delta_c2", "c_conv3")}
    print(name, res[name], flush=True)
    del eng
res["max_abs_diff_split_vs_fused"] = float(np.abs(out["split"] - out["fused"]).max())
res["max_abs_diff_split_vs_split2"] = float(np.abs(out["split"] - out["split2"]).max())
res["logit_range"] = [float(out["fused"].min()), float(out["fused"].max())]
print(json.dumps(res))
