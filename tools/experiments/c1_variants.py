"""Same-run timing of kernel variants of the split Delta path (library built with `make ABLATE=1` for the 1xx ablations).
    python tools/experiments/c1_variants.py c1=0 c1=2 c2=1 ...     (OVN_C1_VARIANT / OVN_C2_VARIANT per run)"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import synthetic as S
from overlapnet_amd.engine import OvnEngine
n = 1024
torch.cuda.set_device(0)
g = torch.Generator(device="cuda").manual_seed(1)
feats = torch.relu(torch.randn((n, 360, 128), device="cuda", generator=g) + 0.1).contiguous()
q = feats[3:4].contiguous()
eng = OvnEngine(64, 900, 4)
eng.load_weights(S.make_trained_like_weights(4, seed=0), S.REFERENCE_MODEL_CFG)
spec, qs = eng.spectrum(feats), eng.spectrum(q)
ref = None
for v in sys.argv[1:]:
    os.environ["OVN_C1_VARIANT"] = "0"
    os.environ["OVN_C2_VARIANT"] = "0"
    for kv in v.split(","):
        k, val = kv.split("=")
        os.environ["OVN_%s_VARIANT" % k.upper()] = val
    for _ in range(2):
        r = eng.heads(feats, q, spec_l=spec, spec_r=qs, want_logit=True)
    torch.cuda.synchronize()
    eng.profile_begin()
    for _ in range(5):
        r = eng.heads(feats, q, spec_l=spec, spec_r=qs, want_logit=True)
    torch.cuda.synchronize()
    p = eng.profile_end()
    lg = r["logit"].cpu().numpy()
    if ref is None: ref = lg
    print(v, {k: round(p[k][0] / max(p[k][1], 1), 4) for k in ("delta_prep", "delta_c12", "delta_c2", "c_conv3")}, "maxdiff", float(np.abs(lg - ref).max()), flush=True)
