import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from tools import synthetic as S
from overlapnet_amd.engine import OvnEngine
torch.cuda.set_device(0)
eng = OvnEngine(64, 900, 4)
eng.load_weights(S.make_trained_like_weights(4, seed=0), S.REFERENCE_MODEL_CFG)
g = torch.Generator(device="cuda").manual_seed(1)
for n in (5, 64, 1024):
    feats = torch.relu(torch.randn((n, 360, 128), device="cuda", generator=g) + 0.1).contiguous()
    q = feats[3:4].contiguous()
    outs = []
    for rep in range(3):
        r = eng.heads(feats, q, want_logit=True)
        o2, o3 = eng.debug_head_activations(min(n, 64))
        outs.append((r["logit"].cpu().numpy().copy(), o2.cpu().numpy().copy()))
    for rep in (1, 2):
        dl = np.abs(outs[rep][0] - outs[0][0]).max(); d2 = np.abs(outs[rep][1] - outs[0][1]).max()
        bad = np.argwhere(np.abs(outs[rep][1] - outs[0][1]).reshape(outs[0][1].shape[0], -1).max(axis=1) > 0).ravel()
        print("n", n, "rep", rep, "max dlogit", dl, "max do2", d2, "pairs with o2 diff", bad[:10], flush=True)
