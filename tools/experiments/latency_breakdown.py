import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from tools import synthetic as S
from overlapnet_amd.engine import OvnEngine
torch.cuda.set_device(0)
eng = OvnEngine(64, 900, 4)
eng.load_weights(S.make_test_weights(4, seed=0), S.REFERENCE_MODEL_CFG)
img = torch.from_numpy(S.candidate_images(1, 4, seed=5)).cuda()
g = torch.Generator(device="cuda").manual_seed(1)
for n in (1, 16, 64):
    feats = torch.relu(torch.randn((n, 360, 128), device="cuda", generator=g) + 0.1).contiguous()
    spec = eng.spectrum(feats)
    qf = torch.empty((1, 360, 128), device="cuda")
    def step():
        eng.leg(img, out=qf); qs = eng.spectrum(qf)
        return eng.heads(feats, qf, spec_l=spec, spec_r=qs)
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    eng.profile_begin()
    for _ in range(10): step()
    torch.cuda.synchronize(); p = eng.profile_end()
    print(n, "ms/query %.3f" % (1e3 * dt), {k: (round(v[0] / 10, 4), v[1] // 10) for k, v in p.items() if v[1]}, flush=True)
