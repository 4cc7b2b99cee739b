import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from tools import synthetic as S
from overlapnet_amd.engine import OvnEngine
torch.cuda.set_device(0)
eng = OvnEngine(64, 900, 4)
eng.load_weights(S.make_test_weights(4, seed=0), S.REFERENCE_MODEL_CFG)
g = torch.Generator(device="cuda").manual_seed(1)
feats = (torch.relu(torch.randn((1025, 360, 128), device="cuda", generator=g) + 0.1) * 7.3).contiguous()
sp = eng.spectrum(feats)
# fp64 reference on a few volumes
x = feats[:3].double().cpu().numpy()
F = np.fft.fft(x, axis=1)[:, :181, :]          # (3, 181, 128)
ref_re = np.transpose(F.real, (0, 2, 1)); ref_im = np.transpose(F.imag, (0, 2, 1))
got = sp[:3].cpu().numpy()
sc = np.abs(F).max()
print("max err re %.3g im %.3g (scale %.3g)" % (np.abs(got[:, :, :181] - ref_re).max() / sc, np.abs(got[:, :, 184:365] - ref_im).max() / sc, sc),
      "pad zero:", float(np.abs(got[:, :, 181:184]).max()), float(np.abs(got[:, :, 365:]).max()))
eng.set_head_precision("f32"); sp32 = eng.spectrum(feats[:3]).cpu().numpy(); eng.set_head_precision("f16x3")
print("fp32 path err re %.3g" % (np.abs(sp32[:, :, :181] - ref_re).max() / sc))
for mode in ("f16x3", "f32", "f16x3"):
    eng.set_head_precision(mode)
    for _ in range(2): eng.spectrum(feats, out=sp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): eng.spectrum(feats, out=sp)
    torch.cuda.synchronize(); print(mode, "spectrum of 1025 volumes: %.3f ms" % (1e3 * (time.perf_counter() - t0) / 5))
eng.set_head_precision("f16x3")
q = feats[3:4].contiguous()
for _ in range(3): eng.spectrum(q)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): eng.spectrum(q)
torch.cuda.synchronize(); print("single volume: %.1f us" % (1e6 * (time.perf_counter() - t0) / 20))
