"""Does the chip sit at its power cap under the leg / the contraction kernel?  Samples rocm-smi while a kernel loop runs.
    python tools/experiments/power_probe.py leg|heads|idle [seconds]"""
import os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import synthetic as S
from overlapnet_amd.engine import OvnEngine
what = sys.argv[1] if len(sys.argv) > 1 else "leg"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
torch.cuda.set_device(0)
eng = OvnEngine(64, 900, 4)
eng.load_weights(S.make_test_weights(4, seed=0), S.REFERENCE_MODEL_CFG)
base = torch.from_numpy(S.candidate_images(64, 4, seed=5)).cuda()
imgs = base[torch.arange(1025) % 64].contiguous()
fv = torch.empty((1025, 360, 128), device="cuda")
eng.leg(imgs, out=fv)
sp = eng.spectrum(fv)
dc = eng.delta_cache(fv[:1024].contiguous())
samples = []
stop = [False]
def sampler():
    while not stop[0]:
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.perf_counter(), o.strip().replace("\n", " | ")))
        except Exception as e:
            samples.append((time.perf_counter(), "ERR %s" % e))
        time.sleep(0.3)
th = threading.Thread(target=sampler)
th.start()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < secs:
    if what == "leg":
        eng.leg(imgs, out=fv)
    elif what == "heads":
        eng.heads(fv[:1024], fv[1024:], spec_l=sp[:1024], spec_r=sp[1024:], dcache_l=dc)
    else:
        time.sleep(0.05)
    n += 1
    if n % 4 == 0:
        torch.cuda.synchronize()
torch.cuda.synchronize()
el = time.perf_counter() - t0
stop[0] = True
th.join()
print("%s: %d calls in %.2f s = %.3f ms per call" % (what, n, el, 1e3 * el / max(n, 1)))
for t, o in samples[1:8]:
    print("  %.2f s  %s" % (t - t0, o[:400]))
