import sys, time, numpy as np, torch
sys.path.insert(0,'/root/repo')
from oracle import overlapnet_oracle as O
from tools import synthetic as S
from overlapnet_amd.engine import OvnEngine
C=4
w=S.make_test_weights(C,0)
eng=OvnEngine(64,900,C); eng.load_weights(w,S.REFERENCE_MODEL_CFG)
imgs=S.candidate_images(6,C,seed=3)
ref=O.leg_forward(imgs,w,S.REFERENCE_MODEL_CFG,np.float64).reshape(6,360,128)
x=torch.from_numpy(imgs).cuda()
for mode in ("f32","f16x3"):
    eng.set_leg_precision(mode)
    fv=eng.leg(x).cpu().numpy()
    print(mode,"rel err vs fp64 oracle", np.max(np.abs(fv-ref))/np.max(np.abs(ref)))
    big=torch.from_numpy(np.repeat(imgs[:2],64,axis=0)).cuda()
    eng.leg(big); torch.cuda.synchronize()
    eng.profile_begin()
    for _ in range(5): eng.leg(big)
    for _ in range(20): eng.leg(x[:1])
    torch.cuda.synchronize()
    p=eng.profile_end()["leg_conv"]
    t0=time.perf_counter()
    for _ in range(5): eng.leg(big)
    torch.cuda.synchronize(); t1=time.perf_counter()
    for _ in range(50): eng.leg(x[:1])
    torch.cuda.synchronize(); t2=time.perf_counter()
    print(mode,"batch128: %.3f ms -> %.0f scans/s ; single scan: %.3f ms"%((t1-t0)/5*1e3, 128*5/(t1-t0), (t2-t1)/50*1e3))
