import cProfile, pstats, sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"), "tools"))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","/root/repo"))
import torch
torch.cuda.set_device(0)
from bench_infer_api import api_sweep
api_sweep(50)
pr = cProfile.Profile()
pr.enable()
r = api_sweep(300)
pr.disable()
print(r)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
