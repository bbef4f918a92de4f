"""Developer tool: one warm-started-like evaluation sequence at a BASELINE configuration's shape under `rocprofv3 --kernel-trace --stats`
(top kernels of the whole evaluation, not just the pair kernel).  python3 tools/config_trace.py c3|c4|c5"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cglb_amd.data import synthetic_problem, trained_like_hypers
from cglb_amd.hip_context import HipContext

which = sys.argv[1]
N, D, M, kind, dt = {"c3": (200_000, 8, 2048, "matern32", torch.float64), "c4": (430_000, 3, 1024, "rbf", torch.float64),
                     "c5": (1_000_000, 16, 4096, "rbf", torch.float32)}[which]
X, y, Z = synthetic_problem(N, D, M, 0)
h = trained_like_hypers(D)
ctx = HipContext(X, y, M, kind, dtype=dt)
ctx.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-4 if dt == torch.float32 else 1e-6)
v = torch.zeros(N, dtype=dt, device=ctx.device)
for it in range(3):
    ctx.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-4 if dt == torch.float32 else 1e-6)
    r = ctx.objective_and_grad(v, True, 1.0, 8, 40)     # 8 CG steps per evaluation: the training-loop regime
torch.cuda.synchronize()
print(which, "steps", r.steps, "bound", r.bound, flush=True)
ctx.close()
