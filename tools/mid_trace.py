"""Developer tool: four cold evaluations at a mid-width shape (N = 50 000, D = 77, M = 1024, RBF) for `rocprofv3 --kernel-trace --stats`;
tools/db_top.py prints the kernel table of the resulting database."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cglb_amd.data import synthetic_problem
from cglb_amd.hip_context import HipContext
N, M, D = 50000, 1024, 77
X, y, Z = synthetic_problem(N, D, M, 0)
ctx = HipContext(X, y, M, "rbf")
v = torch.zeros(N, dtype=torch.float64, device=ctx.device)
for _ in range(4):
    v.zero_()
    ctx.set_hypers(np.full(D, 1.2 * np.sqrt(D)), 1.0, 0.05, 0.0, Z, 1e-6)
    r = ctx.objective_and_grad(v, True, 1.0, 100, 40)
torch.cuda.synchronize()
print("steps", r.steps, "bound", r.bound)
ctx.close()
