"""Developer tool: time the N^2 gradient pass (direct-difference form vs Gram/moment form) at the headline shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cglb_amd.data import synthetic_problem, trained_like_hypers
from cglb_amd.hip_context import HipContext
N, D = int(os.environ.get("N", 100000)), int(os.environ.get("D", 8))
X, y, Z = synthetic_problem(N, D, 64, 0)
h = trained_like_hypers(D)
for kind in ("rbf", "matern32"):
    ctx = HipContext(X, y, 64, kind)
    ctx.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-6)
    ctx.setup()
    for g in (0, 1, 0, 1):
        ctx.set_option("grad_gram", g)
        print(f"{kind} grad_gram={g}: {ctx.time_kernel(2, 3):.3f} ms", flush=True)
