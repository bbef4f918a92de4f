import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cglb_amd.data import synthetic_problem, trained_like_hypers
from cglb_amd.hip_context import HipContext
X, y, Z = synthetic_problem(100000, 8, 1024, 0)
h = trained_like_hypers(8)
ctx = HipContext(X, y, 1024, "rbf")
ctx.set_option("chol_mode", int(os.environ.get("CHOL", "1")))
ctx.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-6)
for _ in range(3):
    ctx.setup()
torch.cuda.synchronize()
