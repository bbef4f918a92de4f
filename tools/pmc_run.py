"""Small fixed workload for rocprofv3 counter passes: a few launches of each hot kernel at the headline size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cglb_amd.data import synthetic_problem, trained_like_hypers
from cglb_amd.hip_context import HipContext
N, D, M = int(os.environ.get("N", 100000)), int(os.environ.get("D", 8)), int(os.environ.get("M", 1024))
dtype = torch.float32 if os.environ.get("DTYPE", "fp64") == "fp32" else torch.float64
X, y, Z = synthetic_problem(N, D, M, 0)
h = trained_like_hypers(D)
ctx = HipContext(X, y, M, os.environ.get("KIND", "rbf"), dtype=dtype)
if "SYM_ORDER" in os.environ:
    ctx.set_option("sym_order", int(os.environ["SYM_ORDER"]))
ctx.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-6)
ctx.setup()
ctx.time_kernel(0, 3)
ctx.time_kernel(1, 3)
ctx.time_kernel(2, 2)
torch.cuda.synchronize()
print("done")
