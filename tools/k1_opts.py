"""Developer tool: time the K_ff mat-vec under different option settings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cglb_amd.data import synthetic_problem, trained_like_hypers
from cglb_amd.hip_context import HipContext
N = int(os.environ.get("N", 100000)); D = int(os.environ.get("D", 8)); kind = os.environ.get("KIND", "rbf")
X, y, Z = synthetic_problem(N, D, 64, 0)
h = trained_like_hypers(D)
dtype = torch.float32 if os.environ.get("DTYPE", "fp64") == "fp32" else torch.float64
ctx = HipContext(X, y, 64, kind, dtype=dtype)
ctx.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-6)
for spec in sys.argv[1:]:
    for kv in spec.split(","):
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    ms3 = ctx.time_kernel(3, 5)
    ms0 = ctx.time_kernel(0, 5)
    print(f"{spec:40s} pair kernel {ms3:7.3f} ms | full mat-vec {ms0:7.3f} ms", flush=True)
