"""Developer tool: per-evaluation time of the fused C path vs the Python-driven cyclic-symmetric driver at world_size 1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cglb_amd.data import synthetic_problem, trained_like_hypers
from cglb_amd.distributed import HipSymLocalOps, SymShardedCGLB
from cglb_amd.hip_context import HipContext
N = int(os.environ.get("N", 100000))
X, y, Z = synthetic_problem(N, 8, 1024, 0)
h = trained_like_hypers(8)
ctx = HipContext(X, y, 1024, "rbf")
ctx.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-6)
v = torch.zeros(N, dtype=torch.float64, device=ctx.device)
for rep in range(3):
    v.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
    r = ctx.objective_and_grad(v, True, 1.0); torch.cuda.synchronize()
    print(f"fused  {1e3*(time.perf_counter()-t0):8.2f} ms steps={r.steps}", flush=True)
drv = SymShardedCGLB(HipSymLocalOps(ctx))
for rep in range(3):
    drv.v.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
    r = drv.objective_and_grad(True, 1.0); torch.cuda.synchronize()
    print(f"driver {1e3*(time.perf_counter()-t0):8.2f} ms steps={r.steps}", flush=True)
