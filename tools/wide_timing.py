"""Developer tool: cost of the wide-input path (D > 32, kernels_wide.hip) at the shapes of the wide Wilson sets."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cglb_amd.data import synthetic_problem
from cglb_amd.hip_context import HipContext

for kind, N, D, M in (("rbf", 50000, 77, 1024), ("matern32", 50000, 77, 1024), ("rbf", 53500, 385, 1024), ("rbf", 100000, 90, 1024), ("rbf", 50000, 32, 1024)):
    X, y, Z = synthetic_problem(N, D, M, 0)
    ctx = HipContext(X, y, M, kind)
    ctx.set_hypers(np.full(D, 1.2 * np.sqrt(D)), 1.0, 0.05, 0.0, Z, 1e-6)
    ctx.setup()
    mv = ctx.time_kernel(0, 3)
    pre = ctx.time_kernel(1, 5)
    gr = ctx.time_kernel(2, 2)
    v = torch.zeros(N, dtype=torch.float64, device=ctx.device)
    ctx.objective_and_grad(v, True, 1.0, 100, 40)
    v.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
    r = ctx.objective_and_grad(v, True, 1.0, 100, 40)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{kind} N={N} D={D} M={M}: mat-vec {mv:.2f} ms ({2.0*N*N*D/mv/1e9:.1f} TFLOP/s on the Gram GEMM alone), preconditioner {pre:.3f} ms, gradient N^2 pass {gr:.1f} ms, "
          f"cold evaluation {1e3*dt:.0f} ms ({r.steps} CG steps)", flush=True)
    ctx.close()
