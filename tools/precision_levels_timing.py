"""Developer tool: pair-kernel and gradient-pass time and the headline evaluation at the three precision levels (same box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cglb_amd.data import synthetic_problem, trained_like_hypers
from cglb_amd.hip_context import HipContext
N, D, M = 100000, 8, 1024
X, y, Z = synthetic_problem(N, D, M, 0)
h = trained_like_hypers(D)
for kind in ("rbf", "matern32"):
    ctx = HipContext(X, y, M, kind)
    out = {}
    for rnd in range(2):
        for prec in (1, 0, 2):
            ctx.set_option("precision", prec)
            ctx.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-6)
            ctx.setup()
            k1 = min(ctx.time_kernel(3, 10) for _ in range(2)); gr = min(ctx.time_kernel(2, 5) for _ in range(2))
            v = torch.zeros(N, dtype=torch.float64, device=ctx.device)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = ctx.objective_and_grad(v, True, 1.0, 100, 40)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            out.setdefault(prec, []).append((k1, gr, dt, r.steps, r.bound))
    for prec in (0, 1, 2):
        k1, gr, dt, st, b = min(out[prec])
        print(f"{kind} precision {prec}: pair kernel {k1:.3f} ms, gradient pass {gr:.3f} ms, evaluation {1e3*dt:.1f} ms ({st} steps, bound {b:.9f})", flush=True)
    ctx.close()
