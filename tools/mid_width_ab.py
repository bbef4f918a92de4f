"""Developer tool: the symmetric K_ff mat-vec of a mid-width input (32 < D <= 96, fp64) register-resident with the Gram chain in slices
(option wide_reg = 1, default) against the Gram tiles through rocBLAS (wide_reg = 0): time per mat-vec, agreement, cold evaluation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cglb_amd.data import synthetic_problem
from cglb_amd.hip_context import HipContext

cases = [("rbf", 50000, 77), ("matern32", 50000, 77), ("rbf", 50000, 40), ("rbf", 50000, 64), ("rbf", 50000, 90), ("matern32", 50000, 90), ("rbf", 100000, 90)]
if len(sys.argv) > 1:
    cases = cases[:int(sys.argv[1])]
M = 1024
for kind, N, D in cases:
    X, y, Z = synthetic_problem(N, D, M, 0)
    p = torch.from_numpy(np.random.default_rng(1).standard_normal(N)).cuda()
    out = {}
    for reg in (1, 0):
        ctx = HipContext(X, y, M, kind)
        ctx.set_option("wide_reg", reg)
        ctx.set_hypers(np.full(D, 1.2 * np.sqrt(D)), 1.0, 0.05, 0.0, Z, 1e-6)
        ctx.setup()
        mv = ctx.time_kernel(0, 3)
        gr = ctx.time_kernel(2, 2)
        Ap = ctx.matvec(p)
        v = torch.zeros(N, dtype=torch.float64, device=ctx.device)
        ctx.objective_and_grad(v, True, 1.0, 100, 40)
        v.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
        r = ctx.objective_and_grad(v, True, 1.0, 100, 40)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out[reg] = (mv, Ap, dt, r, gr)
        ctx.close()
    d = float((out[1][1] - out[0][1]).abs().max() / out[0][1].abs().max())
    gl1, gl0 = np.asarray(out[1][3].grad["lengthscales"]), np.asarray(out[0][3].grad["lengthscales"])
    dg = float(np.abs(gl1 - gl0).max() / np.abs(gl0).max())
    print(f"{kind} N={N} D={D}: mat-vec register-resident {out[1][0]:.2f} ms, Gram tiles {out[0][0]:.2f} ms; max difference {d:.1e}; "
          f"gradient N^2 pass {out[1][4]:.1f} ms against {out[0][4]:.1f} ms, lengthscale gradients differ by {dg:.1e}; "
          f"cold evaluation {1e3*out[1][2]:.0f} ms ({out[1][3].steps} steps) against {1e3*out[0][2]:.0f} ms ({out[0][3].steps} steps); "
          f"bounds {out[1][3].bound:.10g} / {out[0][3].bound:.10g}", flush=True)
