"""Developer tool: what the option "final_matvec" = 0 (K v after the solve taken from the PCG recurrence residual instead of a fresh
mat-vec, models.py:280) changes - bound, gradient, time per evaluation - cold and warm starts, both kernels, both precision levels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cglb_amd.data import synthetic_problem, trained_like_hypers, reference_init_hypers
from cglb_amd.hip_context import HipContext

for (N, D, M) in ((2000, 3, 32), (20000, 8, 256), (100000, 8, 1024)):
    X, y, Z = synthetic_problem(N, D, M, 0)
    for kind in ("rbf", "matern32"):
        for hname, h in (("trained", trained_like_hypers(D)), ("init", reference_init_hypers(D))):
            for prec in (1, 0):
                out = {}
                for fm in (1, 0):
                    ctx = HipContext(X, y, M, kind)
                    ctx.set_option("precision", prec); ctx.set_option("final_matvec", fm)
                    ctx.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-6)
                    v = torch.zeros(N, dtype=torch.float64, device=ctx.device)
                    r_cold = ctx.objective_and_grad(v, True, 1.0, 100, 40)
                    # warm start at slightly moved hypers (what a line-search step of the optimiser sees)
                    ctx.set_hypers(np.asarray(h["lengthscales"]) * 1.02, h["variance"] * 0.98, h["noise"] * 1.03, h["mean"] + 0.01, Z, 1e-6)
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    r_warm = ctx.objective_and_grad(v, True, 1.0, 100, 40)
                    torch.cuda.synchronize(); t_warm = time.perf_counter() - t0
                    out[fm] = (r_cold, r_warm, t_warm)
                    ctx.close()
                def rel(a, b): return abs(a - b) / abs(b)
                def grel(a, b): return max(np.abs(np.asarray(a[k]) - np.asarray(b[k])).max() / (np.abs(np.asarray(b[k])).max() + 1e-300) for k in ("lengthscales", "variance", "noise", "mean", "Z"))
                (c1, w1, t1), (c0, w0, t0_) = out[1], out[0]
                print(f"N={N:6d} D={D} M={M:4d} {kind:8s} {hname:7s} prec={prec}: cold steps {c1.steps}/{c0.steps} bound {rel(c0.bound, c1.bound):.1e} grad {grel(c0.grad, c1.grad):.1e} | "
                      f"warm steps {w1.steps}/{w0.steps} bound {rel(w0.bound, w1.bound):.1e} grad {grel(w0.grad, w1.grad):.1e} | warm eval {1e3*t1:.1f} -> {1e3*t0_:.1f} ms", flush=True)
