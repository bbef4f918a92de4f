"""Developer tool: phases of one cold evaluation at a mid-width shape (cglb_set_option "eval_profile")."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cglb_amd.data import synthetic_problem
from cglb_amd.hip_context import HipContext

N, M = 50000, 1024
for kind, D in (("rbf", 77), ("matern32", 90)):
    X, y, Z = synthetic_problem(N, D, M, 0)
    ctx = HipContext(X, y, M, kind)
    t0 = time.perf_counter(); ctx.set_hypers(np.full(D, 1.2 * np.sqrt(D)), 1.0, 0.05, 0.0, Z, 1e-6); torch.cuda.synchronize(); t_h = time.perf_counter() - t0
    v = torch.zeros(N, dtype=torch.float64, device=ctx.device)
    ctx.objective_and_grad(v, True, 1.0, 100, 40)
    ctx.set_option("eval_profile", 1)
    for _ in range(3):
        v.zero_()
        t0 = time.perf_counter(); ctx.set_hypers(np.full(D, 1.2 * np.sqrt(D)), 1.0, 0.05, 0.0, Z, 1e-6); torch.cuda.synchronize(); t_h = time.perf_counter() - t0
        t0 = time.perf_counter(); r = ctx.objective_and_grad(v, True, 1.0, 100, 40); torch.cuda.synchronize(); t_e = time.perf_counter() - t0
    ctx.set_option("eval_profile", 0)
    n = ctx.get_stat("eval_count")
    print(f"{kind} N={N} D={D} M={M}: set_hypers {1e3*t_h:.1f} ms, evaluation {1e3*t_e:.1f} ms ({r.steps} steps): setup {ctx.get_stat('eval_setup_ms')/n:.1f}, "
          f"pcg {ctx.get_stat('eval_pcg_ms')/n:.1f}, final {ctx.get_stat('eval_final_ms')/n:.2f}, gradient {ctx.get_stat('eval_grad_ms')/n:.1f} ms", flush=True)
    ctx.close()
