"""Developer tool: time the GPU inducing-point selection (cglb_select_inducing) at the headline shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cglb_amd.data import synthetic_problem
from cglb_amd.hip_context import HipContext
for N, D, M, kind in [(100000, 8, 1024, "rbf"), (100000, 8, 1024, "matern32"), (430000, 3, 1024, "rbf"), (200000, 8, 2048, "matern32")]:
    X, y, Z = synthetic_problem(N, D, 8, 0)
    ctx = HipContext(X, y, M, kind)
    ctx.select_inducing(np.ones(D), 1.0)
    t0 = time.perf_counter()
    idx, tr = ctx.select_inducing(np.ones(D), 1.0)
    dt = time.perf_counter() - t0
    traffic = N * 8 * M * (M - 1) / 2 + 3 * N * 8 * M
    print(f"N={N} D={D} M={M} {kind}: {dt*1e3:.1f} ms, {traffic/dt/1e12:.2f} TB/s algorithmic, trace {tr:.1f}", flush=True)
    ctx.close()
