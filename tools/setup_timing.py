"""Developer tool: cost of the fixed part of an evaluation (common terms + gradient algebra) at a given shape.
Prints the blocked Cholesky alone (cglb_time_kernel id 5: K_uu build + factorisation), the whole cglb_setup and one
objective+gradient evaluation with and without the CG solve, so that changes to the small-matrix kernels can be compared."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cglb_amd.data import synthetic_problem
from cglb_amd.hip_context import HipContext

N, D, M = int(os.environ.get("N", 40000)), int(os.environ.get("D", 8)), int(os.environ.get("M", 1024))
kind = os.environ.get("KIND", "rbf")
dtype = torch.float32 if os.environ.get("DTYPE") == "fp32" else torch.float64
X, y, Z = synthetic_problem(N, D, M, 0)
ctx = HipContext(torch.as_tensor(X), torch.as_tensor(y), M, kind, dtype=dtype)
for name in sys.argv[1:]:
    k, v = name.split("="); ctx.set_option(k, int(v))
ctx.set_hypers(np.full(D, 1.5), 1.0, 0.05, 0.0, torch.as_tensor(Z), 1e-6 if dtype == torch.float64 else 1e-5)
ctx.setup()
t_chol = ctx.time_kernel(5, 20)
ctx.setup(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    ctx.set_hypers(np.full(D, 1.5), 1.0, 0.05, 0.0, torch.as_tensor(Z), 1e-6 if dtype == torch.float64 else 1e-5)
    ctx.setup()
torch.cuda.synchronize(); t_setup = (time.perf_counter() - t0) / 10
v = torch.zeros(N, dtype=dtype, device=ctx.device)
res = ctx.objective_and_grad(v, True, 1.0, 100, 40, with_grad=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    ctx.set_hypers(np.full(D, 1.5), 1.0, 0.05, 0.0, torch.as_tensor(Z), 1e-6 if dtype == torch.float64 else 1e-5)
    r2 = ctx.objective_and_grad(v, False, with_grad=True)
torch.cuda.synchronize(); t_eval = (time.perf_counter() - t0) / 5
print(f"N={N} D={D} M={M} {kind} {str(dtype)[6:]}: K_uu + Cholesky {t_chol*1e3:.0f} us; set_hypers + setup {t_setup*1e3:.2f} ms; "
      f"evaluation without CG (setup + bound + gradient) {t_eval*1e3:.2f} ms; steps {res.steps} bound {res.bound:.9f} / {r2.bound:.9f}", flush=True)
