"""Developer tool: run one objective+gradient evaluation of a BASELINE config on ONE GPU (size / overflow smoke test)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cglb_amd.data import synthetic_problem, trained_like_hypers
from cglb_amd.hip_context import HipContext
name, N, D, M, kind, dt = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
dtype = torch.float64 if dt == "fp64" else torch.float32
t0 = time.time()
X, y, Z = synthetic_problem(N, D, M, 0)
h = trained_like_hypers(D)
ctx = HipContext(X, y, M, kind, dtype=dtype)
ctx.set_hypers(h["lengthscales"], h["variance"], 0.2 if dt == "fp32" else h["noise"], h["mean"], Z, 1e-6 if dt == "fp64" else 1e-5)
print(f"[{name}] N={N} D={D} M={M} {kind} {dt}: context ready in {time.time()-t0:.1f}s", flush=True)
v = torch.zeros(N, dtype=dtype, device=ctx.device)
for rep in range(2):
    v.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
    r = ctx.objective_and_grad(v, True, 1.0); torch.cuda.synchronize()
    print(f"[{name}] eval {1e3*(time.perf_counter()-t0):9.1f} ms steps={r.steps} half_rz={r.residual_error:.4f} bound={r.bound:.6g} "
          f"lower={r.lower:.6g} upper={r.upper:.6g} |g_ls|={np.abs(r.grad['lengthscales']).max():.4g} finite={np.isfinite(r.grad['Z']).all()}", flush=True)
ms = ctx.time_kernel(0, 2)
print(f"[{name}] K_ff matvec {ms:.2f} ms ({N*N/ms/1e6:.0f} Gpair/s); mem {torch.cuda.mem_get_info()[0]/2**30:.0f} GiB free", flush=True)
