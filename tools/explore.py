"""Exploratory timing of the hot-path pieces at a given size (developer tool, not the bench contract)."""
import argparse
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from cglb_amd.data import synthetic_problem, trained_like_hypers, reference_init_hypers
from cglb_amd.hip_context import HipContext

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100000)
ap.add_argument("--d", type=int, default=8)
ap.add_argument("--m", type=int, default=1024)
ap.add_argument("--kind", default="rbf")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--opts", default="")
args = ap.parse_args()

X, y, Z = synthetic_problem(args.n, args.d, args.m, 0)
t0 = time.time()
ctx = HipContext(X, y, args.m, args.kind)
for kv in filter(None, args.opts.split(",")):
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
print(f"ctx create+set_data {time.time()-t0:.2f}s", flush=True)
for name, h in (("init", reference_init_hypers(args.d)), ("trained", trained_like_hypers(args.d))):
    ctx.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-6)
    torch.cuda.synchronize()
    t0 = time.time(); ctx.setup(); torch.cuda.synchronize(); t_setup = time.time() - t0
    t0 = time.time(); ctx.setup(); torch.cuda.synchronize(); t_setup = time.time() - t0
    k1 = ctx.time_kernel(0, args.reps)
    pc = ctx.time_kernel(1, args.reps)
    gk = ctx.time_kernel(2, 2)
    pairs = args.n * args.n
    print(f"[{name}] setup {t_setup*1e3:.1f} ms | K_ff matvec {k1:.3f} ms ({pairs/k1/1e6:.1f} Gpair/s) | precond {pc:.3f} ms "
          f"({2*args.m*args.n*8/pc/1e6:.0f} GB/s) | grad_kff {gk:.3f} ms", flush=True)
    for rep in range(2):
        v = torch.zeros(args.n, dtype=torch.float64, device=ctx.device)
        torch.cuda.synchronize(); t0 = time.time()
        res = ctx.objective_and_grad(v, True, 1.0)
        torch.cuda.synchronize(); dt = time.time() - t0
        print(f"[{name}] eval {dt*1e3:.1f} ms steps={res.steps} half_rz={res.residual_error:.4f} bound={res.bound:.6f} "
              f"lower={res.lower:.4f} upper={res.upper:.4f}", flush=True)
