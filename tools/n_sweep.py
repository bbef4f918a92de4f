"""Developer tool: pair-kernel, mat-vec, preconditioner and gradient-pass time across N at fixed D, M (geometry check of the work lists:
chunk halving for small N, slab sizes for large N).  usage: python tools/n_sweep.py [D] [M] [kind]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cglb_amd.data import synthetic_problem, trained_like_hypers
from cglb_amd.hip_context import HipContext
D = int(sys.argv[1]) if len(sys.argv) > 1 else 8
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
kind = sys.argv[3] if len(sys.argv) > 3 else "rbf"
for N in (2000, 5000, 10000, 20000, 40000, 80000, 160000, 320000):
    X, y, Z = synthetic_problem(N, D, min(M, N), 0)
    h = trained_like_hypers(D)
    ctx = HipContext(X, y, min(M, N), kind)
    ctx.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-6)
    ctx.setup()
    k1, mv, pc, gr = (min(ctx.time_kernel(w, r) for _ in range(2)) for w, r in ((3, 5), (0, 5), (1, 10), (2, 3)))
    pairs = ctx.get_stat("k1_pairs_per_launch")
    print(f"N={N:7d}: pair kernel {k1:9.3f} ms ({pairs / k1 / 1e9:6.2f} Tpair/s)  mat-vec {mv:9.3f}  precond {pc:7.3f} ms ({(2*min(M,N)*N*8+3*N*8)/pc/1e9:6.2f} TB/s)  "
          f"gradient {gr:9.3f} ms ({pairs / gr / 1e9:6.2f} Tpair/s)", flush=True)
    ctx.close()
