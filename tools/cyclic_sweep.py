"""Developer tool: per-rank time of the cyclic-symmetric pair kernel for emulated world sizes (one GPU, no collectives)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cglb_amd import _lib
from cglb_amd.data import synthetic_problem, trained_like_hypers
from cglb_amd.hip_context import HipContext
N = int(os.environ.get("N", 100000))
X, y, Z = synthetic_problem(N, 8, 64, 0)
h = trained_like_hypers(8)
ctx = HipContext(X, y, 64, "rbf")
ctx.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-6)
for world in (1, 2, 4, 8):
    for chunk in (128, 256, 512, 1024):
        ctx.set_option("sym_chunk", chunk)
        ts = []
        for rank in sorted({0, world // 2, world - 1}):
            _lib.check(ctx.lib.cglb_set_parallel(ctx._ctx, world, rank), ctx._ctx)
            ts.append(ctx.time_kernel(4, 5))
        print(f"world={world} chunk={chunk:5d}: per-rank pair kernel {max(ts):7.3f} ms (ranks: {', '.join(f'{t:.3f}' for t in ts)}); ideal {3.6/world:.3f}", flush=True)
