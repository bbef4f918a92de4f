"""Developer tool: time K_ff mat-vec variants across D (diagnoses which pipe bounds the kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cglb_amd.data import synthetic_problem, trained_like_hypers
from cglb_amd.hip_context import HipContext
N = int(os.environ.get("N", 100000))
for D in [int(d) for d in os.environ.get("DS", "2,6,8,14").split(",")]:
    X, y, Z = synthetic_problem(N, D, 64, 0)
    h = trained_like_hypers(D)
    for kind in os.environ.get("KINDS", "rbf").split(","):
        ctx = HipContext(X, y, 64, kind)
        line = f"D={D:2d} {kind:8s}"
        for variant in [int(v) for v in os.environ.get("VARIANTS", "0,1").split(",")]:
            ctx.set_option("kff_variant", variant)
            ctx.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-6)
            ms = ctx.time_kernel(3, 5)
            line += f" | v{variant}: {ms:7.3f} ms ({N*N/ms/1e6:7.1f} Gpair/s)"
        print(line, flush=True)
        ctx.close()
