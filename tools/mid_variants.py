"""Developer tool: mat-vec time of the mid-width instances (padded widths 48, 64, 80, 96) for several builds of libcglb_hip.so.
  python tools/mid_variants.py LIB1.so LIB2.so ..."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from cglb_amd.data import synthetic_problem
from cglb_amd.hip_context import HipContext
N, M = 50000, 64
out = []
for D in (40, 64, 77, 90):
    X, y, Z = synthetic_problem(N, D, M, 0)
    for kind in ("rbf", "matern32"):
        ctx = HipContext(X, y, M, kind)
        ctx.set_hypers(np.full(D, 1.2 * np.sqrt(D)), 1.0, 0.05, 0.0, Z, 1e-6)
        ctx.setup()
        out.append((D, kind, round(min(ctx.time_kernel(3, 5) for _ in range(3)), 3), round(min(ctx.time_kernel(2, 3) for _ in range(2)), 3), round(min(ctx.time_kernel(0, 5) for _ in range(3)), 3)))
        ctx.close()
print("RES", out, flush=True)
''' % ROOT
res = {}
for lib in sys.argv[1:]:
    env = dict(os.environ, CGLB_HIP_LIB=os.path.abspath(lib))
    o = subprocess.run([sys.executable, "-c", W], env=env, capture_output=True, text=True)
    line = [l for l in o.stdout.splitlines() if l.startswith("RES")]
    res[os.path.basename(lib)] = eval(line[0][4:]) if line else o.stderr[-400:]
libs = list(res)
print("columns:", libs)
for i in range(len(res[libs[0]])):
    a = res[libs[0]][i]
    print(f"D={a[0]:2d} {a[1]:8s} pair kernel " + " | ".join(f"{res[l][i][2]:7.3f}" for l in libs) + " ms    gradient pass " + " | ".join(f"{res[l][i][3]:7.2f}" for l in libs) + " ms    whole mat-vec " + " | ".join(f"{res[l][i][4]:7.3f}" for l in libs) + " ms")
