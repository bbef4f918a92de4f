"""Developer tool: pair-kernel and gradient-pass time across input dimensions D for TWO builds of libcglb_hip.so (regression check of the
per-D kernel instances: padded widths 1, 2, 3, 4, 6, 8, 12, 16, 24, 32; N = 60 000, RBF and Matern-3/2).
  python tools/d_sweep.py OLD.so NEW.so      (an older build: git archive <rev> cglb_amd/csrc include | tar -x -C build/old && make -C ... OUT=...)"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = r'''
import sys, os, numpy as np, torch
sys.path.insert(0, %r)
from cglb_amd.data import synthetic_problem, trained_like_hypers
from cglb_amd.hip_context import HipContext
N, M = 60000, 64
out = []
for D in (1, 2, 3, 4, 6, 8, 9, 12, 16, 18, 20, 24, 27, 28, 32):
    X, y, Z = synthetic_problem(N, D, M, 0)
    h = trained_like_hypers(D)
    for kind in ("rbf", "matern32"):
        ctx = HipContext(X, y, M, kind)
        ctx.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-6)
        ctx.setup()
        out.append((D, kind, round(min(ctx.time_kernel(3, 5) for _ in range(2)), 3), round(min(ctx.time_kernel(2, 3) for _ in range(2)), 3)))
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
for i in range(len(res[libs[0]])):
    a, b = res[libs[0]][i], res[libs[1]][i]
    print(f"D={a[0]:2d} {a[1]:8s} K1 {a[2]:7.3f} -> {b[2]:7.3f} ms ({100*(b[2]/a[2]-1):+5.1f} %)   grad {a[3]:7.3f} -> {b[3]:7.3f} ms ({100*(b[3]/a[3]-1):+5.1f} %)")
