"""Developer tool: does replaying the small per-iteration kernels from a hipGraph shorten them?  (world-8 share of the panel)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cglb_amd.data import synthetic_problem, trained_like_hypers
from cglb_amd.hip_context import HipContext
from cglb_amd.distributed import HipSymLocalOps, row_partition

N, D, M, world = 100000, 8, 1024, 8
X, y, Z = synthetic_problem(N, D, M, 0)
h = trained_like_hypers(D)
per, parts = row_partition(N, world)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ctx = HipContext(X, y, M, "rbf", row_range=parts[0])
    ctx.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-6)
    ops = HipSymLocalOps(ctx)
    ops.set_parallel(world, 0)
    ops.setup_local(); ops.setup_finish()
    z = lambda n, dt=torch.float64: torch.zeros(n, dtype=dt, device=ctx.device)
    p, r, v, Ap, zf = z(N).normal_(), z(N).normal_(), z(N), z(N).normal_(), z(N)
    u, rz, rz2, pAp, scr = z(M), z(1).fill_(1.0), z(1), z(1), z(1)
    r0, r1 = parts[0]

    def small_kernels():
        ops.vec_dot(N, p, Ap, pAp)
        ops.vec_update_v_r(N, v, r, p, Ap, rz, pAp, True)
        ops.precond_u(r[r0:r1], u)
        ops.precond_z(r[r0:r1], u, zf[r0:r1], scr)
        ops.vec_dot(N, r, zf, rz2)
        ops.vec_update_p(N, p, zf, rz2, rz, False)

    for _ in range(3):
        small_kernels()
    s.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(200):
        small_kernels()
    e1.record(s); e1.synchronize()
    print(f"stream launches: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per group of small kernels", flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        small_kernels()
    g.replay(); s.synchronize()
    e0.record(s)
    for _ in range(200):
        g.replay()
    e1.record(s); e1.synchronize()
    print(f"graph replays  : {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per group of small kernels", flush=True)
