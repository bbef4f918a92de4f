"""Developer tool: per-iteration cost of ONE rank of a `world`-GPU run, measured on a single GPU without collectives.

The cyclic-symmetric driver is run with a communicator that reports (world, rank) but whose collectives do nothing, so the
kernels see exactly the per-rank workload (1/world of the K_ff triangle, 1/world of the panel) and the Python driver issues
the same sequence of calls.  Values are meaningless (partials are never summed); only the timing is.  What is missing from
a real run is the latency of the three collectives per iteration.
usage: python tools/emulate_rank.py [world=8] [iters=40]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cglb_amd.data import synthetic_problem, trained_like_hypers
from cglb_amd.hip_context import HipContext
from cglb_amd.distributed import Comm, HipSymLocalOps, SymShardedCGLB, row_partition

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
N, D, M = int(os.environ.get("N", 100000)), 8, int(os.environ.get("M", 1024))


class FakeComm(Comm):
    def __init__(self, world, rank):
        self.group, self.active, self.world, self.rank, self.force = None, False, world, rank, False

    def allreduce(self, t):
        pass

    def allgather_inplace(self, buf, per):
        pass


X, y, Z = synthetic_problem(N, D, M, 0)
h = trained_like_hypers(D)
per, parts = row_partition(N, world)
rank = 0
ctx = HipContext(X, y, M, "rbf", row_range=parts[rank])
if "SYM_CHUNK" in os.environ:
    ctx.set_option("sym_chunk", int(os.environ["SYM_CHUNK"]))
ctx.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-6)
drv = SymShardedCGLB(HipSymLocalOps(ctx), FakeComm(world, rank))
t0 = time.perf_counter(); drv.setup(); torch.cuda.synchronize(); t_setup = time.perf_counter() - t0
t0 = time.perf_counter(); drv.setup(); torch.cuda.synchronize(); t_setup = time.perf_counter() - t0
ops = drv.ops
drv.p.normal_(); drv.r.normal_(); drv.rz.fill_(1.0)


def iteration(sync=True):
    drv.matvec(drv.p, drv.Ap)
    ops.vec_dot(N, drv.p, drv.Ap, drv.pAp)
    ops.vec_update_v_r(N, drv.v, drv.r, drv.p, drv.Ap, drv.rz, drv.pAp, True)
    drv._precond_and_direction(drv.rz_new, drv.rz, False)
    drv.rz, drv.rz_new = drv.rz_new, drv.rz
    drv.rz.fill_(1.0)   # the partials of the other ranks never arrive here (no-op collectives): keep the scalars finite
    if sync:
        return float(drv.rz.item())


for _ in range(5):
    iteration()
for label, sync in (("host sync every iteration", True), ("no host sync (queue depth only)", False)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        iteration(sync)
    torch.cuda.synchronize()
    print(f"world={world}: {1e3*(time.perf_counter()-t0)/iters:.3f} ms per iteration, {label}", flush=True)
k1 = ctx.time_kernel(4, 10)
pre = ctx.time_kernel(1, 10)
print(f"world={world}: setup {1e3*t_setup:.2f} ms; cyclic pair kernel alone {k1:.3f} ms; local preconditioner kernels {pre:.3f} ms", flush=True)
