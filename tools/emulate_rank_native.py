"""Developer tool: per-iteration cost of ONE rank of a `world`-GPU run of the loops INSIDE the library (cglb_dist_pcg_solve), measured on
a single GPU with collectives that do nothing (cglb_comm_init_callbacks with no-op callbacks).  The kernels see exactly the per-rank
workload (1/world of the K_ff triangle and of the panel) and the library enqueues the same kernels in the same order as in a real
run; values are meaningless (partials are never summed), only the timing and the launch count are.  Missing from a real run: the
latency of the three RCCL collectives per iteration.  Beside it the host-driven twin (tools/emulate_rank.py's loop) on the same box.
usage: python tools/emulate_rank_native.py [world=8] [iters=30]        (under rocprofv3 --kernel-trace --stats for the launch counts)
"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ctypes import byref, c_double, c_int
from cglb_amd import _lib
from cglb_amd.data import synthetic_problem, trained_like_hypers
from cglb_amd.distributed import row_partition
from cglb_amd.hip_context import HipContext, _ptr

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
N, D, M = int(os.environ.get("N", 100000)), int(os.environ.get("D", 8)), int(os.environ.get("M", 1024))
X, y, Z = synthetic_problem(N, D, M, 0)
h = trained_like_hypers(D)
per, parts = row_partition(N, world)
ctx = HipContext(X, y, M, os.environ.get("KIND", "rbf"), row_range=parts[0])
calls = {"ar": 0, "ag": 0}


def _ar(user, buf, count, dtype, stream):
    calls["ar"] += 1
    return 0


def _ag(user, buf, count, dtype, stream):
    calls["ag"] += 1
    return 0


keep = (_lib.ALLREDUCE_FN(_ar), _lib.ALLGATHER_FN(_ag))
_lib.check(ctx.lib.cglb_comm_init_callbacks(ctx._ctx, world, 0, keep[0], keep[1], None), ctx._ctx)
ctx.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-6)
_lib.check(ctx.lib.cglb_dist_setup(ctx._ctx), ctx._ctx)
torch.cuda.synchronize(); t0 = time.perf_counter()
_lib.check(ctx.lib.cglb_dist_setup(ctx._ctx), ctx._ctx)
torch.cuda.synchronize(); t_setup = time.perf_counter() - t0
b = torch.from_numpy(y).to(ctx.device)
steps, half = c_int(), c_double()


def solve(n_it, lookahead):
    ctx.set_option("pcg_lookahead", lookahead)
    v = torch.zeros(N, dtype=torch.float64, device=ctx.device)
    torch.cuda.synchronize(); t = time.perf_counter()
    # tolerance below any value: the loop runs max_iter iterations (the un-reduced partials keep the scalars finite for a few dozen steps)
    rc = ctx.lib.cglb_dist_pcg_solve(ctx._ctx, _ptr(b), _ptr(v), -1e300, n_it, 40, byref(steps), byref(half))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    if rc != 0:
        print("solve ended early:", ctx.lib.cglb_last_error(ctx._ctx).decode(), "after", steps.value, "steps")
    return dt, steps.value


solve(5, 1)
for la in (0, 1):
    c0 = dict(calls)
    dt, st = solve(iters, la)
    print(f"world={world} library loop, look-ahead {la}: {1e3 * dt / max(st, 1):.3f} ms per iteration ({st} iterations; "
          f"{(calls['ar'] - c0['ar']) / max(st, 1):.2f} all-reduce + {(calls['ag'] - c0['ag']) / max(st, 1):.2f} all-gather calls per iteration)", flush=True)
k1 = ctx.time_kernel(4, 10)
pre = ctx.time_kernel(1, 10)
print(f"world={world}: setup {1e3 * t_setup:.2f} ms; cyclic pair kernel alone {k1:.3f} ms; local preconditioner kernels {pre:.3f} ms", flush=True)
ctx.lib.cglb_comm_destroy(ctx._ctx)
ctx.close()
