"""Developer tool: how many mat-vecs the PCG look-ahead wastes in warm-started training evaluations (a stop right after a speculative
mat-vec), for several values of the look-ahead threshold (option "pcg_lookahead": 0 off, k > 0: speculate while 1/2 r^T P r of the
previous iteration > k * max_error)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cglb_amd.backend import interface
from cglb_amd.backend.callbacks import Logger
from cglb_amd.backend.models import CGLB, BaseKernel, GaussianLikelihood, InducingPointKernel, ScaleKernel
from cglb_amd.data import synthetic_problem, trained_like_hypers

N, D, M = int(os.environ.get("N", 100000)), 8, 1024
X, y, Z = synthetic_problem(N, D, M, 0)
h = trained_like_hypers(D)
for la in [int(a) for a in (sys.argv[1:] or ["4", "0", "32", "100"])]:
    base = BaseKernel("rbf", ard_num_dims=D); base.lengthscale = h["lengthscales"]
    scale = ScaleKernel(base); scale.outputscale = h["variance"]
    lik = GaussianLikelihood(lower_bound=1e-6); lik.noise = h["noise"]
    model = CGLB((X, y), lik, InducingPointKernel(scale, Z))
    hip = model.hip
    hip.set_option("pcg_lookahead", la)
    logger = Logger("/tmp/x", lambda: {}, lambda: {}, holdout_interval=-1, include_feval_log=True, verbose=False)
    walls = []
    orig = hip.objective_and_grad
    def timed(*a, **k):
        t = time.perf_counter(); r = orig(*a, **k); walls.append(1e3 * (time.perf_counter() - t)); return r
    hip.objective_and_grad = timed
    hip.set_option("k1_profile", 1)
    res = interface.optimize(model, ((X, y), (X[:8], y[:8])), 30, logger, "scipy")
    launches = hip.get_stat("k1_launches")
    steps = np.asarray(logger.logs["steps-per-feval"], dtype=np.int64)
    nfev = len(steps)
    # needed: per evaluation (after the un-logged warm-up one): steps + restarts + 1 initial mat-vec when v != 0
    needed = int(steps.sum() + (steps // 40).sum() + nfev)
    print(f"pcg_lookahead={la}: {nfev} evaluations, mean CG steps {steps.mean():.2f}, pair-kernel launches {int(launches)} (incl. the warm-up evaluation), "
          f"needed by the logged ones {needed}; median C-ABI call {np.median(walls[1:]):.2f} ms, mean {np.mean(walls[1:]):.2f} ms", flush=True)
    hip.close()
