"""Developer tool: the gradient algebra against L = chol(K_uu) - products with the explicit inverse (grad_trsm 0) or backward-stable
triangular solves (grad_trsm 1) - at realistic shapes: |g0 - g1| per gradient block against the oracle's own noise floor of that block
under eps-sized moves of Z (oracle.grad_roundoff_spread, inducing_only), and the time of the two variants.  Decides the threshold of
the automatic choice (CGLB_LINV_DIAG_RATIO)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import cglb_oracle as orc
from cglb_amd.distributed import HipSymLocalOps
from cglb_amd.hip_context import HipContext

cases = [("rbf", 20000, 8, 512, 1.5, 0.05), ("matern32", 20000, 8, 512, 1.5, 0.05), ("rbf", 30000, 3, 512, 1.5, 0.05), ("rbf", 100000, 8, 1024, 1.5, 0.05),
         ("rbf", 100000, 8, 1024, 3.0, 0.01)]
for kind, N, D, M, ell, noise in cases:
    X, y, Z = orc.synthetic_problem(N, D, M, 0)
    hyp = orc.Hypers(np.full(D, ell), 1.0, noise, 0.0, Z, 1e-6)
    res, secs = {}, {}
    v = w = None
    for mode in (0, 1, 2):
        ctx = HipContext(X, y, M, kind)
        ctx.set_option("grad_trsm", mode)
        ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
        if v is None:
            v = torch.zeros(N, dtype=torch.float64, device=ctx.device)
            ctx.objective_and_grad(v, True, 1.0, 100, 40, with_grad=False)
        ctx.objective_and_grad(v, False); ctx.objective_and_grad(v, False)     # warm the lazily loaded rocBLAS kernels of this variant
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            r = ctx.objective_and_grad(v, False)
        torch.cuda.synchronize(); secs[mode] = (time.perf_counter() - t0) / 3
        res[mode] = r.grad
        if w is None:
            wt = torch.empty(N, dtype=torch.float64, device=ctx.device)
            HipSymLocalOps(ctx).obj_w(wt)
            w = wt.cpu().numpy()
        ratio = ctx.get_stat("L_diag_ratio")
        ctx.close()
    t0 = time.perf_counter()
    floor = orc.grad_roundoff_spread(kind, X, hyp, v.cpu().numpy(), w, probes=2, inducing_only=True)
    tf = time.perf_counter() - t0
    out = " ".join(f"{k}: |g0-g1| {np.abs(np.asarray(res[0][k]) - np.asarray(res[1][k])).max():.1e} |g2-g1| {np.abs(np.asarray(res[2][k]) - np.asarray(res[1][k])).max():.1e} "
                   f"floor {floor[k]:.1e} max|g| {np.abs(np.asarray(res[1][k])).max():.1e};" for k in ("Z", "lengthscales", "variance"))
    print(f"{kind} N={N} D={D} M={M} l={ell} noise={noise}: diag ratio {ratio:.3g}; eval without CG {1e3*secs[0]:.2f} ms (inverse) / {1e3*secs[1]:.2f} ms (solves) / "
          f"{1e3*secs[2]:.2f} ms (refined inverse); {out} [floor took {tf:.0f} s]", flush=True)
