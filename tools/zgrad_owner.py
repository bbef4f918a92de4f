"""Developer tool: which switch owns the Z-gradient error of sweep draw (2024, 186) - precision level or the explicit L^-1 of the
gradient algebra.  Prints |dZ_hip - dZ_oracle|_max for precision x grad_trsm and the oracle's own floor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import cglb_oracle as orc
from cglb_amd.hip_context import HipContext
from tools.fuzz_parity import named_case

for seed, idx in ((2024, 186), (2024, 187), (2024, 0)):
    c = named_case(seed, idx)
    N, D, M, kind = c["N"], c["D"], c["M"], c["kind"]
    X, y, Z = orc.synthetic_problem(max(N, M, 8), D, M, seed=c["data_seed"]); X, y = X[:N], y[:N]
    hyp = orc.Hypers(c["ls"], c["variance"], c["noise"], c["mean"], Z, 1e-6)
    cov = orc.dense_cov(kind, X, hyp)
    terms = orc.common_terms(kind, X, hyp)
    dl = np.diag(terms.L)
    print(f"draw {idx}: N={N} D={D} M={M} {kind}  cond(K_uu)~{np.linalg.cond(terms.L)**2:.2e}  max/min diag L = {dl.max()/dl.min():.2e}")
    for prec in (0, 1):
        for trsm in (0, 1, 2):
            ctx = HipContext(X, y, M, kind)
            ctx.set_option("precision", prec); ctx.set_option("grad_trsm", trsm)
            ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
            v = torch.zeros(N, dtype=torch.float64, device=ctx.device)
            res = ctx.objective_and_grad(v, True, c["tol"], 100, 40, with_grad=True)
            vh = v.cpu().numpy()
            refg = orc.objective(kind, X, y, hyp, vh, run_cg=False, with_grad=True, cov=cov)
            r = (y - hyp.mean) - cov @ vh
            w, _ = orc.nystrom_precond(terms.A, terms.LB, hyp.noise, r)
            floor = orc.grad_roundoff_spread(kind, X, hyp, vh, w)
            out = " ".join(f"{k} {np.abs(np.asarray(res.grad[k]) - np.asarray(refg.grad[k])).max():.2e} (floor {floor[k]:.1e})" for k in ("Z", "lengthscales", "variance", "noise"))
            print(f"  precision={prec} grad_trsm={trsm}: {out}", flush=True)
            ctx.close()
