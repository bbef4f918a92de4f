"""Per-iteration stop statistic 1/2 r^T P r of one named sweep draw: GPU (one solve per iteration cap) against the numpy oracle and the
oracle's own spread under calibrated probes.  Usage: python tools/history_compare.py SEED INDEX [option=value ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from tools.fuzz_parity import named_case  # noqa: E402


def main():
    import torch
    from oracle import cglb_oracle as orc
    from cglb_amd.hip_context import HipContext
    c = named_case(int(sys.argv[1]), int(sys.argv[2]))
    opts = dict(a.split("=") for a in sys.argv[3:])
    N, D, M, kind, prec, tol = c["N"], c["D"], c["M"], c["kind"], c["prec"], c["tol"]
    X, y, Z = orc.synthetic_problem(max(N, M, 8), D, M, seed=c["data_seed"])
    X, y = X[:N], y[:N]
    hyp = orc.Hypers(c["ls"], c["variance"], c["noise"], c["mean"], Z, c["jitter"])
    print(f"N={N} D={D} M={M} {kind} prec={prec} tol={tol} ls={np.asarray(hyp.lengthscales).round(3)} var={hyp.variance:.3g} noise={hyp.noise:.3g}")
    cov = orc.dense_cov(kind, X, hyp)
    print("cond(K + s2 I) = %.3g" % np.linalg.cond(cov))
    sens = orc.roundoff_sensitivity(kind, X, y, hyp, np.zeros(N), tol, 100, 40, delta=2.0 ** -52, cov=cov, calibrate=(c["p"], 7e-15))
    ctx = HipContext(X, y, M, kind, dtype=torch.float64)
    ctx.set_option("precision", int(opts.pop("precision", prec)))
    for k, val in opts.items():
        ctx.set_option(k, float(val))
    ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
    ctx.setup()
    b = torch.from_numpy(y - hyp.mean).to(ctx.device)
    v0 = torch.zeros(N, dtype=torch.float64, device=ctx.device)
    terms = orc.common_terms(kind, X, hyp)
    for k in range(0, len(sens.history) + 2):
        v, steps, half = ctx.pcg(b, v0, 0.0, k, 40)
        vh = v.cpu().numpy()
        r = (y - hyp.mean) - cov @ vh                       # true residual of the GPU's iterate
        _, rz_true = orc.nystrom_precond(terms.A, terms.LB, hyp.noise, r)
        h = sens.history[k] if k < len(sens.history) else float("nan")
        s = sens.stat_rel_spread[k] if k < len(sens.stat_rel_spread) else float("nan")
        print(f"k={k:3d} oracle {h:.6e} (spread {s:.1e})  gpu {half:.6e}  rel {abs(half - h) / h if h == h else float('nan'):.1e}   gpu true-residual statistic {0.5 * rz_true:.6e}")
    ctx.close()


if __name__ == "__main__":
    main()
