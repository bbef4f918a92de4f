#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own solver code.  Runs only in the build
container (needs /root/reference); the produced vectors are committed, this script is the
record of how they were made.  TEST INFRASTRUCTURE ONLY.

What comes from the reference (cglb/backend/pytorch/conjugate_gradient.py, loaded by file path;
its two annotation-only imports `pykeops.torch.LazyTensor` / `gpytorch.lazy.LazyTensor` are
satisfied by empty placeholder modules):
  * ConjugateGradient.__call__      -> v, steps, residual_error          (rows A1, A2)
  * NystromPreconditioner.__call__  -> z, rz on a random r               (row A4)
What is a torch restatement (models.py needs gpytorch and cannot be imported):
  * LowerBoundCG.forward (models.py:151-286) on dense fp64 tensors with the kernel closed forms,
    calling the reference solver above, differentiated by torch.autograd.grad exactly like
    pytorch/optimizer.py:95-98 -> bound, lower, upper, grads wrt constrained hypers.
"""
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import cglb_oracle as orc  # noqa: E402

REF = "/root/reference/cglb/backend/pytorch/conjugate_gradient.py"
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def load_reference_cg():
    for name, attr in (("pykeops", None), ("pykeops.torch", "LazyTensor"),
                       ("gpytorch", None), ("gpytorch.lazy", "LazyTensor")):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if attr:
                setattr(m, attr, type(attr, (), {}))
            sys.modules[name] = m
    spec = importlib.util.spec_from_file_location("_ref_conjugate_gradient", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def t_kernel(kind, x1, x2, ls, var):
    x1s, x2s = x1 / ls, x2 / ls
    d2 = ((x1s[:, None, :] - x2s[None, :, :]) ** 2).sum(-1)
    if orc.kind_id(kind) == orc.RBF:
        return var * torch.exp(-0.5 * d2)
    r = torch.sqrt(d2.clamp_min(1e-36))
    return var * (1.0 + orc.SQRT3 * r) * torch.exp(-orc.SQRT3 * r)


def t_forward(ref, kind, X, y, ls, var, noise, mean, Z, jitter, v0, cg_opt, run_cg=True):
    """Dense torch restatement of LowerBoundCG.forward (models.py:151-286)."""
    N, M = X.shape[0], Z.shape[0]
    sigma = torch.sqrt(noise)
    kuf = t_kernel(kind, Z, X, ls, var)                                  # :196-197
    kuu = t_kernel(kind, Z, Z, ls, var) + jitter * torch.eye(M, dtype=X.dtype)  # :200-201
    L = torch.linalg.cholesky(kuu)                                       # :202
    A = torch.linalg.solve_triangular(L, kuf, upper=False) / sigma       # :206
    AAt = A @ A.T
    B = AAt + torch.eye(M, dtype=X.dtype)
    LB = torch.linalg.cholesky(B)                                        # :210
    AAt_diag_sum = AAt.diagonal().sum()
    kdiag_sum = var * N
    trace = kdiag_sum / noise - AAt_diag_sum                             # :236
    logdet = -LB.diagonal().log().sum() - 0.5 * N * torch.log(noise)     # :239-240
    logdet = logdet - 0.5 * N * torch.log(1.0 + trace / N)               # :243
    const = -0.5 * N * math.log(2.0 * math.pi)
    cov = t_kernel(kind, X, X, ls, var) + noise * torch.eye(N, dtype=X.dtype)  # :251-252
    err = y.reshape(-1, 1) - mean                                        # :253-254
    precon = ref.NystromPreconditioner(A, LB, noise)                     # :260
    with torch.no_grad():                                                # :262-278
        if run_cg:
            v, stats = cg_opt(cov.detach(), err.detach(), v0, precon)
        else:
            v, stats = v0.clone(), ref.ConjugateGradientStats(0, torch.tensor(float("nan")))
    cov_v = cov @ v                                                      # :280
    r = err - cov_v
    w, error_bound = precon(r)                                           # :282
    lower = (v * (r + 0.5 * cov_v)).sum()                                # :283
    upper = lower + 0.5 * error_bound                                    # :284
    bound = -upper + logdet + const                                      # :286,:169
    return bound, lower, upper, logdet, v, stats, (A, LB, L, AAt_diag_sum)


def make_case(ref, name, kind, N, D, M, seed, hyp_kind, max_error, max_cg_iter=100, restart=40, warm=False):
    X, y, Z = orc.synthetic_problem(N, D, M, seed)
    rng = np.random.default_rng(1000 + seed)
    if hyp_kind == "init":
        hyp = orc.reference_init_hypers(D, Z)
    elif hyp_kind == "trained":
        hyp = orc.trained_like_hypers(D, Z)
    elif hyp_kind == "hard":  # low noise: PCG needs > 40 steps at a tolerance well above round-off
        hyp = orc.trained_like_hypers(D, Z)
        hyp.noise = 0.01
    else:  # "random": non-uniform lengthscales, non-zero mean
        hyp = orc.Hypers(lengthscales=0.7 + rng.random(D), variance=0.5 + rng.random(), noise=0.02 + 0.2 * rng.random(),
                         mean=0.3 * rng.standard_normal(), Z=Z.copy(), jitter=1e-6)
    tX, ty = torch.from_numpy(X), torch.from_numpy(y)
    ls = torch.tensor(hyp.lengthscales, requires_grad=True)
    var = torch.tensor(hyp.variance, dtype=torch.float64, requires_grad=True)
    noise = torch.tensor(hyp.noise, dtype=torch.float64, requires_grad=True)
    mean = torch.tensor(hyp.mean, dtype=torch.float64, requires_grad=True)
    tZ = torch.tensor(hyp.Z, requires_grad=True)
    cg_opt = ref.ConjugateGradient(max_error=max_error, max_cg_iter=max_cg_iter, restart_cg_iter=restart)
    v0 = torch.zeros((N, 1), dtype=torch.float64)
    if warm:  # warm start: solution at a perturbed theta
        hyp2 = hyp.copy()
        hyp2.lengthscales = hyp2.lengthscales * 1.1
        ob = orc.objective(kind, X, y, hyp2, np.zeros(N), max_error=max_error)
        v0 = torch.from_numpy(ob.v.reshape(-1, 1).copy())
    bound, lower, upper, logdet, v, stats, (A, LB, L, tr) = t_forward(
        ref, kind, tX, ty, ls, var, noise, mean, tZ, hyp.jitter, v0, cg_opt)
    grads = torch.autograd.grad(bound, [ls, var, noise, mean, tZ])
    # preconditioner alone, on a random r, straight from the reference class
    r_test = torch.from_numpy(rng.standard_normal((N, 1)))
    with torch.no_grad():
        z_test, rz_test = ref.NystromPreconditioner(A.detach(), LB.detach(), noise.detach())(r_test)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        kind=np.int64(orc.kind_id(kind)), X=X, y=y, Z=hyp.Z, lengthscales=hyp.lengthscales,
        variance=np.float64(hyp.variance), noise=np.float64(hyp.noise), mean=np.float64(hyp.mean),
        jitter=np.float64(hyp.jitter), v0=v0.numpy().reshape(-1),
        max_error=np.float64(max_error), max_cg_iter=np.int64(max_cg_iter), restart_cg_iter=np.int64(restart),
        # reference solver outputs
        v=v.numpy().reshape(-1), steps=np.int64(int(stats.steps)), residual_error=np.float64(float(stats.residual_error)),
        r_test=r_test.numpy().reshape(-1), z_test=z_test.numpy().reshape(-1), rz_test=np.float64(float(rz_test)),
        # restated objective (reference solver inside) + autograd
        bound=np.float64(bound.item()), lower=np.float64(lower.item()), upper=np.float64(upper.item()),
        logdet=np.float64(logdet.item()), AAt_diag_sum=np.float64(tr.item()),
        g_lengthscales=grads[0].numpy(), g_variance=np.float64(grads[1].item()), g_noise=np.float64(grads[2].item()),
        g_mean=np.float64(grads[3].item()), g_Z=grads[4].numpy(),
    )
    print(f"{name}: N={N} D={D} M={M} steps={int(stats.steps)} half_rz={float(stats.residual_error):.3e} "
          f"bound={bound.item():.9f}")


def main():
    ref = load_reference_cg()
    torch.set_default_dtype(torch.float64)
    cases = [
        # name, kind, N, D, M, seed, hypers, max_error
        ("c1_snelson_like_m32", "matern32", 200, 1, 16, 0, "init", 1.0),
        ("c1_snelson_like_m32_tight", "matern32", 200, 1, 16, 0, "trained", 1e-10),
        ("rbf_d8_init", "rbf", 384, 8, 32, 1, "init", 1.0),
        ("rbf_d8_trained", "rbf", 384, 8, 32, 1, "trained", 1.0),
        ("rbf_d8_trained_tol1e-3", "rbf", 384, 8, 32, 2, "trained", 1e-3),
        ("m32_d8_trained", "matern32", 384, 8, 32, 1, "trained", 1.0),
        ("m32_d3_random", "matern32", 300, 3, 24, 3, "random", 1e-3),
        ("rbf_d3_random", "rbf", 300, 3, 24, 4, "random", 1e-3),
        ("rbf_d8_restart", "rbf", 512, 8, 8, 5, "trained", 1e-5),      # > 40 steps: hits the i%40==39 restart
        ("m32_d8_restart", "matern32", 512, 8, 8, 5, "hard", 1e-6),
    ]
    for c in cases:
        make_case(ref, *c)
    # never reaches tol: steps == max_cg_iter (12), restart every 5
    make_case(ref, "rbf_d8_maxiter", "rbf", 512, 8, 8, 5, "trained", 1e-30, max_cg_iter=12, restart=5)
    make_case(ref, "rbf_d8_warm", "rbf", 384, 8, 32, 8, "trained", 1.0, warm=True)


if __name__ == "__main__":
    main()
