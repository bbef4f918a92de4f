#!/usr/bin/env python3
"""Generate tests/golden/predict/*.npz: PredictCG.forward (cglb/backend/pytorch/models.py:307-354) with the REFERENCE's own solver.
Runs only in the build container (needs /root/reference); the vectors are committed, this script is the record of how they were
made.  TEST INFRASTRUCTURE ONLY.

From the reference (conjugate_gradient.py loaded by file path like oracle/gen_golden.py): `ConjugateGradient(max_error=1e-3)`
(models.py:291) warm-started at the model's v (models.py:294, :329) and `NystromPreconditioner`.  Restated in torch, statement by
statement, because models.py needs gpytorch: the predictor algebra of models.py:316-352 (err, ksf, cov, common terms :176-213,
cg_mean, res, kus, a_res, the three `torch.triangular_solve` calls, sgpr_mean, f_mean, f_var) on dense fp64 tensors with the
kernel closed forms of oracle/gen_golden.py: t_kernel.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import cglb_oracle as orc  # noqa: E402
from oracle.gen_golden import load_reference_cg, t_kernel  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
OUT = os.path.join(GOLDEN, "predict")


def t_predict(ref, kind, X, y, ls, var, noise, mean, Z, jitter, v_model, xnew, cg_opt):
    N, M = X.shape[0], Z.shape[0]
    err = y.reshape(-1, 1) - mean                                        # :316-318
    sigma_sq = noise
    ksf = t_kernel(kind, xnew, X, ls, var)                               # :320
    cov = t_kernel(kind, X, X, ls, var) + sigma_sq * torch.eye(N, dtype=X.dtype)   # :321
    # terms = self.logdet_and_quad_common_terms((x, y))  (:327 -> :176-213)
    sigma = torch.sqrt(sigma_sq)
    kuf = t_kernel(kind, Z, X, ls, var)
    kuu = t_kernel(kind, Z, Z, ls, var) + jitter * torch.eye(M, dtype=X.dtype)
    L = torch.linalg.cholesky(kuu)
    A = torch.triangular_solve(kuf, L, upper=False)[0] / sigma
    B = A @ A.T + torch.eye(M, dtype=X.dtype)
    LB = torch.linalg.cholesky(B)
    precon = ref.NystromPreconditioner(A, LB, sigma_sq)                  # :328
    new_v, cg_stats = cg_opt(cov, err, v_model.clone(), precon)          # :329 (v_vec: clone of model.v_vec, :294)
    cg_mean = ksf @ new_v                                                # :334
    res = err - cov @ new_v                                              # :335
    kus = t_kernel(kind, Z, xnew, ls, var)                               # :337
    a_res = A @ res                                                      # :340
    trisolve = torch.triangular_solve                                    # :342
    c = trisolve(a_res, LB, upper=False)[0] / sigma                      # :343
    tmp1 = trisolve(kus, L, upper=False)[0]                              # :344
    tmp2 = trisolve(tmp1, LB, upper=False)[0]                            # :345
    sgpr_mean = tmp2.transpose(-1, -2) @ c                               # :347
    f_mean = sgpr_mean + cg_mean + mean                                  # :348
    kss = var * torch.ones(xnew.shape[0], dtype=X.dtype)                 # :350 (k(x, x) = outputscale for both kernels)
    f_var = kss + (tmp2 ** 2).sum(0) - (tmp1 ** 2).sum(0)                # :351
    return f_mean.reshape(-1), f_var.reshape(-1), new_v.reshape(-1), cg_stats


def make_case(ref, source, n_new, seed, extrapolate=False):
    g = dict(np.load(os.path.join(GOLDEN, source + ".npz")))
    kind = int(g["kind"])
    X, y = torch.from_numpy(g["X"]), torch.from_numpy(g["y"])
    D = X.shape[1]
    rng = np.random.default_rng(seed)
    xnew = rng.standard_normal((n_new, D)) * (3.0 if extrapolate else 1.0)   # 3x: points far outside the training range
    xnew[: min(5, n_new)] = g["X"][: min(5, n_new)]                          # and some ON training points (distance exactly 0)
    args = [torch.from_numpy(np.asarray(g[k], dtype=np.float64)) for k in ("lengthscales", "variance", "noise", "mean")]
    cg_opt = ref.ConjugateGradient(max_error=1e-3)                            # models.py:291
    v_model = torch.from_numpy(g["v"]).reshape(-1, 1)                         # the model's v after the training solve of the source case
    with torch.no_grad():
        f_mean, f_var, new_v, stats = t_predict(ref, kind, X, y, *args, torch.from_numpy(g["Z"]), float(g["jitter"]), v_model,
                                                torch.from_numpy(xnew), cg_opt)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, f"predict_{source}.npz"), source=np.array(source), xnew=xnew, f_mean=f_mean.numpy(),
                        f_var=f_var.numpy(), new_v=new_v.numpy(), steps=np.int64(int(stats.steps)),
                        residual_error=np.float64(float(stats.residual_error)))
    print(f"predict_{source}: n_new={n_new} steps={int(stats.steps)} half_rz={float(stats.residual_error):.3e} "
          f"mean[:3]={f_mean[:3].numpy()} var[:3]={f_var[:3].numpy()}")


def main():
    ref = load_reference_cg()
    torch.set_default_dtype(torch.float64)
    make_case(ref, "rbf_d8_trained", 97, 11)
    make_case(ref, "m32_d3_random", 64, 12, extrapolate=True)
    make_case(ref, "c1_snelson_like_m32", 33, 13)
    make_case(ref, "m32_d8_trained", 130, 14)


if __name__ == "__main__":
    main()
