"""Randomised parity sweep: shapes off every tile grid (N, D, M drawn at random, both kernels, both precision levels) against the dense
numpy oracle - mat-vec, bound at the oracle's own v, gradient, and the CG step count.  `tools/fuzz_parity.py` runs longer sweeps."""
import numpy as np
import pytest
import torch

from oracle import cglb_oracle as orc

pytestmark = pytest.mark.gpu


def _draw(rng):
    N = int(rng.choice([rng.integers(2, 300), rng.integers(300, 3000), rng.integers(3000, 7000)]))
    D = int(rng.integers(1, 33))
    M = int(min(N, rng.choice([rng.integers(1, 70), rng.integers(60, 200), rng.integers(200, 600)])))
    return N, D, M, str(rng.choice(["rbf", "matern32"])), int(rng.integers(0, 2)), float(rng.choice([1.0, 1e-2])), int(rng.integers(1 << 30))


@pytest.mark.parametrize("case", range(14))
def test_random_shape_matches_oracle(case):
    from cglb_amd.hip_context import HipContext
    rng = np.random.default_rng(1000 + case)
    N, D, M, kind, prec, tol, seed = _draw(rng)
    X, y, Z = orc.synthetic_problem(max(N, M, 8), D, M, seed=seed)
    X, y = X[:N], y[:N]
    ls = rng.uniform(0.7, 2.5, size=D) * np.sqrt(D / 2.0)
    hyp = orc.Hypers(ls, float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.02, 0.5)), float(rng.normal() * 0.1), Z, 1e-6)
    ctx = HipContext(X, y, M, kind)
    ctx.set_option("precision", prec)
    ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
    v = torch.zeros(N, dtype=torch.float64, device=ctx.device)
    res = ctx.objective_and_grad(v, True, tol, 100, 40, with_grad=True)
    ref = orc.objective(kind, X, y, hyp, np.zeros(N), True, tol, 100, 40)
    refg = orc.objective(kind, X, y, hyp, v.cpu().numpy(), run_cg=False, with_grad=True)   # bound and gradient at the GPU's own v
    p = rng.standard_normal(N)
    Ap = ctx.matvec(torch.from_numpy(p)).cpu().numpy()
    Aref = orc.dense_cov(kind, X, hyp) @ p
    info = f"N={N} D={D} M={M} {kind} precision={prec} tol={tol}"
    np.testing.assert_allclose(Ap, Aref, rtol=0, atol=1e-11 * np.abs(Aref).max(), err_msg=info)
    assert abs(res.steps - ref.steps) <= 1, info
    assert res.bound == pytest.approx(refg.bound, rel=1e-9), info
    if res.steps == ref.steps:
        # the north_star tolerance; beyond a restart (40 steps) two correct solves have drifted apart by round-off and agree only to a
        # fraction of the stopping tolerance itself (the bound moves by 1/2 r^T P r <= max_error between admissible stopping points)
        assert res.bound == pytest.approx(ref.bound, rel=1e-6, abs=(0.5 * tol if res.steps > 40 else 0.0)), info
    for key in ("lengthscales", "Z"):
        # relative to the largest entry, with a floor tied to the bound: with M = N (inducing points on every datum, K_uu as ill
        # conditioned as K_ff) the whole Z gradient is ~1e-11 of the bound and carries cond(K_uu) * eps of absolute error on both sides
        scale = np.abs(refg.grad[key]).max() + 1e-300
        np.testing.assert_allclose(res.grad[key], refg.grad[key], rtol=0, atol=1e-6 * scale + 1e-11 * max(1.0, abs(ref.bound)),
                                   err_msg=info + " grad " + key)
    for key in ("variance", "noise", "mean"):
        assert res.grad[key] == pytest.approx(refg.grad[key], rel=1e-6, abs=1e-9 * abs(ref.bound)), info + " grad " + key
    ctx.close()
