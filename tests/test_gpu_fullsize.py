"""BASELINE.json's full sizes (C2: N=50k D=8 RBF M=1024; headline: N=100k) checked through size-independent properties —
the dense oracle does not fit there: symmetry / linearity / determinism of the implicit operator, a row sample against the
blocked C oracle, the Woodbury identity of the preconditioner, an independent recomputation of the PCG stopping statistic,
monotone bounds under a tighter solve, and a finite-difference check of the analytic gradient."""
import numpy as np
import pytest
import torch

from oracle import cglb_oracle as orc
from oracle import cglb_oracle_c as orcc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[(50_000, "rbf"), (100_000, "rbf"), (100_000, "matern32")],
                ids=["c2_50k_rbf", "headline_100k_rbf", "headline_100k_matern32"])
def big(request):
    from cglb_amd.hip_context import HipContext
    N, kind = request.param
    D, M = 8, 1024
    X, y, Z = orc.synthetic_problem(N, D, M, seed=0)
    hyp = orc.trained_like_hypers(D, Z)
    ctx = HipContext(X, y, M, kind)
    ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
    ctx.setup()
    yield ctx, X, y, hyp, kind
    ctx.close()


def test_operator_properties_and_row_sample(big):
    ctx, X, y, hyp, kind = big
    N = ctx.N
    g = torch.Generator(device="cpu").manual_seed(1)
    p = torch.randn(N, dtype=torch.float64, generator=g).to(ctx.device)
    q = torch.randn(N, dtype=torch.float64, generator=g).to(ctx.device)
    Ap, Aq = ctx.matvec(p), ctx.matvec(q)
    assert float(q @ Ap) == pytest.approx(float(p @ Aq), rel=1e-11)                       # symmetry
    Apq = ctx.matvec(2.0 * p - 0.5 * q)
    assert float((Apq - (2.0 * Ap - 0.5 * Aq)).abs().max()) <= 1e-11 * float(Ap.abs().max())  # linearity
    assert torch.equal(ctx.matvec(p), Ap)                                                  # run-twice determinism
    assert float(p @ Ap) > 0                                                               # positive definite
    rows = slice(12_345, 12_345 + 384)
    ref = orcc.kff_matvec(kind, X, hyp, p.cpu().numpy(), rows.start, rows.stop)            # blocked C oracle, direct differences
    np.testing.assert_allclose(Ap.cpu().numpy()[rows], ref, rtol=0, atol=2e-12 * float(Ap.abs().max()))
    # precision levels (cglb_set_option "precision"): exact (3e-16 kernel values) against the default fast level (<= 1e-13)
    ctx.set_option("precision", 0)
    Ap_exact = ctx.matvec(p)
    np.testing.assert_allclose(Ap_exact.cpu().numpy()[rows], ref, rtol=0, atol=2e-12 * float(Ap.abs().max()))
    assert float((Ap_exact - Ap).abs().max()) <= 2e-13 * float(Ap.abs().max())
    ctx.set_option("precision", 1)
    # plain kernel == symmetric kernel
    ctx.set_option("kff_variant", 0)
    Ap0 = ctx.matvec(p)
    ctx.set_option("kff_variant", 2)
    assert float((Ap0 - Ap).abs().max()) <= 1e-12 * float(Ap.abs().max())


def test_preconditioner_is_woodbury_inverse(big):
    ctx, X, y, hyp, kind = big
    g = torch.Generator(device="cpu").manual_seed(2)
    r = torch.randn(ctx.N, dtype=torch.float64, generator=g).to(ctx.device)
    z, rz = ctx.precond(r)
    A = ctx.get_matrix("A")                                                               # [M, N] on device (test-side algebra only)
    back = hyp.noise * (A.T @ (A @ z)) + hyp.noise * z                                    # (Q_ff + sigma^2 I) z
    assert float((back - r).abs().max()) <= 1e-9 * float(r.abs().max())
    assert rz == pytest.approx(float(r @ z), rel=1e-12) and rz > 0


def test_pcg_stop_statistic_and_monotone_bounds(big):
    ctx, X, y, hyp, kind = big
    N = ctx.N
    b = (ctx.y - hyp.mean)
    v1, steps1, half1 = ctx.pcg(b, torch.zeros(N, dtype=torch.float64), 1.0, 100, 40)
    assert steps1 > 0 and (half1 <= 1.0 or steps1 == 100)
    # the returned statistic is 1/2 r^T P r of the recursively updated residual; recompute it from v alone
    r = b - ctx.matvec(v1)
    _, rz = ctx.precond(r)
    assert 0.5 * rz == pytest.approx(half1, rel=1e-6)
    res1 = ctx.objective_and_grad(v1.clone(), run_cg=False, with_grad=False)
    v2, steps2, half2 = ctx.pcg(b, v1, 1e-2, 100, 40)                                     # warm start, tighter tolerance
    res2 = ctx.objective_and_grad(v2.clone(), run_cg=False, with_grad=False)
    assert res1.lower <= res1.upper and res2.lower <= res2.upper
    assert res2.lower >= res1.lower - 1e-9 * abs(res1.lower)                              # CG only improves the lower bound
    assert res2.upper <= res1.upper + 1e-9 * abs(res1.upper)
    assert res2.upper - res2.lower == pytest.approx(half2, rel=1e-6)                       # gap == 1/2 r^T P r (models.py:284)
    assert res2.bound >= res1.bound


def test_gradient_matches_finite_differences(big):
    ctx, X, y, hyp, kind = big
    N = ctx.N
    v = torch.zeros(N, dtype=torch.float64, device=ctx.device)
    base = ctx.objective_and_grad(v, True, 1.0, 100, 40)                                   # leaves the solution in v

    def bound_at(h):
        ctx.set_hypers(h.lengthscales, h.variance, h.noise, h.mean, h.Z, h.jitter)
        return ctx.objective_and_grad(v, run_cg=False, with_grad=False).bound

    eps = 1e-5
    for name in ("noise", "variance", "mean", "ls0", "ls5", "z"):
        hp, hm = hyp.copy(), hyp.copy()
        if name in ("noise", "variance", "mean"):
            setattr(hp, name, getattr(hyp, name) + eps)
            setattr(hm, name, getattr(hyp, name) - eps)
            ana = base.grad[name]
        elif name.startswith("ls"):
            d = int(name[2:])
            hp.lengthscales[d] += eps
            hm.lengthscales[d] -= eps
            ana = base.grad["lengthscales"][d]
        else:
            hp.Z[7, 3] += eps
            hm.Z[7, 3] -= eps
            ana = base.grad["Z"][7, 3]
        fd = (bound_at(hp) - bound_at(hm)) / (2 * eps)
        assert ana == pytest.approx(fd, rel=2e-5, abs=2e-4), name
    ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, hyp.Z, hyp.jitter)
    ctx.setup()
