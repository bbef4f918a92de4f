"""BASELINE.json's configurations at their full sizes on ONE MI355X (the 8-GPU ones are run single-GPU here: the kernels
see the full problem; the sharded drivers are covered by test_gpu_sharded.py / test_distributed_gloo.py), plus two
end-to-end comparisons at size:

  * a mid-size FULL objective + gradient (N = 16 500: 65 row blocks, ragged last block, > 40 CG steps so the restart
    branch runs) against the dense numpy oracle computed on the box's host;
  * the headline fixtures (N = 100 000 and C2's 50 000, D = 8, M = 1024) produced in the build container by the
    REFERENCE's own ConjugateGradient/NystromPreconditioner code driving the blocked C operator
    (oracle/gen_headline_fixture.py -> tests/golden/headline/*.npz).

Tolerances (fp64): north_star's 1e-6 relative on the bound; step counts exact up to 40 steps and +-1 beyond (CG amplifies
summation-order round-off, DESIGN.md section 6 note); where the step count agrees the bound is held to 1e-10 (<= 40 steps) / 1e-7."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import cglb_oracle as orc
from oracle import cglb_oracle_c as orcc

pytestmark = pytest.mark.gpu

CG = dict(max_error=1.0, max_cg_iter=100, restart_cg_iter=40)   # conjugate_gradient.py:37-39


def _ctx(N, D, M, kind, hyp_kind="trained", dtype=torch.float64, seed=0):
    from cglb_amd.hip_context import HipContext
    X, y, Z = orc.synthetic_problem(N, D, M, seed=seed)
    hyp = orc.trained_like_hypers(D, Z) if hyp_kind == "trained" else orc.reference_init_hypers(D, Z)
    if dtype == torch.float32:
        hyp.jitter = 1e-5                                        # backend.py:77-79
    ctx = HipContext(X, y, M, kind, dtype=dtype)
    ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
    return ctx, X, y, hyp


def _check_matvec_rows(ctx, X, hyp, kind, tol, nrows=256):
    N = ctx.N
    g = torch.Generator(device="cpu").manual_seed(7)
    p = torch.randn(N, dtype=torch.float64, generator=g)
    Ap = ctx.matvec(p.to(ctx.dtype).to(ctx.device)).double().cpu().numpy()
    scale = float(np.abs(Ap).max())
    for start in (0, (N // 2 // 256) * 256 + 37, N - nrows):     # first block, a middle one off the block grid, ragged end
        ref = orcc.kff_matvec(kind, X, hyp, p.to(ctx.dtype).double().numpy(), start, start + nrows)
        np.testing.assert_allclose(Ap[start:start + nrows], ref, rtol=0, atol=tol * scale)
    return p, Ap


def _check_woodbury(ctx, hyp, tol):
    g = torch.Generator(device="cpu").manual_seed(8)
    r = torch.randn(ctx.N, dtype=torch.float64, generator=g).to(ctx.dtype).to(ctx.device)
    z, rz = ctx.precond(r)
    A = ctx.get_matrix("A")
    back = hyp.noise * (A.T @ (A @ z)) + hyp.noise * z           # (Q_ff + sigma^2 I) z, conjugate_gradient.py:95-113
    assert float((back - r).abs().max()) <= tol * float(r.abs().max())
    assert rz == pytest.approx(float(r.double() @ z.double()), rel=tol) and rz > 0
    del A, back


def _check_short_pcg(ctx, hyp, iters, tol):
    """A few PCG iterations (max_cg_iter small) and the returned statistic recomputed from v alone."""
    b = ctx.y - hyp.mean
    v, steps, half = ctx.pcg(b, torch.zeros(ctx.N, dtype=ctx.dtype), 1e-30, iters, 40)
    assert steps == iters
    r = b - ctx.matvec(v)
    _, rz = ctx.precond(r)
    assert 0.5 * rz == pytest.approx(half, rel=tol)
    return v


def test_c3_n200k_matern32_m2048():
    """C3: N = 200 000, D = 8, Matern-3/2, M = 2048, fp64."""
    ctx, X, y, hyp = _ctx(200_000, 8, 2048, "matern32")
    ctx.setup()
    _check_matvec_rows(ctx, X, hyp, "matern32", 2e-12)
    _check_woodbury(ctx, hyp, 1e-8)
    v = _check_short_pcg(ctx, hyp, 3, 1e-7)
    res = ctx.objective_and_grad(v.clone(), run_cg=False, with_grad=True)
    assert res.lower <= res.upper and np.isfinite(res.bound)
    assert all(np.all(np.isfinite(np.asarray(res.grad[k]))) for k in res.grad)
    ctx.close()


def test_c4_n430k_d3_rbf_train_loop(tmp_path):
    """C4 (3droad-like stand-in): N = 430 000, D = 3, RBF, M = 1024, fp64: kernels at size + a short SciPy L-BFGS-B run through
    the cglb.backend mirror (pytorch/interface.py:445-543) whose loss must decrease monotonically over accepted steps."""
    ctx, X, y, hyp = _ctx(430_000, 3, 1024, "rbf", hyp_kind="init")
    ctx.setup()
    _check_matvec_rows(ctx, X, hyp, "rbf", 2e-12)
    _check_woodbury(ctx, hyp, 1e-8)
    _check_short_pcg(ctx, hyp, 3, 1e-7)
    ctx.close()
    del ctx
    # "full CGLB train loop": BACKENDS["hip"] -> create_model (GPU inducing-point selection) -> optimize (SciPy L-BFGS-B)
    from cglb_amd.backend import BACKENDS, CGLBConfig, INDUCING_VARIABLE_CONFIGS, KERNEL_CONFIGS
    from cglb_amd.backend.callbacks import Logger
    from cglb_amd.backend.models import LowerBoundCG
    be = BACKENDS["hip"]
    be.configure_backend(logdir=str(tmp_path), keops=False)
    be.set_default_float("fp64")
    be.set_default_jitter("fp64")
    cfg = CGLBConfig(kernel=KERNEL_CONFIGS["rbf"](), inducing_variable=INDUCING_VARIABLE_CONFIGS["cv"](1024))
    model = be.create_model(cfg, (X, y))
    logger = Logger(str(tmp_path), lambda: {}, lambda: be.model_parameters(model), holdout_interval=-1, include_feval_log=True, verbose=False)
    l_init = float(-LowerBoundCG(model)(None))
    model.v_vec.zero_()
    results = be.optimize(model, ((X, y), (X[:8], y[:8])), 3, logger, "scipy")
    assert 1 <= sum(r.nit for r in results) <= 3 + 3
    l_end = float(results[-1].fun)
    assert np.isfinite(l_end) and l_end < l_init                       # three L-BFGS-B steps lower the loss
    assert len(logger.logs["steps-per-feval"]) == sum(r.nfev for r in results)   # one CG record per objective evaluation (:476)
    model.hip.close()


def test_c5_n1m_d16_rbf_m4096_fp32():
    """C5: N = 1 000 000, D = 16, RBF, M = 4096, fp32 (tolerances relaxed to fp32 round-off against the fp64 oracle)."""
    ctx, X, y, hyp = _ctx(1_000_000, 16, 4096, "rbf", dtype=torch.float32)
    ctx.setup()
    _check_matvec_rows(ctx, X.astype(np.float32).astype(np.float64), hyp, "rbf", 3e-5)
    _check_woodbury(ctx, hyp, 5e-3)
    _check_short_pcg(ctx, hyp, 2, 2e-2)
    ctx.close()


def test_c5_fp32_evaluation_against_the_fp64_path_at_full_size():
    """C5 at its size: the CPU oracle cannot give a bound here (52 CPU-minutes per mat-vec), but the fp64 HIP path - pinned at the
    headline, C2, C3 and C4 sizes by reference-solver fixtures - can: the same evaluation (3 CG steps from v = 0, N = 1 000 000, D = 16,
    M = 4096) in fp64 and in fp32, bound / lower / upper / log-det and the gradient blocks compared at fp32 tolerances."""
    from cglb_amd.hip_context import HipContext
    N, D, M = 1_000_000, 16, 4096
    X, y, Z = orc.synthetic_problem(N, D, M, seed=0)
    hyp = orc.trained_like_hypers(D, Z)
    hyp.jitter = 1e-5
    out = {}
    for dt in (torch.float64, torch.float32):
        ctx = HipContext(X, y, M, "rbf", dtype=dt)
        ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
        v = torch.zeros(N, dtype=dt, device=ctx.device)
        out[dt] = (ctx.objective_and_grad(v, True, 1.0, 3, 40), v.double().cpu().numpy())
        ctx.close()
        del ctx, v
        torch.cuda.empty_cache()
    (r64, v64), (r32, v32) = out[torch.float64], out[torch.float32]
    assert r64.steps == r32.steps == 3
    assert r32.logdet == pytest.approx(r64.logdet, rel=1e-4)
    assert r32.lower == pytest.approx(r64.lower, rel=1e-3) and r32.upper == pytest.approx(r64.upper, rel=1e-3)
    assert r32.bound == pytest.approx(r64.bound, rel=1e-3)
    np.testing.assert_allclose(v32, v64, rtol=0, atol=2e-3 * np.abs(v64).max())
    for key in ("lengthscales", "Z"):
        a, b = np.asarray(r32.grad[key]), np.asarray(r64.grad[key])
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-2 * np.abs(b).max(), err_msg=key)
    for key in ("variance", "noise"):
        assert r32.grad[key] == pytest.approx(r64.grad[key], rel=2e-2, abs=1e-3 * abs(r64.bound)), key


def test_midsize_full_objective_and_gradient_vs_dense_oracle():
    """N = 16 500 (65 row blocks of 256, ragged last one; chunk-halving, XCD-aware item order and the multi-slab combine all
    active), D = 8, M = 256, trained-like hypers: the whole evaluation against the dense numpy oracle."""
    N, D, M = 16_500, 8, 256
    for kind in ("rbf", "matern32"):
        ctx, X, y, hyp = _ctx(N, D, M, kind, seed=3)
        cov = orc.dense_cov(kind, X, hyp)
        ref = orc.objective(kind, X, y, hyp, np.zeros(N), True, with_grad=False, cov=cov, **CG)
        v = torch.zeros(N, dtype=torch.float64, device=ctx.device)
        res = ctx.objective_and_grad(v, True, CG["max_error"], CG["max_cg_iter"], CG["restart_cg_iter"], with_grad=True)
        assert ref.steps > 40, "the case is meant to cross the i % 40 == 39 restart"
        assert abs(res.steps - ref.steps) <= 1
        assert res.bound == pytest.approx(ref.bound, rel=1e-6)                       # north_star
        assert res.lower == pytest.approx(ref.lower, rel=1e-6) and res.upper == pytest.approx(ref.upper, rel=1e-6)
        if res.steps == ref.steps:
            assert res.bound == pytest.approx(ref.bound, rel=1e-7)   # > 40 steps: CG round-off amplification (DESIGN.md section 6 note)
            np.testing.assert_allclose(v.cpu().numpy(), ref.v, rtol=0, atol=1e-4 * np.abs(ref.v).max())
        # bound assembly + analytic gradient at the SAME v (the HIP solution): no CG in between, tight tolerances
        at_v = orc.objective(kind, X, y, hyp, v.cpu().numpy(), run_cg=False, with_grad=True, cov=cov)
        fixed = ctx.objective_and_grad(v, run_cg=False, with_grad=True)
        assert fixed.bound == pytest.approx(at_v.bound, rel=1e-10)
        assert fixed.lower == pytest.approx(at_v.lower, rel=1e-9) and fixed.upper == pytest.approx(at_v.upper, rel=1e-9)
        for key in ("lengthscales", "variance", "noise", "mean", "Z"):
            refg = np.asarray(at_v.grad[key])
            np.testing.assert_allclose(np.asarray(fixed.grad[key]), refg, rtol=1e-6, atol=1e-7 * max(1.0, np.abs(refg).max()), err_msg=f"{kind} {key}")
            np.testing.assert_allclose(np.asarray(res.grad[key]), refg, rtol=1e-6, atol=1e-7 * max(1.0, np.abs(refg).max()), err_msg=f"{kind} {key} (with CG)")
        ctx.close()
        del cov


HEADLINE = sorted(glob.glob(os.path.join(GOLDEN_DIR, "headline", "*.npz")))


@pytest.mark.parametrize("precision", [1, 0], ids=["fast", "exact"])
@pytest.mark.parametrize("path", HEADLINE, ids=[os.path.splitext(os.path.basename(p))[0] for p in HEADLINE])
def test_headline_fixture_from_reference_solver(path, precision):
    """steps / 1/2 r^T P r / bound / lower / upper / gradient at the headline size against the fixture made by the reference's own
    solver loop (conjugate_gradient.py:41-113) over the blocked C operator."""
    g = dict(np.load(path))
    N, D, M, kind = int(g["N"]), int(g["D"]), int(g["M"]), int(g["kind"])
    ctx, X, y, hyp = _ctx(N, D, M, kind, hyp_kind=str(g["hyp_kind"]), seed=int(g["seed"]))
    ctx.set_option("precision", precision)
    np.testing.assert_array_equal(hyp.lengthscales, g["lengthscales"])
    assert hyp.noise == float(g["noise"]) and hyp.variance == float(g["variance"])
    v = torch.zeros(N, dtype=torch.float64, device=ctx.device)
    res = ctx.objective_and_grad(v, True, float(g["max_error"]), int(g["max_cg_iter"]), int(g["restart_cg_iter"]), with_grad=True)
    ref_steps = int(g["steps"])
    assert abs(res.steps - ref_steps) <= (0 if ref_steps <= 40 else 1)
    assert res.logdet == pytest.approx(float(g["logdet"]), rel=1e-10)
    assert res.bound == pytest.approx(float(g["bound"]), rel=1e-6)                       # north_star
    assert res.lower == pytest.approx(float(g["lower"]), rel=1e-6) and res.upper == pytest.approx(float(g["upper"]), rel=1e-6)
    if res.steps != ref_steps:   # only possible beyond 40 steps: say so instead of silently checking less
        import warnings
        warnings.warn(f"{os.path.basename(path)}: {res.steps} steps against the fixture's {ref_steps}: only the 1e-6 assertions were applied")
    if res.steps == ref_steps:
        # Measured agreement with these fixtures (both precision levels, 15 ... 62 steps): bound <= 3e-14, 1/2 r^T P r <= 5e-11,
        # v <= 2e-11 max|v|, gradient <= 3e-12 of its largest entry (d/d mean, a cancelling sum: 3e-8 of its own value).
        # The assertions keep two to three orders of margin on that.
        assert res.bound == pytest.approx(float(g["bound"]), rel=1e-10)
        assert res.lower == pytest.approx(float(g["lower"]), rel=1e-10) and res.upper == pytest.approx(float(g["upper"]), rel=1e-10)
        assert res.residual_error == pytest.approx(float(g["residual_error"]), rel=1e-7)
        vs = v.cpu().numpy()[::int(g["v_stride"])]
        np.testing.assert_allclose(vs, g["v_sample"], rtol=0, atol=1e-8 * np.abs(g["v_sample"]).max())
        for key in ("lengthscales", "variance", "noise", "mean", "Z"):
            refg = np.asarray(g["g_" + key])
            # where the fixture carries the oracle's own noise floor of a block (eps-sized moves of Z; ill-conditioned K_uu, e.g. C4's D = 3
            # at M = 1024), 10x that floor is admissible when it exceeds the fixed tolerance (DESIGN.md section 2 "parity policy")
            floor = 10.0 * float(g["floor_" + key]) if ("floor_" + key) in g else 0.0
            if key == "mean":
                np.testing.assert_allclose(np.asarray(res.grad[key]), refg, rtol=1e-5, atol=floor, err_msg=key)
            else:
                np.testing.assert_allclose(np.asarray(res.grad[key]), refg, rtol=0, atol=max(1e-9 * np.abs(refg).max(), floor), err_msg=key)
    ctx.close()
