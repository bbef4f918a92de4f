"""Edge cases of the boundary: tiny and ragged sizes, duplicated points, error codes mapped to the reference's exceptions."""
import numpy as np
import pytest
import torch

from oracle import cglb_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,D,M", [(1, 1, 1), (2, 3, 2), (17, 1, 3), (63, 2, 5), (65, 8, 8), (257, 4, 16)])
@pytest.mark.parametrize("kind", ["rbf", "matern32"])
def test_tiny_problems_match_oracle(N, D, M, kind):
    from cglb_amd.hip_context import HipContext
    X, y, Z = orc.synthetic_problem(max(N, M, 8), D, M, seed=N + M)  # generated larger (z-normalisation needs > 1 point), then cut
    X, y = X[:N], y[:N]
    hyp = orc.Hypers(np.full(D, 0.9), 1.3, 0.4, 0.2, Z, 1e-6)
    ctx = HipContext(X, y, M, kind)
    ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
    v = torch.zeros(N, dtype=torch.float64, device=ctx.device)
    res = ctx.objective_and_grad(v, True, 1e-6)
    ref = orc.objective(kind, X, y, hyp, np.zeros(N), True, 1e-6)
    # a solve run down to 1/2 r^T P r <= 1e-6 sits at the round-off floor of its last steps: when the oracle's statistic lands within
    # a few 1e-8 of the threshold, a last-bit difference decides one more step (same rule as the duplicate-points test below)
    assert abs(res.steps - ref.steps) <= 1
    # two converged solves (1/2 r^T P r <= 1e-6) agree to within the CG tolerance itself
    assert res.bound == pytest.approx(ref.bound, rel=1e-9, abs=1e-6)
    refg = orc.objective(kind, X, y, hyp, v.cpu().numpy(), run_cg=False, with_grad=True).grad
    np.testing.assert_allclose(res.grad["lengthscales"], refg["lengthscales"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(res.grad["Z"], refg["Z"], rtol=1e-7, atol=1e-9)
    assert res.grad["noise"] == pytest.approx(refg["noise"], rel=1e-8, abs=1e-10)


def test_duplicate_training_points_and_inducing_on_data():
    """Coincident points: distance exactly 0 (Matern sqrt at 0, gradient finite), inducing points equal to data points."""
    from cglb_amd.hip_context import HipContext
    X, y, _ = orc.synthetic_problem(200, 2, 4, seed=3)
    X[10] = X[11] = X[12]
    Z = X[[10, 50, 90, 130]].copy()
    hyp = orc.Hypers(np.array([0.8, 1.2]), 1.0, 0.1, 0.0, Z, 1e-6)
    for kind in ("rbf", "matern32"):
        ctx = HipContext(X, y, 4, kind)
        ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
        v = torch.zeros(200, dtype=torch.float64, device=ctx.device)
        res = ctx.objective_and_grad(v, True, 1e-4)
        ref = orc.objective(kind, X, y, hyp, np.zeros(200), True, 1e-4)
        # the oracle stops with 1/2 r^T P r = 9.8e-5 against the 1e-4 threshold: a last-bit difference decides one more step
        assert abs(res.steps - ref.steps) <= 1 and res.bound == pytest.approx(ref.bound, abs=2e-4)
        same_v = orc.objective(kind, X, y, hyp, v.cpu().numpy(), run_cg=False)
        assert res.bound == pytest.approx(same_v.bound, rel=1e-11)
        assert np.isfinite(res.grad["Z"]).all() and np.isfinite(res.grad["lengthscales"]).all()
        refg = orc.objective(kind, X, y, hyp, v.cpu().numpy(), run_cg=False, with_grad=True).grad
        np.testing.assert_allclose(res.grad["Z"], refg["Z"], rtol=1e-6, atol=1e-8)


def test_error_mapping():
    from cglb_amd.hip_context import HipContext
    X, y, Z = orc.synthetic_problem(100, 2, 4, seed=1)
    ctx = HipContext(X, y, 4, "rbf")
    with pytest.raises(ValueError):
        ctx.set_hypers(np.array([1.0, -1.0]), 1.0, 0.1, 0.0, Z, 1e-6)          # non-positive lengthscale
    with pytest.raises(ValueError):
        ctx.set_hypers(np.ones(2), 1.0, 0.0, 0.0, Z, 1e-6)                       # noise must be positive
    with pytest.raises(RuntimeError):
        ctx.setup()                                                              # call order: hypers not set
    Zdup = Z.copy()
    Zdup[1] = Zdup[0]
    ctx.set_hypers(np.ones(2), 1.0, 0.1, 0.0, Zdup, 0.0)                         # singular K_uu, no jitter
    with pytest.raises(RuntimeError, match="[Cc]holesky"):
        ctx.setup()                                                              # models.py:202 lets torch.cholesky raise
    ctx.set_hypers(np.ones(2), 1.0, 0.1, 0.0, Z, 1e-6)
    ctx.setup()
    with pytest.raises(ValueError):
        ctx.matvec(torch.zeros(99, dtype=torch.float64))                         # wrong length
    with pytest.raises(ValueError):
        HipContext(X, y[:50], 4, "rbf")
    with pytest.raises(KeyError):
        HipContext(X, y, 4, "periodic")


def test_warm_start_is_not_mutated_by_solver_and_reused_by_objective():
    from cglb_amd.hip_context import HipContext
    X, y, Z = orc.synthetic_problem(500, 3, 16, seed=2)
    hyp = orc.trained_like_hypers(3, Z)
    ctx = HipContext(X, y, 16, "rbf")
    ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, 1e-6)
    ctx.setup()
    b = ctx.y - hyp.mean
    v0 = torch.zeros(500, dtype=torch.float64, device=ctx.device)
    v1, s1, _ = ctx.pcg(b, v0, 1.0)
    assert float(v0.abs().max()) == 0.0 and s1 > 0                                # conjugate_gradient.py:55 (clone)
    v2, s2, _ = ctx.pcg(b, v1, 1.0)
    assert s2 == 0 and torch.equal(v2, v1)                                        # already converged: predicate fails before any step
    res = ctx.objective_and_grad(v1.clone(), True, 1.0, with_grad=False)
    assert res.steps == 0


@pytest.mark.parametrize("name,max_error", [("rbf_d8_trained", 1.0), ("rbf_d8_init", 1.0), ("rbf_d8_restart", None), ("rbf_d8_init", 50.0)])
def test_lookahead_stop_test_does_not_change_results(name, max_error):
    """pcg_lookahead=1 enqueues the next mat-vec before the host has seen the stop-test scalar; a wasted speculative mat-vec
    (residual dropping by more than 4x in one iteration - the well-conditioned init case with a loose tolerance) must leave
    v, steps and the residual bitwise unchanged."""
    from conftest import load_golden
    from cglb_amd.hip_context import HipContext
    g = load_golden(name)
    me = float(g["max_error"]) if max_error is None else max_error
    outs = []
    for la in (0, 1):
        ctx = HipContext(g["X"], g["y"], g["Z"].shape[0], int(g["kind"]))
        ctx.set_option("pcg_lookahead", la)
        ctx.set_hypers(g["lengthscales"], float(g["variance"]), float(g["noise"]), float(g["mean"]), g["Z"], float(g["jitter"]))
        ctx.setup()
        b = torch.from_numpy(g["y"] - float(g["mean"]))
        v, steps, half = ctx.pcg(b, torch.from_numpy(g["v0"]), me, int(g["max_cg_iter"]), int(g["restart_cg_iter"]))
        outs.append((v.cpu().numpy(), steps, half))
    assert outs[0][1] == outs[1][1] and outs[0][2] == outs[1][2]
    assert np.array_equal(outs[0][0], outs[1][0])


def test_in_situ_kernel_timing_counts_every_matvec_of_a_solve():
    """k1_profile brackets each launch of the symmetric pair kernel with HIP events: a cold-start solve of `steps` iterations
    launches it steps + (number of restarts) times (the initial mat-vec is skipped for v0 = 0, no look-ahead here)."""
    from conftest import load_golden
    from cglb_amd.hip_context import HipContext
    g = load_golden("rbf_d8_restart")
    ctx = HipContext(g["X"], g["y"], g["Z"].shape[0], int(g["kind"]))
    ctx.set_option("pcg_lookahead", 0)
    ctx.set_hypers(g["lengthscales"], float(g["variance"]), float(g["noise"]), float(g["mean"]), g["Z"], float(g["jitter"]))
    ctx.setup()
    b = torch.from_numpy(g["y"] - float(g["mean"]))
    ctx.set_option("k1_profile", 1)
    v, steps, half = ctx.pcg(b, torch.zeros(len(b), dtype=torch.float64), float(g["max_error"]), int(g["max_cg_iter"]), int(g["restart_cg_iter"]))
    launches, ms = ctx.get_stat("k1_launches"), ctx.get_stat("k1_ms_total")
    ctx.set_option("k1_profile", 0)
    restarts = sum(1 for i in range(steps) if i % int(g["restart_cg_iter"]) == int(g["restart_cg_iter"]) - 1)
    assert launches == steps + restarts and ms > 0
    with pytest.raises(ValueError):
        ctx.get_stat("no_such_statistic")
    ctx.matvec(torch.from_numpy(g["y"]))  # not profiled any more
    assert ctx.get_stat("k1_launches") == launches


@pytest.mark.parametrize("kind", ["rbf", "matern32"])
@pytest.mark.parametrize("ls", [1.0, 0.1, 0.03])
def test_gram_form_gradient_pass_against_direct_differences_and_oracle(kind, ls):
    """The symmetric N^2 gradient pass builds sum_j hv (x_id - x_jd)^2 from moments (x^2 S0 - 2 x S1 + S2), which cancels when the
    lengthscale is far below the data range (here range/l up to ~200: ~4.5 of 16 digits).  Both forms of the pass and the oracle
    must agree on the lengthscale gradient well inside what the optimiser needs."""
    from cglb_amd.hip_context import HipContext
    N, D, M = 1200, 3, 16
    rng = np.random.default_rng(17)
    X = rng.uniform(-3.0, 3.0, size=(N, D))
    y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(N)
    Z = X[:M].copy()
    hyp = orc.Hypers(np.full(D, ls), 1.3, 0.2, 0.1, Z, 1e-6)
    v = rng.standard_normal(N) * 0.1
    ref = orc.objective(kind, X, y, hyp, v, run_cg=False, with_grad=True)
    got = []
    for gram in (0, 1):
        ctx = HipContext(X, y, M, kind)
        ctx.set_option("grad_gram", gram)
        ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
        vd = torch.from_numpy(v).to(ctx.device)
        res = ctx.objective_and_grad(vd, False)
        assert res.bound == pytest.approx(ref.bound, rel=1e-11)
        got.append(res.grad["lengthscales"])
        np.testing.assert_allclose(res.grad["lengthscales"], ref.grad["lengthscales"], rtol=2e-8, atol=1e-9 * np.abs(ref.grad["lengthscales"]).max())
    np.testing.assert_allclose(got[1], got[0], rtol=1e-9, atol=1e-11 * np.abs(got[0]).max())


@pytest.mark.parametrize("M", [64, 65, 100, 191, 320])
@pytest.mark.parametrize("dtype", ["fp64", "fp32"])
def test_ragged_inducing_counts_factorise_like_the_oracle(M, dtype):
    """M that is not a multiple of the 64-column block of the blocked Cholesky (models.py:202, :210): short last block, ragged panel
    rows and trailing tiles; the factors, the bound and the gradient follow the dense oracle."""
    from cglb_amd.hip_context import HipContext
    N, D = 700, 3
    X, y, Z = orc.synthetic_problem(N, D, M, seed=M)
    td = torch.float64 if dtype == "fp64" else torch.float32
    tol = 1e-9 if dtype == "fp64" else 2e-3
    hyp = orc.Hypers(np.array([0.9, 1.1, 1.4]), 1.2, 0.3, 0.1, Z, 1e-6 if dtype == "fp64" else 1e-4)
    ctx = HipContext(X, y, M, "rbf", dtype=td)
    ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
    ctx.setup()
    terms = orc.common_terms("rbf", X, hyp)
    L = ctx.get_matrix("L").double().cpu().numpy()
    LB = ctx.get_matrix("LB").double().cpu().numpy()
    assert np.all(np.triu(L, 1) == 0) and np.all(np.triu(LB, 1) == 0)
    np.testing.assert_allclose(L, terms.L, rtol=0, atol=tol * np.abs(terms.L).max())
    np.testing.assert_allclose(LB, terms.LB, rtol=0, atol=tol * np.abs(terms.LB).max())
    v = torch.zeros(N, dtype=td, device=ctx.device)
    res = ctx.objective_and_grad(v, True, 1e-2)
    ref = orc.objective("rbf", X, y, hyp, np.zeros(N), True, 1e-2)
    if dtype == "fp64":
        assert res.steps == ref.steps
        assert res.bound == pytest.approx(ref.bound, rel=1e-9)
        refg = orc.objective("rbf", X, y, hyp, v.cpu().numpy(), run_cg=False, with_grad=True).grad
        np.testing.assert_allclose(res.grad["Z"], refg["Z"], rtol=1e-6, atol=1e-8 * np.abs(refg["Z"]).max())
        np.testing.assert_allclose(res.grad["lengthscales"], refg["lengthscales"], rtol=1e-7)
    else:
        assert abs(res.steps - ref.steps) <= 1 and res.bound == pytest.approx(ref.bound, rel=2e-3)



@pytest.mark.parametrize("kind", ["rbf", "matern32"])
@pytest.mark.parametrize("D,ls", [(20, 0.6), (8, 0.25), (27, 1.0)])
def test_clamped_exponent_range_bound_and_gradient(kind, D, ls):
    """Large scaled coordinates (wide D at short lengthscales - the reference's initial l = 1 at D >= 17 is such a case): the pair kernels
    switch to their range-clamped 2^x, kernel values underflow to exact zeros far from the diagonal; bound and gradient still follow the oracle."""
    from cglb_amd.hip_context import HipContext
    N, M = 900, 40
    X, y, Z = orc.synthetic_problem(N, D, M, seed=D)
    hyp = orc.Hypers(np.full(D, ls), 1.1, 0.3, 0.05, Z, 1e-6)
    ctx = HipContext(X, y, M, kind)
    ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
    v = torch.zeros(N, dtype=torch.float64, device=ctx.device)
    res = ctx.objective_and_grad(v, True, 1e-2, 100, 40, with_grad=True)
    ref = orc.objective(kind, X, y, hyp, np.zeros(N), True, 1e-2, 100, 40)
    refg = orc.objective(kind, X, y, hyp, v.cpu().numpy(), run_cg=False, with_grad=True)
    assert abs(res.steps - ref.steps) <= 1
    assert res.bound == pytest.approx(refg.bound, rel=1e-10)
    for key in ("lengthscales", "Z"):
        scale = np.abs(refg.grad[key]).max() + 1e-300
        np.testing.assert_allclose(res.grad[key], refg.grad[key], rtol=0, atol=1e-8 * scale + 1e-12 * abs(refg.bound))
    assert res.grad["noise"] == pytest.approx(refg.grad["noise"], rel=1e-8)
    assert res.grad["variance"] == pytest.approx(refg.grad["variance"], rel=1e-8, abs=1e-10 * abs(refg.bound))


@pytest.mark.parametrize("kind", ["rbf", "matern32"])
def test_consecutive_cold_start_solves_on_one_context(kind):
    """Two cold-start solves with different right-hand sides on ONE context (and then a cold-start evaluation): the weighted copy
    p o w that the last direction update of a solve leaves behind must not be taken for the operand of the next solve's first
    mat-vec (the stop without a look-ahead mat-vec - max_iter reached, or the residual dropped by less than 4x - leaves it unconsumed)."""
    from cglb_amd.hip_context import HipContext
    N, D, M = 700, 8, 16
    X, y, Z = orc.synthetic_problem(N, D, M, seed=21)
    hyp = orc.trained_like_hypers(D, Z)
    ctx = HipContext(X, y, M, kind)
    ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
    ctx.setup()
    terms = orc.common_terms(kind, X, hyp)
    cov = orc.dense_cov(kind, X, hyp)
    rng = np.random.default_rng(3)
    for trial, max_iter in enumerate((7, 7, 100)):          # 7: leaves the loop at max_iter, right after a fused direction update
        b = rng.standard_normal(N) * (1.0 + trial)
        v, steps, half = ctx.pcg(b, np.zeros(N), 1e-8, max_iter, 40)
        ref_v, st = orc.pcg(lambda x: cov @ x, b, np.zeros(N), lambda r: orc.nystrom_precond(terms.A, terms.LB, hyp.noise, r), 1e-8, max_iter, 40)
        ref_steps = st.steps
        assert abs(steps - ref_steps) <= (0 if max_iter == 7 else 1), (trial, steps, ref_steps)
        if steps == ref_steps:
            # 7 steps: round-off only.  The run to convergence (tens of steps behind a 16-point preconditioner, stop at 1/2 r^T P r <= 1e-8)
            # pins v no better than the stopping tolerance does: |dv| <~ |K^-1| |r|
            tol_v = 1e-8 if max_iter == 7 else 1e-3
            np.testing.assert_allclose(v.cpu().numpy(), ref_v, rtol=0, atol=tol_v * np.abs(ref_v).max(), err_msg=f"solve {trial}")
    # a cold-start evaluation right after a solve on the same context
    vz = torch.zeros(N, dtype=torch.float64, device=ctx.device)
    res = ctx.objective_and_grad(vz, True, 1.0, 100, 40, with_grad=False)
    ref = orc.objective(kind, X, y, hyp, np.zeros(N), True, 1.0, 100, 40)
    assert res.steps == ref.steps and res.bound == pytest.approx(ref.bound, rel=1e-6)     # 24 steps behind a 16-point preconditioner
    assert res.bound == pytest.approx(orc.objective(kind, X, y, hyp, vz.cpu().numpy(), run_cg=False).bound, rel=1e-11)
    # ... and the used context must give exactly what a fresh one gives: nothing of the earlier solves leaks into this one
    fresh = HipContext(X, y, M, kind)
    fresh.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
    vf = torch.zeros(N, dtype=torch.float64, device=fresh.device)
    rf = fresh.objective_and_grad(vf, True, 1.0, 100, 40, with_grad=False)
    assert (rf.steps, rf.bound) == (res.steps, res.bound)
    np.testing.assert_array_equal(vf.cpu().numpy(), vz.cpu().numpy())
    fresh.close()
    ctx.close()


@pytest.mark.parametrize("kind", ["rbf", "matern32"])
def test_kv_from_the_recurrence_residual_equals_the_recomputed_one(kind):
    """Option "final_matvec": after a solve the evaluation takes K v = e - r from the residual the PCG recurrence carries (default)
    instead of recomputing `cov @ v` (models.py:280, option value 1).  The two agree to the rounding of the mat-vec itself - cold
    start, warm start, a > 40-step solve that crosses the restart - and both agree with the oracle's assembly at the same v."""
    from cglb_amd.hip_context import HipContext
    N, D, M = 3000, 8, 24
    X, y, Z = orc.synthetic_problem(N, D, M, seed=8)
    hyp = orc.trained_like_hypers(D, Z)
    hyp.noise = 0.02
    out = {}
    for fm in (1, 0):
        ctx = HipContext(X, y, M, kind)
        ctx.set_option("final_matvec", fm)
        ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
        v = torch.zeros(N, dtype=torch.float64, device=ctx.device)
        cold = ctx.objective_and_grad(v, True, 1e-3, 100, 40)
        v_cold = v.cpu().numpy().copy()
        ctx.set_hypers(hyp.lengthscales * 1.03, hyp.variance, hyp.noise * 1.05, hyp.mean + 0.01, Z, hyp.jitter)
        warm = ctx.objective_and_grad(v, True, 1e-3, 100, 40)
        out[fm] = (cold, v_cold, warm, v.cpu().numpy().copy())
        ctx.close()
    (c1, vc1, w1, vw1), (c0, vc0, w0, vw0) = out[1], out[0]
    assert c1.steps > 40 and c0.steps == c1.steps and w0.steps == w1.steps
    np.testing.assert_array_equal(vc0, vc1)                      # the solve itself is untouched by the option
    for a, b in ((c0, c1), (w0, w1)):
        assert a.bound == pytest.approx(b.bound, rel=1e-12) and a.lower == pytest.approx(b.lower, rel=1e-11)
        for k in ("lengthscales", "Z"):
            np.testing.assert_allclose(a.grad[k], b.grad[k], rtol=0, atol=1e-9 * np.abs(b.grad[k]).max())
    ref = orc.objective(kind, X, y, hyp, vc0, run_cg=False, with_grad=True)
    assert c0.bound == pytest.approx(ref.bound, rel=1e-11)
    np.testing.assert_allclose(c0.grad["lengthscales"], ref.grad["lengthscales"], rtol=1e-7)
