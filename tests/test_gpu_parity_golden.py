"""HIP path (through the C ABI) against the golden vectors of the reference solver and the numpy oracle.
Tolerances: fp64, north_star asks for 1e-6 relative on the bound; the kernels deliver far better and the
tests hold them to it (1e-9 .. 1e-11) wherever CG's own round-off amplification allows."""
import numpy as np
import pytest
import torch

from conftest import golden_hypers, golden_names, load_golden
from oracle import cglb_oracle as orc

pytestmark = pytest.mark.gpu


def make_ctx(g):
    from cglb_amd.hip_context import HipContext
    hyp = golden_hypers(g)
    ctx = HipContext(g["X"], g["y"], hyp.Z.shape[0], int(g["kind"]))
    ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, hyp.Z, hyp.jitter)
    return ctx, hyp


@pytest.mark.parametrize("name", golden_names())
def test_matvec_vs_dense(name):
    g = load_golden(name)
    ctx, hyp = make_ctx(g)
    cov = orc.dense_cov(int(g["kind"]), g["X"], hyp)
    p = g["r_test"]
    out = ctx.matvec(torch.from_numpy(p)).cpu().numpy()
    ref = cov @ p
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-12 * np.abs(ref).max())


@pytest.mark.parametrize("name", golden_names())
def test_common_terms_and_precond(name):
    g = load_golden(name)
    ctx, hyp = make_ctx(g)
    ctx.setup()
    terms = orc.common_terms(int(g["kind"]), g["X"], hyp)
    A = ctx.get_matrix("A").cpu().numpy()
    L = ctx.get_matrix("L").cpu().numpy()
    LB = ctx.get_matrix("LB").cpu().numpy()
    np.testing.assert_allclose(L, terms.L, rtol=0, atol=1e-9 * np.abs(terms.L).max())
    np.testing.assert_allclose(A, terms.A, rtol=0, atol=1e-8 * np.abs(terms.A).max())
    np.testing.assert_allclose(LB, terms.LB, rtol=0, atol=1e-9 * np.abs(terms.LB).max())
    assert ctx.logdet() == pytest.approx(float(g["logdet"]), rel=1e-11)
    z, rz = ctx.precond(torch.from_numpy(g["r_test"]))
    np.testing.assert_allclose(z.cpu().numpy(), g["z_test"], rtol=0, atol=1e-10 * np.abs(g["z_test"]).max())
    assert rz == pytest.approx(float(g["rz_test"]), rel=1e-10)


@pytest.mark.parametrize("name", golden_names())
def test_pcg_vs_reference_solver(name):
    g = load_golden(name)
    ctx, hyp = make_ctx(g)
    ctx.setup()
    b = torch.from_numpy(g["y"] - hyp.mean)
    v, steps, half_rz = ctx.pcg(b, torch.from_numpy(g["v0"]), float(g["max_error"]), int(g["max_cg_iter"]), int(g["restart_cg_iter"]))
    ref_steps = int(g["steps"])
    assert abs(steps - ref_steps) <= (0 if ref_steps <= 40 else 1)
    assert half_rz <= float(g["max_error"]) or steps == int(g["max_cg_iter"])
    if steps == ref_steps:
        scale = np.abs(g["v"]).max()
        np.testing.assert_allclose(v.cpu().numpy(), g["v"], rtol=0, atol=(1e-8 if ref_steps <= 40 else 1e-4) * scale)
        assert half_rz == pytest.approx(float(g["residual_error"]), rel=1e-5 if ref_steps <= 40 else 0.5)


@pytest.mark.parametrize("name", golden_names())
def test_objective_with_cg(name):
    g = load_golden(name)
    ctx, hyp = make_ctx(g)
    v = torch.from_numpy(g["v0"]).to(ctx.device).clone()
    res = ctx.objective_and_grad(v, True, float(g["max_error"]), int(g["max_cg_iter"]), int(g["restart_cg_iter"]), with_grad=False)
    assert abs(res.steps - int(g["steps"])) <= (0 if int(g["steps"]) <= 40 else 1)
    assert res.bound == pytest.approx(float(g["bound"]), rel=1e-6)   # north_star tolerance
    if res.steps == int(g["steps"]) and res.steps <= 20:
        assert res.bound == pytest.approx(float(g["bound"]), rel=1e-10)


@pytest.mark.parametrize("name", golden_names())
def test_bound_and_gradient_at_reference_v(name):
    g = load_golden(name)
    ctx, hyp = make_ctx(g)
    v = torch.from_numpy(g["v"]).to(ctx.device).clone()
    res = ctx.objective_and_grad(v, run_cg=False)
    assert res.bound == pytest.approx(float(g["bound"]), rel=1e-11)
    assert res.lower == pytest.approx(float(g["lower"]), rel=1e-9)
    assert res.upper == pytest.approx(float(g["upper"]), rel=1e-9)
    for key in ("lengthscales", "variance", "noise", "mean", "Z"):
        ref = g["g_" + key]
        tol = 1e-8 * max(1.0, np.abs(ref).max())
        if key == "mean":
            tol = 1e-11 * np.abs(g["v"]).sum()
        np.testing.assert_allclose(np.asarray(res.grad[key]), ref, rtol=1e-7, atol=tol, err_msg=key)


@pytest.mark.parametrize("precision", [0, 1], ids=["exact", "fast"])
@pytest.mark.parametrize("name", golden_names())
def test_implicit_preconditioner_matches_reference(name, precision):
    """precond_mode = 1: A r and A^T t formed as sigma^-1 L^-1 (K_uf r) / K_fu (L^-T t)/sigma with the tiled pair kernel.
    Long solves (> 40 steps at tolerances 1e-5 / 1e-6, far below the reference's operating point of 1.0) amplify every
    perturbation of the operator: +-1 step at the exact precision level (kernel values to 3e-16), +-2 at the fast one (1e-13)."""
    g = load_golden(name)
    from cglb_amd.hip_context import HipContext
    hyp = golden_hypers(g)
    ctx = HipContext(g["X"], g["y"], hyp.Z.shape[0], int(g["kind"]))
    ctx.set_option("precision", precision)
    ctx.set_option("precond_mode", 1)
    ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, hyp.Z, hyp.jitter)
    ctx.setup()
    z, rz = ctx.precond(torch.from_numpy(g["r_test"]))
    np.testing.assert_allclose(z.cpu().numpy(), g["z_test"], rtol=0, atol=1e-9 * np.abs(g["z_test"]).max())
    assert rz == pytest.approx(float(g["rz_test"]), rel=1e-9)
    v = torch.from_numpy(g["v0"]).to(ctx.device).clone()
    res = ctx.objective_and_grad(v, True, float(g["max_error"]), int(g["max_cg_iter"]), int(g["restart_cg_iter"]))
    assert abs(res.steps - int(g["steps"])) <= (0 if int(g["steps"]) <= 40 else 1 + precision)
    assert res.bound == pytest.approx(float(g["bound"]), rel=1e-6)
    if res.steps == int(g["steps"]) and res.steps <= 20:
        assert res.bound == pytest.approx(float(g["bound"]), rel=1e-9)


@pytest.mark.parametrize("name", golden_names())
def test_precision_levels_agree(name):
    """cglb_set_option("precision"): 0 = degree-4 2^x polynomial + two-step sqrt (kernel values to ~3e-16), 1 (default) = degree-3 +
    one-step sqrt (<= ~1e-13).  The two levels must agree on the operator to 2e-13 and on the golden PCG outputs (exact level:
    same assertions as the default-level tests above)."""
    g = load_golden(name)
    ctx, hyp = make_ctx(g)
    p = torch.from_numpy(g["r_test"])
    fast = ctx.matvec(p).cpu().numpy()
    ctx.set_option("precision", 0)
    exact = ctx.matvec(p).cpu().numpy()
    cov = orc.dense_cov(int(g["kind"]), g["X"], hyp)
    ref = cov @ g["r_test"]
    np.testing.assert_allclose(exact, ref, rtol=0, atol=1e-13 * np.abs(ref).max())
    np.testing.assert_allclose(fast, exact, rtol=0, atol=2e-13 * np.abs(ref).max())
    ctx.setup()
    b = torch.from_numpy(g["y"] - hyp.mean)
    v, steps, half_rz = ctx.pcg(b, torch.from_numpy(g["v0"]), float(g["max_error"]), int(g["max_cg_iter"]), int(g["restart_cg_iter"]))
    ref_steps = int(g["steps"])
    assert abs(steps - ref_steps) <= (0 if ref_steps <= 40 else 1)
    vv = torch.from_numpy(g["v"]).to(ctx.device).clone()
    res = ctx.objective_and_grad(vv, run_cg=False)
    assert res.bound == pytest.approx(float(g["bound"]), rel=1e-11)
    for key in ("lengthscales", "variance", "noise", "Z"):
        refg = g["g_" + key]
        np.testing.assert_allclose(np.asarray(res.grad[key]), refg, rtol=1e-7, atol=1e-8 * max(1.0, np.abs(refg).max()), err_msg=key)


import glob as _glob
import os as _os

PREDICT_CASES = sorted(_glob.glob(_os.path.join(_os.path.dirname(__file__), "golden", "predict", "*.npz")))


@pytest.mark.parametrize("path", PREDICT_CASES, ids=[_os.path.splitext(_os.path.basename(p))[0] for p in PREDICT_CASES])
def test_predict_cg_vs_reference_solver_driven_golden(path):
    """PredictCG (models.py:289-354) through the backend classes against tests/golden/predict/*.npz (the reference's own solver at
    tolerance 1e-3, warm-started at the model's v; predictor algebra restated in torch - oracle/gen_predict_golden.py)."""
    import torch
    from cglb_amd.backend.models import CGLB, BaseKernel, GaussianLikelihood, InducingPointKernel, PredictCG, ScaleKernel
    p = dict(np.load(path))
    g = load_golden(str(p["source"]))
    hyp = golden_hypers(g)
    kind = "rbf" if int(g["kind"]) == 0 else "matern32"
    base = BaseKernel(kind, ard_num_dims=g["X"].shape[1])
    base.lengthscale = hyp.lengthscales
    scale = ScaleKernel(base)
    scale.outputscale = hyp.variance
    lik = GaussianLikelihood(lower_bound=1e-6)
    lik.noise = hyp.noise
    model = CGLB((g["X"], g["y"]), lik, InducingPointKernel(scale, hyp.Z))
    with torch.no_grad():
        model.mean_module.constant.copy_(torch.tensor(hyp.mean, dtype=torch.float64))
        model.v_vec.copy_(torch.from_numpy(g["v"]).reshape(-1, 1))            # the model's v after its training solve
    pred = PredictCG(model)
    f_mean, f_var = pred(torch.from_numpy(p["xnew"]))
    np.testing.assert_allclose(pred.v_vec.cpu().numpy().reshape(-1), p["new_v"], rtol=0, atol=1e-8 * np.abs(p["new_v"]).max())
    np.testing.assert_allclose(f_mean.cpu().numpy().reshape(-1), p["f_mean"], rtol=0, atol=1e-8 * np.abs(p["f_mean"]).max())
    np.testing.assert_allclose(f_var.cpu().numpy().reshape(-1), p["f_var"], rtol=0, atol=1e-8 * np.abs(p["f_var"]).max())
    np.testing.assert_array_equal(model.v_vec.cpu().numpy().reshape(-1), g["v"])      # the model's own v is not touched (models.py:294)
    model.hip.close()


@pytest.mark.parametrize("name", ["rbf_d8_trained", "m32_d8_trained", "m32_d3_random", "rbf_d8_init", "c1_snelson_like_m32"])
def test_opt_in_low_precision_level(name):
    """cglb_set_option("precision", 2): degree-2 table polynomial, kernel values to ~1e-10 (one instruction per pair cheaper than the
    default).  Opt-in: inside north_star's 1e-6 on the bound, outside the 1e-10 of the default level - held to 1e-8 / 5e-10 here.
    Meant for the reference's stopping tolerances (1/2 r^T P r <= 1 in training, 1e-3 in prediction): a solve driven to 1e-5 sits at
    the noise floor an operator known to 1e-10 allows (the > 40-step golden case takes 63 steps instead of 59 there) and is not in this list."""
    g = load_golden(name)
    ctx, hyp = make_ctx(g)
    ctx.set_option("precision", 2)
    p = torch.from_numpy(g["r_test"])
    cov = orc.dense_cov(int(g["kind"]), g["X"], hyp)
    ref = cov @ g["r_test"]
    low = ctx.matvec(p).cpu().numpy()
    err = np.abs(low - ref).max() / np.abs(ref).max()
    assert 1e-13 < err < 5e-10, err                      # it IS the low level (not silently the default) and within its contract
    v = torch.from_numpy(g["v0"]).to(ctx.device).clone()
    res = ctx.objective_and_grad(v, True, float(g["max_error"]), int(g["max_cg_iter"]), int(g["restart_cg_iter"]))
    ref_steps = int(g["steps"])
    assert abs(res.steps - ref_steps) <= (0 if ref_steps <= 40 else 2)
    assert res.bound == pytest.approx(float(g["bound"]), rel=1e-6)
    at_v = orc.objective(int(g["kind"]), g["X"], g["y"], hyp, v.cpu().numpy(), run_cg=False, with_grad=True, cov=cov)
    assert res.bound == pytest.approx(at_v.bound, rel=1e-8)
    for key in ("lengthscales", "Z"):
        np.testing.assert_allclose(np.asarray(res.grad[key]), at_v.grad[key], rtol=0, atol=1e-6 * np.abs(at_v.grad[key]).max(), err_msg=key)
    ctx.close()
