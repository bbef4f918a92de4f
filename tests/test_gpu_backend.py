"""The cglb.backend mirror end to end on the GPU: model creation through the config dispatch, autograd contract of the
objective (optimizer.py:95-98), SciPy training loop, metrics, save/load, plug-in seams and error behaviour."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cglb_oracle as orc

pytestmark = pytest.mark.gpu


def _data(N=300, D=3, seed=5):
    X, y, _ = orc.synthetic_problem(N, D, 8, seed)
    n = int(N * 2 / 3)
    return (X[:n], y[:n]), (X[n:], y[n:])


def _model(kernel="Matern32", M=16, data=None):
    from cglb_amd.backend import BACKENDS, CGLBConfig, INDUCING_VARIABLE_CONFIGS, KERNEL_CONFIGS
    be = BACKENDS["hip"]
    be.configure_backend(logdir="/tmp/cglb_amd_test", keops=False)
    be.set_default_float("fp64")
    be.set_default_jitter("fp64")
    train, test = data or _data()
    cfg = CGLBConfig(kernel=KERNEL_CONFIGS[kernel](), inducing_variable=INDUCING_VARIABLE_CONFIGS["cv"](M))
    return be, be.create_model(cfg, train), (train, test)


def _hyp_of(model):
    from cglb_amd.backend.models import get_cholesky_jitter
    ls, var, noise, mean, Z = [t.detach().cpu().numpy() for t in model.hyper_tensors()]
    return orc.Hypers(ls, float(var), float(noise), float(mean), Z, get_cholesky_jitter())


@pytest.mark.parametrize("kernel,kind", [("Matern32", "matern32"), ("rbf", "rbf")])
def test_objective_value_and_autograd_contract(kernel, kind):
    from cglb_amd.backend.models import LowerBoundCG
    be, model, (train, _) = _model(kernel)
    params = be.model_parameters(model)
    assert set(params) == {".likelihood.variance", ".mean_function.c", ".inducing_variable.Z", ".kernel.lengthscales", ".kernel.variance"}
    assert params[".likelihood.variance"] == pytest.approx(1.0) and params[".kernel.variance"] == pytest.approx(1.0)  # config.py:74-76,:104
    lb = LowerBoundCG(model)
    loss = -lb(None)
    hyp = _hyp_of(model)
    ref = orc.objective(kind, train[0], train[1], hyp, np.zeros(len(train[1])), True, 1.0)
    assert model.cg_stats.steps == ref.steps
    assert float(loss) == pytest.approx(-ref.bound, rel=1e-10)
    # gradient wrt RAW parameters == chain rule of the oracle's constrained gradient (softplus -> sigmoid)
    plist = list(model.parameters())
    grads = torch.autograd.grad(loss, plist)
    v = model.v_vec.cpu().numpy().reshape(-1)
    g = orc.objective(kind, train[0], train[1], hyp, v, run_cg=False, with_grad=True).grad
    named = dict(model.named_parameters())
    got = {n: gr for (n, _), gr in zip(model.named_parameters(), grads)}
    sig = lambda raw: torch.sigmoid(raw).numpy()
    np.testing.assert_allclose(got["likelihood.noise_covar._noise.raw"].numpy(), -g["noise"] * sig(named["likelihood.noise_covar._noise.raw"].detach()), rtol=1e-7)
    np.testing.assert_allclose(got["mean_module.constant"].numpy(), -g["mean"], rtol=1e-6, atol=1e-9 * np.abs(v).sum())
    np.testing.assert_allclose(got["covar_module.inducing_points"].numpy(), -g["Z"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(got["covar_module.base_kernel.base_kernel._lengthscale.raw"].numpy().reshape(-1),
                               -g["lengthscales"] * sig(named["covar_module.base_kernel.base_kernel._lengthscale.raw"].detach()).reshape(-1), rtol=1e-7)
    np.testing.assert_allclose(got["covar_module.base_kernel._outputscale.raw"].numpy(),
                               -g["variance"] * sig(named["covar_module.base_kernel._outputscale.raw"].detach()), rtol=1e-7)


def test_cache_flags_skip_cg_like_reference():
    from cglb_amd.backend.models import LowerBoundCG
    be, model, _ = _model()
    lb = LowerBoundCG(model)
    l1 = float(-lb(None))
    steps1 = model.cg_stats.steps
    assert steps1 > 0
    cached = LowerBoundCG(model, use_cache=True, cached_v_vec_initial=True)   # interface.py:619-625
    with torch.no_grad():
        l2 = float(-cached(None))
    assert l2 == pytest.approx(l1, rel=1e-12)
    assert model.cg_stats.steps == steps1   # no CG ran
    # warm start: second evaluation starts from the stored solution -> 0 or few steps
    float(-lb(None))
    assert model.cg_stats.steps <= steps1


def test_training_loop_metrics_save_load(tmp_path):
    from cglb_amd.backend.callbacks import Logger
    be, model, data = _model("Matern32", M=12)
    mfn = be.metrics_fn(model, data)
    logger = Logger(str(tmp_path), mfn, lambda: be.model_parameters(model), holdout_interval=5, include_feval_log=True, verbose=False)
    loss0 = mfn()["loss"] if False else None
    from cglb_amd.backend.models import LowerBoundCG
    l_init = float(-LowerBoundCG(model)(None))
    results = be.optimize(model, data, 25, logger, "scipy")
    assert sum(r.nit for r in results) <= 25 + 3
    m = mfn()
    assert set(m) == {"cg/steps", "cg/error", "loss", "train/rmse", "test/rmse", "train/nlpd", "test/nlpd"}
    assert m["loss"] < l_init - 1.0            # the bound improved
    assert m["test/rmse"] < 0.9                # better than predicting the mean of z-normalised targets
    assert {"loss", "cg/steps", "cg/error", "steps-per-feval", "residual_error-per-feval", "elapsed_time", "params", "iteration"} <= set(logger.logs)
    be.save(model, str(tmp_path))
    from cglb_amd.backend import jsonio
    saved = jsonio.load(os.path.join(tmp_path, "model.json"))       # json_tricks-shaped reader
    live = be.model_parameters(model)
    assert set(saved) == set(live)
    for k in live:
        assert np.shape(saved[k]) == np.shape(live[k]) and np.allclose(saved[k], live[k], rtol=1e-15)
    be2, model2, _ = _model("Matern32", M=12, data=data)
    be2.load(model2, os.path.join(tmp_path, "model.json"))
    for k, v in be.model_parameters(model).items():
        np.testing.assert_allclose(be2.model_parameters(model2)[k], v, rtol=1e-10, atol=1e-12)


def test_predict_matches_oracle_and_errors():
    from cglb_amd.backend.models import LowerBoundCG, PredictCG
    be, model, (train, test) = _model("rbf", M=16)
    float(-LowerBoundCG(model)(None))
    hyp = _hyp_of(model)
    pred = PredictCG(model)
    f_mean, f_var = pred(torch.as_tensor(test[0]))
    ref_mean, ref_var, _, st = orc.predict("rbf", train[0], train[1], hyp, model.v_vec.cpu().numpy().reshape(-1), test[0], 1e-3)
    assert f_mean.shape == (len(test[1]), 1) and f_var.shape == f_mean.shape
    np.testing.assert_allclose(f_mean.cpu().numpy().reshape(-1), ref_mean, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(f_var.cpu().numpy().reshape(-1), ref_var, rtol=1e-7, atol=1e-9)
    with pytest.raises(NotImplementedError):
        pred(torch.as_tensor(test[0]), full_cov=True)       # models.py:311-314
    with pytest.raises(ValueError):
        LowerBoundCG(object())                              # models.py:112-113


def test_solver_and_preconditioner_seams():
    """cg_opt is any callable (A, b, v, precond) -> (v, stats) (models.py:266-271); A only needs `@`."""
    from cglb_amd.backend.conjugate_gradient import ConjugateGradient, ConjugateGradientStats, KernelOperator, NystromPreconditioner
    from cglb_amd.backend.models import LowerBoundCG
    be, model, (train, _) = _model("rbf", M=16)
    seen = {}

    class MyCG:
        """A foreign solver written against the seam: plain PCG in terms of `A @ p` and `precond(r)` only."""
        def __call__(self, A, b, v, precond):
            v = v.clone()
            r = b - A @ v
            z, rz = precond(r)
            p = z
            i = 0
            while 0.5 * float(rz) > 1.0 and i < 100:
                Ap = A @ p
                gamma = float(rz) / float((p * Ap).sum())
                v += gamma * p
                r = r - gamma * Ap
                z, new_rz = precond(r)
                p = z + p * float(new_rz) / float(rz)
                rz = new_rz
                i += 1
            seen["steps"] = i
            return v, ConjugateGradientStats(i, 0.5 * rz)

    loss_plugin = float(-LowerBoundCG(model, cg_opt=MyCG())(None))
    v_plugin = model.v_vec.clone()
    model.v_vec.zero_()
    loss_fused = float(-LowerBoundCG(model)(None))
    assert seen["steps"] == model.cg_stats.steps
    assert loss_plugin == pytest.approx(loss_fused, rel=1e-10)
    np.testing.assert_allclose(v_plugin.cpu().numpy(), model.v_vec.cpu().numpy(), rtol=0, atol=1e-9 * float(v_plugin.abs().max()))
    # the reference solver object must not mutate its v argument (conjugate_gradient.py:55)
    hip = model.hip
    v0 = torch.zeros(hip.N, 1, dtype=torch.float64, device=hip.device)
    b = (hip.y - 0.0).reshape(-1, 1)
    vout, stats = ConjugateGradient()(KernelOperator(hip), b, v0, NystromPreconditioner(hip))
    assert float(v0.abs().max()) == 0.0 and vout.shape == v0.shape and isinstance(stats.steps, int)
    with pytest.raises(TypeError):
        ConjugateGradient()(torch.eye(hip.N, dtype=torch.float64), b, v0, NystromPreconditioner(hip))


@pytest.mark.parametrize("kernel,kind", [("rbf", "rbf"), ("Matern32", "matern32")])
def test_foreign_preconditioner_through_the_open_seam(kernel, kind):
    """`Preconditioner = Callable[[Tensor], Tuple[Tensor, Tensor]]` (conjugate_gradient.py:22): any callable works, here the identity
    (plain CG) and a Jacobi scaling, against orc.pcg with the same preconditioner on the dense operator."""
    from cglb_amd.backend.conjugate_gradient import ConjugateGradient, KernelOperator
    from cglb_amd.backend.models import LowerBoundCG
    be, model, (train, _) = _model(kernel, M=16)
    float(-LowerBoundCG(model)(None))                       # pushes the hypers and runs setup
    hyp = _hyp_of(model)
    hip = model.hip
    N = hip.N
    cov = orc.dense_cov(kind, train[0], hyp)
    b_np = train[1] - hyp.mean
    b = torch.from_numpy(b_np).to(hip.device).reshape(-1, 1)
    v0 = torch.zeros(N, 1, dtype=torch.float64, device=hip.device)
    dinv = 1.0 / np.diag(cov)
    dinv_t = torch.from_numpy(dinv).to(hip.device).reshape(-1, 1)
    cases = {
        "identity": (lambda r: (r.clone(), (r * r).sum()), lambda r: (r.copy(), float(r @ r))),
        "jacobi": (lambda r: (r * dinv_t, (r * r * dinv_t).sum()), lambda r: (r * dinv, float(r @ (r * dinv)))),
    }
    for name, (hip_pre, np_pre) in cases.items():
        for tol, max_iter, restart in ((1e-2, 100, 40), (1e-30, 7, 3)):
            vout, stats = ConjugateGradient(tol, max_iter, restart)(KernelOperator(hip), b, v0, hip_pre)
            ref_v, ref_stats = orc.pcg(lambda x: cov @ x, b_np, np.zeros(N), np_pre, tol, max_iter, restart)
            assert stats.steps == ref_stats.steps, name
            assert float(stats.residual_error) == pytest.approx(ref_stats.residual_error, rel=1e-6), name
            np.testing.assert_allclose(vout.cpu().numpy().reshape(-1), ref_v, rtol=0, atol=1e-8 * np.abs(ref_v).max(), err_msg=name)
            assert float(v0.abs().max()) == 0.0 and vout.shape == v0.shape      # the argument is not mutated (:55)


def test_lower_bound_checks_the_data_argument():
    """The reference evaluates the bound on the data it is given (models.py:151-169): here that must be the model's training set."""
    from cglb_amd.backend.models import LowerBoundCG
    be, model, (train, test) = _model("rbf", M=8)
    lb = LowerBoundCG(model)
    a = float(-lb(None))
    model.v_vec.zero_()
    assert float(-lb((train[0], train[1]))) == pytest.approx(a, rel=1e-12)
    with pytest.raises(ValueError):
        lb((train[0][:50], train[1][:50]))
    with pytest.raises(ValueError):
        lb((train[0] + 1e-3, train[1]))


def test_cli_train_and_metric_roundtrip(tmp_path):
    """Config C1 plumbing (snelson-like, Matern32, M=16, fp64) through the click command tree of cli.py:60-152."""
    from click.testing import CliRunner
    from cglb_amd.cli import main
    logdir = str(tmp_path / "run")
    r = CliRunner().invoke(main, ["-b", "hip", "-t", "fp64", "-l", logdir, "-s", "0", "train", "-d", "snelson-like", "-n", "15",
                                  "cglb", "-k", "Matern32", "-m", "cglb", "-i", "cv", "-M", "16"], catch_exceptions=False)
    assert r.exit_code == 0, r.output
    for f in ("model.json", "results.json", "logs.json"):
        assert os.path.exists(os.path.join(logdir, f))
    from cglb_amd.backend import jsonio
    results = jsonio.load(os.path.join(logdir, "results.json"))
    logs = jsonio.load(os.path.join(logdir, "logs.json"))
    assert isinstance(logs["params"][0][".kernel.lengthscales"], np.ndarray)        # arrays survive as arrays (json_tricks encoding)
    assert {"loss", "train/rmse", "test/rmse", "train/nlpd", "test/nlpd", "cg/steps", "cg/error", "id"} <= set(results)
    assert {"loss", "elapsed_time", "params", "steps-per-feval", "residual_error-per-feval", "id"} <= set(logs)
    assert results["test/rmse"] < 0.6
    r2 = CliRunner().invoke(main, ["-b", "hip", "-t", "fp64", "-l", logdir, "metric", "-d", "snelson-like", "cglb", "-k", "Matern32",
                                   "-m", "cglb", "-i", "cv", "-M", "16", "-p", os.path.join(logdir, "model.json")], catch_exceptions=False)
    assert r2.exit_code == 0, r2.output
    assert os.path.exists(os.path.join(logdir, "metric.npy"))


@pytest.mark.parametrize("kernel", ["Matern32", "rbf"])
def test_create_model_initialises_inducing_points_on_the_gpu(kernel):
    """create_model -> InducingVariableConfig.init -> cglb_select_inducing: the model's initial Z equals the numpy statement of
    the greedy conditional-variance rule under the initial kernel (unit lengthscales, variance 1; config.py:74-76)."""
    from oracle.cglb_oracle import greedy_conditional_variance
    from cglb_amd.backend.interface import _InitKernel
    be, model, (train, _) = _model(kernel, M=24)
    Z = be.model_parameters(model)[".inducing_variable.Z"]
    ref = greedy_conditional_variance(np.asarray(train[0]), 24, _InitKernel(model.covar_module.base_kernel).__call__)
    np.testing.assert_array_equal(Z, ref)


def test_tf_twin_config_semantics_max_error_vzero_and_joint_optimisation():
    """tensorflow/models.py:31-51,161-164 behind configure_backend(config_semantics="tf"): `max_error` of the config is the CG
    tolerance, `vzero` evaluates the bound at v = 0 without CG, `joint_optimization` makes v a trainable parameter whose gradient
    is d bound / d v = K w - r.  The default ("torch") ignores all three like pytorch/interface.py:315-323."""
    from cglb_amd.backend import BACKENDS, CGLBConfig, INDUCING_VARIABLE_CONFIGS, KERNEL_CONFIGS
    from cglb_amd.backend.models import LowerBoundCG
    be = BACKENDS["hip"]
    train, _ = _data()
    N = len(train[1])

    def build(semantics, **cfg_kw):
        be.configure_backend(logdir="/tmp/cglb_amd_test", keops=False, config_semantics=semantics)
        be.set_default_float("fp64")
        be.set_default_jitter("fp64")
        cfg = CGLBConfig(KERNEL_CONFIGS["rbf"](), INDUCING_VARIABLE_CONFIGS["cv"](12), **cfg_kw)
        return be.create_model(cfg, train)

    try:
        # torch semantics: the config's tolerance is ignored
        m = build("torch", max_error=1e-4)
        float(-LowerBoundCG(m)(None))
        hyp = _hyp_of(m)
        steps_default = orc.objective("rbf", train[0], train[1], hyp, np.zeros(N), True, 1.0).steps
        assert m.cg_stats.steps == steps_default
        # tf semantics: the tolerance is consumed
        m = build("tf", max_error=1e-4)
        loss = float(-LowerBoundCG(m)(None))
        ref = orc.objective("rbf", train[0], train[1], hyp, np.zeros(N), True, 1e-4)
        assert m.cg_stats.steps == ref.steps and ref.steps > steps_default
        assert loss == pytest.approx(-ref.bound, rel=1e-10)
        # vzero: no CG, bound at v = 0
        m = build("tf", vzero=True)
        loss = float(-LowerBoundCG(m)(None))
        ref0 = orc.objective("rbf", train[0], train[1], hyp, np.zeros(N), run_cg=False)
        assert loss == pytest.approx(-ref0.bound, rel=1e-11) and float(m.v_vec.abs().max()) == 0.0 and m.cg_stats is None
        # joint optimisation: v is a parameter; its gradient matches the oracle's K w - r at a non-trivial v
        m = build("tf", joint_optimization=True)
        assert isinstance(m.v_vec, torch.nn.Parameter) and any(p is m.v_vec for p in m.parameters())
        v0 = 0.3 * np.random.default_rng(0).standard_normal(N)
        with torch.no_grad():
            m.v_vec.copy_(torch.from_numpy(v0).reshape(-1, 1))
        loss = -LowerBoundCG(m)(None)
        gv, = torch.autograd.grad(loss, [m.v_vec])
        refj = orc.objective("rbf", train[0], train[1], hyp, v0, run_cg=False, with_grad=True)
        assert float(loss) == pytest.approx(-refj.bound, rel=1e-11)
        np.testing.assert_allclose(gv.cpu().numpy().reshape(-1), -refj.grad["v"], rtol=0, atol=1e-10 * np.abs(refj.grad["v"]).max())
        np.testing.assert_array_equal(m.v_vec.detach().cpu().numpy().reshape(-1), v0)      # no CG ran: v untouched
    finally:
        be.configure_backend(logdir="/tmp/cglb_amd_test", keops=False)                    # back to the default semantics


@pytest.mark.parametrize("cfg_kw", [dict(vzero=True), dict(joint_optimization=True)])
def test_optimize_under_tf_twin_semantics_without_cg(cfg_kw, tmp_path):
    """`optimize` with the TF twin's vzero / joint_optimization: CG never runs, model.cg_stats stays None and the per-evaluation log
    carries zeros (the TF optimize logs the CG statistics of a model that has none as 0, tensorflow/interface.py:296-337)."""
    from cglb_amd.backend import BACKENDS, CGLBConfig, INDUCING_VARIABLE_CONFIGS, KERNEL_CONFIGS
    from cglb_amd.backend.callbacks import Logger
    be = BACKENDS["hip"]
    train, test = _data()
    try:
        be.configure_backend(logdir="/tmp/cglb_amd_test", keops=False, config_semantics="tf")
        be.set_default_float("fp64")
        be.set_default_jitter("fp64")
        model = be.create_model(CGLBConfig(KERNEL_CONFIGS["rbf"](), INDUCING_VARIABLE_CONFIGS["cv"](12), **cfg_kw), train)
        logger = Logger(str(tmp_path), lambda: {}, lambda: be.model_parameters(model), holdout_interval=-1, include_feval_log=True, verbose=False)
        results = be.optimize(model, train, 4, logger, "scipy")
        assert model.cg_stats is None and sum(r.nit for r in results) >= 1
        assert set(logger.logs["steps-per-feval"]) == {0} and set(logger.logs["residual_error-per-feval"]) == {0.0}
        assert np.isfinite(float(model.last_bound))
    finally:
        be.configure_backend(logdir="/tmp/cglb_amd_test", keops=False)


def test_lower_bound_accepts_the_training_set_from_any_device():
    """LowerBoundCG.forward(data): the model's own training set passes whether it arrives as numpy arrays, CPU tensors or CUDA tensors
    (no device-mismatch RuntimeError), anything else is a ValueError; the full comparison runs once per data object."""
    from cglb_amd.backend.models import LowerBoundCG
    be, model, (train, test) = _model("rbf")
    lb = LowerBoundCG(model)
    ref = float(lb(None))
    xd, yd = torch.as_tensor(train[0]).cuda(), torch.as_tensor(train[1]).cuda()
    assert float(lb((xd, yd))) == pytest.approx(ref, rel=1e-12)
    assert model._accepted_data[0]() is xd
    assert float(lb((xd, yd))) == pytest.approx(ref, rel=1e-12)            # second call: identity shortcut
    assert float(lb((train[0], train[1]))) == pytest.approx(ref, rel=1e-12)
    with pytest.raises(ValueError):
        lb((xd + 1.0, yd))
    with pytest.raises(ValueError):
        lb((test[0], test[1]))
