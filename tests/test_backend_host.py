"""Host-side logic of the cglb.backend mirror that needs no GPU: config registries, inducing-point init,
SciPy pack/unpack/assign, constraint transforms, logger keys."""
import numpy as np
import pytest
import torch

from cglb_amd.backend import config as cfg
from cglb_amd.backend.callbacks import Logger, StopWatch
from cglb_amd.backend.models import BaseKernel, GaussianLikelihood, ScaleKernel
from cglb_amd.backend.optimizer import Scipy
from oracle import cglb_oracle as orc


def test_registries_match_reference_keys():
    assert set(cfg.SGPR_CONFIGS) == {"sgpr", "cglb", "sgprn2m", "cglbn2m", "cglbnm2"}          # config.py:144-150
    assert set(cfg.KERNEL_CONFIGS) == {"SquaredExponential", "Matern32", "mat32", "rbf"}         # config.py:152-158
    assert set(cfg.INDUCING_VARIABLE_CONFIGS) == {"InducingVariable", "ConditionalVariance", "iv", "cv"}
    c = cfg.CGLBConfig(kernel=cfg.Matern32Config(), inducing_variable=cfg.InducingVariableConfig(16))
    assert (c.max_error, c.joint_optimization, c.vzero) == (1.0, False, False)                   # config.py:110-114
    data = (np.zeros((5, 3)), np.zeros(5))
    p = c.params(data)
    assert p["noise_variance"] == 1.0 and callable(p["inducing_variable"])                       # config.py:102-107
    assert cfg.Matern32Config().params(data)["lengthscales"].tolist() == [1.0, 1.0, 1.0]          # config.py:74-76
    with pytest.raises(Exception):
        c.max_error = 2.0  # frozen dataclass


def test_greedy_conditional_variance_is_pivoted_cholesky():
    rng = np.random.default_rng(0)
    X = rng.standard_normal((200, 2))

    def kfn(x1, x2=None, full_cov=False):
        if not full_cov:
            return np.ones(len(x1))
        return orc.kernel_matrix("matern32", x1, x1 if x2 is None else x2, np.ones(2), 1.0)

    Z = orc.greedy_conditional_variance(X, 12, kfn)
    K = orc.kernel_matrix("matern32", X, X, np.ones(2), 1.0)
    idx = [int(np.where((X == z).all(1))[0][0]) for z in Z]
    assert len(set(idx)) == 12
    for m in range(1, 12):
        S = idx[:m]
        cond = np.diag(K) - np.einsum("ij,ij->j", K[S], np.linalg.solve(K[np.ix_(S, S)] + 1e-12 * np.eye(m), K[S]))
        assert cond[idx[m]] >= cond.max() * (1 - 1e-9)


def test_scipy_pack_unpack_assign_roundtrip():
    a = torch.nn.Parameter(torch.arange(6, dtype=torch.float64).reshape(2, 3))
    b = torch.nn.Parameter(torch.tensor(7.0, dtype=torch.float64))
    packed = Scipy.pack((a, b))
    assert packed.tolist() == [0, 1, 2, 3, 4, 5, 7]
    vals = Scipy.unpack((a, b), packed * 2)
    Scipy.assign((a, b), vals)
    assert a.data.tolist() == [[0, 2, 4], [6, 8, 10]] and float(b.data) == 14.0
    with pytest.raises(ValueError):
        Scipy.assign((a, b), vals[:1])
    # minimize a quadratic through the same (loss, flat grad) contract as optimizer.py:41-46
    res = Scipy().minimize(lambda: ((a - 1.0) ** 2).sum() + (b + 2.0) ** 2, [a, b], options=dict(maxiter=50))
    assert res.success and np.allclose(a.data.numpy(), 1.0, atol=1e-5) and abs(float(b.data) + 2.0) < 1e-5


def test_constraints_match_gpytorch_conventions():
    lik = GaussianLikelihood(lower_bound=1e-6)          # GreaterThan(1e-6), interface.py:269
    lik.noise = 1.0
    assert float(lik.noise) == pytest.approx(1.0, rel=1e-12)
    assert float(torch.nn.functional.softplus(lik.noise_covar._noise.raw) + 1e-6) == pytest.approx(1.0, rel=1e-12)
    with pytest.raises(ValueError):
        lik.noise = 1e-7
    k = ScaleKernel(BaseKernel("rbf", 3))
    k.base_kernel.lengthscale = np.array([0.5, 1.0, 2.0])
    k.outputscale = 0.3
    assert k.base_kernel.lengthscale.shape == (1, 3)
    np.testing.assert_allclose(k.base_kernel.lengthscale.detach().numpy()[0], [0.5, 1.0, 2.0], rtol=1e-12)
    assert float(k.outputscale) == pytest.approx(0.3, rel=1e-12)


def test_logger_keys_and_stopwatch():
    calls = []
    lg = Logger("/tmp/x", lambda: {"loss": 1.5, "cg/steps": 3, "junk": 0, "train/rmse": 0.1}, lambda: {".kernel.variance": 1.0, ".inducing_point": 2},
                holdout_interval=2, include_feval_log=True, verbose=False)
    lg.timer.start()
    with lg.no_recording():
        lg(0)
        lg.log_for_feval(steps=1)
    assert lg.logs == {} and lg.counter == 1
    lg.counter = 0
    for i in range(4):
        lg.log_for_feval(steps=i, residual_error=0.1)
        lg(i)
    assert lg.logs["iteration"] == [0, 2]
    assert set(lg.logs) >= {"iteration", "elapsed_time", "params", "loss", "cg/steps", "train/rmse", "steps-per-feval", "residual_error-per-feval"}
    assert "junk" not in lg.logs and ".inducing_point" not in lg.logs["params"][0]
    sw = StopWatch()
    assert not sw.started()
    sw.start(); sw.pause(); sw.resume()
    assert sw.stop() >= 0.0 and not sw.started()


def test_inducing_init_dispatches_to_an_accelerated_selection_when_the_callback_offers_one():
    """InducingVariableConfig.init(data, kernel_fn): the kernel callback's `select_inducing` (the hip backend's runs
    cglb_select_inducing on the GPU) does the selection; a plain callable is rejected - the product has no CPU fallback."""
    from cglb_amd.backend.config import InducingVariableConfig
    rng = np.random.default_rng(0)
    X = rng.standard_normal((60, 2))
    y = rng.standard_normal(60)

    def plain(x1, x2=None, full_cov=False):
        x1 = np.asarray(x1)
        if not full_cov:
            return np.ones(len(x1))
        x2 = x1 if x2 is None else np.asarray(x2)
        return np.exp(-0.5 * ((x1[:, None, :] - x2[None, :, :]) ** 2).sum(-1))

    cfg = InducingVariableConfig(7)
    with pytest.raises(TypeError):
        cfg.init((X, y), plain)

    class Accelerated:
        calls = []

        def __call__(self, *a, **k):
            raise AssertionError("the numpy path must not be used when select_inducing is offered")

        def select_inducing(self, X, M):
            self.calls.append((X.shape, M))
            return X[:M] + 100.0

    acc = Accelerated()
    Z_acc = cfg.init((X, y), acc)
    assert acc.calls == [((60, 2), 7)]
    np.testing.assert_array_equal(Z_acc, X[:7] + 100.0)


def test_init_kernel_callback_matches_the_oracle_kernels():
    """The numpy closed form handed to a generic inducing-point initialiser (interface.py:278-284) against the oracle."""
    from oracle import cglb_oracle as orc
    from cglb_amd.backend.interface import _InitKernel
    from cglb_amd.backend.models import BaseKernel, ScaleKernel
    rng = np.random.default_rng(1)
    X1, X2 = rng.standard_normal((9, 3)), rng.standard_normal((5, 3))
    ls = np.array([0.7, 1.3, 2.0])
    for kind in ("rbf", "matern32"):
        base = BaseKernel(kind, ard_num_dims=3)
        base.lengthscale = ls
        k = ScaleKernel(base)
        k.outputscale = 1.9
        fn = _InitKernel(k)
        np.testing.assert_allclose(fn(X1, X2, full_cov=True), orc.kernel_matrix(kind, X1, X2, ls, 1.9), rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(fn(X1), np.full(9, 1.9))
        assert callable(fn.select_inducing)


def test_json_artefacts_use_the_json_tricks_ndarray_encoding(tmp_path):
    """model.json / results.json / logs.json as a json_tricks.load-shaped reader sees them.  Schema derived from the reference's
    writers: model_parameters (pytorch/interface.py:150-178: `.likelihood.variance` a numpy scalar, `.mean_function.c` and
    `.kernel.variance` 0-d arrays, `.kernel.lengthscales` [D], `.inducing_variable.Z` [M, D]) dumped by json_tricks (:546-551);
    Logger.logs (callbacks.py:107-125,139-178: lists per key, `params` a list of dicts of arrays) dumped at cli.py:100-109."""
    import json
    from cglb_amd.backend import jsonio
    params = {".likelihood.variance": np.float64(0.25), ".mean_function.c": np.array(0.5), ".inducing_variable.Z": np.arange(6.0).reshape(3, 2),
              ".kernel.lengthscales": np.array([1.0, 2.0]), ".kernel.variance": np.array(1.5)}
    path = tmp_path / "model.json"
    with open(path, "w") as f:
        jsonio.dump(params, f)
    raw = json.load(open(path))
    assert raw[".likelihood.variance"] == 0.25                                             # numpy scalar -> plain number
    assert raw[".inducing_variable.Z"] == {"__ndarray__": [[0.0, 1.0], [2.0, 3.0], [4.0, 5.0]], "dtype": "float64", "shape": [3, 2], "Corder": True}
    assert raw[".kernel.lengthscales"] == {"__ndarray__": [1.0, 2.0], "dtype": "float64", "shape": [2]}
    assert raw[".mean_function.c"] == {"__ndarray__": 0.5, "dtype": "float64", "shape": []}
    back = jsonio.load(path)
    assert set(back) == set(params)
    for k, v in params.items():
        got = back[k]
        assert np.shape(got) == np.shape(v) and np.allclose(got, v)
        if isinstance(v, np.ndarray):
            assert isinstance(got, np.ndarray) and got.dtype == v.dtype
    logs = {"loss": [np.float64(3.0), np.float64(2.0)], "iteration": [0, 20], "elapsed_time": [0.1, 0.2], "cg/steps": [np.array(5), np.array(3)],
            "params": [{".kernel.lengthscales": np.array([1.0, 2.0])}, {".kernel.lengthscales": np.array([1.1, 2.1])}],
            "steps-per-feval": [5, 3, 0], "id": "run-1"}
    back = jsonio.loads(jsonio.dumps(logs))
    assert back["loss"] == [3.0, 2.0] and back["id"] == "run-1" and back["steps-per-feval"] == [5, 3, 0]
    assert isinstance(back["params"][1][".kernel.lengthscales"], np.ndarray) and back["params"][1][".kernel.lengthscales"].tolist() == [1.1, 2.1]
    assert isinstance(back["cg/steps"][0], np.ndarray) and back["cg/steps"][0].shape == ()


def test_optimize_schedule_four_rounds_without_inducing_points_in_the_last_two(monkeypatch, tmp_path):
    """pytorch/interface.py:445-543 on the host, with the solver and SciPy replaced by recorders: a warm-up evaluation outside the
    recording, then up to four `minimize` rounds, each with maxiter = the steps still left, the third and fourth without the
    inducing points in the variable list (:527-529), the step callback clearing the cache flag and feeding the logger (:479-481)."""
    from types import SimpleNamespace
    from cglb_amd.backend import interface
    from cglb_amd.backend.callbacks import Logger
    from cglb_amd.backend.conjugate_gradient import ConjugateGradientStats

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.noise = torch.nn.Parameter(torch.zeros(1, dtype=torch.float64))
            self.covar_module = torch.nn.Module()
            self.covar_module.inducing_points = torch.nn.Parameter(torch.zeros(4, 2, dtype=torch.float64))
            self.ls = torch.nn.Parameter(torch.zeros(2, dtype=torch.float64))
            self.cg_stats = ConjugateGradientStats(3, 0.5)

    class FakeBound:
        instances = []

        def __init__(self, model):
            self.model, self.cached_v_vec, self.calls = model, True, 0
            FakeBound.instances.append(self)

        def __call__(self, data):
            self.calls += 1
            return -sum((p ** 2).sum() for p in self.model.parameters()) - 1.0

    rounds = []

    class FakeScipy:
        nits = iter([3, 2, 4, 100])

        def minimize(self, closure, variables, options=None, step_callback=None, **kw):
            nit = min(next(FakeScipy.nits), options["maxiter"])
            rounds.append(dict(ids=[id(v) for v in variables], maxiter=options["maxiter"], ftol=options["ftol"], gtol=options["gtol"]))
            for k in range(nit):
                closure()
                step_callback(k, variables, [v.detach() for v in variables])
            return SimpleNamespace(nit=nit, nfev=nit, fun=0.0, status=0)

    monkeypatch.setattr(interface, "LowerBoundCG", FakeBound)
    monkeypatch.setattr(interface, "Scipy", FakeScipy)
    model = Model()
    logger = Logger(str(tmp_path), lambda: {"loss": 1.0}, lambda: {}, holdout_interval=1, include_feval_log=True, verbose=False)
    results = interface._optimize_cglb(model, None, 20, logger, "scipy")
    ips = id(model.covar_module.inducing_points)
    assert [r.nit for r in results] == [3, 2, 4, 11]                      # never more than four rounds; the last takes what is left
    assert [r["maxiter"] for r in rounds] == [20, 17, 15, 11]
    assert all(r["ftol"] == 0.0 and r["gtol"] == 0.0 for r in rounds)
    assert ips in rounds[0]["ids"] and ips in rounds[1]["ids"] and ips not in rounds[2]["ids"] and ips not in rounds[3]["ids"]
    assert len(rounds[2]["ids"]) == len(rounds[0]["ids"]) - 1
    bound = FakeBound.instances[-1]
    assert bound.calls == 1 + 20 and bound.cached_v_vec is False            # warm-up + one per closure call; flag reset by the callback
    assert len(logger.logs["steps-per-feval"]) == 20                        # the warm-up evaluation is not recorded (:494-501)
    assert len(logger.logs["loss"]) == 20                                   # holdout_interval = 1: metrics at every accepted step
    # the schedule stops as soon as the budget is used up
    FakeScipy.nits = iter([20])
    rounds.clear()
    assert [r.nit for r in interface._optimize_cglb(Model(), None, 20, logger, "scipy")] == [20] and len(rounds) == 1


def test_optimize_narrows_and_restores_the_host_thread_pools(monkeypatch):
    """`optimize` runs with narrow OpenMP/BLAS pools (idle workers of wide pools starve the HIP runtime's threads) and restores the
    previous width afterwards; CGLB_HOST_THREADS=0 leaves the pools alone."""
    import torch
    from cglb_amd.backend.interface import _narrow_host_pools
    before = torch.get_num_threads()
    monkeypatch.setenv("CGLB_HOST_THREADS", "2")
    with _narrow_host_pools():
        assert torch.get_num_threads() == min(before, 2)
    assert torch.get_num_threads() == before
    monkeypatch.setenv("CGLB_HOST_THREADS", "0")
    with _narrow_host_pools():
        assert torch.get_num_threads() == before
    assert torch.get_num_threads() == before
