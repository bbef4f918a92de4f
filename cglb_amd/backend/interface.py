"""Backend interface for the HIP path — mirror of cglb/backend/pytorch/interface.py restricted to what the CGLB
path needs: configure_backend, set_default_float/jitter, get_default_float(_str), create_kernel,
create_model(CGLBConfig), model_parameters, optimize (SciPy L-BFGS-B, four-round schedule :445-543), save, load,
metrics_fn (:607-658).  Exact-GP / Adam / MultiDeviceKernel branches are out of scope and raise NotImplementedError,
like the reference's unregistered singledispatch defaults (:120-147).
"""
from __future__ import annotations

import json
import os
from contextlib import contextmanager
from dataclasses import asdict
from functools import singledispatch
from pathlib import Path
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import jsonio, metric
from .callbacks import Logger
from .config import CGLBConfig, KernelConfig, Matern32Config, ModelConfig, SGPRConfig, SquaredExponentialConfig
from .models import (CGLB, GPR, BaseKernel, GaussianLikelihood, InducingPointKernel, LowerBoundCG, PredictCG, ScaleKernel,
                     get_cholesky_jitter, log_density, set_cholesky_jitter)
from .optimizer import Scipy

__all__ = ["create_kernel", "create_model", "optimize", "save", "load", "metrics_fn"]

Tensor = torch.Tensor
Data = Tuple[np.ndarray, np.ndarray]

_STATE = {"dtype": torch.float64, "logdir": None, "config_semantics": "torch"}


def configure_backend(logdir: Optional[str] = None, keops: Optional[bool] = None, config_semantics: str = "torch", **kwargs):
    """interface.py:66-88.  `keops` is accepted and ignored: the implicit K_ff mat-vec is always the HIP kernel.
    config_semantics: "torch" (default) mirrors pytorch/interface.py:315-323, which ignores `max_error` / `joint_optimization` / `vzero`
    of CGLBConfig; "tf" consumes them like the TF twin's create_model (tensorflow/interface.py:244-258, models.py:31-51,161-164)."""
    assert logdir is not None
    if config_semantics not in ("torch", "tf"):
        raise ValueError("config_semantics must be 'torch' or 'tf'")
    _STATE["logdir"] = logdir
    _STATE["config_semantics"] = config_semantics
    if not torch.cuda.is_available():
        raise RuntimeError("the hip backend needs an MI355X (HIP device); there is no CPU fallback")


def set_default_jitter(jitter):  # interface.py:90-91
    set_cholesky_jitter(jitter)


def set_default_float(float_type: str) -> None:  # interface.py:94-104
    types = {"fp32": torch.float32, "float32": torch.float32, "fp64": torch.float64, "float64": torch.float64}
    if float_type not in types:
        raise NotImplementedError(f"Unknown float type {float_type}")
    _STATE["dtype"] = types[float_type]


def get_default_float_str() -> str:  # interface.py:107-113
    return {torch.float32: "fp32", torch.float64: "fp64"}[_STATE["dtype"]]


def get_default_float() -> np.dtype:  # interface.py:116-117
    return torch.tensor(1, dtype=_STATE["dtype"]).numpy().dtype


@singledispatch
def create_model(model_cfg: ModelConfig, data: Data):
    raise NotImplementedError()


@singledispatch
def create_kernel(cfg: KernelConfig, data: Data):
    raise NotImplementedError()


@singledispatch
def optimize(model: GPR, dataset, num_steps: int, logger: Logger, optimizer: str):
    raise NotImplementedError()


@singledispatch
def save(model: GPR, logdir: str):
    raise NotImplementedError()


@singledispatch
def load(model: GPR, filepath: str):
    raise NotImplementedError()


@singledispatch
def metrics_fn(model: GPR, dataset_bundle):
    raise NotImplementedError()


def model_parameters(model) -> Dict[str, np.ndarray]:
    """interface.py:150-178 — same keys."""
    kernel = model.covar_module
    params = {
        ".likelihood.variance": _numpy(model.likelihood.noise_covar.noise)[0],
        ".mean_function.c": _numpy(model.mean_module.constant),
    }
    if isinstance(kernel, InducingPointKernel):
        params.update({".inducing_variable.Z": _numpy(kernel.inducing_points)})
        kernel = kernel.base_kernel
    params.update({
        ".kernel.lengthscales": _numpy(kernel.base_kernel.lengthscale)[0, :],
        ".kernel.variance": _numpy(kernel.outputscale).squeeze(),
    })
    return params


def _make_kernel(kind: str, cfg, data: Data) -> ScaleKernel:
    params = cfg.params(data)
    lengthscales = np.asarray(params["lengthscales"], dtype=np.float64)
    base = BaseKernel(kind, ard_num_dims=len(lengthscales))
    base.lengthscale = lengthscales
    kernel = ScaleKernel(base)
    kernel.outputscale = params["variance"]
    return kernel


@create_kernel.register
def _create_kernel_m32(cfg: Matern32Config, data: Data):  # interface.py:220-230 (registered first: subclass of SE config)
    return _make_kernel("matern32", cfg, data)


@create_kernel.register
def _create_kernel_se(cfg: SquaredExponentialConfig, data: Data):  # interface.py:207-217
    if isinstance(cfg, Matern32Config):
        return _make_kernel("matern32", cfg, data)
    return _make_kernel("rbf", cfg, data)


def _kernel_numpy(kernel: ScaleKernel, x1, x2, full_cov: bool):
    """init_kernel_fn of interface.py:278-284 (numpy closed form; only used to pick the initial inducing points)."""
    x1 = np.asarray(x1, dtype=np.float64)
    var = float(kernel.outputscale.detach())
    if not full_cov:
        return np.full(x1.shape[0], var)
    ls = _numpy(kernel.base_kernel.lengthscale).reshape(-1)
    x2 = x1 if x2 is None else np.asarray(x2, dtype=np.float64)
    a, b = x1 / ls, x2 / ls
    d2 = np.maximum((a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2.0 * a @ b.T, 0.0)
    if kernel.base_kernel.kind == "rbf":
        return var * np.exp(-0.5 * d2)
    r = np.sqrt(d2)
    return var * (1.0 + np.sqrt(3.0) * r) * np.exp(-np.sqrt(3.0) * r)


class _InitKernel:
    """The init_kernel_fn callback of interface.py:278-284, plus the HIP implementation of the greedy selection it feeds
    (config.py:62-65 -> cglb_select_inducing): the O(N M^2) pivoted Cholesky runs on the GPU, not through this callback."""

    def __init__(self, kernel: ScaleKernel):
        self.kernel = kernel

    def __call__(self, x1, x2=None, full_cov: bool = False):
        return _kernel_numpy(self.kernel, x1, x2, full_cov)

    def select_inducing(self, X: np.ndarray, num_variables: int) -> np.ndarray:
        from ..hip_context import HipContext
        X = np.asarray(X, dtype=np.float64).reshape(len(X), -1)
        ctx = HipContext(X, np.zeros(X.shape[0]), num_variables, self.kernel.base_kernel.kind, dtype=_STATE["dtype"])
        try:
            ls = _numpy(self.kernel.base_kernel.lengthscale).reshape(-1)
            indices, _ = ctx.select_inducing(ls, float(self.kernel.outputscale.detach()))
        finally:
            ctx.close()
        return X[indices].copy()


def _likelihood_and_kernel_for_sgpr(model_cfg: SGPRConfig, data: Data):
    """interface.py:263-301"""
    params = model_cfg.params(data)
    likelihood = GaussianLikelihood(lower_bound=1e-6)
    likelihood.noise = params["noise_variance"]
    base_kernel = create_kernel(model_cfg.kernel, data)

    inducing_variable = params["inducing_variable"](_InitKernel(base_kernel))
    return likelihood, InducingPointKernel(base_kernel, inducing_variable)


@create_model.register
def _create_model_cglb(model_cfg: CGLBConfig, data: Data):
    """interface.py:315-323.  Like the reference's torch path, max_error / joint_optimization / vzero of the config
    are not consumed here (the objective uses ConjugateGradient() defaults) unless the backend was configured with
    config_semantics="tf"."""
    likelihood, kernel = _likelihood_and_kernel_for_sgpr(model_cfg, data)
    extra = {}
    if _STATE["config_semantics"] == "tf":  # the TF twin's create_model hands these to the model (tensorflow/interface.py:244-258)
        extra = dict(max_error=model_cfg.max_error, joint_optimization=model_cfg.joint_optimization, vzero=model_cfg.vzero)
    model = CGLB((np.asarray(data[0]), np.asarray(data[1]).reshape(-1)), likelihood, kernel, dtype=_STATE["dtype"], **extra)
    _broadcast_parameters(model)
    return model


def _broadcast_parameters(model: CGLB):
    """N ranks: every rank built the model from the same data and config (the greedy inducing-point selection is deterministic); the
    initial parameters are broadcast from rank 0 all the same, so that the replicas start from identical bits by construction."""
    ctx = model.hip
    if getattr(ctx, "world", 1) <= 1:
        return
    import torch.distributed as dist
    group = getattr(ctx, "group", None)
    src = dist.get_global_rank(group, 0) if group is not None else 0
    with torch.no_grad():
        for p in model.parameters():
            buf = p.detach().to(ctx.device).contiguous()
            dist.broadcast(buf, src=src, group=group)
            p.copy_(buf.to(p.device))


@contextmanager
def _narrow_host_pools():
    """The optimiser's host side is a handful of small vectors (M D + D + 3 numbers), all heavy work runs on the GPU.  With their default
    widths (one thread per visible core: 128 on a box that grants this process 16) the OpenMP / BLAS pools of torch and numpy spin between
    calls and starve the HIP runtime's own threads: at N = 57k, D = 27 kernel launches stalled for ~70 ms at a time and an evaluation took
    99 ms of wall time for 58 ms of GPU work (47 ms with narrow pools).  CGLB_HOST_THREADS overrides the width (default 4; 0 = leave alone)."""
    n = int(os.environ.get("CGLB_HOST_THREADS", "4"))
    if n <= 0:
        yield
        return
    prev = torch.get_num_threads()
    torch.set_num_threads(min(prev, n))
    try:
        try:
            from threadpoolctl import threadpool_info, threadpool_limits
        except Exception:  # pragma: no cover
            threadpool_limits = None
        if threadpool_limits is None:
            yield
        else:
            # only ever NARROW a pool: a launcher may have set OMP_NUM_THREADS=1 (torch.distributed.run does), and widening a BLAS pool
            # beyond the width it was initialised with crashed SciPy's L-BFGS-B core (SIGSEGV inside _lbfgsb.setulb on both ranks of the
            # first CLI run under torch.distributed.run, round 3)
            limits = {}
            for info in threadpool_info():
                api, cur = info.get("user_api"), int(info.get("num_threads") or 1)
                if api and cur > n:
                    limits[api] = n
            if limits:
                with threadpool_limits(limits=limits):
                    yield
            else:
                yield
    finally:
        torch.set_num_threads(prev)


@optimize.register
def _optimize_cglb(model: CGLB, dataset, num_steps: int, logger: Logger, optimize: str = "scipy"):
    with _narrow_host_pools():
        return _optimize_cglb_impl(model, dataset, num_steps, logger, optimize)


def _assert_ranks_agree(model: CGLB, what: str):
    """N ranks each run the reference's single-process optimiser on what must be identical (loss, gradient) sequences: the loss of the
    last evaluation and a checksum of the parameters are all-gathered once per round; a rank that drifted raises on EVERY rank (all
    see the same gathered numbers), instead of the job running on with replicas that no longer describe one model."""
    ctx = getattr(model, "hip", None)
    if getattr(ctx, "world", 1) <= 1:
        return
    import torch.distributed as dist
    digest = 0.0
    for p in model.parameters():
        digest += float(p.detach().double().abs().sum())
    mine = torch.tensor([float(model.last_bound), digest], dtype=torch.float64, device=ctx.device)
    group = getattr(ctx, "group", None)
    if group is None and getattr(ctx, "comm", None) is not None:
        group = ctx.comm.group
    gathered = [torch.empty_like(mine) for _ in range(ctx.world)]
    dist.all_gather(gathered, mine, group=group)
    vals = torch.stack(gathered).cpu().numpy()
    if not (np.all(vals[:, 0] == vals[0, 0]) and np.all(vals[:, 1] == vals[0, 1])):
        raise RuntimeError(f"ranks disagree {what}: (bound, parameter checksum) per rank = {vals.tolist()}")


def _optimize_cglb_impl(model: CGLB, dataset, num_steps: int, logger: Logger, optimize: str = "scipy"):
    """interface.py:445-543: warm-up evaluation outside the clock, then up to four L-BFGS-B rounds, the last two
    without the inducing points."""
    assert optimize == "scipy"
    lbfgs = Scipy()
    lower_bound = LowerBoundCG(model)
    results = []

    def lbfgs_closure() -> Tensor:
        loss = -lower_bound(None)
        stats = model.cg_stats                              # None when CG never ran (TF-twin vzero / joint_optimization)
        # steps-per-feval / residual_error-per-feval (:476); without CG the TF optimize logs zeros (tensorflow/interface.py:296-337)
        logger.log_for_feval(**(asdict(stats) if stats is not None else dict(steps=0, residual_error=0.0)))
        return loss

    def step_callback(*args):
        lower_bound.cached_v_vec = False                    # :480
        logger(*args)

    def optimize_fn(params, maxiter: int, ftol: float = 0.0, gtol: float = 0.0, disp: bool = False):
        options = dict(maxiter=maxiter, ftol=ftol, gtol=gtol, disp=disp)
        return lbfgs.minimize(lbfgs_closure, params, options=options, step_callback=step_callback)

    params = list(model.parameters())
    with logger.no_recording():                             # :494-501
        _loss = lbfgs_closure()
        _grads = torch.autograd.grad(_loss, params)
        if torch.cuda.is_available():                       # :499-501
            torch.cuda.synchronize()
    logger.timer.reset()
    logger.timer.start()

    remaining = num_steps
    for round_id in range(4):                               # :507-543
        if remaining <= 0:
            break
        if round_id == 2:
            ips = model.covar_module.inducing_points
            params = [p for p in model.parameters() if id(p) != id(ips)]
        result = optimize_fn(params, remaining)
        remaining -= result.nit
        results.append(result)
        _assert_ranks_agree(model, f"after optimisation round {round_id}")
    return results


@save.register
def _save(model: GPR, logdir: str):  # interface.py:546-551: json_tricks.dump(model_parameters(model)) -> same encoding (jsonio.py)
    os.makedirs(logdir, exist_ok=True)
    params = model_parameters(model)
    with open(Path(logdir, "model.json"), "w") as file:
        jsonio.dump(params, file)


@load.register
def _load(model: GPR, filepath: str):
    """Reads a model.json written by `save` (the reference's torch `load` expects a state_dict and is asymmetric
    with its own `save`; here the pair round-trips)."""
    params = jsonio.load(filepath)   # decodes the __ndarray__ objects json_tricks / `save` write; plain lists work too
    model.likelihood.noise = params[".likelihood.variance"]
    with torch.no_grad():
        model.mean_module.constant.copy_(torch.as_tensor(np.asarray(params[".mean_function.c"]), dtype=torch.float64).reshape(()))
        model.covar_module.inducing_points.copy_(torch.as_tensor(np.asarray(params[".inducing_variable.Z"]), dtype=torch.float64))
    model.covar_module.base_kernel.base_kernel.lengthscale = params[".kernel.lengthscales"]
    model.covar_module.base_kernel.outputscale = params[".kernel.variance"]
    return model


@metrics_fn.register
def _compute_metrics_cglb(model: CGLB, dataset_bundle):
    """interface.py:607-658"""

    def cglb_cg_params():
        if model.cg_stats is not None:
            return {"cg/steps": _numpy(model.cg_stats.steps), "cg/error": _numpy(model.cg_stats.residual_error)}
        return {}

    train, test = dataset_bundle

    def cglb_metrics():
        with torch.no_grad():
            lower_bound = LowerBoundCG(model, use_cache=True, cached_v_vec_initial=True)  # no CG: reuse model.v_vec (:619-625)
            loss = -lower_bound(None)
            return dict(loss=_numpy(loss))

    x_full = np.concatenate([np.asarray(train[0]), np.asarray(test[0])], axis=0)
    y_full = np.concatenate([np.asarray(train[1]).reshape(-1), np.asarray(test[1]).reshape(-1)], axis=0).reshape(-1, 1)
    total = int(x_full.shape[0])

    def error_and_logdensity():
        predict_f = PredictCG(model)
        lpds, errs = [], []
        max_batch = int(1e6)
        with torch.no_grad():
            for i in range(0, total, max_batch):
                f_mean, f_var = predict_f(torch.as_tensor(x_full[i: i + max_batch]))
                y_batch = torch.as_tensor(y_full[i: i + max_batch], dtype=f_mean.dtype, device=f_mean.device)
                lpds.append(_numpy(log_density(model, y_batch, f_mean, f_var)))
                errs.append(_numpy(y_batch - f_mean))
        err, lpd = np.concatenate(errs, axis=0), np.concatenate(lpds, axis=0)
        n = np.asarray(train[0]).shape[0]
        return (err[:n], err[n:]), (lpd[:n], lpd[n:])

    rmse_lpd_metrics = metric.rmse_and_lpd_fn(error_and_logdensity)
    return lambda: metric.call_metric_fns(cglb_cg_params, cglb_metrics, rmse_lpd_metrics)


def _numpy(tensor) -> np.ndarray:
    if isinstance(tensor, torch.Tensor):
        return tensor.detach().cpu().numpy()
    return np.asarray(tensor)
