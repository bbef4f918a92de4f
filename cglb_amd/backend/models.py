"""Model and objective classes — mirror of the reference's cglb/backend/pytorch/models.py for the CGLB path.

Same class names and call contracts (`CGLB`, `CommonTerms`-free `LowerBoundCG`, `PredictCG`, `PredictLogdensityCG`,
`log_density`, `gaussian`); GPyTorch's module tree (likelihood / mean_module / covar_module with raw, softplus-
constrained parameters) is restated with plain torch.nn modules so that `model.parameters()`, the parameter
dictionary keys (interface.model_parameters) and `torch.autograd.grad(loss, params)` (optimizer.py:95-98) behave
the same.  All numerical work is done by libcglb_hip.so through `HipContext`; the bound is exposed to autograd
by a `torch.autograd.Function` whose backward returns the analytic gradient computed on the GPU (v detached,
models.py:257-274).
"""
from __future__ import annotations

import math
import weakref
from dataclasses import dataclass
from typing import Optional, Tuple, Union

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from ..dist_context import make_context
from ..hip_context import HipContext
from .conjugate_gradient import ConjugateGradient, ConjugateGradientStats, KernelOperator, NystromPreconditioner

Tensor = torch.Tensor
GenericTensor = Union[np.ndarray, Tensor]
Data = Tuple[GenericTensor, GenericTensor]


def _inv_softplus(x: Tensor) -> Tensor:
    return x + torch.log(-torch.expm1(-x))


class _Constrained(nn.Module):
    """value = softplus(raw) + lower_bound  (gpytorch Positive / GreaterThan constraints)."""

    def __init__(self, shape, lower_bound: float = 0.0):
        super().__init__()
        self.lower_bound = float(lower_bound)
        self.raw = nn.Parameter(torch.zeros(shape, dtype=torch.float64))

    @property
    def value(self) -> Tensor:
        return F.softplus(self.raw) + self.lower_bound

    def set(self, value):
        v = torch.as_tensor(value, dtype=torch.float64).reshape(self.raw.shape) - self.lower_bound
        if (v <= 0).any():
            raise ValueError(f"value must exceed the lower bound {self.lower_bound}")
        with torch.no_grad():
            self.raw.copy_(_inv_softplus(v))


class BaseKernel(nn.Module):
    """RBFKernel / MaternKernel(nu=1.5) with ARD lengthscales (pytorch/interface.py:207-230)."""

    def __init__(self, kind: str, ard_num_dims: int):
        super().__init__()
        self.kind = kind  # "rbf" | "matern32"
        self._lengthscale = _Constrained((1, ard_num_dims))

    @property
    def lengthscale(self) -> Tensor:
        return self._lengthscale.value

    @lengthscale.setter
    def lengthscale(self, value):
        self._lengthscale.set(value)


class ScaleKernel(nn.Module):
    def __init__(self, base_kernel: BaseKernel):
        super().__init__()
        self.base_kernel = base_kernel
        self._outputscale = _Constrained(())

    @property
    def outputscale(self) -> Tensor:
        return self._outputscale.value

    @outputscale.setter
    def outputscale(self, value):
        self._outputscale.set(value)


class InducingPointKernel(nn.Module):
    """gpytorch.kernels.InducingPointKernel stand-in: base kernel + trainable inducing points (interface.py:297-299)."""

    def __init__(self, base_kernel: ScaleKernel, inducing_points):
        super().__init__()
        self.base_kernel = base_kernel
        self.inducing_points = nn.Parameter(torch.as_tensor(inducing_points, dtype=torch.float64).clone())


class _NoiseCovar(nn.Module):
    def __init__(self, lower_bound):
        super().__init__()
        self._noise = _Constrained((1,), lower_bound)

    @property
    def noise(self) -> Tensor:
        return self._noise.value


class GaussianLikelihood(nn.Module):
    """noise >= 1e-6 for SGPR/CGLB (pytorch/interface.py:269-273)."""

    def __init__(self, lower_bound: float = 1e-6):
        super().__init__()
        self.noise_covar = _NoiseCovar(lower_bound)

    @property
    def noise(self) -> Tensor:
        return self.noise_covar.noise

    @noise.setter
    def noise(self, value):
        self.noise_covar._noise.set(value)


class ConstantMean(nn.Module):
    def __init__(self):
        super().__init__()
        self.constant = nn.Parameter(torch.zeros((), dtype=torch.float64))


class GPR(nn.Module):
    """models.py:38-47"""

    def __init__(self, data: Data, likelihood: GaussianLikelihood, kernel: nn.Module):
        super().__init__()
        x = torch.as_tensor(data[0])
        self.train_inputs = (x.reshape(x.shape[0], -1),)
        self.train_targets = torch.as_tensor(data[1]).reshape(-1)
        self.likelihood = likelihood
        self.mean_module = ConstantMean()
        self.covar_module = kernel


class SGPR(GPR):
    ...


class CGLB(SGPR):
    """models.py:54-87: SGPR + the persistent warm-start vector v_vec (zeros[N,1], no grad) and cg_stats."""

    def __init__(self, data: Data, likelihood: GaussianLikelihood, kernel: InducingPointKernel, dtype: torch.dtype = torch.float64,
                 device: Optional[torch.device] = None, max_error: Optional[float] = None, joint_optimization: bool = False,
                 vzero: bool = False, context=None):
        """`context`: the engine behind the model - by default `make_context` builds it: a `HipContext` in a single-process run, one rank
        of a `DistHipContext` (rows of K_ff dealt over the ranks, collectives inside libcglb_hip.so) when a torch.distributed process
        group with more than one rank is initialised; every rank then holds the full replicated `v_vec` and evaluates identical
        (loss, gradient) pairs.  Tests inject `distributed.PyDistContext` here.
        The three arguments before it are the TF twin's (tensorflow/models.py:31-51): the torch reference never passes them
        (pytorch/interface.py:315-323) and neither does `create_model` unless `configure_backend(config_semantics="tf")`.
        `max_error` becomes the default tolerance of `LowerBoundCG`; `joint_optimization` (without `vzero`) makes `v_vec` a trainable
        parameter and skips CG; `vzero` keeps v = 0 and skips CG (tensorflow/models.py:161-164)."""
        super().__init__(data, likelihood, kernel)
        self.dtype = dtype
        self.max_error, self.joint_optimization, self.vzero = max_error, bool(joint_optimization), bool(vzero)
        kind = kernel.base_kernel.base_kernel.kind
        if context is None:
            context = make_context(self.train_inputs[0], self.train_targets, kernel.inducing_points.shape[0], kind, dtype=dtype, device=device)
        self.hip = context
        if getattr(context, "world", 1) > 1 and self.joint_optimization and not self.vzero:
            raise NotImplementedError("joint optimisation of v (the TF twin's opt-in) is not available on more than one rank")
        v0 = self._build_v_vec()
        # v0 trainable only under joint optimisation (tensorflow/models.py:47-48); otherwise a plain buffer without grad (models.py:59-68)
        self._v_vec = nn.Parameter(v0) if (self.joint_optimization and not self.vzero) else v0
        self._hyper_token = None

    def check_same_data(self, data: Data) -> None:
        """ValueError unless (x, y) is the training set the HIP context was built on (same shapes and content).  The full comparison
        runs once per data object: the last accepted pair is remembered (weakly), so an
        optimiser that hands over the same arrays at every evaluation does not pay N*D comparisons inside its loop."""
        x, y = data
        if x is self.train_inputs[0] and y is self.train_targets:
            return
        accepted = getattr(self, "_accepted_data", None)   # weak references to the last pair that passed the full comparison
        if accepted is not None and accepted[0]() is x and accepted[1]() is y:
            return
        x_in, y_in = x, y
        # everything on the host in fp64, whatever device the model or the caller keep their tensors on
        x = torch.as_tensor(x).detach().cpu().to(torch.float64)
        y = torch.as_tensor(y).detach().cpu().to(torch.float64).reshape(-1)
        x = x.reshape(x.shape[0], -1) if x.ndim else x.reshape(1, 1)
        tx = self.train_inputs[0].detach().cpu().to(torch.float64)
        ty = self.train_targets.detach().cpu().to(torch.float64).reshape(-1)
        if x.shape != tx.shape or y.shape != ty.shape:
            raise ValueError(f"LowerBoundCG was given data of shape {tuple(x.shape)}/{tuple(y.shape)} but the model holds the training set "
                             f"{tuple(tx.shape)}/{tuple(ty.shape)}: the bound is only defined on the model's own training data")
        if not (torch.equal(y, ty) and torch.equal(x, tx)):
            raise ValueError("LowerBoundCG was given data that differs from the model's training set")
        try:
            self._accepted_data = (weakref.ref(x_in), weakref.ref(y_in))
        except TypeError:  # lists / scalars cannot be weakly referenced: compared in full every time
            self._accepted_data = None

    def _build_v_vec(self) -> Tensor:  # models.py:59-68
        return torch.zeros((self.hip.N, 1), dtype=self.dtype, device=self.hip.device, requires_grad=False)

    @property
    def v_vec(self) -> Tensor:
        return self._v_vec

    @property
    def cg_stats(self) -> Optional[ConjugateGradientStats]:
        return getattr(self, "_cg_stats", None)

    @cg_stats.setter
    def cg_stats(self, value: ConjugateGradientStats):  # models.py:80-87
        steps, error = value.steps, value.residual_error
        if isinstance(steps, torch.Tensor):
            steps = steps.detach().cpu().numpy()
        if isinstance(error, torch.Tensor):
            error = error.detach().cpu().numpy()
        self._cg_stats = ConjugateGradientStats(steps, error)

    # constrained hyper-parameters, as tensors attached to the raw parameters
    def hyper_tensors(self):
        k = self.covar_module
        return (k.base_kernel.base_kernel.lengthscale.reshape(-1), k.base_kernel.outputscale.reshape(()),
                self.likelihood.noise.reshape(()), self.mean_module.constant.reshape(()), k.inducing_points)

    def push_hypers(self, jitter: float):
        ls, var, noise, mean, Z = [t.detach() for t in self.hyper_tensors()]
        self.hip.set_hypers(ls.cpu().numpy(), float(var), float(noise), float(mean), Z.cpu(), jitter)


@dataclass
class Bounds:
    upper_bound: Tensor
    lower_bound: Tensor


_DEFAULT_JITTER = {"value": 1e-6}


def set_cholesky_jitter(value: float):
    _DEFAULT_JITTER["value"] = float(value)


def get_cholesky_jitter() -> float:
    return _DEFAULT_JITTER["value"]


class _BoundFunction(torch.autograd.Function):
    """bound(lengthscales, variance, noise, mean, Z) with the analytic gradient from the GPU (row G of SURVEY 8a)."""

    @staticmethod
    def forward(ctx, owner, ls, var, noise, mean, Z, v_param=None):
        model = owner.model
        hip = model.hip
        hip.set_hypers(ls.detach().cpu().numpy(), float(var), float(noise), float(mean), Z.detach().cpu(), get_cholesky_jitter())
        need_grad = any(ctx.needs_input_grad[1:])
        v = model.v_vec.detach().reshape(-1)
        run_cg = not (owner._use_cache and owner.cached_v_vec)          # models.py:263
        if model.joint_optimization or model.vzero:                     # tensorflow/models.py:161-164: v0 is used as it stands
            run_cg = False
        cg = owner.cg_opt
        if run_cg and type(cg) is not ConjugateGradient:
            # foreign plug-in solver through the seam (models.py:266-271): cg_opt(A, b, v, precond)
            hip.setup()
            err = (hip.y - float(mean)).reshape(-1, 1)
            new_v, stats = cg(KernelOperator(hip), err, model.v_vec, NystromPreconditioner(hip))
            model.cg_stats = stats
            model.v_vec.data.copy_(new_v.reshape(model.v_vec.shape))
            res = hip.objective_and_grad(v, False, with_grad=need_grad)
        else:
            res = hip.objective_and_grad(v, run_cg, cg.max_error, cg.max_cg_iter, cg.restart_cg_iter, with_grad=need_grad)
            if run_cg:
                model.cg_stats = ConjugateGradientStats(res.steps, torch.tensor(res.residual_error, dtype=torch.float64))
        if run_cg:
            owner.cached_v_vec = owner._use_cache                       # models.py:278
        owner.last_bounds = Bounds(upper_bound=torch.tensor(-res.upper), lower_bound=torch.tensor(-res.lower))  # models.py:286
        model.last_bound = float(res.bound)  # value of the most recent evaluation (diagnostics / tests)
        ctx.grads = res.grad
        ctx.grad_v = None
        if v_param is not None and ctx.needs_input_grad[6]:            # joint optimisation: d bound / d v = K w - r, one more mat-vec
            ctx.grad_v = hip.objective_grad_v().reshape(v_param.shape)
        return torch.tensor(res.bound, dtype=torch.float64)

    @staticmethod
    def backward(ctx, gout):
        g = ctx.grads
        if g is None:
            raise RuntimeError("gradient was not requested in forward")
        gout = gout.to(torch.float64)
        gv = None if ctx.grad_v is None else gout.to(ctx.grad_v.device) * ctx.grad_v
        return (None, gout * torch.from_numpy(g["lengthscales"]), gout * g["variance"], gout * g["noise"], gout * g["mean"],
                gout * torch.from_numpy(g["Z"]), gv)


class LowerBoundCG(nn.Module):
    """models.py:104-286.  `LowerBoundCG(model)(data)` returns the lower bound on the log marginal likelihood."""

    def __init__(self, model: SGPR, cg_opt: Optional[ConjugateGradient] = None, use_cache: bool = False,
                 cached_v_vec_initial: bool = False):
        if not isinstance(model, SGPR):
            raise ValueError(f"CGLB model expected in the constructor of the {self.__class__}")  # models.py:112-113
        super().__init__()
        object.__setattr__(self, "model", model)  # not a sub-module: parameters stay owned by the model
        if cg_opt is None:  # the model carries a tolerance only under the TF twin's config semantics (tensorflow/models.py:36-51)
            tol = getattr(model, "max_error", None)
            cg_opt = ConjugateGradient() if tol is None else ConjugateGradient(max_error=float(tol))
        self.cg_opt = cg_opt
        self._cached_v_vec = cached_v_vec_initial
        self._use_cache = use_cache
        self.last_bounds: Optional[Bounds] = None

    @property
    def cached_v_vec(self) -> bool:
        return self._cached_v_vec

    @cached_v_vec.setter
    def cached_v_vec(self, value: bool):
        self._cached_v_vec = value

    @property
    def likelihood(self):
        return self.model.likelihood

    @property
    def kernel(self):
        return self.model.covar_module.base_kernel

    @property
    def inducing_points(self) -> Tensor:
        return self.model.covar_module.inducing_points

    @property
    def noise(self) -> Tensor:
        return self.likelihood.noise.squeeze()

    def forward(self, data: Optional[Tuple[Tensor, Tensor]] = None, *params) -> Tensor:
        """The reference evaluates the bound on the `data` it is given (models.py:151-169); here the training set lives in the
        model's HIP context, so `data` must be None or that same training set: anything else (a subset, a held-out set) raises
        instead of silently returning the bound of the training data."""
        if data is not None:
            self.model.check_same_data(data)
        ls, var, noise, mean, Z = self.model.hyper_tensors()
        v_param = self.model.v_vec if isinstance(self.model.v_vec, nn.Parameter) else None
        return _BoundFunction.apply(self, ls, var, noise, mean, Z, v_param)


class PredictCG(LowerBoundCG):
    """models.py:289-354: posterior mean/variance with the CG-corrected SGPR predictor (tolerance 1e-3)."""

    def __init__(self, model: SGPR, cg_opt: Optional[ConjugateGradient] = None):
        cg_opt = ConjugateGradient(max_error=1e-3) if cg_opt is None else cg_opt
        super().__init__(model, cg_opt)
        self._v_vec = model.v_vec.detach().clone()
        self.cached = False

    @property
    def v_vec(self):
        return self._v_vec

    def clear_cache(self):
        self.v_vec.copy_(self.model.v_vec.detach().clone())
        self.cached = False

    def forward(self, xnew: Tensor, full_cov: bool = False, full_output_cov: bool = False) -> Tuple[Tensor, Tensor]:
        if full_cov:
            raise NotImplementedError("The predict_f method currently  supports only `full_cov=False` option")  # models.py:311-314
        model, hip = self.model, self.model.hip
        with torch.no_grad():
            if not self.cached:
                model.push_hypers(get_cholesky_jitter())
                hip.setup()                                                            # models.py:327
                ls, var, noise, mean, Z = model.hyper_tensors()
                err = (hip.y - float(mean)).reshape(-1, 1)
                new_v, _stats = self.cg_opt(KernelOperator(hip), err, self.v_vec, NystromPreconditioner(hip))  # :329
                self.v_vec.data.copy_(new_v.reshape(self.v_vec.shape))
                self.cached = True
            f_mean, f_var = hip.predict(self.v_vec.reshape(-1), xnew)                  # models.py:334-352
        return f_mean.reshape(-1, 1), f_var.reshape(-1, 1)


class PredictLogdensityCG(PredictCG):
    def forward(self, data: Tuple[Tensor, Tensor], full_cov: bool = False, full_output_cov: bool = False):
        if full_cov or full_output_cov:
            raise NotImplementedError(
                "The predict_log_density method currently supports only the argument values full_cov=False and full_output_cov=False")
        x, y = data
        f_mean, f_var = super().forward(x, full_cov=full_cov, full_output_cov=full_output_cov)
        y = torch.as_tensor(y, dtype=f_mean.dtype, device=f_mean.device)
        return gaussian(y, f_mean, f_var + self.noise.to(f_mean.device)).sum(axis=-1)


def log_density(m, y, f_mean, f_var) -> Tensor:  # models.py:370-372
    noise = m.likelihood.noise.squeeze().detach().to(f_mean.device)
    y = torch.as_tensor(y, dtype=f_mean.dtype, device=f_mean.device)
    return gaussian(y, f_mean, f_var + noise).sum(axis=-1)


def gaussian(x, mu, var):  # models.py:375-379
    pi2 = math.log(2 * math.pi)
    x = x.reshape(*mu.shape)
    return -0.5 * (pi2 + torch.log(var) + (mu - x) ** 2 / var)
