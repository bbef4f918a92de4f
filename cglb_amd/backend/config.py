"""Configuration dataclasses — same names, fields, defaults and registries as the reference's
cglb/backend/config.py (:45-166), so that the types keep working as dispatch keys.

Difference: the reference initialises inducing points through the third-party `robustgp.ConditionalVariance`
(config.py:62-65; requirements.txt:15, un-pinned git URL, absent here).  `InducingVariableConfig.init` restates that
published algorithm (greedy maximisation of the conditional variance == pivoted Cholesky of K_ff, without
sampling) in numpy; parity with robustgp itself is unpinned.
"""
from __future__ import annotations

import dataclasses
from functools import partial
from typing import Callable, Dict, Tuple, Union

import numpy as np

__all__ = [
    "Config", "ModelConfig", "KernelConfig", "SquaredExponentialConfig", "Matern32Config", "CGLBConfig", "CGLBN2MConfig",
    "CGLBNM2Config", "SGPRN2MConfig", "GPRConfig", "SGPRConfig", "InducingVariableConfig", "GPR_CONFIGS", "SGPR_CONFIGS",
    "KERNEL_CONFIGS", "INDUCING_VARIABLE_CONFIGS", "greedy_conditional_variance",
]

Data = Tuple[np.ndarray, np.ndarray]
dataclass_frozen = partial(dataclasses.dataclass, frozen=True)


def greedy_conditional_variance(X: np.ndarray, M: int, kernel_fn: Callable, jitter: float = 1e-12) -> np.ndarray:
    """Greedy inducing-point selection: repeatedly take the point with the largest conditional variance
    given the points chosen so far (the deterministic `sample=False` rule the reference requests,
    config.py:63).  kernel_fn(x1, x2, full_cov) follows the reference callback (pytorch/interface.py:278-284):
    x2=None, full_cov=False -> diag; full_cov=True -> matrix.  O(N M^2) time, O(N M) memory."""
    N = X.shape[0]
    M = min(M, N)
    d = np.asarray(kernel_fn(X, None, full_cov=False), dtype=np.float64).reshape(-1) + jitter
    ci = np.zeros((M, N))
    chosen = np.zeros(M, dtype=np.int64)
    chosen[0] = int(np.argmax(d))
    for m in range(M - 1):
        j = chosen[m]
        dj = np.sqrt(d[j])
        col = np.asarray(kernel_fn(X, X[j:j + 1], full_cov=True), dtype=np.float64).reshape(-1)
        col[j] += jitter
        ei = (col - ci[:m].T @ ci[:m, j]) / dj
        ci[m] = ei
        d = np.maximum(d - ei * ei, 0.0)
        d[chosen[: m + 1]] = 0.0  # a chosen point has no conditional variance left
        chosen[m + 1] = int(np.argmax(d))
    return X[chosen].copy()


class Config:
    def params(self, **kwargs) -> Dict[str, Union[float, np.ndarray]]:
        pass


@dataclass_frozen
class ModelConfig(Config):
    pass


@dataclass_frozen
class InducingVariableConfig(Config):
    num_variables: int

    def params(self, data: Data) -> Dict[str, Union[float, np.ndarray]]:
        ...

    def init(self, data: Data, kernel_fn: Callable):
        return greedy_conditional_variance(np.asarray(data[0]), self.num_variables, kernel_fn)


class KernelConfig(Config):
    pass


@dataclass_frozen
class SquaredExponentialConfig(KernelConfig):
    def params(self, data: Data) -> Dict[str, Union[float, np.ndarray]]:
        vecdim = data[0].shape[-1]
        return {"variance": 1.0, "lengthscales": np.repeat(1.0, vecdim)}


@dataclass_frozen
class Matern32Config(SquaredExponentialConfig):
    pass


@dataclass_frozen
class GPRConfig(ModelConfig):
    kernel: KernelConfig

    def params(self, data: Data) -> Dict[str, Union[float, np.ndarray]]:
        return {"noise_variance": 1.0}


@dataclass_frozen
class ExactGPConfig(GPRConfig):
    ...


@dataclass_frozen
class SGPRConfig(ModelConfig):
    kernel: KernelConfig
    inducing_variable: InducingVariableConfig

    def params(self, data: Data) -> Dict[str, Union[float, np.ndarray, Callable]]:
        inducing_variable_fn = partial(self.inducing_variable.init, data)
        return {"noise_variance": 1.0, "inducing_variable": inducing_variable_fn}


@dataclass_frozen
class CGLBConfig(SGPRConfig):
    max_error: float = 1.0
    joint_optimization: bool = False
    vzero: bool = False

    def params(self, data: Data) -> Dict[str, Union[float, np.ndarray]]:
        param_dict = super().params(data)
        param_dict["max_error"] = self.max_error
        param_dict["joint_optimization"] = self.joint_optimization
        param_dict["vzero"] = self.vzero
        return param_dict


@dataclass_frozen
class CGLBN2MConfig(CGLBConfig):
    pass


@dataclass_frozen
class CGLBNM2Config(CGLBConfig):
    pass


@dataclass_frozen
class SGPRN2MConfig(SGPRConfig):
    pass


GPR_CONFIGS = {"gpr": GPRConfig, "exactgp": ExactGPConfig}
SGPR_CONFIGS = {"sgpr": SGPRConfig, "cglb": CGLBConfig, "sgprn2m": SGPRN2MConfig, "cglbn2m": CGLBN2MConfig, "cglbnm2": CGLBNM2Config}
KERNEL_CONFIGS = {"SquaredExponential": SquaredExponentialConfig, "Matern32": Matern32Config, "mat32": Matern32Config, "rbf": SquaredExponentialConfig}
INDUCING_VARIABLE_CONFIGS = {"InducingVariable": InducingVariableConfig, "ConditionalVariance": InducingVariableConfig,
                             "iv": InducingVariableConfig, "cv": InducingVariableConfig}
