"""Configuration dataclasses — same names, fields, defaults and registries as the reference's
cglb/backend/config.py (:45-166), so that the types keep working as dispatch keys.

Difference: the reference initialises inducing points through the third-party `robustgp.ConditionalVariance`
(config.py:62-65; requirements.txt:15, un-pinned git URL, absent here).  `InducingVariableConfig.init` hands the selection to
the kernel callback's device implementation (`select_inducing`: the hip backend's callback runs cglb_select_inducing, the
greedy maximisation of the conditional variance == pivoted Cholesky of K_ff without sampling, on the GPU).  There is no CPU
fallback: a plain callable without `select_inducing` is rejected.  The numpy statement of the rule that checks the kernel lives
in oracle/cglb_oracle.py; parity with robustgp itself is unpinned (its published form shuffles the inputs first, so its first
pick is a random row where this rule starts from row 0).
"""
from __future__ import annotations

import dataclasses
from functools import partial
from typing import Callable, Dict, Tuple, Union

import numpy as np

__all__ = [
    "Config", "ModelConfig", "KernelConfig", "SquaredExponentialConfig", "Matern32Config", "CGLBConfig", "CGLBN2MConfig",
    "CGLBNM2Config", "SGPRN2MConfig", "GPRConfig", "SGPRConfig", "InducingVariableConfig", "GPR_CONFIGS", "SGPR_CONFIGS",
    "KERNEL_CONFIGS", "INDUCING_VARIABLE_CONFIGS",
]

Data = Tuple[np.ndarray, np.ndarray]


class Config:
    """Base of all configuration records.  `params(data)` returns the initial values a backend needs."""

    def params(self, **kwargs) -> Dict[str, Union[float, np.ndarray]]:
        return {}


def _record(name: str, bases, fields=(), namespace=None, doc: str = ""):
    """Frozen dataclass factory: the reference's config classes are frozen dataclasses used as dispatch keys."""
    ns = dict(namespace or {})
    ns["__doc__"] = doc
    ns["__module__"] = __name__
    return dataclasses.make_dataclass(name, fields, bases=bases, namespace=ns, frozen=True)


ModelConfig = _record("ModelConfig", (Config,), doc="Marker base of model configurations.")


class KernelConfig(Config):
    """Marker base of kernel configurations."""


def _unit_kernel_params(self, data: Data):
    # variance 1, unit ARD lengthscales (one per input column) — reference defaults, config.py:74-76
    return {"variance": 1.0, "lengthscales": np.ones(np.asarray(data[0]).shape[-1])}


SquaredExponentialConfig = _record("SquaredExponentialConfig", (KernelConfig,), namespace={"params": _unit_kernel_params},
                                   doc="ARD squared-exponential (RBF) kernel.")
Matern32Config = _record("Matern32Config", (SquaredExponentialConfig,), doc="ARD Matern-3/2 kernel (same initial values).")


def _inducing_init(self, data: Data, kernel_fn: Callable):
    # The kernel callback brings the device implementation of the selection rule (the hip backend's does: cglb_select_inducing).
    accelerated = getattr(kernel_fn, "select_inducing", None)
    if accelerated is None:
        raise TypeError("InducingVariableConfig.init needs a kernel callback with a `select_inducing(X, M)` method (the hip backend's "
                        "init kernel runs the greedy conditional-variance selection on the GPU); there is no CPU fallback")
    return accelerated(np.asarray(data[0]), self.num_variables)


InducingVariableConfig = _record("InducingVariableConfig", (Config,), [("num_variables", int)], namespace={"init": _inducing_init},
                                 doc="Number of inducing points and their initialisation (greedy conditional variance).")

_NOISE_INIT = 1.0  # reference config.py:89, :104


def _gpr_params(self, data: Data):
    return {"noise_variance": _NOISE_INIT}


GPRConfig = _record("GPRConfig", (ModelConfig,), [("kernel", KernelConfig)], namespace={"params": _gpr_params}, doc="Exact GP regression.")
ExactGPConfig = _record("ExactGPConfig", (GPRConfig,), doc="Exact GP (iterative baseline; out of scope here).")


def _sgpr_params(self, data: Data):
    return {"noise_variance": _NOISE_INIT, "inducing_variable": partial(self.inducing_variable.init, data)}


SGPRConfig = _record("SGPRConfig", (ModelConfig,), [("kernel", KernelConfig), ("inducing_variable", InducingVariableConfig)],
                     namespace={"params": _sgpr_params}, doc="Sparse GP regression (Titsias).")


def _cglb_params(self, data: Data):
    out = _sgpr_params(self, data)
    out.update(max_error=self.max_error, joint_optimization=self.joint_optimization, vzero=self.vzero)
    return out


CGLBConfig = _record("CGLBConfig", (SGPRConfig,),
                     [("max_error", float, dataclasses.field(default=1.0)), ("joint_optimization", bool, dataclasses.field(default=False)),
                      ("vzero", bool, dataclasses.field(default=False))],
                     namespace={"params": _cglb_params}, doc="Conjugate-gradient lower bound model (reference config.py:110-121).")
CGLBN2MConfig = _record("CGLBN2MConfig", (CGLBConfig,), doc="Log-det ablation (out of scope).")
CGLBNM2Config = _record("CGLBNM2Config", (CGLBConfig,), doc="Log-det ablation (out of scope).")
SGPRN2MConfig = _record("SGPRN2MConfig", (SGPRConfig,), doc="Log-det ablation (out of scope).")

# registries: the keys are the reference's CLI choices (config.py:139-166)
GPR_CONFIGS = dict(gpr=GPRConfig, exactgp=ExactGPConfig)
SGPR_CONFIGS = dict(sgpr=SGPRConfig, cglb=CGLBConfig, sgprn2m=SGPRN2MConfig, cglbn2m=CGLBN2MConfig, cglbnm2=CGLBNM2Config)
KERNEL_CONFIGS = {"SquaredExponential": SquaredExponentialConfig, "Matern32": Matern32Config, "mat32": Matern32Config,
                  "rbf": SquaredExponentialConfig}
INDUCING_VARIABLE_CONFIGS = {key: InducingVariableConfig for key in ("InducingVariable", "ConditionalVariance", "iv", "cv")}
