"""Solver plug-in seam — mirror of the reference's cglb/backend/pytorch/conjugate_gradient.py.

Same names, arguments and return values: `ConjugateGradient(max_error, max_cg_iter, restart_cg_iter)` is a
callable `(A, b, v, precond) -> (v, ConjugateGradientStats)` (:31-86) and `NystromPreconditioner` is a callable
`r -> (z, rz)` (:89-113).  The operands are HIP-backed handles instead of torch matrices: `A` is a
`KernelOperator` (implicit K_ff + sigma^2 I, only `A @ x` is defined, as the reference only uses that) and the
preconditioner wraps the common terms held by the same context.  The loop itself runs inside libcglb_hip.so
(cglb_pcg_solve); anything that is not HIP-backed is rejected loudly — there is no torch/CPU fallback.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Tuple, Union

import torch

from ..hip_context import HipContext

Tensor = torch.Tensor
Preconditioner = Callable[[Tensor], Tuple[Tensor, Tensor]]


@dataclass
class ConjugateGradientStats:
    steps: Union[Tensor, float]
    residual_error: Union[Tensor, float]


class KernelOperator:
    """kernel(x).add_diag(sigma_sq) as an operator (models.py:251-252): supports `A @ x` for x of shape [N] or [N, k]."""

    def __init__(self, ctx: HipContext):
        self.ctx = ctx
        self.shape = (ctx.N, ctx.N)

    def __matmul__(self, x: Tensor) -> Tensor:
        x = torch.as_tensor(x)
        if x.shape[0] != self.ctx.N:
            raise ValueError(f"operator is {self.shape}, operand has {x.shape[0]} rows")
        if x.ndim == 1:
            return self.ctx.matvec(x)
        cols = [self.ctx.matvec(x[:, k].contiguous()) for k in range(x.shape[1])]
        return torch.stack(cols, dim=1)

    def detach(self):
        return self


@dataclass
class NystromPreconditioner:
    """(Q_ff + sigma^2 I)^-1 r by Woodbury, A = sigma^-1 L^-1 K_uf and LB = chol(A A^T + I) living in the HIP
    context (computed by cglb_setup).  Call: r [N] or [N,1] -> (z, rz) like conjugate_gradient.py:95-113."""

    ctx: HipContext

    def __call__(self, r: Tensor) -> Tuple[Tensor, Tensor]:
        shape = r.shape
        z, rz = self.ctx.precond(r.reshape(-1))
        return z.reshape(shape), torch.tensor(rz, dtype=torch.float64)


@dataclass
class ConjugateGradient:
    """CG stops if: 0.5 * r^T Q^-1 r < max_error || i > max_cg_iter  (conjugate_gradient.py:31-39)."""

    max_error: float = 1.0
    max_cg_iter: int = 100
    restart_cg_iter: int = 40

    def __call__(self, A: KernelOperator, b: Tensor, v: Tensor, precond: Preconditioner) -> Tuple[Tensor, ConjugateGradientStats]:
        if not isinstance(A, KernelOperator):
            raise TypeError("cglb_amd ConjugateGradient needs a HIP KernelOperator as A (no dense/torch fallback)")
        if not isinstance(precond, NystromPreconditioner) or precond.ctx is not A.ctx:
            raise TypeError("cglb_amd ConjugateGradient needs the NystromPreconditioner of the same HIP context")
        shape = v.shape
        # the context clones v (conjugate_gradient.py:55): the caller's tensor is not mutated
        vout, steps, half_rz = A.ctx.pcg(b.reshape(-1), v.reshape(-1), self.max_error, self.max_cg_iter, self.restart_cg_iter)
        stats = ConjugateGradientStats(steps, torch.tensor(half_rz, dtype=torch.float64))  # :83 (i: int, 0.5*rz: CPU tensor)
        return vout.reshape(shape), stats
