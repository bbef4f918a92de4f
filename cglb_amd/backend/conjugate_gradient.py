"""Solver plug-in seam — mirror of the reference's cglb/backend/pytorch/conjugate_gradient.py.

Same names, arguments and return values: `ConjugateGradient(max_error, max_cg_iter, restart_cg_iter)` is a
callable `(A, b, v, precond) -> (v, ConjugateGradientStats)` (:31-86) and `NystromPreconditioner` is a callable
`r -> (z, rz)` (:89-113).  The operands are HIP-backed handles instead of torch matrices: `A` is a
`KernelOperator` (implicit K_ff + sigma^2 I, only `A @ x` is defined, as the reference only uses that) and the
preconditioner wraps the common terms held by the same context.  With that pair the loop itself runs inside libcglb_hip.so
(cglb_pcg_solve).  The preconditioner seam is open like the reference's (`Preconditioner = Callable[[Tensor], Tuple[Tensor,
Tensor]]`, :22): any other callable `r -> (z, rz)` is driven from a host loop that follows :41-86 line by line, with `A @ p` and
the vector updates still on the GPU (cglb_matvec_dot, cglb_vec_*).  The operator `A` must be the HIP `KernelOperator`: there is
no dense/torch fallback for the N^2 work.
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
        if not callable(precond):
            raise TypeError("precond must be callable: r -> (z, rz)  (conjugate_gradient.py:22)")
        if not isinstance(precond, NystromPreconditioner) or precond.ctx is not A.ctx:
            return self._solve_with_foreign_preconditioner(A, b, v, precond)
        shape = v.shape
        # the context clones v (conjugate_gradient.py:55): the caller's tensor is not mutated
        vout, steps, half_rz = A.ctx.pcg(b.reshape(-1), v.reshape(-1), self.max_error, self.max_cg_iter, self.restart_cg_iter)
        stats = ConjugateGradientStats(steps, torch.tensor(half_rz, dtype=torch.float64))  # :83 (i: int, 0.5*rz: CPU tensor)
        return vout.reshape(shape), stats

    def _solve_with_foreign_preconditioner(self, A: KernelOperator, b: Tensor, v: Tensor, precond: Preconditioner):
        """conjugate_gradient.py:41-86 on the host, for a preconditioner that is any callable `r -> (z, rz)`: the mat-vec (with the
        fused p^T A p), the residual and the v / r / p updates are the library's kernels; only `precond(r)` is the caller's code."""
        from ..distributed import HipSymLocalOps
        ctx = A.ctx
        ops, N, shape = HipSymLocalOps(ctx), ctx.N, v.shape
        if (ctx.r0, ctx.r1) != (0, N):
            raise TypeError("a foreign preconditioner needs a single-shard context")
        scalar = lambda: torch.zeros(1, dtype=torch.float64, device=ctx.device)
        rz_t, nrz_t, pAp_t = scalar(), scalar(), scalar()

        def apply(r_vec):
            z, rz = precond(r_vec.reshape(shape))
            z = ctx._dev(torch.as_tensor(z).reshape(-1), N)
            return z, float(rz)

        bb = ctx._dev(b.reshape(-1), N)
        vv = ctx._dev(v.reshape(-1), N).clone()                      # :55
        Kv, Ap, r = ctx.empty(N), ctx.empty(N), ctx.empty(N)
        ops.matvec(vv, Kv)                                           # :57
        ops.vec_residual(N, r, bb, Kv)                               # :58
        z, rz = apply(r)                                             # :59
        p = z.clone()                                                # :61
        rz_t.fill_(rz)
        i = 0
        while 0.5 * rz > self.max_error and i < self.max_cg_iter:    # :65
            ops.matvec_dot(p, Ap, pAp_t)                             # :66 and (p * Ap).sum() of :67
            restart = self.restart_cg_iter > 0 and (i % self.restart_cg_iter == self.restart_cg_iter - 1)  # :70
            ops.vec_update_v_r(N, vv, r, p, Ap, rz_t, pAp_t, not restart)   # :67-68, :72
            if restart:
                ops.matvec(vv, Kv)
                ops.vec_residual(N, r, bb, Kv)                       # :72 exact residual
            z, new_rz = apply(r)                                     # :73
            nrz_t.fill_(new_rz)
            ops.vec_update_p(N, p, z, nrz_t, rz_t, restart)          # :75
            rz_t, nrz_t, rz = nrz_t, rz_t, new_rz                    # :76
            i += 1
        return vv.reshape(shape), ConjugateGradientStats(i, torch.tensor(0.5 * rz, dtype=torch.float64))  # :83-86
