"""Row-sharded CGLB evaluation: one process per GPU, collectives through torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no distributed code; this is new design.  Rank g owns the row block
I_g of K_ff: slices of y, v, r, z, p, Ap and the column shard A[:, I_g] of the Nystrom panel; X, Z, L, LB
and the hyper-parameters are replicated.  One PCG iteration (conjugate_gradient.py:65-81) exchanges

    all-gather   p            N/G elements per rank   (every rank needs the full direction for K[I_g,:] p)
    all-reduce   p^T A p      1 scalar
    all-reduce   u = A r      M elements             (before the replicated small triangular products)
    all-reduce   r^T P r      1 scalar               (stop test, every iteration, as the reference does)

all latency-bound on xGMI.  The local arithmetic is behind `LocalOps`: in production that is
`HipLocalOps` (the C ABI of libcglb_hip.so); the CPU tests inject an oracle-backed implementation to
exercise this driver at world_size 2 over gloo.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch

try:
    import torch.distributed as dist
except Exception:  # pragma: no cover
    dist = None


def row_partition(n: int, world: int) -> Tuple[int, list]:
    """Contiguous equal blocks of `per = ceil(n/world)` rows (the last ones may be short or empty)."""
    per = (n + world - 1) // world
    return per, [(min(g * per, n), min((g + 1) * per, n)) for g in range(world)]


class Comm:
    """The two collectives the path needs.  world == 1 -> no-ops."""

    def __init__(self, group=None, force: bool = False):
        """force=True issues the collectives even at world_size 1 (used to exercise the RCCL calls on a one-GPU box)."""
        self.group = group
        self.active = dist is not None and dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        self.force = bool(force) and self.active

    def allreduce(self, t: torch.Tensor) -> None:
        if self.world > 1 or self.force:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def allgather_inplace(self, buf: torch.Tensor, per: int) -> None:
        """buf has world*per elements; rank g's slice buf[g*per:(g+1)*per] is valid on rank g."""
        if self.world == 1 and not self.force:
            return
        mine = buf[self.rank * per:(self.rank + 1) * per]
        if buf.is_cuda:
            # RCCL's in-place form: the send buffer is this rank's slot of the receive buffer (no staging copy)
            dist.all_gather_into_tensor(buf, mine, group=self.group)
            return
        try:
            dist.all_gather_into_tensor(buf, mine.clone(), group=self.group)
        except (RuntimeError, NotImplementedError):
            chunks = [buf[g * per:(g + 1) * per] for g in range(self.world)]
            dist.all_gather(chunks, mine.clone(), group=self.group)


class LocalOps:
    """What the sharded driver needs from one rank.  All tensors live on `device` with `dtype`; scalar
    slots are float64 tensors of one element; *_full vectors have N elements, *_local have nloc."""

    device: torch.device
    dtype: torch.dtype
    N: int
    M: int
    D: int
    r0: int
    r1: int

    def setup_local(self): raise NotImplementedError
    def aat_tensor(self) -> torch.Tensor: raise NotImplementedError
    def setup_finish(self): raise NotImplementedError
    def rhs(self, out_local): raise NotImplementedError
    def matvec(self, p_full, out_local): raise NotImplementedError
    def matvec_dot(self, p_full, out_local, pdot): raise NotImplementedError
    def precond_u(self, r_local, u_out): raise NotImplementedError
    def precond_z(self, r_local, u, z_local, rz_out): raise NotImplementedError
    def update_v_r(self, v_local, r_local, p_local, Ap_local, rz, pAp, update_r: bool): raise NotImplementedError
    def residual(self, r_local, b_local, Kv_local): raise NotImplementedError
    def update_p(self, p_local, z_local, new_rz, rz, restart: bool): raise NotImplementedError
    def obj_phase1(self, v_full, u_out): raise NotImplementedError
    def obj_phase2(self, v_full, u, sc_out, aw_out): raise NotImplementedError
    def obj_phase3(self, v_full, sc, aw, grad_out): raise NotImplementedError
    def obj_finish(self, sc) -> Tuple[float, float, float, float]: raise NotImplementedError


class HipLocalOps(LocalOps):
    """LocalOps over the C ABI (cglb_shard_* entry points of include/cglb_hip.h)."""

    def __init__(self, ctx):
        from . import _lib
        self._lib = _lib
        self.ctx = ctx
        self.lib = ctx.lib
        self.h = ctx._ctx
        self.device, self.dtype = ctx.device, ctx.dtype
        self.N, self.M, self.D, self.r0, self.r1 = ctx.N, ctx.M, ctx.D, ctx.r0, ctx.r1

    @staticmethod
    def _p(t):
        from ctypes import c_void_p
        return None if t is None else c_void_p(t.data_ptr())

    def _ck(self, rc):
        self._lib.check(rc, self.h)

    def setup_local(self): self.ctx.setup_local()
    def aat_tensor(self): return self.ctx.aat_tensor()
    def setup_finish(self): self.ctx.setup_finish()
    def rhs(self, out_local): self._ck(self.lib.cglb_shard_rhs(self.h, self._p(out_local)))
    def matvec(self, p_full, out_local): self._ck(self.lib.cglb_matvec(self.h, self._p(p_full), self._p(out_local)))
    def matvec_dot(self, p_full, out_local, pdot): self._ck(self.lib.cglb_matvec_dot(self.h, self._p(p_full), self._p(out_local), self._p(pdot)))
    def precond_u(self, r_local, u_out): self._ck(self.lib.cglb_shard_precond_u(self.h, self._p(r_local), self._p(u_out)))
    def precond_z(self, r_local, u, z_local, rz_out): self._ck(self.lib.cglb_shard_precond_z(self.h, self._p(r_local), self._p(u), self._p(z_local), self._p(rz_out)))
    def update_v_r(self, v_local, r_local, p_local, Ap_local, rz, pAp, update_r): self._ck(self.lib.cglb_shard_update_v_r(self.h, self._p(v_local), self._p(r_local), self._p(p_local), self._p(Ap_local), self._p(rz), self._p(pAp), int(update_r)))
    def residual(self, r_local, b_local, Kv_local): self._ck(self.lib.cglb_shard_residual(self.h, self._p(r_local), self._p(b_local), self._p(Kv_local)))
    def update_p(self, p_local, z_local, new_rz, rz, restart): self._ck(self.lib.cglb_shard_update_p(self.h, self._p(p_local), self._p(z_local), self._p(new_rz), self._p(rz), int(restart)))
    def obj_phase1(self, v_full, u_out): self._ck(self.lib.cglb_shard_obj_phase1(self.h, self._p(v_full), self._p(u_out)))
    def obj_phase2(self, v_full, u, sc_out, aw_out): self._ck(self.lib.cglb_shard_obj_phase2(self.h, self._p(v_full), self._p(u), self._p(sc_out), self._p(aw_out)))
    def obj_phase3(self, v_full, sc, aw, grad_out): self._ck(self.lib.cglb_shard_obj_phase3(self.h, self._p(v_full), self._p(sc), self._p(aw), self._p(grad_out)))

    def obj_finish(self, sc):
        from ctypes import c_double
        out4 = (c_double * 4)()
        self._ck(self.lib.cglb_shard_obj_finish(self.h, self._p(sc), out4))
        return out4[0], out4[1], out4[2], out4[3]


@dataclass
class ShardedResult:
    bound: float
    lower: float
    upper: float
    logdet: float
    steps: int
    residual_error: float
    grad: Optional[np.ndarray]  # packed constrained-space gradient, identical on every rank


class ShardedCGLB:
    """Sharded objective (+gradient) evaluation.  Every rank calls the same methods in the same order."""

    def __init__(self, ops: LocalOps, comm: Optional[Comm] = None):
        self.ops = ops
        self.comm = comm if comm is not None else Comm()
        self.N, self.M, self.D = ops.N, ops.M, ops.D
        self.per, parts = row_partition(self.N, self.comm.world)
        r0, r1 = parts[self.comm.rank]
        if (ops.r0, ops.r1) != (r0, r1):
            raise ValueError(f"rank {self.comm.rank}: local ops own rows [{ops.r0},{ops.r1}) but the partition says [{r0},{r1})")
        self.r0, self.r1, self.nloc = r0, r1, r1 - r0
        dev, dt = ops.device, ops.dtype
        G, per = self.comm.world, self.per
        z = lambda n, d=dt: torch.zeros(max(n, 1), dtype=d, device=dev)
        # gathered vectors (padded to G*per so that equal-sized all-gathers line up with row order)
        self.vbuf, self.pbuf = z(G * per), z(G * per)
        self.r, self.z, self.Ap, self.Kv, self.b = z(self.nloc), z(self.nloc), z(self.nloc), z(self.nloc), z(self.nloc)
        self.u, self.aw = z(self.M), z(self.M)
        self.rz, self.rz_new, self.pAp = z(1, torch.float64), z(1, torch.float64), z(1, torch.float64)
        self.sc = z(8, torch.float64)
        self.grad = z(self.D + 3 + self.M * self.D, torch.float64)

    # views --------------------------------------------------------------------------------------------
    def _own(self, buf):
        return buf[self.comm.rank * self.per: self.comm.rank * self.per + self.nloc]

    @property
    def v_local(self):
        """This rank's slice of the persistent warm-start vector (models.py:59-72)."""
        return self._own(self.vbuf)

    def v_full(self):
        return self.vbuf[: self.N]

    # common terms --------------------------------------------------------------------------------------
    def setup(self):
        """models.py:176-213 with B = I + sum_g A_g A_g^T (one all-reduce of M x M per evaluation)."""
        self.ops.setup_local()
        if self.comm.world > 1:
            self.comm.allreduce(self.ops.aat_tensor())
        self.ops.setup_finish()

    def _precond(self, rz_out):
        self.ops.precond_u(self.r, self.u)
        self.comm.allreduce(self.u)
        self.ops.precond_z(self.r, self.u, self.z, rz_out)
        self.comm.allreduce(rz_out)

    # PCG (conjugate_gradient.py:41-86) ------------------------------------------------------------------
    def pcg(self, max_error=1.0, max_cg_iter=100, restart_cg_iter=40) -> Tuple[int, float]:
        """Solves (K_ff + noise I) v = y - mean from the current v (in place).  Returns (steps, 1/2 r^T P r)."""
        ops, comm, per = self.ops, self.comm, self.per
        p_local, v_local = self._own(self.pbuf), self._own(self.vbuf)
        ops.rhs(self.b)
        comm.allgather_inplace(self.vbuf, per)
        ops.matvec(self.v_full(), self.Kv)                      # :57
        ops.residual(self.r, self.b, self.Kv)                   # :58
        self._precond(self.rz)                                  # :59
        ops.update_p(p_local, self.z, self.rz, self.rz, True)   # :61  p = z
        comm.allgather_inplace(self.pbuf, per)
        rz = float(self.rz.item())
        i = 0
        while 0.5 * rz > max_error and i < max_cg_iter:         # :65
            ops.matvec_dot(self.pbuf[: self.N], self.Ap, self.pAp)   # :66-67
            comm.allreduce(self.pAp)
            restart = restart_cg_iter > 0 and (i % restart_cg_iter == restart_cg_iter - 1)  # :70
            ops.update_v_r(v_local, self.r, p_local, self.Ap, self.rz, self.pAp, not restart)  # :68, :72
            if restart:
                comm.allgather_inplace(self.vbuf, per)
                ops.matvec(self.v_full(), self.Kv)
                ops.residual(self.r, self.b, self.Kv)
            self._precond(self.rz_new)                          # :73
            ops.update_p(p_local, self.z, self.rz_new, self.rz, restart)  # :75
            self.rz.copy_(self.rz_new)                          # :76
            comm.allgather_inplace(self.pbuf, per)
            rz = float(self.rz.item())                          # host test of :65 (the sync of :80-81)
            i += 1
        return i, 0.5 * rz

    # objective + gradient (models.py:151-174, :246-286; optimizer.py:95-98) -------------------------------
    def objective_and_grad(self, run_cg=True, max_error=1.0, max_cg_iter=100, restart_cg_iter=40, with_grad=True) -> ShardedResult:
        ops, comm = self.ops, self.comm
        self.setup()
        steps, half = 0, float("nan")
        if run_cg:
            steps, half = self.pcg(max_error, max_cg_iter, restart_cg_iter)
        comm.allgather_inplace(self.vbuf, self.per)
        v_full = self.v_full()
        ops.obj_phase1(v_full, self.u)
        comm.allreduce(self.u)
        ops.obj_phase2(v_full, self.u, self.sc, self.aw)
        comm.allreduce(self.sc)
        grad = None
        if with_grad:
            comm.allreduce(self.aw)
            ops.obj_phase3(v_full, self.sc, self.aw, self.grad)
            comm.allreduce(self.grad)
            grad = self.grad.detach().cpu().numpy().copy()
        bound, lower, upper, logdet = ops.obj_finish(self.sc)
        return ShardedResult(bound, lower, upper, logdet, steps, half, grad)


# =====================================================================================================================
# Cyclic-symmetric scheme: keeps the factor 2 of the symmetric pair kernel under sharding.
# =====================================================================================================================
class SymLocalOps(LocalOps):
    """Additional local operations of the cyclic-symmetric driver (see SymShardedCGLB)."""

    noise: float

    def set_parallel(self, world: int, rank: int): raise NotImplementedError
    def rhs_full(self, out_full): raise NotImplementedError
    def matvec_cyclic(self, p_full, out_full_partial): raise NotImplementedError
    def vec_dot(self, n, a, b, out): raise NotImplementedError
    def vec_update_v_r(self, n, v, r, p, Ap, rz, pAp, update_r: bool): raise NotImplementedError
    def vec_residual(self, n, r, b, Kv): raise NotImplementedError
    def vec_update_p(self, n, p, z, new_rz, rz, restart: bool): raise NotImplementedError
    def vec_axpy(self, n, alpha, x, y): raise NotImplementedError
    def precond_z_seg(self, r_local, u, z_slot, per: int): raise NotImplementedError
    def vec_update_p_seg(self, n, per: int, world: int, p, zseg, new_rz, rz, restart: bool): raise NotImplementedError
    def obj_phase1_kv(self, Kv_local, u_out): raise NotImplementedError
    def obj_w(self, w_local_out): raise NotImplementedError
    def obj_phase3_cyclic(self, v_full, u_full, sc, aw, grad_out): raise NotImplementedError
    # hyper-parameter update and the prediction pieces (PredictCG, models.py:307-354) - needed behind the backend API
    def set_hypers(self, lengthscales, variance, noise, mean, Z, jitter): raise NotImplementedError
    def y_full(self) -> torch.Tensor: raise NotImplementedError
    def drop_weighted_operand(self): pass   # HIP: the vector vec_update_p_seg wrote is handed out and may change before the next mat-vec
    def predict_u(self, Kv_local, u_out): raise NotImplementedError
    def predict_rows(self, v_full, u, xnew) -> Tuple[torch.Tensor, torch.Tensor]: raise NotImplementedError


class HipSymLocalOps(HipLocalOps, SymLocalOps):
    @property
    def noise(self):
        return self.ctx.noise

    def set_parallel(self, world, rank): self._ck(self.lib.cglb_set_parallel(self.h, int(world), int(rank)))
    def rhs_full(self, out_full): self._ck(self.lib.cglb_rhs_full(self.h, self._p(out_full)))
    def matvec_cyclic(self, p_full, out): self._ck(self.lib.cglb_matvec_cyclic(self.h, self._p(p_full), self._p(out)))
    def vec_dot(self, n, a, b, out): self._ck(self.lib.cglb_vec_dot(self.h, int(n), self._p(a), self._p(b), self._p(out)))
    def vec_update_v_r(self, n, v, r, p, Ap, rz, pAp, update_r): self._ck(self.lib.cglb_vec_update_v_r(self.h, int(n), self._p(v), self._p(r), self._p(p), self._p(Ap), self._p(rz), self._p(pAp), int(update_r)))
    def vec_residual(self, n, r, b, Kv): self._ck(self.lib.cglb_vec_residual(self.h, int(n), self._p(r), self._p(b), self._p(Kv)))
    def vec_update_p(self, n, p, z, new_rz, rz, restart): self._ck(self.lib.cglb_vec_update_p(self.h, int(n), self._p(p), self._p(z), self._p(new_rz), self._p(rz), int(restart)))
    def vec_axpy(self, n, alpha, x, y): self._ck(self.lib.cglb_vec_axpy(self.h, int(n), float(alpha), self._p(x), self._p(y)))
    def precond_z_seg(self, r_local, u, z_slot, per): self._ck(self.lib.cglb_shard_precond_z_seg(self.h, self._p(r_local), self._p(u), self._p(z_slot), int(per)))
    def vec_update_p_seg(self, n, per, world, p, zseg, new_rz, rz, restart): self._ck(self.lib.cglb_vec_update_p_seg(self.h, int(n), int(per), int(world), self._p(p), self._p(zseg), self._p(new_rz), self._p(rz), int(restart)))
    def obj_phase1_kv(self, Kv_local, u_out): self._ck(self.lib.cglb_shard_obj_phase1_kv(self.h, self._p(Kv_local), self._p(u_out)))
    def obj_w(self, w_local_out): self._ck(self.lib.cglb_shard_obj_w(self.h, self._p(w_local_out)))
    def obj_phase3_cyclic(self, v_full, u_full, sc, aw, grad_out): self._ck(self.lib.cglb_shard_obj_phase3_cyclic(self.h, self._p(v_full), self._p(u_full), self._p(sc), self._p(aw), self._p(grad_out)))
    def set_hypers(self, lengthscales, variance, noise, mean, Z, jitter): self.ctx.set_hypers(lengthscales, variance, noise, mean, Z, jitter)
    def y_full(self): return self.ctx.y
    def drop_weighted_operand(self): self.ctx.set_option("drop_weighted_operand", 1)
    def predict_u(self, Kv_local, u_out): self._ck(self.lib.cglb_shard_predict_u(self.h, self._p(Kv_local), self._p(u_out)))

    def predict_rows(self, v_full, u, xnew):
        xn = torch.as_tensor(xnew, dtype=self.dtype).reshape(-1, self.D).contiguous().to(self.device)
        n = int(xn.shape[0])
        mean, var = torch.empty(max(n, 1), dtype=self.dtype, device=self.device), torch.empty(max(n, 1), dtype=self.dtype, device=self.device)
        self._ck(self.lib.cglb_shard_predict_rows(self.h, self._p(v_full), self._p(u), self._p(xn) if n else None, n, self._p(mean), self._p(var)))
        return mean[:n], var[:n]


class SymShardedCGLB:
    """Multi-GPU evaluation with the symmetric K_ff work dealt cyclically over the ranks.

    * K_ff p: every rank evaluates the (row block, column chunk) cells of the GLOBAL upper triangle whose 256-row block index
      is == rank (mod world), using each kernel value for both out_i and out_j, and produces a full-length partial vector;
      one all-reduce (N elements) gives every rank the full product.  Work per rank is N^2/(2G) pair evaluations, against
      ~N^2/G for row sharding (where only the diagonal block of a shard is symmetric).
    * p, Ap, v, r, b live in full on every rank; their O(N) updates are done redundantly (identical inputs, identical
      kernels -> identical bits).
    * The Nystrom panel A stays column-sharded over contiguous rows [r0, r1): u = A r is all-reduced (M elements) and the
      preconditioned residual z is all-gathered in slices of per + 1 elements: the extra element of a slice carries that rank's
      partial of r^T z over its own rows.  r^T P r - the scalar the host's stop test reads (conjugate_gradient.py:65) - is the sum
      of those `world` partials in rank order, so every rank derives it from the SAME gathered numbers: the loop's control flow is
      identical on all ranks by construction, even if a replicated vector should ever differ in a bit between ranks (which would
      otherwise strand the ranks in different collectives).  A non-finite value raises on every rank together.
    Collectives per PCG iteration: all-reduce(N), all-reduce(M), all-gather(N/G + 1) - three instead of four, no scalar ones.
    """

    def __init__(self, ops: SymLocalOps, comm: Optional[Comm] = None):
        self.ops = ops
        self.comm = comm if comm is not None else Comm()
        self.N, self.M, self.D = ops.N, ops.M, ops.D
        G = self.comm.world
        self.per, parts = row_partition(self.N, G)
        r0, r1 = parts[self.comm.rank]
        if (ops.r0, ops.r1) != (r0, r1):
            raise ValueError(f"rank {self.comm.rank}: local ops own rows [{ops.r0},{ops.r1}) but the partition says [{r0},{r1})")
        self.r0, self.r1, self.nloc = r0, r1, r1 - r0
        dev, dt = ops.device, ops.dtype
        z = lambda n, d=dt: torch.zeros(max(n, 1), dtype=d, device=dev)
        N, per = self.N, self.per
        self.v, self.p, self.r, self.Ap, self.Kv, self.b = z(N), z(N), z(N), z(N), z(N), z(N)
        self.zseg = z(G * (per + 1))                    # all-gather target of z: G slices of per rows + 1 partial of r^T z
        self.ubuf = z(G * per)                          # all-gather target of u = w + v/2 (gradient phase)
        self.u, self.aw = z(self.M), z(self.M)
        self.rz, self.rz_new, self.pAp, self.scratch = z(1, torch.float64), z(1, torch.float64), z(1, torch.float64), z(1, torch.float64)
        self.sc = z(8, torch.float64)
        self.grad = z(self.D + 3 + self.M * self.D, torch.float64)
        ops.set_parallel(G, self.comm.rank)
        self.lookahead = True
        # False (default, like the library's option "final_matvec" = 0): after a solve K v = b - r comes from the residual the PCG
        # recurrence carries instead of one more mat-vec + all-reduce (models.py:280 recomputes it; difference at round-off level)
        self.final_matvec = False
        self._pinned = None
        self._event = None

    def _own(self, buf):
        return buf[self.comm.rank * self.per: self.comm.rank * self.per + self.nloc]

    def _read_scalar_async(self, t):
        """Starts the device->host copy of a 1-element tensor; the returned callable waits for it and gives the float."""
        if t.device.type != "cuda":
            value = float(t.item())
            return lambda: value
        if self._pinned is None:
            self._pinned = torch.empty(1, dtype=torch.float64).pin_memory()
            self._event = torch.cuda.Event()
        self._pinned.copy_(t, non_blocking=True)
        self._event.record(torch.cuda.current_stream(t.device))

        def wait():
            self._event.synchronize()
            return float(self._pinned[0])
        return wait

    def v_full(self):
        return self.v

    def setup(self):
        self.ops.setup_local()
        if self.comm.world > 1 or self.comm.force:
            self.comm.allreduce(self.ops.aat_tensor())
        self.ops.setup_finish()

    def matvec(self, x_full, out_full):
        """out = (K_ff + noise I) x, full length on every rank."""
        self.ops.matvec_cyclic(x_full, out_full)    # rank 0's partial carries the noise term
        self.comm.allreduce(out_full)

    def _precond_and_direction(self, rz_new, rz_old, restart: bool):
        """z = P r (conjugate_gradient.py:73), gathered in segments; then rz_new = r^T z from the gathered partials and
        p = z + p rz_new / rz_old, or p = z (:75)."""
        ops, per, G = self.ops, self.per, self.comm.world
        r_loc = self.r[self.r0:self.r1]
        ops.precond_u(r_loc, self.u)
        self.comm.allreduce(self.u)
        slot = self.zseg[self.comm.rank * (per + 1): (self.comm.rank + 1) * (per + 1)]
        ops.precond_z_seg(r_loc, self.u, slot, per)
        self.comm.allgather_inplace(self.zseg, per + 1)
        ops.vec_update_p_seg(self.N, per, G, self.p, self.zseg, rz_new, rz_old, restart)

    @staticmethod
    def _check_finite(rz: float, where: str):
        if not np.isfinite(rz):
            # the value is a function of all-gathered numbers only: every rank sees the same one and leaves the loop here together
            raise FloatingPointError(f"r^T P r is not finite {where}: {rz}")

    def precond_apply(self, r_full) -> Tuple[torch.Tensor, float]:
        """z = (Q_ff + noise I)^-1 r on full replicated vectors (the preconditioner seam, conjugate_gradient.py:95-113)."""
        saved = self.r
        self.r = r_full
        z = torch.zeros_like(self.p)
        p_saved, self.p = self.p, z
        try:
            self._precond_and_direction(self.scratch, self.scratch, True)   # p = z, scratch = r^T z from the gathered partials
            self.ops.drop_weighted_operand()
        finally:
            self.r, self.p = saved, p_saved
        return z, float(self.scratch.item())

    def pcg(self, max_error=1.0, max_cg_iter=100, restart_cg_iter=40, b=None) -> Tuple[int, float]:
        """conjugate_gradient.py:41-86 on replicated full vectors; v is updated in place.  b: right-hand side (default y - mean)."""
        ops, N = self.ops, self.N
        if b is None:
            ops.rhs_full(self.b)
        else:
            self.b.copy_(b)
        ops.vec_dot(N, self.v, self.v, self.scratch)
        if float(self.scratch.item()) == 0.0:
            # cold start (models.py:59-68): A v == 0 and r == b exactly, so the mat-vec and its all-reduce are skipped
            # (bit-identical; v is replicated, every rank takes the same branch) - same rule as the fused pcg_impl
            self.r.copy_(self.b)
        else:
            self.matvec(self.v, self.Kv)                           # :57
            ops.vec_residual(N, self.r, self.b, self.Kv)           # :58
        self._precond_and_direction(self.rz, self.rz, True)        # :59, :61
        rz = float(self.rz.item())
        self._check_finite(rz, "at the start of the solve")
        i = 0
        ahead = False
        while 0.5 * rz > max_error and i < max_cg_iter:            # :65
            if not ahead:
                self.matvec(self.p, self.Ap)                       # :66
            ops.vec_dot(N, self.p, self.Ap, self.pAp)              # :67
            restart = restart_cg_iter > 0 and (i % restart_cg_iter == restart_cg_iter - 1)  # :70
            ops.vec_update_v_r(N, self.v, self.r, self.p, self.Ap, self.rz, self.pAp, not restart)  # :68, :72
            if restart:
                self.matvec(self.v, self.Kv)
                ops.vec_residual(N, self.r, self.b, self.Kv)
            self._precond_and_direction(self.rz_new, self.rz, restart)   # :73, :75
            self.rz, self.rz_new = self.rz_new, self.rz            # :76 (the two scalar slots swap roles: no copy kernel)
            # Host test of :65 with look-ahead (same rule as the fused pcg_impl): while the residual is still far above the
            # tolerance the next mat-vec (kernel + all-reduce) is enqueued before the host waits for this iteration's scalar;
            # if the test then says stop it was wasted work on Ap only.  rz comes from gathered partials -> same decision everywhere.
            pending = self._read_scalar_async(self.rz)
            ahead = self.lookahead and (i + 1 < max_cg_iter) and (0.5 * rz > 32.0 * max_error)   # same factor as the library (CGLB_LOOKAHEAD_FACTOR)
            if ahead:
                self.matvec(self.p, self.Ap)
            rz = pending()
            self._check_finite(rz, f"after iteration {i}")
            i += 1
        return i, 0.5 * rz

    def objective_and_grad(self, run_cg=True, max_error=1.0, max_cg_iter=100, restart_cg_iter=40, with_grad=True) -> ShardedResult:
        ops, comm = self.ops, self.comm
        self.setup()
        steps, half = 0, float("nan")
        if run_cg:
            steps, half = self.pcg(max_error, max_cg_iter, restart_cg_iter)
        if run_cg and not self.final_matvec:
            ops.vec_residual(self.N, self.Kv, self.b, self.r)                 # K v = e - r
        else:
            self.matvec(self.v, self.Kv)                                      # models.py:280
        ops.obj_phase1_kv(self.Kv[self.r0:self.r1], self.u)
        comm.allreduce(self.u)
        ops.obj_phase2(self.v, self.u, self.sc, self.aw)
        comm.allreduce(self.sc)
        grad = None
        if with_grad:
            comm.allreduce(self.aw)
            u_loc = self._own(self.ubuf)
            ops.obj_w(u_loc)                                                   # w = P r (local slice)
            ops.vec_axpy(self.nloc, 0.5, self.v[self.r0:self.r1], u_loc)       # u = w + v/2
            comm.allgather_inplace(self.ubuf, self.per)
            ops.obj_phase3_cyclic(self.v, self.ubuf[: self.N], self.sc, self.aw, self.grad)
            comm.allreduce(self.grad)
            grad = self.grad.detach().cpu().numpy().copy()
        bound, lower, upper, logdet = ops.obj_finish(self.sc)
        return ShardedResult(bound, lower, upper, logdet, steps, half, grad)

    def predict(self, xnew) -> Tuple[torch.Tensor, torch.Tensor]:
        """PredictCG.forward after the solve (models.py:334-352) at self.v: K v by the cyclic mat-vec, a_res = A res all-reduced (M),
        then the new points are dealt to the ranks in contiguous slices (each rank evaluates k(x*, X) v over ALL columns and the SGPR
        terms for its slice) and the per-slice (mean, variance) pairs are all-gathered.  Returns full-length tensors on every rank."""
        ops, comm, G = self.ops, self.comm, self.comm.world
        xn = torch.as_tensor(xnew, dtype=ops.dtype).reshape(-1, self.D)
        n_new = int(xn.shape[0])
        self.matvec(self.v, self.Kv)                                          # :335
        ops.predict_u(self.Kv[self.r0:self.r1], self.u)                       # :340
        comm.allreduce(self.u)
        pern = (n_new + G - 1) // G
        a, b = min(comm.rank * pern, n_new), min((comm.rank + 1) * pern, n_new)
        gat = torch.zeros(max(G * 2 * pern, 1), dtype=ops.dtype, device=ops.device)
        if b > a:
            mean, var = ops.predict_rows(self.v, self.u, xn[a:b])
            base = comm.rank * 2 * pern
            gat[base: base + (b - a)].copy_(mean)
            gat[base + pern: base + pern + (b - a)].copy_(var)
        if pern > 0:
            comm.allgather_inplace(gat, 2 * pern)
        seg = gat[: G * 2 * pern].reshape(G, 2, pern) if pern > 0 else gat.reshape(0, 2, 0)
        return seg[:, 0, :].reshape(-1)[:n_new].clone(), seg[:, 1, :].reshape(-1)[:n_new].clone()


class PyDistContext:
    """The `HipContext` surface over the HOST-DRIVEN twin of the N-rank path (SymShardedCGLB + any SymLocalOps): what the backend's
    model classes talk to when the collectives are issued from Python.  With `HipSymLocalOps` it is the step-by-step form of
    `dist_context.DistHipContext` (same kernels, same collectives, one C call per phase); with CPU local ops it lets the backend API
    (LowerBoundCG, optimize, PredictCG, metrics_fn) run at world size > 1 over gloo without a GPU in the tests."""

    def __init__(self, ops: SymLocalOps, comm: Optional[Comm] = None):
        self.ops = ops
        self.drv = SymShardedCGLB(ops, comm)
        self.comm = self.drv.comm
        self.world, self.rank = self.comm.world, self.comm.rank
        self.N, self.M, self.D = ops.N, ops.M, ops.D
        self.device, self.dtype = ops.device, ops.dtype
        self.r0, self.r1 = 0, ops.N       # vectors handed in / out are full length
        self.y = ops.y_full()
        self.noise = None

    def close(self):
        ctx = getattr(self.ops, "ctx", None)
        if ctx is not None:
            ctx.close()

    def _dev(self, t, n=None):
        t = torch.as_tensor(t, dtype=self.dtype, device=self.device).reshape(-1).contiguous()
        if n is not None and t.numel() != n:
            raise ValueError(f"expected a vector of length {n}, got {t.numel()}")
        return t

    def empty(self, n):
        return torch.empty(n, dtype=self.dtype, device=self.device)

    def set_hypers(self, lengthscales, variance, noise, mean, Z, jitter=1e-6):
        self.ops.set_hypers(np.asarray(lengthscales, dtype=np.float64).reshape(-1), float(variance), float(noise), float(mean), Z, float(jitter))
        self.noise = float(noise)

    def setup(self):
        self.drv.setup()

    def matvec(self, p_full, out=None):
        out = self.empty(self.N) if out is None else out
        self.drv.matvec(self._dev(p_full, self.N), out)
        return out

    def precond(self, r):
        return self.drv.precond_apply(self._dev(r, self.N))

    def pcg(self, b, v0, max_error=1.0, max_cg_iter=100, restart_cg_iter=40):
        self.drv.v.copy_(self._dev(v0, self.N))      # the driver's own v: the caller's tensor is not mutated (conjugate_gradient.py:55)
        steps, half = self.drv.pcg(max_error, max_cg_iter, restart_cg_iter, b=self._dev(b, self.N))
        return self.drv.v.clone(), steps, half

    def objective_and_grad(self, v_inout, run_cg=True, max_error=1.0, max_cg_iter=100, restart_cg_iter=40, with_grad=True):
        from .hip_context import ObjectiveResult
        if v_inout.numel() != self.N:
            raise ValueError("v_inout must be a vector of length N")
        self.drv.v.copy_(v_inout.reshape(-1))
        res = self.drv.objective_and_grad(run_cg, max_error, max_cg_iter, restart_cg_iter, with_grad)
        if run_cg:
            v_inout.reshape(-1).copy_(self.drv.v)
        grad = None
        if res.grad is not None:
            D, M, g = self.D, self.M, res.grad
            grad = {"lengthscales": g[:D].copy(), "variance": float(g[D]), "noise": float(g[D + 1]), "mean": float(g[D + 2]),
                    "Z": g[D + 3:].reshape(M, D).copy()}
        return ObjectiveResult(res.bound, res.lower, res.upper, res.logdet, res.steps, res.residual_error, grad)

    def objective_grad_v(self):
        raise NotImplementedError("joint optimisation of v (the TF twin's opt-in) is not available on more than one rank")

    def predict(self, v_full, xnew):
        self.drv.v.copy_(self._dev(v_full, self.N))
        return self.drv.predict(xnew)
