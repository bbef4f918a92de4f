"""N-rank contexts behind the backend API: the same surface as `HipContext`, one process per GPU.

`DistHipContext` keeps the whole evaluation inside libcglb_hip.so (cglb_dist_* entry points, include/cglb_hip.h): the library issues
its collectives itself on the context stream - RCCL (communicator created from an id that rank 0 obtains and torch.distributed's
store hands to the others), or, where RCCL cannot run (several ranks sharing one GPU in the tests, a gloo-only process group),
callbacks into torch.distributed.  Vectors handed to / returned by it are full length and replicated: every rank calls the same
methods with the same arguments and obtains the same results, so the ranks of a job can each run the reference's single-process
control flow (SciPy L-BFGS-B, the metric callbacks) on identical numbers (BASELINE config C4: "8 x MI355X, full CGLB train loop").

`make_context` is what the model classes call: a plain `HipContext` in a single-process run, a `DistHipContext` when a process group
with more than one rank is initialised.  The reference has no distributed code; the partitioning is north_star's.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import byref, c_double, c_int, c_void_p
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from .distributed import Comm, row_partition
from .hip_context import HipContext, ObjectiveResult, _ptr, _wrap_device_pointer, grad_len

try:
    import torch.distributed as dist
except Exception:  # pragma: no cover
    dist = None


def dist_world(group=None) -> int:
    if dist is None or not dist.is_available() or not dist.is_initialized():
        return 1
    return dist.get_world_size(group)


class DistHipContext(HipContext):
    """One rank of an N-rank evaluation.  collectives: "rccl" | "callbacks" | "auto" (RCCL when the process group's backend is nccl)."""

    def __init__(self, X, y, num_inducing: int, kind, dtype: torch.dtype = torch.float64, device: Optional[torch.device] = None,
                 group=None, collectives: str = "auto", force: bool = False):
        if dist is None or not dist.is_initialized():
            raise RuntimeError("DistHipContext needs an initialised torch.distributed process group")
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        n = len(X)
        self.per, parts = row_partition(n, self.world)
        super().__init__(X, y, num_inducing, kind, dtype=dtype, device=device, row_range=parts[self.rank])
        if collectives == "auto":
            collectives = os.environ.get("CGLB_COLLECTIVES") or ("rccl" if dist.get_backend(group) == "nccl" else "callbacks")
        self.collectives = collectives
        self._cb_error = None
        if collectives == "rccl":
            ident = [None]
            if self.rank == 0:
                buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
                _lib.check(self.lib.cglb_comm_get_unique_id(buf), None)
                ident[0] = bytes(buf.raw)
            if self.world > 1:
                dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            with torch.cuda.device(self.device):
                _lib.check(self.lib.cglb_comm_init_rccl(self._ctx, ctypes.c_char_p(ident[0]), self.world, self.rank), self._ctx)
        elif collectives == "callbacks":
            self._comm = Comm(group, force=force)
            self._keep = (_lib.ALLREDUCE_FN(self._cb_allreduce), _lib.ALLGATHER_FN(self._cb_allgather))  # must outlive the C context
            _lib.check(self.lib.cglb_comm_init_callbacks(self._ctx, self.world, self.rank, self._keep[0], self._keep[1], None), self._ctx)
        else:
            raise ValueError("collectives must be 'rccl', 'callbacks' or 'auto'")

    # -- collective callbacks (host-provided fabric: torch.distributed on views of the library's device buffers) ---------------
    def _stream_ctx(self, stream):
        s = torch.cuda.ExternalStream(int(stream), device=self.device) if stream else torch.cuda.default_stream(self.device)
        return torch.cuda.stream(s)

    def _cb_allreduce(self, _user, buf, count, dtype, stream):
        try:
            with self._stream_ctx(stream):
                t = _wrap_device_pointer(buf, (int(count),), torch.float64 if dtype == _lib.F64 else torch.float32, self.device)
                self._comm.allreduce(t)
            return 0
        except Exception as exc:  # surfaces as CGLB_ERR_COMM -> RuntimeError from the C call; the cause is attached there
            self._cb_error = exc
            return 1

    def _cb_allgather(self, _user, buf, count, dtype, stream):
        try:
            with self._stream_ctx(stream):
                t = _wrap_device_pointer(buf, (int(count) * self.world,), torch.float64 if dtype == _lib.F64 else torch.float32, self.device)
                self._comm.allgather_inplace(t, int(count))
            return 0
        except Exception as exc:
            self._cb_error = exc
            return 1

    def _check(self, rc):
        if rc != _lib.OK and self._cb_error is not None:
            exc, self._cb_error = self._cb_error, None
            raise RuntimeError(f"collective callback failed: {exc!r}") from exc
        _lib.check(rc, self._ctx)

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self.lib.cglb_comm_destroy(self._ctx)
        super().close()

    # -- the HipContext surface on full replicated vectors -----------------------------------------------------------------
    def setup(self):
        self._check(self.lib.cglb_dist_setup(self._ctx))

    def matvec(self, p_full: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        p = self._dev(p_full, self.N)
        out = self.empty(self.N) if out is None else out
        self._check(self.lib.cglb_dist_matvec(self._ctx, _ptr(p), _ptr(out)))
        return out

    def precond(self, r) -> Tuple[torch.Tensor, float]:
        r = self._dev(r, self.N)
        z = self.empty(self.N)
        rz = c_double()
        self._check(self.lib.cglb_dist_precond_apply(self._ctx, _ptr(r), _ptr(z), byref(rz)))
        return z, rz.value

    def pcg(self, b, v0, max_error=1.0, max_cg_iter=100, restart_cg_iter=40):
        b = self._dev(b, self.N)
        v = self._dev(v0, self.N).clone()  # conjugate_gradient.py:55
        steps, half = c_int(), c_double()
        self._check(self.lib.cglb_dist_pcg_solve(self._ctx, _ptr(b), _ptr(v), float(max_error), int(max_cg_iter), int(restart_cg_iter),
                                                 byref(steps), byref(half)))
        return v, steps.value, half.value

    def objective_and_grad(self, v_inout: torch.Tensor, run_cg=True, max_error=1.0, max_cg_iter=100, restart_cg_iter=40,
                           with_grad=True) -> ObjectiveResult:
        if v_inout.device != self.device or v_inout.dtype != self.dtype or v_inout.numel() != self.N or not v_inout.is_contiguous():
            raise ValueError("v_inout must be a contiguous device vector of length N in the context dtype")
        out4 = (c_double * 4)()
        g = np.empty(grad_len(self.D, self.M), dtype=np.float64) if with_grad else None
        steps, half = c_int(), c_double()
        self._check(self.lib.cglb_dist_objective_and_grad(
            self._ctx, _ptr(v_inout), int(bool(run_cg)), float(max_error), int(max_cg_iter), int(restart_cg_iter), out4,
            g.ctypes.data_as(ctypes.POINTER(c_double)) if with_grad else None, byref(steps), byref(half)))
        return ObjectiveResult(out4[0], out4[1], out4[2], out4[3], steps.value, half.value, self.unpack_grad(g) if with_grad else None)

    def objective_grad_v(self):
        raise NotImplementedError("joint optimisation of v (the TF twin's opt-in) is not available on more than one rank")

    def predict(self, v_full, xnew):
        xn = torch.as_tensor(xnew, dtype=self.dtype).reshape(-1, self.D).contiguous().to(self.device)
        v = self._dev(v_full, self.N)
        mean, var = self.empty(xn.shape[0]), self.empty(xn.shape[0])
        self._check(self.lib.cglb_dist_predict(self._ctx, _ptr(v), _ptr(xn), xn.shape[0], _ptr(mean), _ptr(var)))
        return mean, var

    def get_matrix(self, which: str):
        raise NotImplementedError("common-term matrices are inspected on single-shard contexts")


def make_context(X, y, num_inducing: int, kind, dtype: torch.dtype = torch.float64, device: Optional[torch.device] = None, group=None):
    """The context a model builds on: single-process -> HipContext; inside a torch.distributed job -> one rank of a DistHipContext.
    CGLB_FORCE_DIST=1 takes the N-rank path (and issues its collectives) even at world size 1 - rehearsal on a one-GPU box."""
    force = os.environ.get("CGLB_FORCE_DIST") == "1"
    if dist_world(group) > 1 or (force and dist is not None and dist.is_initialized()):
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        return DistHipContext(X, y, num_inducing, kind, dtype=dtype, device=device, group=group, force=force)
    return HipContext(X, y, num_inducing, kind, dtype=dtype, device=device)
