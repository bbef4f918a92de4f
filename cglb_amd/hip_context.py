"""Thin object wrapper over the C ABI: one HipContext = one cglb_ctx = one GPU row shard.

torch is used for device memory and streams only; every computation is a call into libcglb_hip.so.
"""
from __future__ import annotations

import ctypes
from ctypes import byref, c_double, c_int, c_void_p
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib

KINDS = {"rbf": _lib.RBF, "SquaredExponential": _lib.RBF, "matern32": _lib.MATERN32, "Matern32": _lib.MATERN32,
         "mat32": _lib.MATERN32, 0: _lib.RBF, 1: _lib.MATERN32}


def grad_len(D: int, M: int) -> int:
    return D + 3 + M * D


@dataclass
class ObjectiveResult:
    bound: float
    lower: float
    upper: float
    logdet: float
    steps: int
    residual_error: float
    grad: Optional[dict]  # constrained-space gradient of `bound`


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else c_void_p(t.data_ptr())


class HipContext:
    def __init__(self, X, y, num_inducing: int, kind, dtype: torch.dtype = torch.float64, device: Optional[torch.device] = None,
                 row_range: Optional[Tuple[int, int]] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("cglb_amd needs a HIP device (MI355X); there is no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.dtype = dtype
        self.kind = KINDS[kind]
        X = torch.as_tensor(X, dtype=dtype).reshape(len(X), -1).contiguous()
        y = torch.as_tensor(y, dtype=dtype).reshape(-1).contiguous()
        self.N, self.D = int(X.shape[0]), int(X.shape[1])
        if y.shape[0] != self.N:
            raise ValueError("X and y disagree on the number of rows")
        self.M = int(num_inducing)
        self.r0, self.r1 = (0, self.N) if row_range is None else (int(row_range[0]), int(row_range[1]))
        self.nloc = self.r1 - self.r0
        self._ctx = c_void_p()
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = self.lib.cglb_ctx_create(byref(self._ctx), self.N, self.r0, self.r1, self.D, self.M,
                                          _lib.F64 if dtype == torch.float64 else _lib.F32, self.kind,
                                          self.device.index or 0, c_void_p(stream))
        _lib.check(rc, None)
        Xd, yd = X.to(self.device), y.to(self.device)
        _lib.check(self.lib.cglb_set_data(self._ctx, _ptr(Xd), _ptr(yd)), self._ctx)
        torch.cuda.synchronize(self.device)
        self.y = yd

    # -- lifetime ------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self.lib.cglb_ctx_destroy(self._ctx)
            self._ctx = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers -------------------------------------------------------------------------------------
    def _dev(self, t, n=None) -> torch.Tensor:
        t = torch.as_tensor(t, dtype=self.dtype, device=self.device).reshape(-1).contiguous()
        if n is not None and t.numel() != n:
            raise ValueError(f"expected a vector of length {n}, got {t.numel()}")
        return t

    def empty(self, n) -> torch.Tensor:
        return torch.empty(n, dtype=self.dtype, device=self.device)

    def set_option(self, name: str, value: int):
        _lib.check(self.lib.cglb_set_option(self._ctx, name.encode(), int(value)), self._ctx)

    # -- hypers / common terms -----------------------------------------------------------------------
    def set_hypers(self, lengthscales, variance, noise, mean, Z, jitter=1e-6):
        ls = np.ascontiguousarray(np.broadcast_to(np.asarray(lengthscales, dtype=np.float64).reshape(-1), (self.D,)))
        Zd = torch.as_tensor(Z, dtype=self.dtype).reshape(self.M, self.D).contiguous().to(self.device)
        rc = self.lib.cglb_set_hypers(self._ctx, ls.ctypes.data_as(ctypes.POINTER(c_double)), float(variance), float(noise),
                                      float(mean), _ptr(Zd), float(jitter))
        _lib.check(rc, self._ctx)
        torch.cuda.synchronize(self.device)
        self.noise = float(noise)

    def setup(self):
        _lib.check(self.lib.cglb_setup(self._ctx), self._ctx)

    def setup_local(self):
        _lib.check(self.lib.cglb_shard_setup_local(self._ctx), self._ctx)

    def aat_tensor(self) -> torch.Tensor:
        """Zero-copy view of the library's partial A A^T buffer (for the all-reduce between setup phases)."""
        ptr = self.lib.cglb_aat_buffer(self._ctx)
        return _wrap_device_pointer(ptr, (self.M * self.M,), self.dtype, self.device)

    def setup_finish(self):
        _lib.check(self.lib.cglb_shard_setup_finish(self._ctx), self._ctx)

    def logdet(self) -> float:
        out = c_double()
        _lib.check(self.lib.cglb_logdet(self._ctx, byref(out)), self._ctx)
        return out.value

    def get_matrix(self, which: str) -> torch.Tensor:
        idx = {"A": 0, "L": 1, "LB": 2}[which]
        shape = (self.M, self.nloc) if idx == 0 else (self.M, self.M)
        out = torch.empty(shape, dtype=self.dtype, device=self.device)
        _lib.check(self.lib.cglb_get_matrix(self._ctx, idx, _ptr(out)), self._ctx)
        return out

    # -- operator / preconditioner / solver ----------------------------------------------------------
    def matvec(self, p_full: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        p = self._dev(p_full, self.N)
        out = self.empty(self.nloc) if out is None else out
        _lib.check(self.lib.cglb_matvec(self._ctx, _ptr(p), _ptr(out)), self._ctx)
        return out

    def cross_matvec(self, xnew, v_full) -> torch.Tensor:
        xn = torch.as_tensor(xnew, dtype=self.dtype).reshape(-1, self.D).contiguous().to(self.device)
        v = self._dev(v_full, self.N)
        out = self.empty(xn.shape[0])
        _lib.check(self.lib.cglb_cross_matvec(self._ctx, _ptr(xn), xn.shape[0], _ptr(v), _ptr(out)), self._ctx)
        return out

    def precond(self, r) -> Tuple[torch.Tensor, float]:
        r = self._dev(r, self.N)
        z = self.empty(self.N)
        rz = c_double()
        _lib.check(self.lib.cglb_precond_apply(self._ctx, _ptr(r), _ptr(z), byref(rz)), self._ctx)
        return z, rz.value

    def pcg(self, b, v0, max_error=1.0, max_cg_iter=100, restart_cg_iter=40) -> Tuple[torch.Tensor, int, float]:
        b = self._dev(b, self.N)
        v = self._dev(v0, self.N).clone()  # the reference clones v (conjugate_gradient.py:55)
        steps, half = c_int(), c_double()
        rc = self.lib.cglb_pcg_solve(self._ctx, _ptr(b), _ptr(v), float(max_error), int(max_cg_iter), int(restart_cg_iter),
                                     byref(steps), byref(half))
        _lib.check(rc, self._ctx)
        return v, steps.value, half.value

    def objective_and_grad(self, v_inout: torch.Tensor, run_cg=True, max_error=1.0, max_cg_iter=100, restart_cg_iter=40,
                           with_grad=True) -> ObjectiveResult:
        """v_inout (device, length N) is the persistent warm-start vector: updated in place when run_cg."""
        if v_inout.device != self.device or v_inout.dtype != self.dtype or v_inout.numel() != self.N or not v_inout.is_contiguous():
            raise ValueError("v_inout must be a contiguous device vector of length N in the context dtype")
        out4 = (c_double * 4)()
        g = np.empty(grad_len(self.D, self.M), dtype=np.float64) if with_grad else None
        steps, half = c_int(), c_double()
        rc = self.lib.cglb_objective_and_grad(
            self._ctx, _ptr(v_inout), int(bool(run_cg)), float(max_error), int(max_cg_iter), int(restart_cg_iter), out4,
            g.ctypes.data_as(ctypes.POINTER(c_double)) if with_grad else None, byref(steps), byref(half))
        _lib.check(rc, self._ctx)
        return ObjectiveResult(out4[0], out4[1], out4[2], out4[3], steps.value, half.value, self.unpack_grad(g) if with_grad else None)

    def objective_grad_v(self) -> torch.Tensor:
        """d bound / d v at the v of the evaluation just made (TF twin's joint optimisation of v, tensorflow/models.py:161-164)."""
        out = self.empty(self.N)
        _lib.check(self.lib.cglb_objective_grad_v(self._ctx, _ptr(out)), self._ctx)
        return out

    def unpack_grad(self, g: np.ndarray) -> dict:
        D, M = self.D, self.M
        return {"lengthscales": g[:D].copy(), "variance": float(g[D]), "noise": float(g[D + 1]), "mean": float(g[D + 2]),
                "Z": g[D + 3:].reshape(M, D).copy()}

    def select_inducing(self, lengthscales, variance, jitter=1e-12, return_Z=False):
        """Greedy conditional-variance choice of the M inducing points under the given (initial) kernel - config.py:55-65.
        Returns (indices int64 [min(M, N)], remaining trace) and, if asked, the device tensor Z = X[indices].
        Must be followed by set_hypers before any other call."""
        ls = np.ascontiguousarray(np.broadcast_to(np.asarray(lengthscales, dtype=np.float64).reshape(-1), (self.D,)))
        msel = min(self.M, self.N)
        idx = np.empty(msel, dtype=np.int64)
        Z = torch.empty((msel, self.D), dtype=self.dtype, device=self.device) if return_Z else None
        trace = c_double()
        rc = self.lib.cglb_select_inducing(self._ctx, ls.ctypes.data_as(ctypes.POINTER(c_double)), float(variance), float(jitter),
                                           idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), _ptr(Z), byref(trace))
        _lib.check(rc, self._ctx)
        return (idx, trace.value, Z) if return_Z else (idx, trace.value)

    def predict(self, v_full, xnew) -> Tuple[torch.Tensor, torch.Tensor]:
        xn = torch.as_tensor(xnew, dtype=self.dtype).reshape(-1, self.D).contiguous().to(self.device)
        v = self._dev(v_full, self.N)
        mean, var = self.empty(xn.shape[0]), self.empty(xn.shape[0])
        _lib.check(self.lib.cglb_predict(self._ctx, _ptr(v), _ptr(xn), xn.shape[0], _ptr(mean), _ptr(var)), self._ctx)
        return mean, var

    def get_stat(self, name: str) -> float:
        out = c_double()
        _lib.check(self.lib.cglb_get_stat(self._ctx, name.encode(), byref(out)), self._ctx)
        return out.value

    def time_kernel(self, which: int, reps: int) -> float:
        ms = c_double()
        _lib.check(self.lib.cglb_time_kernel(self._ctx, int(which), int(reps), byref(ms)), self._ctx)
        return ms.value


class _DevPtr:
    """Minimal __cuda_array_interface__ carrier so torch can view library-owned device memory."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def _wrap_device_pointer(ptr, shape, dtype, device) -> torch.Tensor:
    typestr = "<f8" if dtype == torch.float64 else "<f4"
    return torch.as_tensor(_DevPtr(ptr, shape, typestr), device=device)
