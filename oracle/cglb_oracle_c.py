"""ctypes wrapper of oracle/_build/libcglb_oracle.so (C/OpenMP blocked oracle).  TEST INFRASTRUCTURE ONLY.

Used by tests at sizes where the dense numpy oracle does not fit, and by bench.py's `cpu_baseline` leg
(kind "port": the reference's Python cannot run on the GPU box)."""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_double, c_int, c_int64

import numpy as np

from . import cglb_oracle as orc

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcglb_oracle.so")
_lib = None
_dp = POINTER(c_double)


def build():
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        lib = ctypes.CDLL(_SO)
        lib.orc_kff_matvec.argtypes = [c_int, c_int64, c_int, _dp, _dp, c_double, c_double, _dp, c_int64, c_int64, _dp]
        lib.orc_cross.argtypes = [c_int, c_int64, c_int64, c_int, _dp, _dp, _dp, c_double, _dp, _dp]
        lib.orc_kernel_block.argtypes = [c_int, c_int64, c_int64, c_int, _dp, _dp, _dp, c_double, _dp]
        lib.orc_grad_kff.argtypes = [c_int, c_int64, c_int, _dp, _dp, c_double, _dp, _dp, c_int64, c_int64, _dp]
        lib.orc_num_threads.restype = c_int
        lib.orc_set_num_threads.argtypes = [c_int]
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def num_threads() -> int:
    return load().orc_num_threads()


def set_num_threads(n: int):
    load().orc_set_num_threads(int(n))


def kff_matvec(kind, X, hyp: orc.Hypers, p, r0=0, r1=None):
    """(K_ff + noise I)[r0:r1, :] @ p without forming K_ff."""
    X, p, ls = _c(X), _c(p), _c(hyp.lengthscales)
    N, D = X.shape
    r1 = N if r1 is None else r1
    out = np.empty(r1 - r0)
    rc = load().orc_kff_matvec(orc.kind_id(kind), N, D, _p(X), _p(ls), hyp.variance, hyp.noise, _p(p), r0, r1, _p(out))
    assert rc == 0
    return out


def cross(kind, X1, X2, hyp: orc.Hypers, v):
    X1, X2, v, ls = _c(X1), _c(X2), _c(v), _c(hyp.lengthscales)
    out = np.empty(X1.shape[0])
    rc = load().orc_cross(orc.kind_id(kind), X1.shape[0], X2.shape[0], X1.shape[1], _p(X1), _p(X2), _p(ls), hyp.variance, _p(v), _p(out))
    assert rc == 0
    return out


def kernel_block(kind, X1, X2, hyp: orc.Hypers):
    X1, X2, ls = _c(X1), _c(X2), _c(hyp.lengthscales)
    out = np.empty((X1.shape[0], X2.shape[0]))
    rc = load().orc_kernel_block(orc.kind_id(kind), X1.shape[0], X2.shape[0], X1.shape[1], _p(X1), _p(X2), _p(ls), hyp.variance, _p(out))
    assert rc == 0
    return out


def grad_kff(kind, X, hyp: orc.Hypers, u, v, r0=0, r1=None):
    X, u, v, ls = _c(X), _c(u), _c(v), _c(hyp.lengthscales)
    N, D = X.shape
    r1 = N if r1 is None else r1
    out = np.empty(D)
    rc = load().orc_grad_kff(orc.kind_id(kind), N, D, _p(X), _p(ls), hyp.variance, _p(u), _p(v), r0, r1, _p(out))
    assert rc == 0
    return out


def common_terms(kind, X, hyp: orc.Hypers) -> orc.CommonTerms:
    """models.py:176-213 with the kernel blocks from the C oracle and LAPACK (scipy) for the dense algebra."""
    import math
    import scipy.linalg as sla
    M = hyp.Z.shape[0]
    kuf = kernel_block(kind, hyp.Z, X, hyp)
    kuu = kernel_block(kind, hyp.Z, hyp.Z, hyp) + hyp.jitter * np.eye(M)
    L = np.linalg.cholesky(kuu)
    A = sla.solve_triangular(L, kuf, lower=True, overwrite_b=True) / math.sqrt(hyp.noise)
    AAt = A @ A.T
    LB = np.linalg.cholesky(AAt + np.eye(M))
    return orc.CommonTerms(A=A, LB=LB, AAt_diag_sum=float(np.trace(AAt)), L=L)


def objective_blocked(kind, X, y, hyp: orc.Hypers, v0, run_cg=True, max_error=1.0, max_cg_iter=100, restart_cg_iter=40):
    """LowerBoundCG.forward (models.py:151-286) with the implicit blocked operator: value only."""
    import math
    N = X.shape[0]
    terms = common_terms(kind, X, hyp)
    logdet = orc.logdet_estimator(kind, X, hyp, terms)
    err = y.reshape(-1) - hyp.mean
    matvec = lambda p: kff_matvec(kind, X, hyp, p)
    precon = lambda r: orc.nystrom_precond(terms.A, terms.LB, hyp.noise, r)
    if run_cg:
        v, stats = orc.pcg(matvec, err, np.asarray(v0, dtype=np.float64).reshape(-1), precon, max_error, max_cg_iter, restart_cg_iter)
    else:
        v, stats = np.asarray(v0, dtype=np.float64).reshape(-1).copy(), orc.PCGStats(0, float("nan"))
    cov_v = matvec(v)
    r = err - cov_v
    w, eb = precon(r)
    lower = float((v * (r + 0.5 * cov_v)).sum())
    upper = lower + 0.5 * eb
    const = -0.5 * N * math.log(2 * math.pi)
    return orc.Objective(bound=-upper + logdet + const, lower=lower, upper=upper, logdet=logdet, const=const,
                         steps=stats.steps, residual_error=stats.residual_error, v=v)
