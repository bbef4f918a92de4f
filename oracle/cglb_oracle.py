"""CPU oracle for the CGLB quadratic-term hot path.  TEST INFRASTRUCTURE ONLY.

This module is a plain numpy (fp64, dense) restatement of the reference algorithm.
It is the checker the HIP path is compared with; nothing under ``cglb_amd/`` may
import it.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it.

Parity status
-------------
* PCG loop and Nystrom preconditioner: PINNED.  ``tests/golden/*.npz`` were produced
  by the reference's own ``ConjugateGradient.__call__`` / ``NystromPreconditioner.__call__``
  (cglb/backend/pytorch/conjugate_gradient.py, loaded by file path inside the build
  container by ``oracle/gen_golden.py``); ``tests/test_oracle_golden.py`` checks this
  restatement against them.
* Objective / gradient assembly (cglb/backend/pytorch/models.py:151-286): restated here and
  checked against a torch-autograd dense restatement that calls the reference solver
  (same golden files).  models.py itself cannot be imported (needs gpytorch).
* Kernel closed forms (RBF / Matern-3/2, ARD) live in third-party gpytorch / gpflow
  (requirements.txt:12-13, unpinned, not in the container): **parity unpinned** for the
  kernel evaluation itself; the published closed forms are restated in ``kernel_matrix``.

All citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, Optional, Tuple

import numpy as np
import scipy.linalg as sla

RBF = 0
MATERN32 = 1
KINDS = {"rbf": RBF, "SquaredExponential": RBF, "matern32": MATERN32, "Matern32": MATERN32, "mat32": MATERN32}

SQRT3 = math.sqrt(3.0)


def kind_id(kind) -> int:
    if isinstance(kind, str):
        return KINDS[kind]
    return int(kind)


# --------------------------------------------------------------------------- kernels
def scaled_sqdist(X1: np.ndarray, X2: np.ndarray, ls: np.ndarray) -> np.ndarray:
    """r^2_{ij} = sum_d ((x1_id - x2_jd)/l_d)^2 by direct differences (never negative)."""
    X1s = X1 / ls
    X2s = X2 / ls
    d2 = np.zeros((X1.shape[0], X2.shape[0]), dtype=X1.dtype)
    for d in range(X1.shape[1]):
        diff = X1s[:, d][:, None] - X2s[:, d][None, :]
        d2 += diff * diff
    return d2


def kernel_from_sqdist(kind, d2: np.ndarray, var) -> np.ndarray:
    """K-def.  RBF: var*exp(-r^2/2); Matern32: var*(1+sqrt3 r)exp(-sqrt3 r).

    Call sites in the reference: pytorch/interface.py:207-230 (ScaleKernel(RBF|Matern nu=1.5)),
    models.py:196,200,233,251.
    """
    k = kind_id(kind)
    if k == RBF:
        return var * np.exp(-0.5 * d2)
    r = np.sqrt(d2)
    return var * (1.0 + SQRT3 * r) * np.exp(-SQRT3 * r)


def kernel_matrix(kind, X1, X2, ls, var) -> np.ndarray:
    return kernel_from_sqdist(kind, scaled_sqdist(X1, X2, np.asarray(ls, dtype=X1.dtype)), var)


def kernel_diag(kind, X, var) -> np.ndarray:
    """k(x,x) = var for both kernels (models.py:233)."""
    return np.full((X.shape[0],), var, dtype=X.dtype)


# --------------------------------------------------------------------------- parameters
@dataclass
class Hypers:
    """Constrained hyper-parameters (what the reference keeps behind softplus transforms)."""

    lengthscales: np.ndarray  # [D]
    variance: float  # sigma_f^2  (ScaleKernel.outputscale)
    noise: float  # sigma^2    (likelihood.noise)
    mean: float  # ConstantMean.constant
    Z: np.ndarray  # [M, D] inducing points
    jitter: float = 1e-6  # backend.py:77-79

    def copy(self) -> "Hypers":
        return Hypers(np.array(self.lengthscales, copy=True), float(self.variance), float(self.noise),
                      float(self.mean), np.array(self.Z, copy=True), float(self.jitter))


@dataclass
class CommonTerms:
    """models.py:90-95"""

    A: np.ndarray  # [M, N]
    LB: np.ndarray  # [M, M] lower
    AAt_diag_sum: float
    L: np.ndarray  # [M, M] lower


def common_terms(kind, X, hyp: Hypers) -> CommonTerms:
    """models.py:176-213 (TF twin tensorflow/models.py:58-75)."""
    M = hyp.Z.shape[0]
    sigma = math.sqrt(hyp.noise)
    kuf = kernel_matrix(kind, hyp.Z, X, hyp.lengthscales, hyp.variance)  # :196-197
    kuu = kernel_matrix(kind, hyp.Z, hyp.Z, hyp.lengthscales, hyp.variance)
    kuu = kuu + hyp.jitter * np.eye(M, dtype=X.dtype)  # :200-201
    L = np.linalg.cholesky(kuu)  # :202
    A = sla.solve_triangular(L, kuf, lower=True) / sigma  # :206
    AAt = A @ A.T  # :207
    B = AAt + np.eye(M, dtype=X.dtype)  # :208-209
    LB = np.linalg.cholesky(B)  # :210
    return CommonTerms(A=A, LB=LB, AAt_diag_sum=float(np.trace(AAt)), L=L)  # :211-213


def logdet_estimator(kind, X, hyp: Hypers, terms: CommonTerms) -> float:
    """models.py:215-244."""
    N = X.shape[0]
    kdiag_sum = float(kernel_diag(kind, X, hyp.variance).sum())
    trace = kdiag_sum / hyp.noise - terms.AAt_diag_sum  # :236
    logdet = -float(np.log(np.diag(terms.LB)).sum())  # :239
    logdet -= 0.5 * N * math.log(hyp.noise)  # :240
    logdet -= 0.5 * N * math.log(1.0 + trace / N)  # :243
    return logdet


# --------------------------------------------------------------------------- preconditioner / PCG
def nystrom_precond(A: np.ndarray, LB: np.ndarray, sigma_sq: float, r: np.ndarray) -> Tuple[np.ndarray, float]:
    """conjugate_gradient.py:95-113.  r: [N] -> (z [N], rz)."""
    Ar = A @ r  # :105
    LBinvAr = sla.solve_triangular(LB, Ar, lower=True)  # :106
    t = sla.solve_triangular(LB.T, LBinvAr, lower=False)  # :107
    p = t @ A  # :110
    rp = r - p  # :111
    rpr = float((rp * r).sum())  # :112
    return rp / sigma_sq, rpr / sigma_sq  # :113


@dataclass
class PCGStats:
    """conjugate_gradient.py:25-28"""

    steps: int
    residual_error: float


def pcg(matvec: Callable[[np.ndarray], np.ndarray], b: np.ndarray, v0: np.ndarray,
        precond: Callable[[np.ndarray], Tuple[np.ndarray, float]],
        max_error: float = 1.0, max_cg_iter: int = 100, restart_cg_iter: int = 40,
        history: Optional[list] = None) -> Tuple[np.ndarray, PCGStats]:
    """conjugate_gradient.py:41-86, operation for operation (vectors are [N]).  `history` (test aid, not in the reference): receives
    the stop statistic 1/2 r^T P r every time the predicate of :65 reads it, so history[k] is the value tested before iteration k."""
    v = v0.copy()  # :55
    Av = matvec(v)  # :57
    r = b - Av  # :58
    z, rz = precond(r)  # :59
    p = z  # :61
    i = 0
    if history is not None:
        history.append(0.5 * rz)
    while (0.5 * rz > max_error) and (i < max_cg_iter):  # :65
        Ap = matvec(p)  # :66
        gamma = rz / float((p * Ap).sum())  # :67
        v = v + gamma * p  # :68
        restart = i % restart_cg_iter == restart_cg_iter - 1  # :70
        r = (b - matvec(v)) if restart else (r - gamma * Ap)  # :72
        z, new_rz = precond(r)  # :73
        p = z if restart else (z + p * new_rz / rz)  # :75
        rz = new_rz  # :76
        i += 1  # :77
        if history is not None:
            history.append(0.5 * rz)
    return v, PCGStats(steps=i, residual_error=0.5 * rz)  # :83-86


# --------------------------------------------------------------------------- objective
@dataclass
class Objective:
    bound: float  # LowerBoundCG.forward return (models.py:169)
    lower: float  # lower bound on 1/2 e^T K^-1 e (models.py:283)
    upper: float  # upper bound (models.py:284)
    logdet: float
    const: float
    steps: int
    residual_error: float
    v: np.ndarray
    grad: Optional[Dict[str, np.ndarray]] = None


def dense_cov(kind, X, hyp: Hypers) -> np.ndarray:
    """K_ff + sigma^2 I (models.py:251-252)."""
    K = kernel_matrix(kind, X, X, hyp.lengthscales, hyp.variance)
    K[np.diag_indices_from(K)] += hyp.noise
    return K


def objective(kind, X, y, hyp: Hypers, v0: np.ndarray, run_cg: bool = True,
              max_error: float = 1.0, max_cg_iter: int = 100, restart_cg_iter: int = 40,
              with_grad: bool = False, cov: Optional[np.ndarray] = None) -> Objective:
    """LowerBoundCG.forward, models.py:151-174, with quad_estimator :246-286 inlined."""
    N = X.shape[0]
    terms = common_terms(kind, X, hyp)  # :155
    const = -0.5 * N * math.log(2.0 * math.pi)  # :162-163
    logdet = logdet_estimator(kind, X, hyp, terms)  # :165
    if cov is None:
        cov = dense_cov(kind, X, hyp)
    err = y.reshape(-1) - hyp.mean  # :253-254
    precon = lambda r: nystrom_precond(terms.A, terms.LB, hyp.noise, r)  # :260
    if run_cg:  # :262-278
        v, stats = pcg(lambda x: cov @ x, err, v0.reshape(-1), precon, max_error, max_cg_iter, restart_cg_iter)
    else:
        v, stats = v0.reshape(-1).copy(), PCGStats(0, float("nan"))
    cov_v = cov @ v  # :280
    r = err - cov_v  # :281
    w, error_bound = precon(r)  # :282
    lower = float((v * (r + 0.5 * cov_v)).sum())  # :283
    upper = lower + 0.5 * error_bound  # :284
    bound = -upper + logdet + const  # :286, :169
    out = Objective(bound=bound, lower=lower, upper=upper, logdet=logdet, const=const,
                    steps=stats.steps, residual_error=stats.residual_error, v=v)
    if with_grad:
        out.grad = objective_grad(kind, X, hyp, terms, v, w)
        # TF twin (tensorflow/models.py:161-164, `joint_optimization`): v itself is a variable there; d bound / d v = K w - r
        out.grad["v"] = cov @ w - r
    return out


# --------------------------------------------------------------------------- round-off sensitivity of the oracle itself
@dataclass
class Sensitivity:
    """What the oracle's OWN answers do when its inputs move at the round-off level (`roundoff_sensitivity`)."""

    bound: float  # unperturbed
    steps: int
    history: list  # 1/2 r^T P r read by the stop test before iteration k (unperturbed run)
    bound_spread: float  # max over probes |bound_p - bound|
    steps_spread: int  # max over probes |steps_p - steps|
    stat_rel_spread: np.ndarray  # per k: max over probes |h_p[k] - h[k]| / h[k] on the common prefix of the histories


def roundoff_sensitivity(kind, X, y, hyp: Hypers, v0: np.ndarray, max_error: float = 1.0, max_cg_iter: int = 100,
                         restart_cg_iter: int = 40, delta: float = 2.0 ** -52, probes: int = 4, seed: int = 0,
                         cov: Optional[np.ndarray] = None, calibrate: Optional[Tuple[np.ndarray, float]] = None) -> Sensitivity:
    """Parity criterion for CG-path quantities, derived instead of tuned.  A PCG solve is a chaotic map once Lanczos orthogonality is
    lost (weak preconditioner, tens of steps): two CORRECT implementations whose mat-vecs agree to `delta` relative end at bounds that
    differ by far more than `delta`, and may stop one step apart when the stop statistic passes the tolerance within its own noise.
    How much is measured on the oracle itself: the solve of conjugate_gradient.py:41-86 is repeated `probes` times with the dense
    operator multiplied entry-wise by (1 + delta E), E symmetric with entries uniform in [-1, 1], and the bound (models.py:280-286,
    assembled with the unperturbed operator at each probe's v) and step count are compared with the unperturbed run.
    A test may then accept |bound_hip - bound| <= k * bound_spread and a step difference that a probe shows as well - and otherwise
    holds the HIP path to north_star's 1e-6 and to the exact step count.  `delta`: the relative accuracy of the kernel values the
    implementation under test documents (2^-52 for exactly rounded values)."""
    N = X.shape[0]
    if cov is None:
        cov = dense_cov(kind, X, hyp)
    terms = common_terms(kind, X, hyp)
    err = y.reshape(-1) - hyp.mean
    precon = lambda r: nystrom_precond(terms.A, terms.LB, hyp.noise, r)

    def bound_at(v):
        cov_v = cov @ v
        r = err - cov_v
        _, eb = precon(r)
        lower = float((v * (r + 0.5 * cov_v)).sum())
        return -(lower + 0.5 * eb)  # the v-dependent part of the bound (logdet and const do not depend on v)

    h0: list = []
    v, st = pcg(lambda x: cov @ x, err, v0.reshape(-1), precon, max_error, max_cg_iter, restart_cg_iter, history=h0)
    b0 = bound_at(v)
    rng = np.random.default_rng(seed)
    b_spread, s_spread = 0.0, 0
    rel = np.zeros(len(h0))
    for probe in range(probes):
        E = rng.uniform(-1.0, 1.0, size=(N, N))
        E = np.triu(E) + np.triu(E, 1).T
        if probe == 0 and calibrate is not None:
            pc, target = calibrate
            ref_mv = cov @ pc
            dev = float(np.abs((cov * E) @ pc).max() / np.abs(ref_mv).max())   # deviation per unit amplitude
            if dev > 0.0:
                delta = max(delta, float(target) / dev)
        covp = cov * (1.0 + delta * E)
        del E
        hp: list = []
        vp, stp = pcg(lambda x: covp @ x, err, v0.reshape(-1), precon, max_error, max_cg_iter, restart_cg_iter, history=hp)
        b_spread = max(b_spread, abs(bound_at(vp) - b0))
        s_spread = max(s_spread, abs(stp.steps - st.steps))
        n = min(len(hp), len(h0))
        rel[:n] = np.maximum(rel[:n], np.abs(np.asarray(hp[:n]) - np.asarray(h0[:n])) / np.maximum(np.abs(h0[:n]), 1e-300))
        del covp
    logdet = logdet_estimator(kind, X, hyp, terms)
    const = -0.5 * N * math.log(2.0 * math.pi)
    return Sensitivity(bound=b0 + logdet + const, steps=st.steps, history=h0, bound_spread=b_spread, steps_spread=s_spread,
                       stat_rel_spread=rel)


def grad_roundoff_spread(kind, X, hyp: Hypers, v: np.ndarray, w: np.ndarray, probes: int = 3, seed: int = 0,
                         delta: float = 2.0 ** -52, inducing_only: bool = False) -> Dict[str, float]:
    """Absolute noise floor of the analytic gradient at a FIXED (v, w): the largest change of each gradient block when the inducing
    points and lengthscales move by `delta` relative (the backward error any Cholesky / triangular solve of K_uu commits).  With
    cond(K_uu) ~ 1e8 (inducing points on nearly every datum) the Z gradient carries ~cond * eps of absolute error in ANY
    implementation; a test accepts a deviation of k times this floor and otherwise holds the HIP gradient to 1e-6 relative.
    inducing_only: perturb Z alone and leave the N^2 K_ff term (which does not depend on Z) out of both sides - the same floor for the
    inducing-point part at sizes where the dense N x N form cannot be built (the headline-size fixtures)."""
    if inducing_only:
        base = objective_grad(kind, X, hyp, common_terms(kind, X, hyp), v, w, skip_kff=True)
        rng = np.random.default_rng(seed)
        out = {k: 0.0 for k in base}
        for _ in range(probes):
            hp = hyp.copy()
            hp.Z = hp.Z * (1.0 + delta * rng.uniform(-1.0, 1.0, size=hp.Z.shape))
            g = objective_grad(kind, X, hp, common_terms(kind, X, hp), v, w, skip_kff=True)
            for k in base:
                out[k] = max(out[k], float(np.max(np.abs(np.asarray(g[k]) - np.asarray(base[k])))))
        return out
    base = objective_grad(kind, X, hyp, common_terms(kind, X, hyp), v, w)
    rng = np.random.default_rng(seed)
    out = {k: 0.0 for k in base}
    for _ in range(probes):
        hp = hyp.copy()
        hp.Z = hp.Z * (1.0 + delta * rng.uniform(-1.0, 1.0, size=hp.Z.shape))
        hp.lengthscales = hp.lengthscales * (1.0 + delta * rng.uniform(-1.0, 1.0, size=hp.lengthscales.shape))
        g = objective_grad(kind, X, hp, common_terms(kind, X, hp), v, w)
        for k in base:
            out[k] = max(out[k], float(np.max(np.abs(np.asarray(g[k]) - np.asarray(base[k])))))
    return out


# --------------------------------------------------------------------------- analytic gradient (row G)
def kernel_grad_factor(kind, d2: np.ndarray, var) -> np.ndarray:
    """h_ij such that dk/dl_d = h*delta_d^2/l_d and dk/dx1_d = -h*delta_d/l_d,
    delta_d = (x1_d - x2_d)/l_d.  RBF: h = k.  Matern32: h = 3 var exp(-sqrt3 r)."""
    if kind_id(kind) == RBF:
        return var * np.exp(-0.5 * d2)
    return 3.0 * var * np.exp(-SQRT3 * np.sqrt(d2))


def objective_grad(kind, X, hyp: Hypers, terms: CommonTerms, v: np.ndarray, w: np.ndarray, blocked=None, skip_kff: bool = False) -> Dict[str, np.ndarray]:
    """Gradient of ``bound`` wrt the constrained hypers with v held constant.

    This is what ``torch.autograd.grad(loss, variables)`` (pytorch/optimizer.py:95-98) yields
    for ``-loss`` when v is detached (models.py:257-274), written analytically:

      d bound = (v+w)^T 1 dmu + (w+v/2)^T dKff v + [(w+v/2)^T v + w^T w/2] ds
                + c^T dKuf w - c^T dKuu c/2  - 1/2 dlog|B| - N/(2s) ds - N/(2tau) dtau

    with w = P r, c = Kuu^-1 Kuf w, tau = 1 + f/s - tr(AA^T)/N, s = noise, f = variance.
    Returns d bound / d{lengthscales, variance, noise, mean, Z}.

    `blocked`: optional module with `grad_kff` / `kff_matvec` (oracle/cglb_oracle_c.py) — the two N^2 pieces are then
    streamed instead of formed densely, so that the same formula runs at N = 100k.
    """
    N, D = X.shape
    M = hyp.Z.shape[0]
    s, f, ls = hyp.noise, hyp.variance, np.asarray(hyp.lengthscales, dtype=X.dtype)
    sigma = math.sqrt(s)
    A, LB, L, T = terms.A, terms.LB, terms.L, terms.AAt_diag_sum
    tau = 1.0 + f / s - T / N
    u = w + 0.5 * v

    eyeM = np.eye(M, dtype=X.dtype)
    LBinv = sla.solve_triangular(LB, eyeM, lower=True)
    Binv = LBinv.T @ LBinv
    Linv = sla.solve_triangular(L, eyeM, lower=True)
    c = sigma * (Linv.T @ (A @ w))  # Kuu^-1 Kuf w
    # adjoints of Kuu and Kuf
    inner = 0.5 * (eyeM - Binv) - (0.5 / tau) * (A @ A.T)
    Guu = -0.5 * np.outer(c, c) + Linv.T @ inner @ Linv
    Guf = np.outer(c, w) + (Linv.T @ ((eyeM / tau - Binv) @ A)) / sigma

    g_ls = np.zeros(D, dtype=X.dtype)
    g_Z = np.zeros((M, D), dtype=X.dtype)

    Xs, Zs = X / ls, hyp.Z / ls
    # N^2 bilinear form (w + v/2)^T dKff v  (skip_kff: left out - for DIFFERENCES of gradients under perturbations of Z, on which it does not depend)
    if skip_kff:
        g_f = 0.0
    elif blocked is not None:
        g_ls += blocked.grad_kff(kind, X, hyp, u, v)
        g_f = float(u @ (blocked.kff_matvec(kind, X, hyp, v) - s * v)) / f
    else:
        d2 = scaled_sqdist(X, X, ls)
        h = kernel_grad_factor(kind, d2, f)
        k_ff = kernel_from_sqdist(kind, d2, f)
        Wff = h * np.outer(u, v)
        for d in range(D):
            delta = Xs[:, d][:, None] - Xs[:, d][None, :]
            g_ls[d] += float((Wff * delta * delta).sum()) / ls[d]
        g_f = float(u @ (k_ff @ v)) / f
        del d2, h, k_ff, Wff
    # Kuf part
    d2 = scaled_sqdist(hyp.Z, X, ls)
    h = kernel_grad_factor(kind, d2, f)
    k_uf = kernel_from_sqdist(kind, d2, f)
    Wuf = h * Guf
    for d in range(D):
        delta = Zs[:, d][:, None] - Xs[:, d][None, :]
        g_ls[d] += float((Wuf * delta * delta).sum()) / ls[d]
        g_Z[:, d] += -(Wuf * delta).sum(axis=1) / ls[d]
    g_f += float((Guf * k_uf).sum()) / f
    # Kuu part (symmetric adjoint; z_m appears in row and column)
    d2 = scaled_sqdist(hyp.Z, hyp.Z, ls)
    h = kernel_grad_factor(kind, d2, f)
    k_uu = kernel_from_sqdist(kind, d2, f)
    Wuu = h * Guu
    for d in range(D):
        delta = Zs[:, d][:, None] - Zs[:, d][None, :]
        g_ls[d] += float((Wuu * delta * delta).sum()) / ls[d]
        g_Z[:, d] += -2.0 * (Wuu * delta).sum(axis=1) / ls[d]
    g_f += float((Guu * k_uu).sum()) / f

    g_f += -N / (2.0 * tau * s)
    g_s = float(u @ v) + 0.5 * float(w @ w) + (M - float(np.trace(Binv))) / (2.0 * s) - N / (2.0 * s) \
        + N * f / (2.0 * tau * s * s) - T / (2.0 * tau * s)
    g_mu = float((v + w).sum())
    return {"lengthscales": g_ls, "variance": np.float64(g_f), "noise": np.float64(g_s),
            "mean": np.float64(g_mu), "Z": g_Z}


# --------------------------------------------------------------------------- prediction
def predict(kind, X, y, hyp: Hypers, v0: np.ndarray, Xnew: np.ndarray, max_error: float = 1e-3,
            max_cg_iter: int = 100, restart_cg_iter: int = 40):
    """PredictCG.forward, models.py:307-354 (TF twin tensorflow/models.py:194-246)."""
    err = y.reshape(-1) - hyp.mean  # :318
    ksf = kernel_matrix(kind, Xnew, X, hyp.lengthscales, hyp.variance)  # :320
    cov = dense_cov(kind, X, hyp)  # :321
    terms = common_terms(kind, X, hyp)  # :327
    precon = lambda r: nystrom_precond(terms.A, terms.LB, hyp.noise, r)
    new_v, stats = pcg(lambda x: cov @ x, err, v0.reshape(-1), precon, max_error, max_cg_iter, restart_cg_iter)
    cg_mean = ksf @ new_v  # :334
    res = err - cov @ new_v  # :335
    kus = kernel_matrix(kind, hyp.Z, Xnew, hyp.lengthscales, hyp.variance)  # :337
    sigma = math.sqrt(hyp.noise)
    a_res = terms.A @ res  # :340
    c = sla.solve_triangular(terms.LB, a_res, lower=True) / sigma  # :343
    tmp1 = sla.solve_triangular(terms.L, kus, lower=True)  # :344
    tmp2 = sla.solve_triangular(terms.LB, tmp1, lower=True)  # :345
    sgpr_mean = tmp2.T @ c  # :347
    f_mean = sgpr_mean + cg_mean + hyp.mean  # :348
    f_var = kernel_diag(kind, Xnew, hyp.variance) + (tmp2 ** 2).sum(0) - (tmp1 ** 2).sum(0)  # :350-351
    return f_mean, f_var, new_v, stats


def gaussian_log_density(x, mu, var):
    """models.py:375-379"""
    return -0.5 * (math.log(2 * math.pi) + np.log(var) + (mu - x) ** 2 / var)


# --------------------------------------------------------------------------- inducing-point initialisation
def greedy_conditional_variance(X: np.ndarray, M: int, kernel_fn: Callable, jitter: float = 1e-12) -> np.ndarray:
    """Greedy inducing-point selection: repeatedly take the point with the largest conditional variance given the points chosen
    so far == pivoted Cholesky of K_ff, lowest index on ties - the deterministic `sample=False` rule the reference requests from
    robustgp.ConditionalVariance (config.py:62-65).  kernel_fn(x1, x2, full_cov) follows the reference callback
    (pytorch/interface.py:278-284): x2=None, full_cov=False -> diag; full_cov=True -> matrix.  O(N M^2) time, O(N M) memory.

    PARITY UNPINNED: robustgp is third-party (requirements.txt:15, un-pinned git URL) and absent from the container; this restates
    its published algorithm.  Known difference: as published, ConditionalVariance shuffles the inputs before the greedy pass, so with
    a constant k(x, x) its FIRST pick is a random row; this statement (and the HIP kernel it checks) starts from row 0.  The later
    picks are the arg-max of the conditional variance either way.
    """
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


# --------------------------------------------------------------------------- synthetic inputs
def synthetic_problem(N: int, D: int, M: int, seed: int = 0, dtype=np.float64):
    """Deterministic synthetic regression set: X~N(0,1), y=sin(Xa)+0.1eps z-normalised,
    Z = first M rows of a seeded permutation of X (stand-in for robustgp greedy init, config.py:62-65)."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D))
    a = rng.standard_normal(D) / math.sqrt(D)
    y = np.sin(X @ a) + 0.1 * rng.standard_normal(N)
    y = (y - y.mean()) / y.std()
    perm = rng.permutation(N)
    Z = X[perm[:M]].copy()
    return X.astype(dtype), y.astype(dtype), Z.astype(dtype)


def reference_init_hypers(D: int, Z: np.ndarray, jitter: float = 1e-6) -> Hypers:
    """config.py:74-76 (variance=1, lengthscales=1), :104-107 (noise=1); mean 0."""
    return Hypers(lengthscales=np.ones(D), variance=1.0, noise=1.0, mean=0.0, Z=Z.copy(), jitter=jitter)


def trained_like_hypers(D: int, Z: np.ndarray, jitter: float = 1e-6) -> Hypers:
    """SURVEY 8d 'trained-like' point: l=1.5, var=1, noise=0.05."""
    return Hypers(lengthscales=np.full(D, 1.5), variance=1.0, noise=0.05, mean=0.0, Z=Z.copy(), jitter=jitter)
