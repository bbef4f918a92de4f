"""Inducing-point initialisation on the GPU (cglb_select_inducing) against the numpy statement of the same greedy rule
(oracle.cglb_oracle.greedy_conditional_variance, the restated rule of config.py:55-65) and through properties at full size."""
import numpy as np
import pytest
import torch

from oracle import cglb_oracle as orc

pytestmark = pytest.mark.gpu


def _kernel_fn(kind, ls, var):
    def fn(x1, x2=None, full_cov=False):
        x1 = np.asarray(x1, dtype=np.float64)
        if not full_cov:
            return np.full(x1.shape[0], var)
        return orc.kernel_matrix(kind, x1, x1 if x2 is None else np.asarray(x2, dtype=np.float64), ls, var)
    return fn


@pytest.mark.parametrize("kind", ["rbf", "matern32"])
@pytest.mark.parametrize("N,D,M", [(3000, 3, 100), (700, 8, 64), (1, 2, 1), (50, 1, 50), (257, 16, 9)])
def test_selection_matches_numpy_greedy(kind, N, D, M):
    from oracle.cglb_oracle import greedy_conditional_variance
    from cglb_amd.hip_context import HipContext
    rng = np.random.default_rng(N + D)
    X = rng.standard_normal((N, D))
    ls = 0.7 + rng.random(D)
    var = 1.7
    ctx = HipContext(X, np.zeros(N), M, kind)
    idx, trace, Z = ctx.select_inducing(ls, var, return_Z=True)
    ref = greedy_conditional_variance(X, M, _kernel_fn(kind, ls, var))
    assert len(np.unique(idx)) == min(M, N)
    np.testing.assert_array_equal(X[idx], ref)
    np.testing.assert_array_equal(Z.cpu().numpy(), ref)
    # remaining trace = tr(K_ff - K_fu K_uu^-1 K_uf) with the selected points (the quantity the greedy rule minimises step by step)
    Kuf = orc.kernel_matrix(kind, X[idx], X, ls, var)
    Kuu = orc.kernel_matrix(kind, X[idx], X[idx], ls, var) + 1e-12 * np.eye(len(idx))
    resid = N * var - np.trace(Kuf.T @ np.linalg.solve(Kuu, Kuf))
    assert trace == pytest.approx(resid, rel=1e-6, abs=1e-6 * N * var)
    # the context is usable afterwards (set_hypers re-creates the scaled operands)
    ctx.set_hypers(ls, var, 0.5, 0.0, X[idx], 1e-6)
    ctx.setup()
    assert np.isfinite(ctx.logdet())


def test_selection_fp32_and_duplicates():
    """fp32 contexts and data with fewer distinct points than M: finite, in range; duplicates are only taken once the distinct
    points are exhausted."""
    from cglb_amd.hip_context import HipContext
    rng = np.random.default_rng(5)
    base = rng.standard_normal((20, 2))
    X = np.concatenate([base, base, base])
    ctx = HipContext(X, np.zeros(len(X)), 32, "rbf")
    idx, trace = ctx.select_inducing(np.ones(2), 1.0)
    assert ((0 <= idx) & (idx < len(X))).all() and np.isfinite(trace)
    first = X[idx[:20]]
    assert len(np.unique(first.round(12), axis=0)) == 20
    ctx32 = HipContext(rng.standard_normal((2000, 4)), np.zeros(2000), 64, "matern32", dtype=torch.float32)
    idx32, trace32 = ctx32.select_inducing(np.ones(4), 1.0, jitter=1e-6)
    assert len(np.unique(idx32)) == 64 and np.isfinite(trace32) and 0 <= trace32 < 2000


def test_selection_error_mapping():
    from cglb_amd.hip_context import HipContext
    ctx = HipContext(np.zeros((10, 2)) + np.arange(10)[:, None], np.zeros(10), 4, "rbf")
    with pytest.raises(ValueError):
        ctx.select_inducing(np.array([1.0, -1.0]), 1.0)
    with pytest.raises(ValueError):
        ctx.select_inducing(np.ones(2), 0.0)


def test_selection_full_size_properties():
    """N = 100k, M = 1024 (headline shape): unique indices, the M = 256 choice is a prefix of the M = 1024 one, the trace
    decreases as M grows and equals the Nystrom residual trace recomputed from the common terms."""
    from cglb_amd.hip_context import HipContext
    N, D = 100_000, 8
    X, y, _ = orc.synthetic_problem(N, D, 8, seed=0)
    ls, var = np.ones(D), 1.0
    traces = {}
    for M in (256, 1024):
        ctx = HipContext(X, y, M, "rbf")
        idx, traces[M] = ctx.select_inducing(ls, var)
        assert len(np.unique(idx)) == M
        if M == 256:
            first = idx
        else:
            np.testing.assert_array_equal(idx[:256], first)  # greedy: the M = 256 choice is a prefix of the M = 1024 one
        ctx.close()
    assert 0 < traces[1024] < traces[256] < N * var
    # the reported trace is tr(K_ff - Q_ff) of the selected set: recompute it through the library's own common terms,
    # tr(Q_ff) = noise tr(A A^T) (jitter 1e-6 in K_uu there against 1e-12 in the selection, hence the tolerance)
    ctx = HipContext(X, y, 1024, "rbf")
    ctx.set_hypers(ls, var, 1.0, 0.0, X[idx], 1e-6)
    ctx.setup()
    A = ctx.get_matrix("A")
    resid = N * var - float((A * A).sum())
    assert traces[1024] == pytest.approx(resid, rel=1e-5)
