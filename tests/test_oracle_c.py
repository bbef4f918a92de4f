"""C/OpenMP blocked oracle against the numpy oracle (itself pinned to the reference's golden vectors)."""
import numpy as np
import pytest

from conftest import golden_hypers, load_golden
from oracle import cglb_oracle as orc
from oracle import cglb_oracle_c as orcc


@pytest.mark.parametrize("name", ["rbf_d8_trained", "m32_d8_trained", "m32_d3_random", "c1_snelson_like_m32"])
def test_blocked_pieces_match_dense(name):
    g = load_golden(name)
    hyp = golden_hypers(g)
    kind = int(g["kind"])
    X = g["X"]
    cov = orc.dense_cov(kind, X, hyp)
    p = g["r_test"]
    np.testing.assert_allclose(orcc.kff_matvec(kind, X, hyp, p), cov @ p, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(orcc.kff_matvec(kind, X, hyp, p, 10, 77), (cov @ p)[10:77], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(orcc.kernel_block(kind, hyp.Z, X, hyp), orc.kernel_matrix(kind, hyp.Z, X, hyp.lengthscales, hyp.variance),
                               rtol=1e-14, atol=1e-15)
    Xn = X[:17] + 0.1
    np.testing.assert_allclose(orcc.cross(kind, Xn, X, hyp, p), orc.kernel_matrix(kind, Xn, X, hyp.lengthscales, hyp.variance) @ p,
                               rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("name", ["rbf_d8_trained", "m32_d3_random"])
def test_blocked_objective_and_grad_kff(name):
    g = load_golden(name)
    hyp = golden_hypers(g)
    kind = int(g["kind"])
    ob = orcc.objective_blocked(kind, g["X"], g["y"], hyp, g["v0"], True, float(g["max_error"]))
    assert ob.steps == int(g["steps"])
    assert ob.bound == pytest.approx(float(g["bound"]), rel=1e-10)
    # N^2 gradient piece vs the dense formula
    rng = np.random.default_rng(0)
    u, v = rng.standard_normal(len(g["y"])), rng.standard_normal(len(g["y"]))
    d2 = orc.scaled_sqdist(g["X"], g["X"], hyp.lengthscales)
    W = orc.kernel_grad_factor(kind, d2, hyp.variance) * np.outer(u, v)
    Xs = g["X"] / hyp.lengthscales
    ref = np.array([(W * (Xs[:, d][:, None] - Xs[:, d][None, :]) ** 2).sum() / hyp.lengthscales[d] for d in range(Xs.shape[1])])
    np.testing.assert_allclose(orcc.grad_kff(kind, g["X"], hyp, u, v), ref, rtol=1e-11, atol=1e-11)
