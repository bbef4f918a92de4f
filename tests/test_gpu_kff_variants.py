"""All K_ff mat-vec kernel variants against the dense oracle / the blocked C oracle, including ragged sizes
(N not a multiple of 16/64/256), D in {1,3,5,8,12,16,24,32} (every padded dimension of dispatch.h up to CGLB_MAX_D), both kernels."""
import numpy as np
import pytest
import torch

from oracle import cglb_oracle as orc
from oracle import cglb_oracle_c as orcc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("kind", ["rbf", "matern32"])
@pytest.mark.parametrize("N,D", [(257, 1), (1000, 3), (4099, 8), (2500, 16), (333, 5), (700, 12), (600, 24), (520, 32)])
def test_matvec_variants(variant, kind, N, D):
    from cglb_amd.hip_context import HipContext
    X, y, Z = orc.synthetic_problem(N, D, 8, seed=N + D)
    rng = np.random.default_rng(1)
    hyp = orc.Hypers(0.6 + rng.random(D), 0.7, 0.3, 0.1, Z, 1e-6)
    ctx = HipContext(X, y, 8, kind)
    ctx.set_option("kff_variant", variant)
    ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, 1e-6)
    p = rng.standard_normal(N)
    out = ctx.matvec(torch.from_numpy(p)).cpu().numpy()
    ref = orcc.kff_matvec(kind, X, hyp, p)
    np.testing.assert_allclose(out, ref, rtol=0, atol=2e-12 * np.abs(ref).max())
    # duplicated points (distance exactly 0) and a far outlier (kernel underflow)
    X2 = X.copy()
    X2[1] = X2[0]
    X2[2] = 50.0
    ctx2 = HipContext(X2, y, 8, kind)
    ctx2.set_option("kff_variant", variant)
    ctx2.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, 1e-6)
    out2 = ctx2.matvec(torch.from_numpy(p)).cpu().numpy()
    ref2 = orcc.kff_matvec(kind, X2, hyp, p)
    np.testing.assert_allclose(out2, ref2, rtol=0, atol=5e-10 * np.abs(ref2).max())  # Gram form with a far outlier: |x|^2 eps
    assert np.isfinite(out2).all()


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_matvec_row_shard_and_linearity(variant):
    from cglb_amd.hip_context import HipContext
    N, D = 3000, 8
    X, y, Z = orc.synthetic_problem(N, D, 8, seed=3)
    h = orc.trained_like_hypers(D, Z)
    rng = np.random.default_rng(2)
    p, q = rng.standard_normal(N), rng.standard_normal(N)
    full = HipContext(X, y, 8, "rbf")
    full.set_option("kff_variant", variant)
    full.set_hypers(h.lengthscales, h.variance, h.noise, h.mean, Z, 1e-6)
    ref = full.matvec(torch.from_numpy(p)).cpu().numpy()
    for r0, r1 in [(0, 1504), (1504, 3000), (1600, 1601)]:
        sh = HipContext(X, y, 8, "rbf", row_range=(r0, r1))
        sh.set_option("kff_variant", variant)
        sh.set_hypers(h.lengthscales, h.variance, h.noise, h.mean, Z, 1e-6)
        np.testing.assert_allclose(sh.matvec(torch.from_numpy(p)).cpu().numpy(), ref[r0:r1], rtol=0, atol=1e-12 * np.abs(ref).max())
    # linearity and symmetry: q^T (A p) == p^T (A q)
    Aq = full.matvec(torch.from_numpy(q)).cpu().numpy()
    Apq = full.matvec(torch.from_numpy(2.0 * p - 0.5 * q)).cpu().numpy()
    np.testing.assert_allclose(Apq, 2.0 * ref - 0.5 * Aq, rtol=0, atol=1e-11 * np.abs(ref).max())
    assert q @ ref == pytest.approx(p @ Aq, rel=1e-11)
    # run-twice determinism (fixed-order combine, no atomics)
    again = full.matvec(torch.from_numpy(p)).cpu().numpy()
    assert np.array_equal(again, ref)


@pytest.mark.parametrize("N,world", [(2999, 2), (2999, 3), (5000, 8), (300, 4)])
def test_cyclic_partials_sum_to_full_matvec(N, world):
    """cglb_matvec_cyclic: the per-rank partial vectors of the cyclic-symmetric split add up to (K_ff + noise I) p (all ranks
    emulated in one process, no collectives; the noise term is in rank 0's partial)."""
    from ctypes import c_void_p
    from cglb_amd import _lib
    from cglb_amd.hip_context import HipContext
    X, y, Z = orc.synthetic_problem(N, 8, 8, seed=N)
    h = orc.trained_like_hypers(8, Z)
    rng = np.random.default_rng(3)
    p = torch.from_numpy(rng.standard_normal(N)).cuda()
    ctx = HipContext(X, y, 8, "rbf")
    ctx.set_hypers(h.lengthscales, h.variance, h.noise, h.mean, Z, 1e-6)
    ref = ctx.matvec(p).cpu().numpy()
    total = np.zeros(N)
    for rank in range(world):
        _lib.check(ctx.lib.cglb_set_parallel(ctx._ctx, world, rank), ctx._ctx)
        out = torch.empty(N, dtype=torch.float64, device=ctx.device)
        _lib.check(ctx.lib.cglb_matvec_cyclic(ctx._ctx, c_void_p(p.data_ptr()), c_void_p(out.data_ptr())), ctx._ctx)
        total += out.cpu().numpy()
    np.testing.assert_allclose(total, ref, rtol=0, atol=1e-12 * np.abs(ref).max())


@pytest.mark.parametrize("half_width", [12.3, 12.7, 40.0])
@pytest.mark.parametrize("kind", ["rbf", "matern32"])
def test_symmetric_matvec_at_the_exponent_range_limit(kind, half_width):
    """The symmetric RBF kernel folds 2^(a_j) into the column operand, so its unweighted factor spans up to +-1000 octaves;
    cglb_set_hypers switches to the clamped variant beyond 0.95 of that.  12.3 / 12.7 straddle the switch for D=2, l=1
    (2 |x|^2 log2(e) = 950 octaves at 12.83 -> with the data's own max), 40 is far inside the clamped regime."""
    from cglb_amd.hip_context import HipContext
    N, D = 1500, 2
    rng = np.random.default_rng(11)
    X = rng.uniform(-half_width, half_width, size=(N, D))
    X[0] = [half_width, half_width]
    X[1] = [-half_width, -half_width]
    X[2] = [half_width, -half_width]
    X[3] = X[0] * (1 - 1e-9)  # near-duplicate of the farthest point: largest unweighted factor times smallest weight
    y = rng.standard_normal(N)
    Z = X[:8].copy()
    hyp = orc.Hypers(np.ones(D), 1.3, 0.2, 0.0, Z, 1e-6)
    ctx = HipContext(X, y, 8, kind)
    ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, 1e-6)
    p = rng.standard_normal(N)
    out = ctx.matvec(torch.from_numpy(p)).cpu().numpy()
    ref = orcc.kff_matvec(kind, X, hyp, p)
    assert np.isfinite(out).all()
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-10 * np.abs(ref).max())
