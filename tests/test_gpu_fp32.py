"""fp32 path (config C5 "tolerance relaxed"): same kernels instantiated for float (hardware v_exp_f32), checked against the
fp64 oracle at fp32-level tolerances."""
import numpy as np
import pytest
import torch

from oracle import cglb_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["rbf", "matern32"])
@pytest.mark.parametrize("variant,D", [(0, 8), (2, 8), (2, 16), (2, 12), (2, 24), (2, 3)])
def test_fp32_matvec_and_objective(kind, variant, D):
    """D = 16 is config C5's input dimension; 12/16 run 4 rows per lane in fp32 (2 in fp64), 24 runs 2."""
    from cglb_amd.hip_context import HipContext
    N, M = 1500, 32
    X, y, Z = orc.synthetic_problem(N, D, M, seed=21)
    hyp = orc.trained_like_hypers(D, Z)
    hyp.noise = 0.5
    hyp.jitter = 1e-5  # backend.py:77-79 fp32 jitter
    ctx = HipContext(X, y, M, kind, dtype=torch.float32)
    ctx.set_option("kff_variant", variant)
    ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
    rng = np.random.default_rng(0)
    p = rng.standard_normal(N)
    out = ctx.matvec(torch.from_numpy(p).float()).cpu().numpy().astype(np.float64)
    ref = orc.dense_cov(kind, X, hyp) @ p
    assert np.abs(out - ref).max() <= 2e-4 * np.abs(ref).max()
    v = torch.zeros(N, dtype=torch.float32, device=ctx.device)
    res = ctx.objective_and_grad(v, True, 1.0)
    refo = orc.objective(kind, X, y, hyp, np.zeros(N), True, 1.0)
    assert abs(res.steps - refo.steps) <= 1
    assert res.bound == pytest.approx(refo.bound, rel=2e-3)
    refg = orc.objective(kind, X, y, hyp, v.cpu().numpy().astype(np.float64), run_cg=False, with_grad=True).grad
    np.testing.assert_allclose(res.grad["lengthscales"], refg["lengthscales"], rtol=5e-2, atol=5e-2 * np.abs(refg["lengthscales"]).max())
    assert res.grad["noise"] == pytest.approx(refg["noise"], rel=5e-2)
