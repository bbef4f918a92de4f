"""Inputs wider than 32 dimensions (the reference accepts any Wilson UCI set, datasets.py:47-76: buzz 77, song 90, slice 385 ...): the
Gram part of the pair value goes through rocBLAS in tiles (cglb_amd/csrc/kernels_wide.hip); up to 96 dimensions (fp64) the two N^2
passes - the symmetric K_ff mat-vec and the K_ff gradient pass - stay register-resident, their column operands handed out by
v_fmac_f64 row_newbcast (kernels_kff_sym.hip / kernels_grad.hip, option "wide_reg").  Every seam against the dense oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import cglb_oracle as orc

pytestmark = pytest.mark.gpu


def _problem(N, D, M, seed, ell_factor=1.0):
    X, y, Z = orc.synthetic_problem(N, D, M, seed=seed)
    rng = np.random.default_rng(seed + 100)
    ls = rng.uniform(0.8, 1.6, size=D) * np.sqrt(D) * ell_factor       # lengthscales ~ sqrt(D): correlated over the data range
    return X, y, Z, orc.Hypers(ls, 1.3, 0.08, 0.15, Z, 1e-6)


@pytest.mark.parametrize("kind", ["rbf", "matern32"])
@pytest.mark.parametrize("N,D,M", [(700, 33, 24), (1500, 77, 48), (600, 385, 32)])
def test_wide_inputs_every_seam_against_the_oracle(kind, N, D, M):
    from cglb_amd.hip_context import HipContext
    X, y, Z, hyp = _problem(N, D, M, seed=D)
    ctx = HipContext(X, y, M, kind)
    ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
    ctx.setup()
    cov = orc.dense_cov(kind, X, hyp)
    terms = orc.common_terms(kind, X, hyp)
    # operator and preconditioner seams
    p = np.random.default_rng(1).standard_normal(N)
    Ap = ctx.matvec(torch.from_numpy(p)).cpu().numpy()
    np.testing.assert_allclose(Ap, cov @ p, rtol=0, atol=1e-11 * np.abs(cov @ p).max())
    np.testing.assert_allclose(ctx.get_matrix("A").cpu().numpy(), terms.A, rtol=0, atol=1e-9 * np.abs(terms.A).max())
    np.testing.assert_allclose(ctx.get_matrix("LB").cpu().numpy(), terms.LB, rtol=0, atol=1e-9 * np.abs(terms.LB).max())
    z, rz = ctx.precond(p)
    zr, rzr = orc.nystrom_precond(terms.A, terms.LB, hyp.noise, p)
    np.testing.assert_allclose(z.cpu().numpy(), zr, rtol=0, atol=1e-10 * np.abs(zr).max())
    # evaluation: solve, bound, gradient
    v = torch.zeros(N, dtype=torch.float64, device=ctx.device)
    res = ctx.objective_and_grad(v, True, 1e-2, 100, 40)
    ref = orc.objective(kind, X, y, hyp, np.zeros(N), True, 1e-2, 100, 40, cov=cov)
    assert abs(res.steps - ref.steps) <= (0 if ref.steps <= 40 else 1)
    refg = orc.objective(kind, X, y, hyp, v.cpu().numpy(), run_cg=False, with_grad=True, cov=cov)
    assert res.bound == pytest.approx(refg.bound, rel=1e-10)
    if res.steps == ref.steps and res.steps <= 40:
        assert res.bound == pytest.approx(ref.bound, rel=1e-8)
    for key in ("lengthscales", "Z", "variance", "noise", "mean"):
        a, b = np.asarray(res.grad[key]), np.asarray(refg.grad[key])
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-7 * max(np.abs(b).max(), 1e-3 * abs(ref.bound)), err_msg=key)
    # prediction and cross mat-vec (new points far outside the data as well)
    xnew = np.random.default_rng(2).standard_normal((57, D)) * np.linspace(0.5, 3.0, 57)[:, None]
    ctx.setup()
    pm, pv = ctx.predict(v, xnew)
    om, ov, _, _ = orc.predict(kind, X, y, hyp, v.cpu().numpy(), xnew, max_error=1e300)
    np.testing.assert_allclose(pm.cpu().numpy(), om, rtol=0, atol=1e-9 * np.abs(om).max())
    np.testing.assert_allclose(pv.cpu().numpy(), ov, rtol=0, atol=1e-9 * np.abs(ov).max())
    cm = ctx.cross_matvec(xnew, torch.from_numpy(p)).cpu().numpy()
    np.testing.assert_allclose(cm, orc.kernel_matrix(kind, xnew, X, hyp.lengthscales, hyp.variance) @ p, rtol=0, atol=1e-11 * np.abs(cm).max())
    ctx.close()


@pytest.mark.parametrize("kind", ["rbf", "matern32"])
@pytest.mark.parametrize("D,ell_factor,precision", [(40, 1.0, 1), (64, 1.0, 0), (77, 1.0, 1), (90, 1.0, 1), (90, 1.0, 0), (77, 0.12, 1), (96, 0.1, 0)])
def test_mid_width_register_resident_passes_against_the_gram_tiles_and_the_oracle(kind, D, ell_factor, precision):
    """32 < D <= 96: the register-resident mat-vec and gradient pass (padded widths 48, 64, 80, 96; the last with a row shared by four
    lanes in the gradient pass) against the same context run through the Gram tiles (wide_reg = 0) and against the dense oracle; ragged N
    (not a multiple of the 64 / 128-row blocks), both precision levels, and short lengthscales that select the range-clamped variants."""
    from cglb_amd.hip_context import HipContext
    N, M = 1867, 40
    X, y, Z, hyp = _problem(N, D, M, seed=D, ell_factor=ell_factor)
    cov = orc.dense_cov(kind, X, hyp)
    p = np.random.default_rng(5).standard_normal(N)
    out = {}
    for reg in (1, 0):
        ctx = HipContext(X, y, M, kind)
        ctx.set_option("wide_reg", reg)
        ctx.set_option("precision", precision)
        ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
        ctx.setup()
        Ap = ctx.matvec(torch.from_numpy(p)).cpu().numpy()
        v = torch.zeros(N, dtype=torch.float64, device=ctx.device)
        res = ctx.objective_and_grad(v, True, 1e-2, 100, 40)
        out[reg] = (Ap, res, v.cpu().numpy())
        ctx.close()
    ref_mv = cov @ p
    for reg in (1, 0):
        np.testing.assert_allclose(out[reg][0], ref_mv, rtol=0, atol=1e-11 * np.abs(ref_mv).max())
    np.testing.assert_allclose(out[1][0], out[0][0], rtol=0, atol=2e-12 * np.abs(ref_mv).max())
    # Each variant against the oracle AT ITS OWN v: the two solves differ by the rounding of their mat-vecs (1e-14), which a 37-step solve
    # behind a 40-point preconditioner amplifies to a different v within the stopping tolerance (and, at larger N, to other step counts)
    for reg in (1, 0):
        res, vh = out[reg][1], out[reg][2]
        refg = orc.objective(kind, X, y, hyp, vh, run_cg=False, with_grad=True, cov=cov)
        assert res.bound == pytest.approx(refg.bound, rel=1e-10)
        for key in ("lengthscales", "Z", "variance", "noise", "mean"):
            a, b = np.asarray(res.grad[key]), np.asarray(refg.grad[key])
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-9 * max(np.abs(b).max(), 1e-3 * abs(refg.bound)), err_msg=f"{key} (wide_reg {reg})")

@pytest.mark.parametrize("kind", ["rbf", "matern32"])
@pytest.mark.parametrize("N,D,M", [(1, 40, 1), (2, 90, 2), (63, 64, 5), (64, 77, 64), (65, 90, 7), (129, 40, 16), (257, 96, 9)])
def test_mid_width_edge_shapes(kind, N, D, M):
    """Fewer rows than a 64-row block, exactly one block, one row beyond a block boundary (64-row blocks of the mat-vec, 128-row blocks
    of the gradient pass), N = 1, M = N: every shape against the dense oracle."""
    from cglb_amd.hip_context import HipContext
    X, y, Z = orc.synthetic_problem(max(N, M, 8), D, M, seed=N + D)
    X, y = X[:N], y[:N]
    rng = np.random.default_rng(N)
    hyp = orc.Hypers(rng.uniform(0.8, 1.6, size=D) * np.sqrt(D), 1.3, 0.08, 0.15, Z, 1e-6)
    ctx = HipContext(X, y, M, kind)
    ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
    ctx.setup()
    cov = orc.dense_cov(kind, X, hyp)
    p = rng.standard_normal(N)
    Ap = ctx.matvec(torch.from_numpy(p)).cpu().numpy()
    np.testing.assert_allclose(Ap, cov @ p, rtol=0, atol=1e-12 * np.abs(cov @ p).max())
    v = torch.zeros(N, dtype=torch.float64, device=ctx.device)
    res = ctx.objective_and_grad(v, True, 1e-2, 100, 40)
    ref = orc.objective(kind, X, y, hyp, np.zeros(N), True, 1e-2, 100, 40, cov=cov)
    assert res.steps == ref.steps
    refg = orc.objective(kind, X, y, hyp, v.cpu().numpy(), run_cg=False, with_grad=True, cov=cov)
    assert res.bound == pytest.approx(refg.bound, rel=1e-10, abs=1e-10)
    for key in ("lengthscales", "Z", "variance", "noise", "mean"):
        a, b = np.asarray(res.grad[key]), np.asarray(refg.grad[key])
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-8 * max(np.abs(b).max(), 1e-3 * abs(refg.bound), 1e-6), err_msg=key)
    ctx.close()


def test_mid_width_low_precision_level_and_switching_the_path_on_one_context():
    """precision 2 (kernel values to 1e-10) has its own mid-width mat-vec instances (the gradient pass runs level 1 there); and the option
    wide_reg may be flipped on a live context - both paths serve the same operands."""
    from cglb_amd.hip_context import HipContext
    N, D, M = 1500, 77, 32
    X, y, Z, hyp = _problem(N, D, M, seed=21)
    cov = orc.dense_cov("rbf", X, hyp)
    p = np.random.default_rng(3).standard_normal(N)
    ctx = HipContext(X, y, M, "rbf")
    ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
    ctx.setup()
    a1 = ctx.matvec(torch.from_numpy(p)).cpu().numpy()
    ctx.set_option("wide_reg", 0)
    a0 = ctx.matvec(torch.from_numpy(p)).cpu().numpy()
    ctx.set_option("wide_reg", 1)
    a1b = ctx.matvec(torch.from_numpy(p)).cpu().numpy()
    ref = cov @ p
    np.testing.assert_allclose(a1, ref, rtol=0, atol=1e-11 * np.abs(ref).max())
    np.testing.assert_allclose(a0, ref, rtol=0, atol=1e-11 * np.abs(ref).max())
    assert np.array_equal(a1, a1b)
    ctx.set_option("precision", 2)
    ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
    a2 = ctx.matvec(torch.from_numpy(p)).cpu().numpy()
    err = np.abs(a2 - ref).max() / np.abs(ref).max()
    assert 1e-13 < err < 2e-9, err     # the level-2 polynomial is really the one that ran, and it is as accurate as documented
    v = torch.zeros(N, dtype=torch.float64, device=ctx.device)
    res = ctx.objective_and_grad(v, True, 1.0, 100, 40)
    refg = orc.objective("rbf", X, y, hyp, v.cpu().numpy(), run_cg=False, with_grad=True, cov=cov)
    assert res.bound == pytest.approx(refg.bound, rel=1e-6)
    np.testing.assert_allclose(res.grad["lengthscales"], refg.grad["lengthscales"], rtol=0, atol=1e-6 * np.abs(refg.grad["lengthscales"]).max())
    ctx.close()


@pytest.mark.parametrize("kind", ["rbf", "matern32"])
def test_wide_inputs_in_fp32(kind):
    """fp32 contexts keep every product of a wide input on the Gram tiles (sgemm); against the fp64 oracle at single-precision tolerances."""
    from cglb_amd.hip_context import HipContext
    N, D, M = 3000, 50, 40
    X, y, Z = orc.synthetic_problem(N, D, M, seed=4)
    rng = np.random.default_rng(1)
    hyp = orc.Hypers(rng.uniform(0.8, 1.6, size=D) * np.sqrt(D), 1.3, 0.3, 0.15, Z, 1e-4)
    ctx = HipContext(X, y, M, kind, dtype=torch.float32)
    ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
    ctx.setup()
    cov = orc.dense_cov(kind, X, hyp)
    p = rng.standard_normal(N)
    Ap = ctx.matvec(torch.from_numpy(p).float()).double().cpu().numpy()
    np.testing.assert_allclose(Ap, cov @ p, rtol=0, atol=2e-5 * np.abs(cov @ p).max())
    v = torch.zeros(N, dtype=torch.float32, device=ctx.device)
    res = ctx.objective_and_grad(v, True, 1.0, 100, 40)
    ref = orc.objective(kind, X, y, hyp, np.zeros(N), True, 1.0, 100, 40, cov=cov)
    assert abs(res.steps - ref.steps) <= 1
    refg = orc.objective(kind, X, y, hyp, v.double().cpu().numpy(), run_cg=False, with_grad=True, cov=cov)
    assert res.bound == pytest.approx(refg.bound, rel=2e-6)
    for key in ("lengthscales", "Z", "variance", "noise"):
        b = np.asarray(refg.grad[key])
        np.testing.assert_allclose(np.asarray(res.grad[key]), b, rtol=0, atol=5e-4 * max(np.abs(b).max(), 1e-3 * abs(refg.bound)), err_msg=key)
    ctx.close()


def test_wide_inducing_point_selection_and_backend_training_step():
    """Greedy conditional-variance selection and three L-BFGS-B iterations through the backend mirror at D = 50."""
    from cglb_amd.backend import BACKENDS, CGLBConfig, INDUCING_VARIABLE_CONFIGS, KERNEL_CONFIGS
    from cglb_amd.backend.callbacks import Logger
    from cglb_amd.backend.interface import _InitKernel
    from oracle.cglb_oracle import greedy_conditional_variance
    N, D, M = 900, 50, 20
    X, y, _ = orc.synthetic_problem(N, D, M, seed=9)
    be = BACKENDS["hip"]
    be.configure_backend(logdir="/tmp/cglb_amd_test", keops=False)
    be.set_default_float("fp64")
    be.set_default_jitter("fp64")
    model = be.create_model(CGLBConfig(kernel=KERNEL_CONFIGS["Matern32"](), inducing_variable=INDUCING_VARIABLE_CONFIGS["cv"](M)), (X, y))
    Zsel = be.model_parameters(model)[".inducing_variable.Z"]
    np.testing.assert_array_equal(Zsel, greedy_conditional_variance(X, M, _InitKernel(model.covar_module.base_kernel).__call__))
    logger = Logger("/tmp/cglb_amd_test", lambda: {}, lambda: be.model_parameters(model), holdout_interval=-1, verbose=False)
    from cglb_amd.backend.models import LowerBoundCG
    with torch.no_grad():
        l0 = float(-LowerBoundCG(model)(None))
    results = be.optimize(model, ((X, y), (X[:5], y[:5])), 3, logger, "scipy")
    assert sum(r.nit for r in results) == 3 and np.isfinite(results[-1].fun) and results[-1].fun < l0
    model.hip.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _dist_worker(rank, world, port, q, shape=(9000, 40, 128), wide_reg=1):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cglb_amd.dist_context import DistHipContext
        N, D, M = shape              # (9000, 40, 128): three 4096-row tiles, 22 CG steps
        X, y, Z, hyp = _problem(N, D, M, seed=3)
        ctx = DistHipContext(X, y, M, "rbf", collectives="callbacks")
        ctx.set_option("wide_reg", wide_reg)
        ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
        v = torch.zeros(N, dtype=torch.float64, device=ctx.device)
        r = ctx.objective_and_grad(v, True, 1.0, 100, 40)
        q.put((rank, r.steps, r.bound, v.cpu().numpy(), r.grad))
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("wide_reg", [1, 0])
def test_wide_inputs_on_two_ranks(wide_reg):
    """wide_reg = 1: the register-resident passes in their cyclic form; 0: the Gram tiles, three row tiles dealt 2 + 1 over the two ranks
    (symmetric use of the tiles in the mat-vec and in the gradient pass, mirrored sums landing in the other rank's rows)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, q, (9000, 40, 128), wide_reg)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    N, D, M = 9000, 40, 128
    X, y, Z, hyp = _problem(N, D, M, seed=3)
    cov = orc.dense_cov("rbf", X, hyp)
    ref = orc.objective("rbf", X, y, hyp, np.zeros(N), True, 1.0, 100, 40, cov=cov)
    for rank, steps, bound, v, grad in out:
        assert steps == ref.steps and ref.steps == 22, rank
        refg = orc.objective("rbf", X, y, hyp, v, run_cg=False, with_grad=True, cov=cov)
        assert bound == pytest.approx(refg.bound, rel=1e-10)
        np.testing.assert_allclose(grad["lengthscales"], refg.grad["lengthscales"], rtol=0, atol=1e-7 * np.abs(refg.grad["lengthscales"]).max())
        np.testing.assert_allclose(grad["Z"], refg.grad["Z"], rtol=0, atol=1e-7 * np.abs(refg.grad["Z"]).max())
    assert out[0][1:3] == out[1][1:3]



def test_mid_width_on_three_ranks_with_uneven_block_shares():
    """N = 300 at D = 77: five 64-row blocks of the mat-vec and three 128-row blocks of the gradient pass dealt over three ranks (2 + 2 + 1
    and 1 + 1 + 1), loops and collectives inside the library (callbacks over gloo, all ranks on cuda:0)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    shape = (300, 77, 24)
    procs = [ctx.Process(target=_dist_worker, args=(r, 3, port, q, shape)) for r in range(3)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=600) for _ in range(3)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    N, D, M = shape
    X, y, Z, hyp = _problem(N, D, M, seed=3)
    cov = orc.dense_cov("rbf", X, hyp)
    ref = orc.objective("rbf", X, y, hyp, np.zeros(N), True, 1.0, 100, 40, cov=cov)
    for rank, steps, bound, v, grad in out:
        assert steps == ref.steps, rank
        refg = orc.objective("rbf", X, y, hyp, v, run_cg=False, with_grad=True, cov=cov)
        assert bound == pytest.approx(refg.bound, rel=1e-10)
        for key in ("lengthscales", "Z", "variance", "noise"):
            b = np.asarray(refg.grad[key])
            np.testing.assert_allclose(np.asarray(grad[key]), b, rtol=0, atol=1e-8 * max(np.abs(b).max(), 1e-3 * abs(refg.bound)), err_msg=key)
    assert out[0][1:3] == out[1][1:3] == out[2][1:3]


def _row_sharded_worker(rank, world, port, shape, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cglb_amd.distributed import Comm, HipLocalOps, ShardedCGLB, row_partition
        from cglb_amd.hip_context import HipContext
        N, D, M = shape
        X, y, Z, hyp = _problem(N, D, M, seed=5)
        per, parts = row_partition(N, world)
        ctx = HipContext(X, y, M, "rbf", row_range=parts[rank])
        ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
        drv = ShardedCGLB(HipLocalOps(ctx), Comm())
        res = drv.objective_and_grad(True, 1.0, 100, 40)
        v = drv.v_full().cpu().numpy()
        if rank == 0:
            q.put((res.bound, res.steps, res.grad, v))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("D", [100, 60])
def test_wide_inputs_row_sharded_driver_on_two_ranks(D):
    """The older row-sharded scheme (each rank owns a contiguous block of rows and all columns: tiles of a row range that does not start
    at row 0, no symmetric use) still serves wide inputs: D = 100 on the Gram tiles, D = 60 likewise (the register-resident passes
    cover the full square and the cyclic deal only)."""
    shape = (5000, D, 48)
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_row_sharded_worker, args=(r, 2, port, shape, q)) for r in range(2)]
    for p in procs:
        p.start()
    bound, steps, grad, v = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    N, D, M = shape
    X, y, Z, hyp = _problem(N, D, M, seed=5)
    cov = orc.dense_cov("rbf", X, hyp)
    ref = orc.objective("rbf", X, y, hyp, np.zeros(N), True, 1.0, 100, 40, cov=cov)
    assert abs(steps - ref.steps) <= (0 if ref.steps <= 40 else 1)
    # at the driver's own v (the solve behind a 48-point preconditioner amplifies the rounding of the mat-vec into v)
    refg = orc.objective("rbf", X, y, hyp, v, run_cg=False, with_grad=True, cov=cov)
    assert bound == pytest.approx(refg.bound, rel=1e-10)
    gl, gz = np.asarray(refg.grad["lengthscales"]), np.asarray(refg.grad["Z"])
    np.testing.assert_allclose(grad[:D], gl, rtol=0, atol=1e-8 * np.abs(gl).max())
    assert grad[D] == pytest.approx(refg.grad["variance"], rel=1e-7, abs=1e-8 * abs(refg.bound))
    assert grad[D + 1] == pytest.approx(refg.grad["noise"], rel=1e-7, abs=1e-8 * abs(refg.bound))
    np.testing.assert_allclose(grad[D + 3:], gz.reshape(-1), rtol=0, atol=1e-8 * np.abs(gz).max())
