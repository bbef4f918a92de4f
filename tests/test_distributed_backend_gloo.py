"""The N-rank path THROUGH the backend API (BASELINE config C4: "full CGLB train loop" on N ranks) on CPU: world_size 2 and 4 over
gloo, every rank holding a `CGLB` model whose engine is the host-driven twin of the library's N-rank loops
(`distributed.PyDistContext` over `SymShardedCGLB`) with oracle-backed local arithmetic.  Checked:

  * `optimize` (pytorch/interface.py:445-543) reproduces, on EVERY rank, the trajectory recorded from the reference's own `Scipy`
    wrapper and solver (tests/golden/train/*.npz): loss and CG step count of every evaluation, nit / nfev, final parameters;
  * `PredictCG` / `metrics_fn` (models.py:289-354, interface.py:607-658) with the new points dealt over the ranks equal the
    single-process oracle;
  * the solver / operator / preconditioner seams (`ConjugateGradient(A, b, v, precond)`, `A @ x`, `precond(r)`) on full replicated vectors.

The same model code runs on the GPU box with `dist_context.DistHipContext` (tests/test_gpu_dist_backend.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN_DIR
from oracle import cglb_oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build_model(g, rank, world):
    from cglb_amd.backend.models import CGLB, BaseKernel, GaussianLikelihood, InducingPointKernel, ScaleKernel
    from cglb_amd.distributed import Comm, PyDistContext, row_partition
    from sharded_oracle_ops import OracleSymLocalOps
    kind = "rbf" if int(g["kind"]) == 0 else "matern32"
    X, y = g["X"], g["y"]
    N, D = X.shape
    hyp = orc.Hypers(g["init_lengthscales"].copy(), float(g["init_variance"]), float(g["init_noise"]), float(g["init_mean"]), g["init_Z"].copy(), 1e-6)
    per, parts = row_partition(N, world)
    ops = OracleSymLocalOps(kind, X, y, hyp, *parts[rank])
    base = BaseKernel(kind, ard_num_dims=D)
    base.lengthscale = hyp.lengthscales
    scale = ScaleKernel(base)
    scale.outputscale = hyp.variance
    lik = GaussianLikelihood(lower_bound=1e-6)
    lik.noise = hyp.noise
    model = CGLB((X, y), lik, InducingPointKernel(scale, hyp.Z), context=PyDistContext(ops, Comm()))
    with torch.no_grad():
        model.mean_module.constant.copy_(torch.tensor(hyp.mean, dtype=torch.float64))
    return model, kind


def _train_worker(rank, world, port, path, tmp, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CGLB_HOST_THREADS="0")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cglb_amd.backend import interface
        from cglb_amd.backend.callbacks import Logger
        g = dict(np.load(path))
        model, kind = _build_model(g, rank, world)
        interface._broadcast_parameters(model)
        logger = Logger(tmp, lambda: {}, lambda: interface.model_parameters(model), holdout_interval=-1, include_feval_log=True, verbose=False)
        losses = []
        orig = logger.log_for_feval

        def record(**entries):
            orig(**entries)
            losses.append(-float(model.last_bound))
        logger.log_for_feval = record
        results = interface._optimize_cglb_impl(model, ((g["X"], g["y"]), (g["X"][:4], g["y"][:4])), int(g["num_steps"]), logger, "scipy")
        params = interface.model_parameters(model)
        q.put((rank, losses, list(logger.logs["steps-per-feval"]), [int(r.nit) for r in results], [int(r.nfev) for r in results],
               {k: np.asarray(v) for k, v in params.items()}))
    finally:
        dist.destroy_process_group()


def _spawn(target, world, args, n_results):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=600) for _ in range(n_results)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return out


@pytest.mark.parametrize("world", [2, 4])
def test_optimize_on_n_ranks_reproduces_the_reference_trajectory(world, tmp_path):
    path = os.path.join(GOLDEN_DIR, "train", "train_rbf_d8_trained.npz")
    g = dict(np.load(path))
    out = sorted(_spawn(_train_worker, world, (path, str(tmp_path)), world), key=lambda t: t[0])
    ref_loss, ref_steps = g["loss"], g["steps"][1:]
    for rank, losses, steps, nit, nfev, params in out:
        assert len(losses) == len(ref_loss), rank
        np.testing.assert_allclose(losses, ref_loss, rtol=1e-6, atol=1e-6, err_msg=f"rank {rank}")
        np.testing.assert_array_equal(np.asarray(steps, dtype=np.int64), ref_steps, err_msg=f"rank {rank}")
        assert nit == g["nit"].tolist() and nfev == g["nfev"].tolist(), rank
        np.testing.assert_allclose(params[".kernel.lengthscales"], g["final_lengthscales"], rtol=1e-5)
        np.testing.assert_allclose(params[".kernel.variance"], g["final_variance"], rtol=1e-5)
        np.testing.assert_allclose(params[".likelihood.variance"], g["final_noise"].reshape(()), rtol=1e-5)
        np.testing.assert_allclose(params[".inducing_variable.Z"], g["final_Z"], rtol=1e-5, atol=1e-6)
    # identical trajectories on every rank: same losses to the last bit, same parameters
    for rank, losses, steps, nit, nfev, params in out[1:]:
        assert losses == out[0][1] and steps == out[0][2]
        for k in params:
            np.testing.assert_array_equal(params[k], out[0][5][k])


def _predict_worker(rank, world, port, path, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CGLB_HOST_THREADS="0")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cglb_amd.backend import interface
        from cglb_amd.backend.conjugate_gradient import ConjugateGradient, KernelOperator, NystromPreconditioner
        from cglb_amd.backend.models import LowerBoundCG, PredictCG
        g = dict(np.load(path))
        model, kind = _build_model(g, rank, world)
        loss = float(-LowerBoundCG(model)(None))
        steps = int(model.cg_stats.steps)
        xnew = np.random.default_rng(5).standard_normal((37, g["X"].shape[1]))   # 37: ragged slices over the ranks
        f_mean, f_var = PredictCG(model)(torch.from_numpy(xnew))
        # metrics on the model's own training set + a held-out part (collective: every rank evaluates them)
        metrics = interface.metrics_fn(model, ((g["X"], g["y"]), (xnew, np.sin(xnew[:, 0]))))()
        assert np.isfinite(metrics["loss"]) and np.isfinite(metrics["train/rmse"]) and np.isfinite(metrics["test/nlpd"])
        assert float(metrics["loss"]) == pytest.approx(loss, rel=1e-12)     # cached v: no CG, same bound (interface.py:619-625)
        # seams on full replicated vectors
        hip = model.hip
        hip.setup()
        A, P = KernelOperator(hip), NystromPreconditioner(hip)
        xr = torch.from_numpy(np.random.default_rng(6).standard_normal(hip.N))
        Ax = (A @ xr).numpy().copy()
        z, rz = P(xr)
        v, st = ConjugateGradient(max_error=1e-6)(A, xr.reshape(-1, 1), torch.zeros(hip.N, 1, dtype=torch.float64), P)
        q.put((rank, loss, steps, f_mean.numpy().reshape(-1), f_var.numpy().reshape(-1), Ax, z.numpy().copy(), float(rz), v.numpy().reshape(-1), int(st.steps)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_predict_and_seams_on_n_ranks(world):
    path = os.path.join(GOLDEN_DIR, "train", "train_rbf_d8_trained.npz")
    g = dict(np.load(path))
    out = sorted(_spawn(_predict_worker, world, (path,), world), key=lambda t: t[0])
    X, y = g["X"], g["y"]
    N = X.shape[0]
    hyp = orc.Hypers(g["init_lengthscales"].copy(), float(g["init_variance"]), float(g["init_noise"]), float(g["init_mean"]), g["init_Z"].copy(), 1e-6)
    ref = orc.objective("rbf", X, y, hyp, np.zeros(N), True, 1.0)
    xnew = np.random.default_rng(5).standard_normal((37, X.shape[1]))
    pm, pv, _, _ = orc.predict("rbf", X, y, hyp, ref.v, xnew, max_error=1e-3)
    cov = orc.dense_cov("rbf", X, hyp)
    terms = orc.common_terms("rbf", X, hyp)
    xr = np.random.default_rng(6).standard_normal(N)
    zr, rzr = orc.nystrom_precond(terms.A, terms.LB, hyp.noise, xr)
    vr, str_ = orc.pcg(lambda t: cov @ t, xr, np.zeros(N), lambda r: orc.nystrom_precond(terms.A, terms.LB, hyp.noise, r), 1e-6)
    for rank, loss, steps, f_mean, f_var, Ax, z, rz, v, st in out:
        assert steps == ref.steps and loss == pytest.approx(-ref.bound, rel=1e-10), rank
        np.testing.assert_allclose(f_mean, pm, rtol=0, atol=1e-8 * np.abs(pm).max(), err_msg=f"rank {rank}")
        np.testing.assert_allclose(f_var, pv, rtol=0, atol=1e-8 * np.abs(pv).max())
        np.testing.assert_allclose(Ax, cov @ xr, rtol=0, atol=1e-11 * np.abs(cov @ xr).max())
        np.testing.assert_allclose(z, zr, rtol=0, atol=1e-10 * np.abs(zr).max())
        assert rz == pytest.approx(rzr, rel=1e-10)
        assert abs(st - str_.steps) <= 1
        np.testing.assert_allclose(v, vr, rtol=0, atol=1e-6 * np.abs(vr).max())
    for t in out[1:]:
        np.testing.assert_array_equal(t[3], out[0][3])      # replicated outputs are identical on every rank
        np.testing.assert_array_equal(t[4], out[0][4])
