"""The training loop against trajectories recorded from the REFERENCE's own `Scipy` wrapper
(cglb/backend/pytorch/optimizer.py:20-98) and solver (conjugate_gradient.py), oracle/gen_train_fixture.py ->
tests/golden/train/*.npz: warm-up evaluation, then the `minimize` rounds of pytorch/interface.py:505-543.

The hip backend's `optimize` must reproduce, evaluation by evaluation, the loss and the CG step count, the nit / nfev of every
round and the final constrained parameters.  Tolerances: loss 1e-6 relative (north_star), parameters 1e-5; L-BFGS-B is a
deterministic function of the (loss, gradient) sequence, so agreement of the trace implies the same path through the schedule."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu

CASES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "train", "*.npz")))


@pytest.mark.parametrize("path", CASES, ids=[os.path.splitext(os.path.basename(p))[0] for p in CASES])
def test_optimize_reproduces_reference_trajectory(path, tmp_path):
    from cglb_amd.backend import BACKENDS, CGLBConfig, INDUCING_VARIABLE_CONFIGS, KERNEL_CONFIGS
    from cglb_amd.backend.callbacks import Logger
    g = dict(np.load(path))
    be = BACKENDS["hip"]
    be.configure_backend(logdir=str(tmp_path), keops=False)
    be.set_default_float("fp64")
    be.set_default_jitter("fp64")
    assert float(g["jitter"]) == 1e-6
    kernel = "rbf" if int(g["kind"]) == 0 else "Matern32"
    M = g["init_Z"].shape[0]
    cfg = CGLBConfig(kernel=KERNEL_CONFIGS[kernel](), inducing_variable=INDUCING_VARIABLE_CONFIGS["cv"](M))
    model = be.create_model(cfg, (g["X"], g["y"]))
    # start from the fixture's hypers (the reference's greedy robustgp initialisation of Z is third-party and unpinned)
    model.likelihood.noise = float(g["init_noise"])
    model.covar_module.base_kernel.base_kernel.lengthscale = g["init_lengthscales"]
    model.covar_module.base_kernel.outputscale = float(g["init_variance"])
    with torch.no_grad():
        model.mean_module.constant.copy_(torch.tensor(float(g["init_mean"]), dtype=torch.float64))
        model.covar_module.inducing_points.copy_(torch.from_numpy(g["init_Z"]))
    logger = Logger(str(tmp_path), lambda: {}, lambda: be.model_parameters(model), holdout_interval=-1, include_feval_log=True, verbose=False)
    losses = []
    orig = logger.log_for_feval

    def record(**entries):
        orig(**entries)
        losses.append(-float(model.last_bound))
    logger.log_for_feval = record
    results = be.optimize(model, ((g["X"], g["y"]), (g["X"][:4], g["y"][:4])), int(g["num_steps"]), logger, "scipy")
    # `losses` holds every evaluation including the warm-up one (index 0, as in the fixture); the warm-up runs under
    # logger.no_recording(), so the logged CG statistics start at the first evaluation of round 1
    ref_loss, ref_steps = g["loss"], g["steps"][1:]
    got_steps = np.asarray(logger.logs["steps-per-feval"], dtype=np.int64)
    n = min(len(ref_loss), len(losses))
    np.testing.assert_allclose(losses[:n], ref_loss[:n], rtol=1e-6, atol=1e-6)
    m = min(len(ref_steps), len(got_steps))
    np.testing.assert_array_equal(got_steps[:m], ref_steps[:m])
    assert len(got_steps) == len(ref_steps)
    assert [int(r.nit) for r in results] == g["nit"].tolist()
    assert [int(r.nfev) for r in results] == g["nfev"].tolist()
    assert len(losses) == len(ref_loss)
    params = be.model_parameters(model)
    np.testing.assert_allclose(params[".likelihood.variance"], g["final_noise"].reshape(()), rtol=1e-5)
    np.testing.assert_allclose(params[".kernel.lengthscales"], g["final_lengthscales"], rtol=1e-5)
    np.testing.assert_allclose(params[".kernel.variance"], g["final_variance"], rtol=1e-5)
    np.testing.assert_allclose(params[".mean_function.c"], g["final_mean"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(params[".inducing_variable.Z"], g["final_Z"], rtol=1e-5, atol=1e-6)
    model.hip.close()
