#!/usr/bin/env python3
"""Training-trajectory fixtures from the REFERENCE's own optimiser wrapper and solver.  TEST INFRASTRUCTURE ONLY.

Runs only in the build container (needs /root/reference); writes tests/golden/train/*.npz, which are committed.

What comes from the reference (both files import numpy / scipy / torch only and are loaded by file path):
  * cglb/backend/pytorch/optimizer.py : `Scipy.minimize` (pack / unpack / assign, `torch.autograd.grad(loss, variables)`,
    the step callback, SciPy L-BFGS-B)                                                        -> rows b5 / f1
  * cglb/backend/pytorch/conjugate_gradient.py : `ConjugateGradient`, `NystromPreconditioner`  -> rows A1 / A2 / A4
What is restated here because `pytorch/interface.py` / `models.py` need gpytorch:
  * the model's raw parameters (softplus transforms; likelihood noise >= 1e-6, interface.py:269-273) in the order of
    `model.parameters()`, the warm-started `v_vec` (models.py:59-72, :274), `LowerBoundCG.forward` as the dense torch
    restatement of oracle/gen_golden.py (t_forward), and the optimise schedule of interface.py:445-543: warm-up evaluation,
    then up to four `Scipy().minimize` rounds with options {maxiter, ftol = 0, gtol = 0}, the last two without the inducing
    points, `step_callback` resetting the cache flag and counting accepted steps.

Stored: per objective evaluation the loss, CG steps and 1/2 r^T P r; per round nit / nfev / final loss; the final constrained
parameters; the inputs (X, y, initial hypers incl. Z).
"""
import importlib.util
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import cglb_oracle as orc  # noqa: E402
from oracle.gen_golden import load_reference_cg, t_forward  # noqa: E402

REF_OPT = "/root/reference/cglb/backend/pytorch/optimizer.py"
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
OUT = os.path.join(GOLDEN, "train")
NOISE_FLOOR = 1e-6  # interface.py:269


def load_reference_optimizer():
    spec = importlib.util.spec_from_file_location("_ref_optimizer", REF_OPT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def inv_softplus(x):
    x = torch.as_tensor(x, dtype=torch.float64)
    return x + torch.log(-torch.expm1(-x))


def make_case(ref_cg, ref_opt, name, source, num_steps):
    g = dict(np.load(os.path.join(GOLDEN, source + ".npz")))
    kind = int(g["kind"])
    X, y = torch.from_numpy(g["X"]), torch.from_numpy(g["y"])
    N = X.shape[0]
    jitter = float(g["jitter"])
    # raw parameters in the order of model.parameters(): likelihood noise, mean constant, inducing points, outputscale, lengthscales
    raw_noise = inv_softplus(torch.tensor([float(g["noise"]) - NOISE_FLOOR])).requires_grad_(True)
    mean = torch.tensor(float(g["mean"]), dtype=torch.float64, requires_grad=True)
    Z = torch.from_numpy(g["Z"].copy()).requires_grad_(True)
    raw_var = inv_softplus(torch.tensor(float(g["variance"]))).requires_grad_(True)
    raw_ls = inv_softplus(torch.from_numpy(g["lengthscales"].copy()).reshape(1, -1)).requires_grad_(True)
    params = [raw_noise, mean, Z, raw_var, raw_ls]

    v_vec = torch.zeros((N, 1), dtype=torch.float64)                       # models.py:59-68
    cg_opt = ref_cg.ConjugateGradient()                                    # torch create_model ignores the config's max_error
    trace = {"loss": [], "steps": [], "residual_error": [], "round": []}
    state = {"round": -1, "accepted": 0}

    def lbfgs_closure():                                                   # interface.py:474-477
        noise = F.softplus(raw_noise).reshape(()) + NOISE_FLOOR
        var = F.softplus(raw_var)
        ls = F.softplus(raw_ls).reshape(-1)
        bound, lower, upper, logdet, v, stats, _ = t_forward(ref_cg, kind, X, y, ls, var, noise, mean, Z, jitter, v_vec, cg_opt)
        v_vec.copy_(v)                                                     # models.py:274
        trace["loss"].append(float(-bound)); trace["steps"].append(int(stats.steps))
        trace["residual_error"].append(float(stats.residual_error)); trace["round"].append(state["round"])
        return -bound

    def step_callback(step, variables, values):                            # interface.py:479-481
        state["accepted"] += 1

    def optimize_fn(variables, maxiter, ftol=0.0, gtol=0.0, disp=False):   # interface.py:483-491
        options = dict(maxiter=maxiter, ftol=ftol, gtol=gtol, disp=disp)
        return ref_opt.Scipy().minimize(lbfgs_closure, variables, options=options, step_callback=step_callback)

    # warm-up evaluation outside the clock (interface.py:494-501); it does move v_vec
    loss0 = lbfgs_closure()
    torch.autograd.grad(loss0, params)
    results, remaining, variables = [], num_steps, params
    for round_id in range(4):                                              # interface.py:507-543
        if remaining <= 0:
            break
        if round_id == 2:
            variables = [p for p in params if p is not Z]                  # :527-529
        state["round"] = round_id
        res = optimize_fn(variables, remaining)
        remaining -= res.nit
        results.append(res)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        source=np.array(source), kind=np.int64(kind), num_steps=np.int64(num_steps), X=g["X"], y=g["y"], jitter=np.float64(jitter),
        init_Z=g["Z"], init_lengthscales=g["lengthscales"], init_variance=g["variance"], init_noise=g["noise"], init_mean=g["mean"],
        loss=np.array(trace["loss"]), steps=np.array(trace["steps"], dtype=np.int64), residual_error=np.array(trace["residual_error"]),
        feval_round=np.array(trace["round"], dtype=np.int64), accepted_steps=np.int64(state["accepted"]),
        nit=np.array([r.nit for r in results], dtype=np.int64), nfev=np.array([r.nfev for r in results], dtype=np.int64),
        fun=np.array([float(r.fun) for r in results]), status=np.array([int(r.status) for r in results], dtype=np.int64),
        final_noise=(F.softplus(raw_noise) + NOISE_FLOOR).detach().numpy(), final_mean=mean.detach().numpy(), final_Z=Z.detach().numpy(),
        final_variance=F.softplus(raw_var).detach().numpy(), final_lengthscales=F.softplus(raw_ls).detach().numpy().reshape(-1),
    )
    print(f"{name}: rounds nit={[r.nit for r in results]} nfev={[r.nfev for r in results]} status={[int(r.status) for r in results]} "
          f"loss {trace['loss'][0]:.6f} -> {trace['loss'][-1]:.6f}; CG steps per feval {trace['steps']}", flush=True)


def main():
    torch.set_default_dtype(torch.float64)
    ref_cg, ref_opt = load_reference_cg(), load_reference_optimizer()
    make_case(ref_cg, ref_opt, "train_c1_snelson_like_m32", "c1_snelson_like_m32", 15)
    make_case(ref_cg, ref_opt, "train_rbf_d8_trained", "rbf_d8_trained", 12)
    # SciPy ends the first round early here (30 of 40 iterations): the second `minimize` round of interface.py:517-523 runs
    make_case(ref_cg, ref_opt, "train_m32_d3_random_two_rounds", "m32_d3_random", 40)


if __name__ == "__main__":
    main()
