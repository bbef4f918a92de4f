"""Developer tool: randomized parity sweep of the HIP path against the dense numpy oracle (shapes off every tile grid).
usage: python tools/fuzz_parity.py [cases] [seed] [fp32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import cglb_oracle as orc
from cglb_amd.hip_context import HipContext

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fp32 = len(sys.argv) > 3 and sys.argv[3] == "fp32"
td = torch.float32 if fp32 else torch.float64
F = 3e6 if fp32 else 1.0   # tolerance factor of the fp32 run (round-off 1e-7 against 1e-16, with some slack for sums over N)
bad = 0
t0 = time.time()
for c in range(cases):
    N = int(rng.choice([rng.integers(2, 300), rng.integers(300, 3000), rng.integers(3000, 9000)]))
    D = int(rng.integers(1, 33))
    M = int(min(N, rng.choice([rng.integers(1, 70), rng.integers(60, 200), rng.integers(200, 700)])))
    kind = str(rng.choice(["rbf", "matern32"]))
    prec = int(rng.integers(0, 2))
    X, y, Z = orc.synthetic_problem(max(N, M, 8), D, M, seed=int(rng.integers(1 << 30)))
    X, y = X[:N], y[:N]
    ls = rng.uniform(0.7, 2.5, size=D) * np.sqrt(D / 2.0)
    hyp = orc.Hypers(ls, float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.1, 0.5) if fp32 else rng.uniform(0.02, 0.5)), float(rng.normal() * 0.1), Z, 1e-4 if fp32 else 1e-6)
    tol = float(rng.choice([1.0, 1e-2]))
    ctx = HipContext(X, y, M, kind, dtype=td)
    ctx.set_option("precision", prec)
    ctx.set_hypers(hyp.lengthscales, hyp.variance, hyp.noise, hyp.mean, Z, hyp.jitter)
    v = torch.zeros(N, dtype=td, device=ctx.device)
    res = ctx.objective_and_grad(v, True, tol, 100, 40, with_grad=True)
    ref = orc.objective(kind, X, y, hyp, np.zeros(N), True, tol, 100, 40)
    refg = orc.objective(kind, X, y, hyp, v.double().cpu().numpy(), run_cg=False, with_grad=True)
    p = rng.standard_normal(N)
    Ap = ctx.matvec(torch.from_numpy(p).to(td)).double().cpu().numpy()
    Aref = orc.dense_cov(kind, X, hyp) @ p
    e_mv = np.abs(Ap - Aref).max() / np.abs(Aref).max()
    e_b = abs(res.bound - ref.bound) / abs(ref.bound)
    e_b2 = abs(res.bound - refg.bound) / abs(refg.bound)      # same v: no CG-path dependence
    gl = np.abs(res.grad["lengthscales"] - refg.grad["lengthscales"]).max() / (np.abs(refg.grad["lengthscales"]).max() + 1e-300)
    gz = np.abs(res.grad["Z"] - refg.grad["Z"]).max() / (np.abs(refg.grad["Z"]).max() + 1e-300)
    # the Z gradient carries cond(K_uu) eps of absolute error on both sides (M = N: K_uu as ill conditioned as K_ff): floor tied to the bound;
    # two correct CG runs agree on the bound to 1e-6 or to a fraction of the stopping tolerance (the bound moves by 1/2 r^T P r <= tol
    # between admissible stopping points and long solves with a weak preconditioner drift apart by round-off)
    gz_abs = np.abs(res.grad["Z"] - refg.grad["Z"]).max()
    ok = (e_mv < 1e-11 * F and e_b2 < 1e-9 * F and gl < 1e-6 * (3e3 if fp32 else 1) and (gz < 1e-6 * (3e3 if fp32 else 1) or gz_abs < 1e-9 * F * max(1.0, abs(ref.bound)))
          and abs(res.steps - ref.steps) <= (3 if fp32 else 2)
          and (e_b < 1e-6 * (1e3 if fp32 else 1) or abs(res.steps - ref.steps) > 0 or abs(res.bound - ref.bound) < 0.5 * tol))
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} N={N:5d} D={D:2d} M={M:3d} {kind:8s} prec={prec} tol={tol:g} steps {res.steps}/{ref.steps} matvec {e_mv:.1e} bound {e_b:.1e} "
          f"bound@v {e_b2:.1e} grad ls {gl:.1e} Z {gz:.1e}", flush=True)
    ctx.close()
print(f"{cases} cases, {bad} bad, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
