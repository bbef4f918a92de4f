#!/usr/bin/env python3
"""Benchmark of the CGLB hot path on MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is ONE objective + gradient evaluation (common terms + PCG from a cold start v = 0 + bound
assembly + analytic gradient) of the headline workload: synthetic N = 100 000,
D = 8, M = 1024, fp64, RBF.  Nothing is cached between steps: v is reset to zero and the common
terms are recomputed every step.  Inputs are resident in HBM before the timed region.  With N > 1 the
SAME problem is dealt over the ranks (cyclic-symmetric K_ff blocks + column-sharded Nystrom panel: strong scaling) and three RCCL
collectives run every PCG iteration.

Rank 0 prints one JSON line (see the `record` dict at the end for the fields).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the host driver of this pool only supports dmabuf IPC: without this RCCL's cross-process buffer sharing fails (hipIpcGetMemHandle)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X vector fp64 (SURVEY 8d / BASELINE.md 3): 256 CU x 4 SIMD x 16 fma lanes x 2.4 GHz
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec


def pair_flops(kind: str, D: int) -> int:
    """Algorithmic flops per kernel pair evaluation: RBF 3D+4, Matern-3/2 3D+7."""
    return 3 * D + (4 if kind == "rbf" else 7)


def cpu_baseline(kind, X, y, Z, hyp, N, D, M, steps_cg, restarts):
    """Times the CPU oracle (C/OpenMP restatement, kind "port") on a bounded sample of the same workload and
    scales to one evaluation.  Only this leg touches oracle/."""
    from oracle import cglb_oracle as orc
    from oracle import cglb_oracle_c as orcc
    import scipy.linalg as sla

    cores = orcc.num_threads()
    h = orc.Hypers(np.asarray(hyp["lengthscales"], dtype=np.float64), hyp["variance"], hyp["noise"], hyp["mean"], Z, 1e-6)
    rng = np.random.default_rng(0)
    p = rng.standard_normal(N)
    rows = min(N, 1536)
    orcc.kff_matvec(kind, X, h, p, 0, 64)  # warm the thread pool
    t0 = time.perf_counter(); orcc.kff_matvec(kind, X, h, p, 0, rows); t_mv = (time.perf_counter() - t0) * N / rows
    grows = min(N, 768)
    t0 = time.perf_counter(); orcc.grad_kff(kind, X, h, p, p, 0, grows); t_gk = (time.perf_counter() - t0) * N / grows
    # panel work on a 1/8 column sample: K_uf block, trsm, syrk, the two GEMVs of the preconditioner, the adjoint GEMM
    ncol = max(M, N // 8)
    scale = N / ncol
    t0 = time.perf_counter(); kuf = orcc.kernel_block(kind, Z, X[:ncol], h); t_kuf = (time.perf_counter() - t0) * scale
    kuu = orcc.kernel_block(kind, Z, Z, h) + 1e-6 * np.eye(M)
    t0 = time.perf_counter(); L = np.linalg.cholesky(kuu); t_chol = time.perf_counter() - t0
    t0 = time.perf_counter(); A = sla.solve_triangular(L, kuf, lower=True, overwrite_b=True); t_trsm = (time.perf_counter() - t0) * scale
    t0 = time.perf_counter(); AAt = A @ A.T; t_syrk = (time.perf_counter() - t0) * scale
    LB = np.linalg.cholesky(AAt * scale / h.noise + np.eye(M))
    r = rng.standard_normal(ncol)
    t0 = time.perf_counter()
    for _ in range(3):
        orc.nystrom_precond(A, LB, h.noise, r)
    t_pc = (time.perf_counter() - t0) / 3 * scale
    Hm = rng.standard_normal((M, M))
    t0 = time.perf_counter(); _ = Hm @ A; t_gemm = (time.perf_counter() - t0) * scale
    n_mv = steps_cg + 2 + restarts
    t_eval = n_mv * t_mv + (steps_cg + 2) * t_pc + t_gk + t_kuf + 2 * t_chol + t_trsm + t_syrk + t_gemm
    return {
        "value": 1.0 / t_eval, "unit": "evals/s", "cores": cores, "kind": "port",
        "sample": (f"oracle/cglb_oracle.c (OpenMP, {cores} threads): K_ff mat-vec timed on {rows} of {N} rows "
                   f"({t_mv:.2f} s/mat-vec scaled), gradient pass on {grows} rows ({t_gk:.2f} s scaled), panel work "
                   f"(K_uf, trsm, syrk, preconditioner GEMVs {t_pc*1e3:.0f} ms/apply, adjoint GEMM) on {ncol} of {N} columns; "
                   f"evaluation = {n_mv} mat-vecs + {steps_cg + 2} preconditioner applies + gradient + common terms "
                   f"= {t_eval:.1f} s"),
    }


def training_secondary(X, y, Z, hyp, kind, iters=30):
    """The reference's REAL workload (pytorch/interface.py:445-543; xpert-main.toml: 2 000 L-BFGS-B steps): warm-started evaluations
    through the backend mirror.  `iters` SciPy L-BFGS-B iterations from the headline hypers with Z as given (the greedy selection is
    not part of the loop); reports wall time per evaluation, the mean CG step count, and where the device time of an evaluation goes
    (HIP events inside cglb_objective_and_grad, cglb_set_option "eval_profile")."""
    import tempfile
    from cglb_amd.backend import interface
    from cglb_amd.backend.callbacks import Logger
    from cglb_amd.backend.models import CGLB, BaseKernel, GaussianLikelihood, InducingPointKernel, ScaleKernel
    D = X.shape[1]
    base = BaseKernel(kind, ard_num_dims=D)
    base.lengthscale = hyp["lengthscales"]
    scale = ScaleKernel(base)
    scale.outputscale = hyp["variance"]
    lik = GaussianLikelihood(lower_bound=1e-6)
    lik.noise = hyp["noise"]
    model = CGLB((X, y), lik, InducingPointKernel(scale, Z))
    tmp = tempfile.mkdtemp(prefix="cglb_bench_")
    logger = Logger(tmp, lambda: {}, lambda: {}, holdout_interval=-1, include_feval_log=True, verbose=False)
    hip = model.hip
    walls = []
    orig = hip.objective_and_grad

    def timed(*a, **k):
        t = time.perf_counter()
        r = orig(*a, **k)
        walls.append(1e3 * (time.perf_counter() - t))
        return r
    hip.objective_and_grad = timed
    hip.set_option("eval_profile", 1)      # the warm-up evaluation of optimize() is inside: dropped from the host-side medians below
    t0 = time.perf_counter()
    results = interface.optimize(model, ((X, y), (X[:8], y[:8])), iters, logger, "scipy")
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    n = hip.get_stat("eval_count")
    phases = {k: hip.get_stat(f"eval_{k}_ms") / max(n, 1.0) for k in ("setup", "pcg", "final", "grad")}
    hip.set_option("eval_profile", 0)
    steps = np.asarray(logger.logs["steps-per-feval"], dtype=np.float64)
    nfev, nit = int(sum(r.nfev for r in results)), int(sum(r.nit for r in results))
    out = {"workload": f"{nit} L-BFGS-B iterations / {nfev} warm-started evaluations through cglb_amd.backend.optimize, same shape and start hypers",
           "ms_per_evaluation_wall": 1e3 * wall / max(nfev + 1, 1), "ms_per_evaluation_c_abi_median": float(np.median(walls[1:])),
           "mean_cg_steps": float(steps.mean()), "max_cg_steps": float(steps.max()),
           "device_ms_per_evaluation": phases, "device_ms_step_independent": phases["setup"] + phases["final"] + phases["grad"],
           "evaluations": nfev, "iterations": nit, "final_loss": float(results[-1].fun), "L_diag_ratio": hip.get_stat("L_diag_ratio")}
    hip.close()
    return out


def check_against_fixture(N, D, M, kind, hyp_name, res, steps=None, bound=None):
    """Compare steps / bound of a cold-start evaluation with tests/golden/headline/*.npz (data, not oracle code).  north_star: 1e-6
    relative on the bound; step count exact up to 40 steps, +-1 beyond (DESIGN.md section 6 note).  Raises on a mismatch."""
    name = {("rbf", "trained"): "headline_rbf_trained", ("rbf", "init"): "headline_rbf_init", ("matern32", "trained"): "headline_m32_trained"}
    path = os.path.join(ROOT, "tests", "golden", "headline", name.get((kind, hyp_name), "none") + ".npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    if (int(g["N"]), int(g["D"]), int(g["M"])) != (N, D, M):
        return None
    steps = res.steps if res is not None else steps
    bound = res.bound if res is not None else bound
    ref_steps, ref_bound = int(g["steps"]), float(g["bound"])
    rel = abs(bound - ref_bound) / abs(ref_bound)
    ok = abs(steps - ref_steps) <= (0 if ref_steps <= 40 else 1) and rel <= 1e-6
    out = {"fixture": os.path.relpath(path, ROOT), "ref_steps": ref_steps, "steps": steps, "ref_bound": ref_bound, "bound": bound,
           "bound_rel_err": rel, "ok": bool(ok)}
    if not ok:
        raise SystemExit(f"bench.py: result disagrees with the reference-solver fixture: {out}")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=int, default=100_000, help="N (training rows)")
    ap.add_argument("--dims", type=int, default=8, help="D (input dimension)")
    ap.add_argument("--inducing", type=int, default=1024, help="M (inducing points)")
    ap.add_argument("--kernel", default="rbf", choices=["rbf", "matern32"])
    ap.add_argument("--hypers", default="trained", choices=["trained", "init"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    args = ap.parse_args()

    from cglb_amd.data import reference_init_hypers, synthetic_problem, trained_like_hypers
    from cglb_amd.distributed import Comm, HipSymLocalOps, SymShardedCGLB, row_partition
    from cglb_amd.hip_context import HipContext

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; the product path has no CPU fallback")
    # Rehearsal knobs for a one-GPU box only (the real multi-GPU run uses neither): CGLB_BENCH_BACKEND=gloo and
    # CGLB_BENCH_SHARE_GPU=1 let several ranks share cuda:0 so that the N > 1 code path can be exercised end to end.
    backend = os.environ.get("CGLB_BENCH_BACKEND", "nccl")
    if os.environ.get("CGLB_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # CGLB_BENCH_FORCE_DIST=1: run the distributed driver and issue its collectives even at world_size 1 (rehearses the RCCL calls
    # on a one-GPU box; never set by the real runs)
    force_dist = os.environ.get("CGLB_BENCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    stdout_fd = None
    if use_dist:
        # RCCL prints a version banner on the process's stdout at communicator creation: keep fd 1 pointed at stderr until the JSON
        # line is printed, so that rank 0's stdout carries exactly that one line
        sys.stdout.flush()
        stdout_fd = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # 2-minute collective timeout: a rank that falls out of step ends the job with a non-zero exit instead of hanging it
        from datetime import timedelta
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=timedelta(minutes=2))  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend, timeout=timedelta(minutes=2))

    N, D, M, kind = args.rows, args.dims, args.inducing, args.kernel
    X, y, Z = synthetic_problem(N, D, M, seed=0)
    hypers = {"trained": trained_like_hypers(D), "init": reference_init_hypers(D)}
    cg = dict(max_error=1.0, max_cg_iter=100, restart_cg_iter=40)  # conjugate_gradient.py:37-39

    per, parts = row_partition(N, world)
    comm = Comm(force=force_dist)
    # N > 1: cyclic-symmetric scheme (each kernel value used twice, global upper triangle dealt to the ranks by 256-row blocks).
    # Default driver: the loops INSIDE the library with RCCL on the context stream (cglb_dist_*, cglb_amd/dist_context.py) - one host
    # round trip per PCG iteration.  CGLB_BENCH_DRIVER=python selects the host-driven twin (torch.distributed collectives between the
    # C calls); if the in-library communicator cannot be created the bench falls back to it and says so in the record.
    driver = os.environ.get("CGLB_BENCH_DRIVER", "native") if use_dist else "fused"
    driver_note = None
    ctx = drv = None
    if driver == "native":
        try:
            from cglb_amd.dist_context import DistHipContext
            ctx = DistHipContext(X, y, M, kind, device=dev, collectives="rccl" if backend == "nccl" else "callbacks", force=force_dist)
        except Exception as exc:  # noqa: BLE001 - any failure to set the communicator up: use the host-driven twin, keep the reason
            driver_note = f"native driver unavailable ({type(exc).__name__}: {exc}); host-driven twin used"
            driver, ctx = "python", None
        ok = torch.tensor([1.0 if driver == "native" else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)                 # all ranks take the same driver
        if float(ok.item()) == 0.0 and driver == "native":
            ctx.close()
            driver, ctx, driver_note = "python", None, "native driver unavailable on another rank; host-driven twin used"
    if ctx is None:
        ctx = HipContext(X, y, M, kind, device=dev, row_range=parts[rank])
        if use_dist:
            drv = SymShardedCGLB(HipSymLocalOps(ctx), comm)
    v = torch.zeros(N, dtype=torch.float64, device=dev)

    def barrier():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def one_step():
        if drv is None:   # single GPU (fused C call) or N ranks with the loops inside the library: same call, full replicated v
            v.zero_()
            return ctx.objective_and_grad(v, True, cg["max_error"], cg["max_cg_iter"], cg["restart_cg_iter"], with_grad=True)
        drv.v.zero_()
        return drv.objective_and_grad(True, cg["max_error"], cg["max_cg_iter"], cg["restart_cg_iter"], with_grad=True)

    k1_stats = {}

    def run(name, steps, warmup):
        h = hypers[name]
        ctx.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-6)
        res = None
        for _ in range(warmup):
            res = one_step()
        barrier()
        ctx.set_option("k1_profile", 1)   # HIP events around every launch of the dominant kernel inside the timed region
        t0 = time.perf_counter()
        for _ in range(steps):
            res = one_step()
        barrier()
        dt = time.perf_counter() - t0
        k1_stats[name] = (ctx.get_stat("k1_ms_total"), ctx.get_stat("k1_launches"))
        ctx.set_option("k1_profile", 0)
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, res

    dt, res = run(args.hypers, args.steps, args.warmup)
    secondary = None
    if not args.no_secondary:
        other = "init" if args.hypers == "trained" else "trained"
        dt2, res2 = run(other, max(1, min(args.steps, 3)), 1)
        k2 = max(1, min(args.steps, 3))
        secondary = {"workload_hypers": other, "value": k2 / dt2, "unit": "evals/s", "ms_per_step": dt2 / k2 * 1e3,
                     "cg_steps": res2.steps, "bound": res2.bound}
        # restore the headline hypers for the kernel timings below
        h = hypers[args.hypers]
        ctx.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-6)
        if not use_dist and kind == "rbf":
            # SURVEY 8(d) names both kernels at the headline shape: the same evaluation with Matern-3/2 (single GPU only)
            ctx_m = HipContext(X, y, M, "matern32", device=dev)
            ctx_m.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-6)
            vm = torch.zeros(N, dtype=torch.float64, device=dev)
            resm = ctx_m.objective_and_grad(vm, True, cg["max_error"], cg["max_cg_iter"], cg["restart_cg_iter"], with_grad=True)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(2):
                vm.zero_()
                resm = ctx_m.objective_and_grad(vm, True, cg["max_error"], cg["max_cg_iter"], cg["restart_cg_iter"], with_grad=True)
            torch.cuda.synchronize(dev)
            dtm = (time.perf_counter() - t0) / 2
            secondary["matern32"] = {"workload_hypers": args.hypers, "value": 1.0 / dtm, "unit": "evals/s", "ms_per_step": dtm * 1e3,
                                     "cg_steps": resm.steps, "bound": resm.bound, "kff_matvec_ms": ctx_m.time_kernel(0, 5),
                                     "k1_ms": ctx_m.time_kernel(3, 5)}
            ctx_m.close()
            del ctx_m, vm
        if not use_dist:
            secondary["training"] = training_secondary(X, y, Z, hypers[args.hypers], kind)
        if not use_dist and (N, D, M, kind) == (100_000, 8, 1024, "rbf"):
            # inputs of 33-96 dimensions (the wide Wilson sets: buzz 77, song 90): the two N^2 passes stay register-resident there
            # (DESIGN.md section 4 "Mid widths"); one cold evaluation at N = 50 000, D = 77 for the record
            Nw, Dw = 50_000, 77
            Xw, yw, Zw = synthetic_problem(Nw, Dw, M, seed=0)
            ctx_w = HipContext(Xw, yw, M, "rbf", device=dev)
            lsw = np.full(Dw, 1.2 * np.sqrt(Dw))
            ctx_w.set_hypers(lsw, 1.0, 0.05, 0.0, Zw, 1e-6)
            vw = torch.zeros(Nw, dtype=torch.float64, device=dev)
            resw = ctx_w.objective_and_grad(vw, True, cg["max_error"], cg["max_cg_iter"], cg["restart_cg_iter"], with_grad=True)
            vw.zero_()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            ctx_w.set_hypers(lsw, 1.0, 0.05, 0.0, Zw, 1e-6)
            resw = ctx_w.objective_and_grad(vw, True, cg["max_error"], cg["max_cg_iter"], cg["restart_cg_iter"], with_grad=True)
            torch.cuda.synchronize(dev)
            dtw = time.perf_counter() - t0
            secondary["mid_width"] = {"workload": f"N={Nw} D={Dw} M={M} rbf fp64, cold start", "value": 1.0 / dtw, "unit": "evals/s", "ms_per_step": dtw * 1e3,
                                      "cg_steps": resw.steps, "bound": resw.bound, "kff_matvec_ms": ctx_w.time_kernel(0, 3), "grad_pass_ms": ctx_w.time_kernel(2, 2)}
            ctx_w.close()
            del ctx_w, vw, Xw, yw, Zw

    # ---- roofline of the dominant kernel, measured live with HIP events on the library's stream -------------
    if driver == "native":
        ctx.setup()
    else:
        ctx.setup_local()
        if use_dist:
            comm.allreduce(ctx.aat_tensor())
        ctx.setup_finish()
    reps = 10
    # dominant kernel: average duration of its launches INSIDE the timed region (HIP events on the library's stream, rank 0's share
    # of the triangle when N > 1); the stand-alone figure (back-to-back launches outside the solver) is kept beside it
    k1_ms_total, k1_launches = k1_stats[args.hypers]
    ms_pair = k1_ms_total / max(k1_launches, 1.0)
    ms_pair_standalone = ctx.time_kernel(3 if world == 1 else 4, reps)
    ms_prec = ctx.time_kernel(1, reps)   # preconditioner apply (gemv_u + triangular products + gemv_t + epilogue)
    nloc = parts[rank][1] - parts[rank][0]
    # Work of ONE launch of the pair kernel = (flops per pair evaluation) x (pairs the launch evaluates).  The kernel is
    # the symmetric form: each unordered pair {i, j} is evaluated once and used for out_i and out_j, so a launch evaluates
    # ~N(N + 256)/2 pairs on one GPU (the library reports the exact count of its work list), not N^2.
    pairs_eval = ctx.get_stat("k1_pairs_per_launch")
    fpp = pair_flops(kind, D)
    achieved = fpp * pairs_eval / (ms_pair * 1e-3) / 1e12
    esz = 8
    prec_bytes = 2.0 * M * nloc * esz + 3.0 * nloc * esz
    # Counter-derived figures come from committed rocprofv3 --pmc passes of this same command (tools/pmc_collect.sh); they are
    # quoted only for the workload they were collected on and name their source file.
    traffic_k1 = traffic_prec = executed = None
    def newest(pattern):   # the most recent round's committed counter summary (profiles/rNN_*.json)
        import glob
        found = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
        return os.path.relpath(found[-1], ROOT) if found else os.path.join("profiles", pattern)
    pmc_hbm = newest("r[0-9][0-9]_pmc_hbm_traffic.json")
    pmc_sq = newest("r[0-9][0-9]_pmc_sq_counters.json")
    on_profiled_workload = world == 1 and (N, D, M, kind) == (100_000, 8, 1024, "rbf")
    if on_profiled_workload and os.path.exists(os.path.join(ROOT, pmc_hbm)):
        pk = json.load(open(os.path.join(ROOT, pmc_hbm)))["kernels"]
        traffic_k1 = pk["kff_sym_kernel"]["hbm_bytes_per_launch"]
        traffic_prec = sum(pk[k]["hbm_bytes_per_launch"] for k in ("gemv_u_kernel", "gemv_t_kernel", "precond_z_kernel") if k in pk)
    if on_profiled_workload and os.path.exists(os.path.join(ROOT, pmc_sq)):
        sq = json.load(open(os.path.join(ROOT, pmc_sq)))["kernels"].get("kff_sym_kernel", {})
        if "executed_flop_per_launch" in sq:
            executed = {"flop_per_launch": sq["executed_flop_per_launch"],
                        "tflops": sq["executed_flop_per_launch"] / (ms_pair * 1e-3) / 1e12,
                        "frac_of_peak": sq["executed_flop_per_launch"] / (ms_pair * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                        "valu_wave_instructions_per_launch": sq.get("valu_insts_per_launch"),
                        # every fp64 / VOP3 instruction occupies its SIMD for 4 cycles: issue cycles per SIMD / launch time = the clock
                        # the kernel would need if it did nothing but issue -> ~the sustained clock means the issue pipe is full
                        "valu_issue_ghz_needed": (sq.get("valu_insts_per_launch", 0.0) * 4.0 / 1024.0) / (ms_pair * 1e-3) / 1e9,
                        "source": pmc_sq}
    roofline = {
        "kernel": "kff_sym_kernel (pair kernel of the implicit K_ff mat-vec)", "bound": "valu_fp64", "achieved": achieved,
        "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_VECTOR_PEAK_TFLOPS,
        "traffic": traffic_k1, "traffic_source": pmc_hbm if traffic_k1 is not None else None,
        "ms_per_launch": ms_pair, "launches_timed": int(k1_launches), "ms_per_launch_standalone": ms_pair_standalone,
        "pairs_evaluated_per_launch": pairs_eval, "flop_per_pair": fpp,
        # time of a non-symmetric kernel doing all N^2/world pairs at the vector-fp64 peak, over the measured time: how the launch
        # compares with SURVEY 8(d)'s floor for this mat-vec (NOT a fraction of peak; > 1 is the gain of the symmetric form)
        "vs_nonsymmetric_floor": (fpp * float(N) * float(N) / world / (FP64_VECTOR_PEAK_TFLOPS * 1e12)) / (ms_pair * 1e-3),
        "executed": executed,
        "algorithmic": (f"{fpp} flop/pair x {pairs_eval:.4g} evaluated pairs per launch (symmetric form: each unordered pair once); K_ff is never "
                        f"materialised, algorithmic HBM bytes N(D+2)w = {N * (D + 2) * 8 / 1e6:.1f} MB"),
    }
    assert 0.0 < roofline["frac"] <= 1.0, roofline
    roofline_hbm = {
        "kernel": "nystrom preconditioner apply (gemv_u_kernel + gemv_t_kernel over the stored panel A + precond_z_kernel)", "bound": "hbm",
        "achieved": prec_bytes / (ms_prec * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": prec_bytes / (ms_prec * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic_prec,
        "traffic_source": pmc_hbm if traffic_prec is not None else None, "ms_per_apply": ms_prec,
        "algorithmic": "2 M n_local w + 3 n_local w bytes per apply",
    }
    assert 0.0 < roofline_hbm["frac"] <= 1.0, roofline_hbm
    # gradient bilinear pass ((5D+6) flop per pair), same symmetric enumeration of the pairs
    roofline_grad = None
    if world == 1:
        ms_grad = ctx.time_kernel(2, 5)
        gflop = (5 * D + 6) * pairs_eval
        roofline_grad = {"kernel": "grad_kff_gram_kernel (N^2 bilinear form of the lengthscale gradient)", "bound": "valu_fp64",
                         "achieved": gflop / (ms_grad * 1e-3) / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": gflop / (ms_grad * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS, "traffic": None, "ms_per_launch": ms_grad,
                         "algorithmic": f"{5 * D + 6} flop/pair x {pairs_eval:.4g} evaluated pairs (stand-alone launches)"}
        assert 0.0 < roofline_grad["frac"] <= 1.0, roofline_grad
    if secondary and "matern32" in secondary:
        mm = secondary["matern32"]
        mflop = pair_flops("matern32", D) * pairs_eval
        mm["roofline"] = {"kernel": "kff_sym_kernel<matern32>", "bound": "valu_fp64", "achieved": mflop / (mm["k1_ms"] * 1e-3) / 1e12,
                          "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": mflop / (mm["k1_ms"] * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS, "traffic": None, "ms_per_launch": mm["k1_ms"],
                          "algorithmic": f"{pair_flops('matern32', D)} flop/pair x {pairs_eval:.4g} evaluated pairs (stand-alone launches)"}
        assert 0.0 < mm["roofline"]["frac"] <= 1.0, mm["roofline"]

    # ---- self-check of the timed result against the committed headline fixture (made by the reference's own solver loop over the
    # blocked C operator, oracle/gen_headline_fixture.py): a numerically broken kernel must not produce a bench line
    parity = check_against_fixture(N, D, M, kind, args.hypers, res)
    if secondary is not None and parity is not None:
        other_fx = check_against_fixture(N, D, M, kind, secondary["workload_hypers"], None, steps=secondary["cg_steps"], bound=secondary["bound"])
        if other_fx is not None:
            parity["secondary"] = other_fx
        if "matern32" in secondary:
            m_fx = check_against_fixture(N, D, M, "matern32", secondary["matern32"]["workload_hypers"], None,
                                         steps=secondary["matern32"]["cg_steps"], bound=secondary["matern32"]["bound"])
            if m_fx is not None:
                parity["matern32"] = m_fx

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        restarts = res.steps // cg["restart_cg_iter"]
        cpu = cpu_baseline(kind, X, y, Z, hypers[args.hypers], N, D, M, res.steps, restarts)

    if rank == 0:
        record = {
            "metric": "cglb_objective_grad_evals_per_sec", "value": args.steps / dt, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"CGLB objective+gradient, synthetic N={N} D={D} M={M} {kind} fp64, hypers={args.hypers}, cold start v=0",
                "N": N, "D": D, "M": M, "kernel": kind, "hypers": {k: (np.asarray(val).tolist()) for k, val in hypers[args.hypers].items()},
                "cg": cg, "parallelism": (f"cyclic-symmetric K_ff blocks + column-sharded Nystrom panel x{world}, "
                                f"{'loops and RCCL collectives inside libcglb_hip.so' if driver == 'native' else 'host-driven collectives (torch.distributed)'}")
                if world > 1 else "single GPU",
            },
            "driver": driver, "driver_note": driver_note,
            "cg_steps": res.steps, "cg_residual_error": res.residual_error, "bound": res.bound,
            "roofline": roofline, "roofline_hbm": roofline_hbm, "roofline_grad": roofline_grad, "cpu_baseline": cpu, "secondary": secondary,
            "parity_check": parity,
        }
        if stdout_fd is not None:
            sys.stdout.flush()
            os.dup2(stdout_fd, 1)
        print(json.dumps(record), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except (FloatingPointError, RuntimeError) as exc:   # non-finite stop statistic / failed HIP or RCCL call: fresh-process semantics
        print(f"bench.py: rank {os.environ.get('RANK', '0')} failed: {exc}", file=sys.stderr, flush=True)
        sys.exit(3)
