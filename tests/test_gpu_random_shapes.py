"""Randomised parity sweep: shapes off every tile grid (N, D, M drawn at random, both kernels, both precision levels) against the dense
numpy oracle - mat-vec, bound at the oracle's own v, gradient, and the CG step count - plus the named draws of earlier long sweeps
that needed an explanation.  `tools/fuzz_parity.py` runs longer sweeps with the same acceptance rules (`check_case`): fixed
tolerances for everything that does not depend on the CG path; for the CG path exact agreement unless the oracle's OWN answer moves
under a kernel-accuracy-sized perturbation of its operator, in which case a small multiple of that measured spread is admissible
(oracle.roundoff_sensitivity; DESIGN.md section 2 "parity policy")."""
import numpy as np
import pytest

from tools.fuzz_parity import check_case, named_case

pytestmark = pytest.mark.gpu


def _draw(rng):
    N = int(rng.choice([rng.integers(2, 300), rng.integers(300, 3000), rng.integers(3000, 7000)]))
    D = int(rng.integers(1, 33))
    M = int(min(N, rng.choice([rng.integers(1, 70), rng.integers(60, 200), rng.integers(200, 600)])))
    kind, prec, tol, seed = str(rng.choice(["rbf", "matern32"])), int(rng.integers(0, 2)), float(rng.choice([1.0, 1e-2])), int(rng.integers(1 << 30))
    ls = rng.uniform(0.7, 2.5, size=D) * np.sqrt(D / 2.0)
    variance, noise, mean = float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.02, 0.5)), float(rng.normal() * 0.1)
    return dict(N=N, D=D, M=M, kind=kind, prec=prec, tol=tol, data_seed=seed, ls=ls, variance=variance, noise=noise, mean=mean,
                p=rng.standard_normal(N), jitter=1e-6)


@pytest.mark.parametrize("case", range(14))
def test_random_shape_matches_oracle(case):
    ok, line, d = check_case(_draw(np.random.default_rng(1000 + case)))
    assert ok, line
    # beyond the shared rules: none of these 14 draws is in the chaotic regime, so they are held to the tight form outright
    assert abs(d["res"].steps - d["ref"].steps) <= 1, line
    assert d["e_b2"] < 1e-9, line


def test_named_draw_weak_preconditioner_is_cg_chaos_not_a_kernel_defect():
    """Sweep seed 2024, draw 169 (N=2398, D=26, M=7, RBF, fast level, tol 1): 31-32 steps with a 7-point preconditioner
    (cond 6.6e3, top eigenvalue 1264 against 98).  The bound after the solve differed from the oracle's by 1.0e-4 relative while the
    bound re-assembled at the GPU's own v agreed to 2e-15.  The oracle itself is not reproducible here: under an eps-sized perturbation
    of its operator its bound moves by 5e-5 relative and its step count by one - asserted below, so that the acceptance of the GPU
    result rests on a measurement, not on a comment."""
    c = named_case(2024, 169)
    assert (c["N"], c["D"], c["M"], c["kind"], c["prec"], c["tol"]) == (2398, 26, 7, "rbf", 1, 1.0)
    ok, line, d = check_case(c)
    assert ok, line
    assert d["e_b2"] < 1e-12 and d["e_mv"] < 1e-12, line
    from oracle import cglb_oracle as orc
    X, y, Z = orc.synthetic_problem(c["N"], c["D"], c["M"], seed=c["data_seed"])
    hyp = orc.Hypers(c["ls"], c["variance"], c["noise"], c["mean"], Z, 1e-6)
    s = orc.roundoff_sensitivity(c["kind"], X, y, hyp, np.zeros(c["N"]), 1.0, 100, 40, delta=2.0 ** -52)
    assert s.bound_spread > 1e-6 * abs(s.bound), "the oracle became reproducible on this draw: tighten the test"


@pytest.mark.parametrize("precision", [0, 1])
def test_named_draw_inducing_points_on_nearly_every_datum(precision):
    """Sweep seed 2024, draw 186 (N=561, D=1, M=533, RBF): cond(K_uu) = 3e8.  The Z gradient is ~1e-11 of the bound and carries
    cond * eps of absolute error in any implementation; the GPU's must stay within 10x the oracle's own floor under eps-level
    perturbations of Z (oracle.grad_roundoff_spread), at both precision levels."""
    c = named_case(2024, 186)
    assert (c["N"], c["D"], c["M"], c["kind"]) == (561, 1, 533, "rbf")
    c["prec"] = precision
    ok, line, _ = check_case(c)
    assert ok, line


@pytest.mark.parametrize("index,shape", [(48, (6661, 20, 1, "matern32", 0, 1.0)), (83, (6441, 2, 6, "matern32", 0, 0.01))])
def test_named_draw_summation_order_moves_the_stopping_point(index, shape):
    """Sweep seed 7, draws 48 and 83 (Matern-3/2, exact level, 1 and 6 inducing points: next to no preconditioner, 32-38 steps).  The GPU
    stopped one step away from the oracle although four probes of the oracle did not move its step count.  Measured on draw 83
    (tools/history_compare.py, profiles/r03_history_7_83.log): the GPU's 1/2 r^T P r equals the oracle's to 1e-14 through iteration 10;
    from there any eps-sized change of the operator - one entry pair, Fortran instead of C order of the same matrix - moves the
    ORACLE's statistic by 5e-12, 3e-9, 2e-6, 7e-3, 0.7 at iterations 11-15 (x 600 per iteration, the extreme Ritz values having
    converged), the trajectories re-approach to 1e-6 at iteration 19 and part again.  So the rule is: the GPU tracks the oracle for as
    long as the oracle reproduces itself (asserted through `tracked`), and past that point a stopping step is admissible when the
    oracle's statistic has moved by enough, at some iteration up to there, to cover its distance from the tolerance."""
    c = named_case(7, index)
    assert (c["N"], c["D"], c["M"], c["kind"], c["prec"], c["tol"]) == shape
    ok, line, d = check_case(c)
    assert ok, line
    assert d["e_b2"] < 1e-13 and d["e_mv"] < 1e-13, line
    k0, track = d["tracked"]
    assert k0 >= 8 and track < 1e-8, (k0, track, line)
