"""Developer tool: coefficients of the table-driven 2^x polynomials in cglb_amd/csrc/devmath.h.

  centred form : P(s) ~ 2^(s / T)          on |s| <= 1/2, P(0) = 1 exactly   (range reduction by round-to-nearest)
  floor form   : P(s) ~ 2^((s - 1/2) / T)  on 0 <= s < 1, free constant term  (range reduction by floor/fract; the table
                 then holds 2^((k + 1/2) / T))
Fit: interpolation at Chebyshev nodes in 60-digit arithmetic (within a few per cent of the minimax error at these tiny
arguments), coefficients rounded to fp64, error re-measured with the rounded coefficients evaluated in exact arithmetic.

usage: python tools/exp2_poly_fit.py [TAB_BITS ...]
"""
import sys

import mpmath as mp

mp.mp.dps = 60


def cheb_nodes(a, b, n):
    return [(a + b) / 2 + (b - a) / 2 * mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]


def fit(f, a, b, deg, fixed_c0=None):
    """Coefficients c[0..deg] (ascending).  With fixed_c0 the polynomial is c0 + s * q(s), q fitted to (f - c0) / s."""
    if fixed_c0 is None:
        xs = cheb_nodes(a, b, deg + 1)
        A = mp.matrix([[x ** j for j in range(deg + 1)] for x in xs])
        y = mp.matrix([f(x) for x in xs])
        c = mp.lu_solve(A, y)
        return [c[j] for j in range(deg + 1)]
    xs = cheb_nodes(a, b, deg)
    A = mp.matrix([[x ** j for j in range(deg)] for x in xs])
    y = mp.matrix([(f(x) - fixed_c0) / x if abs(x) > mp.mpf(10) ** -40 else mp.diff(f, 0) for x in xs])
    c = mp.lu_solve(A, y)
    return [mp.mpf(fixed_c0)] + [c[j] for j in range(deg)]


def max_rel_err(f, a, b, coef):
    worst = mp.mpf(0)
    for k in range(4001):
        x = a + (b - a) * k / 4000
        p = sum(mp.mpf(float(c)) * x ** j for j, c in enumerate(coef))
        worst = max(worst, abs(p / f(x) - 1))
    return worst


def main():
    bits_list = [int(a) for a in sys.argv[1:]] or [8]
    for bits in bits_list:
        T = mp.mpf(2) ** bits
        for deg in (2, 3, 4, 5):
            f = lambda s: mp.mpf(2) ** (s / T)
            c = fit(f, mp.mpf(-0.5), mp.mpf(0.5), deg, fixed_c0=1)
            print(f"TAB_BITS={bits} centred degree {deg}: max rel err {mp.nstr(max_rel_err(f, mp.mpf(-0.5), mp.mpf(0.5), c), 3)}")
            print("   ", ", ".join(float(x).hex() for x in c[1:]), " (c1..)")
            g = lambda s: mp.mpf(2) ** ((s - mp.mpf(0.5)) / T)
            c = fit(g, mp.mpf(0), mp.mpf(1), deg)
            print(f"TAB_BITS={bits} floor   degree {deg}: max rel err {mp.nstr(max_rel_err(g, mp.mpf(0), mp.mpf(1), c), 3)}")
            print("   ", ", ".join(float(x).hex() for x in c), " (c0..)")


if __name__ == "__main__":
    main()
