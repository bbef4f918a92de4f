/*
 * CPU oracle, C/OpenMP part.  TEST INFRASTRUCTURE ONLY (checker + timed CPU baseline "port"):
 * nothing under cglb_amd/ links or loads this.
 *
 * Blocked restatement of the streaming pieces of the reference hot path, so that K_ff (80 GB at
 * N = 100k) is never formed on the CPU either:
 *   orc_kff_matvec  —  `A @ p`, A = kernel(x).add_diag(sigma^2)   (reference cglb/backend/pytorch/
 *                      models.py:251-252; conjugate_gradient.py:57,66,72)
 *   orc_cross       —  rectangular k(X1, X2) @ v                   (models.py:320,334)
 *   orc_kernel_block—  dense K(X1, X2) block (K_uf, K_uu)          (models.py:196-201)
 *   orc_grad_kff    —  sum_ij u_i dK_ij/dl_d v_j                   (autograd of models.py:280)
 * Kernel closed forms follow oracle/cglb_oracle.py (direct differences, fp64, libm exp/sqrt).
 * Parity: checked against the numpy oracle (which is pinned to the reference's golden vectors) in
 * tests/test_oracle_c.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_RBF 0
#define ORC_MATERN32 1
#define SQRT3 1.7320508075688772935
#define MAXD 64
#define JB 256 /* column block held in L1 */

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

static inline double kappa(int kind, double d2) {
    if (kind == ORC_RBF) return exp(-0.5 * d2);
    const double r = sqrt(d2);
    return (1.0 + SQRT3 * r) * exp(-SQRT3 * r);
}
static inline double hfac(int kind, double d2) {
    if (kind == ORC_RBF) return exp(-0.5 * d2);
    return 3.0 * exp(-SQRT3 * sqrt(d2));
}

/* scaled copy Xs[i][d] = X[i][d] / l_d */
static double* scaled(const double* X, int64_t n, int D, const double* ls) {
    double* Xs = (double*)malloc(sizeof(double) * (size_t)n * D);
    for (int64_t i = 0; i < n; ++i)
        for (int d = 0; d < D; ++d) Xs[i * D + d] = X[i * D + d] / ls[d];
    return Xs;
}

/* out[i - r0] = var * sum_j kappa(x_i, x_j) p_j + noise * p_i, rows r0 <= i < r1 of an N x N operator */
int orc_kff_matvec(int kind, int64_t N, int D, const double* X, const double* ls, double var, double noise, const double* p,
                   int64_t r0, int64_t r1, double* out) {
    if (D > MAXD) return 1;
    double* Xs = scaled(X, N, D, ls);
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = r0; i < r1; ++i) {
        const double* xi = Xs + i * D;
        double acc = 0.0;
        for (int64_t jb = 0; jb < N; jb += JB) {
            const int64_t je = jb + JB < N ? jb + JB : N;
            double d2[JB];
            for (int64_t j = jb; j < je; ++j) {
                const double* xj = Xs + j * D;
                double s = 0.0;
                for (int d = 0; d < D; ++d) {
                    const double df = xi[d] - xj[d];
                    s += df * df;
                }
                d2[j - jb] = s;
            }
            for (int64_t j = jb; j < je; ++j) acc += kappa(kind, d2[j - jb]) * p[j];
        }
        out[i - r0] = var * acc + noise * p[i];
    }
    free(Xs);
    return 0;
}

/* out[i] = var * sum_j kappa(x1_i, x2_j) v_j */
int orc_cross(int kind, int64_t n1, int64_t n2, int D, const double* X1, const double* X2, const double* ls, double var,
              const double* v, double* out) {
    double* A = scaled(X1, n1, D, ls);
    double* B = scaled(X2, n2, D, ls);
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < n1; ++i) {
        double acc = 0.0;
        for (int64_t j = 0; j < n2; ++j) {
            double s = 0.0;
            for (int d = 0; d < D; ++d) {
                const double df = A[i * D + d] - B[j * D + d];
                s += df * df;
            }
            acc += kappa(kind, s) * v[j];
        }
        out[i] = var * acc;
    }
    free(A);
    free(B);
    return 0;
}

/* dense block out[i * n2 + j] = var * kappa(x1_i, x2_j) */
int orc_kernel_block(int kind, int64_t n1, int64_t n2, int D, const double* X1, const double* X2, const double* ls, double var,
                     double* out) {
    double* A = scaled(X1, n1, D, ls);
    double* B = scaled(X2, n2, D, ls);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n1; ++i)
        for (int64_t j = 0; j < n2; ++j) {
            double s = 0.0;
            for (int d = 0; d < D; ++d) {
                const double df = A[i * D + d] - B[j * D + d];
                s += df * df;
            }
            out[i * n2 + j] = var * kappa(kind, s);
        }
    free(A);
    free(B);
    return 0;
}

/* dl[d] = sum_{r0<=i<r1} sum_j u_i (dK_ij/dl_d) v_j,  dK/dl_d = var * h * delta_d^2 / l_d */
int orc_grad_kff(int kind, int64_t N, int D, const double* X, const double* ls, double var, const double* u, const double* v,
                 int64_t r0, int64_t r1, double* dl) {
    if (D > MAXD) return 1;
    double* Xs = scaled(X, N, D, ls);
    for (int d = 0; d < D; ++d) dl[d] = 0.0;
#pragma omp parallel
    {
        double loc[MAXD];
        for (int d = 0; d < D; ++d) loc[d] = 0.0;
#pragma omp for schedule(dynamic, 16)
        for (int64_t i = r0; i < r1; ++i) {
            const double* xi = Xs + i * D;
            double acc[MAXD];
            for (int d = 0; d < D; ++d) acc[d] = 0.0;
            for (int64_t j = 0; j < N; ++j) {
                const double* xj = Xs + j * D;
                double sq[MAXD], s = 0.0;
                for (int d = 0; d < D; ++d) {
                    const double df = xi[d] - xj[d];
                    sq[d] = df * df;
                    s += sq[d];
                }
                const double hv = hfac(kind, s) * v[j];
                for (int d = 0; d < D; ++d) acc[d] += hv * sq[d];
            }
            for (int d = 0; d < D; ++d) loc[d] += u[i] * acc[d];
        }
#pragma omp critical
        for (int d = 0; d < D; ++d) dl[d] += loc[d] * var / ls[d];
    }
    free(Xs);
    return 0;
}
