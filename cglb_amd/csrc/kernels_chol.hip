// Blocked lower Cholesky of the two small M x M matrices of the common terms (K_uu + jitter I and B = A A^T + I,
// reference models.py:202, :210).  rocSOLVER's potrf spends 2.9 ms on a 1024^2 matrix (latency of ~40 small kernels with a
// 155-us unblocked panel step); this right-looking version factors the 64 x 64 diagonal block in the registers of one wave, solves the
// panel below it with one row per thread against scalar-loaded factors, and leaves the trailing rank-64 update to rocBLAS syrk (a true dense contraction).
// Column-major, in place; the strict upper triangle is not referenced or modified.
#include "devmath.h"
#include "dispatch.h"

#ifndef CHOL_NB
#define CHOL_NB 64
#endif

__device__ __forceinline__ double readlane_t(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float readlane_t(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Factor the nb x nb diagonal block at (k0, k0) with ONE wave: lane r owns row r of the block in registers (fully unrolled,
// static indices).  What another lane needs from row j — L[j][t], t < j — is taken straight out of lane j's registers with
// v_readlane into SGPRs and used as the scalar operand of the fma: no LDS, no barrier (an LDS broadcast per term exposed its
// full latency 2016 times: 58 us; this form is issue-bound).
//   step j:  s_r = a_r[j] - sum_{t<j} L[r][t] L[j][t]  (all lanes);  L[j][j] = sqrt(s_j);  L[r][j] = s_r / L[j][j]  (r > j)
// A short block (nb < 64, last block of a ragged matrix) is padded with the identity.  info[0] receives k0 + j + 1 for the
// first non-positive pivot (only if still 0); the block is then left unwritten.  Dblk receives a dense transposed copy of the
// factored block (Dblk[t * 64 + c] = L[c][t] for c > t, zero elsewhere) plus the reciprocal diagonal for the panel kernel.
template <typename T>
__global__ __launch_bounds__(64) void chol_diag_kernel(T* __restrict__ A, int n, int k0, int nb, int* __restrict__ info,
                                                       T* __restrict__ Dblk) {
    const int r = threadIdx.x;
    T a[CHOL_NB];
#pragma unroll
    for (int cc = 0; cc < CHOL_NB; ++cc) {
        T v = (cc == r) ? T(1) : T(0);  // identity padding
        if (r < nb && cc < nb) v = (cc <= r) ? A[(size_t)(k0 + cc) * n + k0 + r] : T(0);
        a[cc] = v;
    }
    int bad = 0;
    // (a right-looking order - scale column j, then update all later columns with readlane'd factors - measured 72 us against
    //  38 us for this left-looking one: its readlanes depend on the value written just before)
#pragma unroll
    for (int j = 0; j < CHOL_NB; ++j) {
        T s0 = a[j], s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
        for (int t = 0; t < j; ++t) {
            const T ljt = readlane_t(a[t], j);  // wave-uniform: L[j][t]
            if ((t & 3) == 0) s0 = tfma<T>(-a[t], ljt, s0);
            else if ((t & 3) == 1) s1 = tfma<T>(-a[t], ljt, s1);
            else if ((t & 3) == 2) s2 = tfma<T>(-a[t], ljt, s2);
            else s3 = tfma<T>(-a[t], ljt, s3);
        }
        const T s = (s0 + s1) + (s2 + s3);
        const T sj = readlane_t(s, j);
        T pv = T(1);
        if (sj > T(0)) pv = sqrt(sj);
        else if (bad == 0) bad = j + 1;
        a[j] = (r == j) ? pv : (r > j ? s / pv : T(0));
    }
    if (bad != 0) {
        if (r == 0 && info[0] == 0) info[0] = k0 + bad;
        return;
    }
#pragma unroll
    for (int cc = 0; cc < CHOL_NB; ++cc) {
        if (r < nb && cc <= r && cc < nb) A[(size_t)(k0 + cc) * n + k0 + r] = a[cc];
        if (r < CHOL_NB) Dblk[cc * CHOL_NB + r] = (cc < r) ? a[cc] : T(0);  // strictly lower part, transposed: Dblk[t][c] = L[c][t]
    }
    T diag = T(1);
#pragma unroll
    for (int cc = 0; cc < CHOL_NB; ++cc)
        if (cc == r) diag = a[cc];
    if (r < CHOL_NB) Dblk[CHOL_NB * CHOL_NB + r] = T(1) / diag;
}

// Panel below the diagonal block: X L_kk^T = P, one row of P per thread in registers (fully unrolled).  L_kk comes from the
// dense read-only copy Dblk with wave-uniform indices, i.e. through scalar loads into SGPRs that feed the fma directly.
// FULL: nb == 64 (every block but a ragged last one): loads and stores are fully static.
template <typename T, bool FULL>
__global__ __launch_bounds__(64) void chol_panel_kernel(T* __restrict__ A, int n, int k0, int nb, const T* __restrict__ Dblk) {
    const int row = k0 + nb + blockIdx.x * 64 + threadIdx.x;
    if (row >= n) return;
    T x[CHOL_NB];
    if (FULL) {
#pragma unroll
        for (int cc = 0; cc < CHOL_NB; ++cc) x[cc] = A[(size_t)(k0 + cc) * n + row];
    } else {
#pragma unroll
        for (int cc = 0; cc < CHOL_NB; ++cc) x[cc] = T(0);
        for (int cc = 0; cc < nb; ++cc) {  // the select chain keeps x[] in registers despite the dynamic bound
            const T v = A[(size_t)(k0 + cc) * n + row];
#pragma unroll
            for (int q = 0; q < CHOL_NB; ++q)
                if (q == cc) x[q] = v;
        }
    }
    // right-looking substitution: once x[t] is final every later entry is updated independently (63 - t parallel fmas per
    // step, so a lone wave has no dependent chain to wait on); column t of L_kk is contiguous in Dblk (stored transposed)
#pragma unroll
    for (int t = 0; t < CHOL_NB; ++t) {
        x[t] *= Dblk[CHOL_NB * CHOL_NB + t];
#pragma unroll
        for (int cc = t + 1; cc < CHOL_NB; ++cc) x[cc] = tfma<T>(-x[t], Dblk[t * CHOL_NB + cc], x[cc]);
    }
    if (FULL) {
#pragma unroll
        for (int cc = 0; cc < CHOL_NB; ++cc) A[(size_t)(k0 + cc) * n + row] = x[cc];
    } else {
        for (int cc = 0; cc < nb; ++cc) {
            T v = T(0);
#pragma unroll
            for (int q = 0; q < CHOL_NB; ++q)
                if (q == cc) v = x[q];
            A[(size_t)(k0 + cc) * n + row] = v;
        }
    }
}

static inline rocblas_status xsyrk2(rocblas_handle h, int n, int k, const double* A, int lda, double* C, int ldc) {
    const double a = -1.0, b = 1.0;
    return rocblas_dsyrk(h, rocblas_fill_lower, rocblas_operation_none, n, k, &a, A, lda, &b, C, ldc);
}
static inline rocblas_status xsyrk2(rocblas_handle h, int n, int k, const float* A, int lda, float* C, int ldc) {
    const float a = -1.0f, b = 1.0f;
    return rocblas_ssyrk(h, rocblas_fill_lower, rocblas_operation_none, n, k, &a, A, lda, &b, C, ldc);
}

template <typename T>
static int cholesky_impl(cglb_ctx* c, T* A, int* info_slot) {
    const int n = c->M;
    if (!c->chol_blk) HIP_CHECK(c, hipMalloc(&c->chol_blk, (CHOL_NB * CHOL_NB + CHOL_NB) * sizeof(double)));
    T* Dblk = (T*)c->chol_blk;
    HIP_CHECK(c, hipMemsetAsync(info_slot, 0, sizeof(int), c->stream));
    for (int k0 = 0; k0 < n; k0 += CHOL_NB) {
        const int nb = n - k0 < CHOL_NB ? n - k0 : CHOL_NB;
        hipLaunchKernelGGL((chol_diag_kernel<T>), dim3(1), dim3(64), 0, c->stream, A, n, k0, nb, info_slot, Dblk);
        CGLB_LAUNCH_CHECK(c);
        const int rest = n - k0 - nb;
        if (rest > 0) {
            if (nb == CHOL_NB)
                hipLaunchKernelGGL((chol_panel_kernel<T, true>), dim3((rest + 63) / 64), dim3(64), 0, c->stream, A, n, k0, nb, (const T*)Dblk);
            else
                hipLaunchKernelGGL((chol_panel_kernel<T, false>), dim3((rest + 63) / 64), dim3(64), 0, c->stream, A, n, k0, nb, (const T*)Dblk);
            CGLB_LAUNCH_CHECK(c);
            // A22 -= P P^T (lower)
            BLAS_CHECK(c, xsyrk2(c->blas, rest, nb, A + (size_t)k0 * n + k0 + nb, n, A + (size_t)(k0 + nb) * n + k0 + nb, n));
        }
    }
    return CGLB_OK;
}

// In-place lower Cholesky of the column-major M x M matrix `A`; *info_slot (device int) = 0 or 1 + index of the first
// non-positive pivot.  A failed factorisation leaves garbage below the failing block (callers check info before use).
int launch_cholesky_lower(cglb_ctx* c, void* A, int* info_slot) {
    CGLB_DISPATCH_T(c->dtype, return cholesky_impl<T>(c, (T*)A, info_slot));
}
