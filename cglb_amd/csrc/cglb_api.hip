// C ABI of libcglb_hip.so: context, common terms (rocSOLVER/rocBLAS for the true dense contractions),
// preconditioner, PCG loop, objective assembly and analytic gradient.  See include/cglb_hip.h.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

#include <rccl/rccl.h>

#include "devmath.h"
#include "dispatch.h"

namespace {

thread_local std::string g_create_error;

// scalar slots in ctx->scal
enum { S_RZ = 0, S_PAP = 1, S_NRZ = 2, S_TRACE = 3, S_SUMLOG = 4, S_TRBINV = 5, S_TMP = 6, S_SC = 8, S_TMP2 = 16, S_LDIAG = 56 };

// ---- typed rocBLAS / rocSOLVER wrappers ---------------------------------------------------------------
inline rocblas_status xpotrf(rocblas_handle h, rocblas_fill u, int n, double* A, int lda, rocblas_int* info) { return rocsolver_dpotrf(h, u, n, A, lda, info); }
inline rocblas_status xpotrf(rocblas_handle h, rocblas_fill u, int n, float* A, int lda, rocblas_int* info) { return rocsolver_spotrf(h, u, n, A, lda, info); }
inline rocblas_status xtrtri(rocblas_handle h, rocblas_fill u, rocblas_diagonal d, int n, double* A, int lda, rocblas_int* info) { return rocsolver_dtrtri(h, u, d, n, A, lda, info); }
inline rocblas_status xtrtri(rocblas_handle h, rocblas_fill u, rocblas_diagonal d, int n, float* A, int lda, rocblas_int* info) { return rocsolver_strtri(h, u, d, n, A, lda, info); }
inline rocblas_status xtrsm(rocblas_handle h, rocblas_side s, rocblas_fill u, rocblas_operation t, rocblas_diagonal d, int m, int n, const double* al, const double* A, int lda, double* B, int ldb) { return rocblas_dtrsm(h, s, u, t, d, m, n, al, A, lda, B, ldb); }
inline rocblas_status xtrsm(rocblas_handle h, rocblas_side s, rocblas_fill u, rocblas_operation t, rocblas_diagonal d, int m, int n, const float* al, const float* A, int lda, float* B, int ldb) { return rocblas_strsm(h, s, u, t, d, m, n, al, A, lda, B, ldb); }
inline rocblas_status xsyrk(rocblas_handle h, rocblas_fill u, rocblas_operation t, int n, int k, const double* al, const double* A, int lda, const double* be, double* C, int ldc) { return rocblas_dsyrk(h, u, t, n, k, al, A, lda, be, C, ldc); }
inline rocblas_status xsyrk(rocblas_handle h, rocblas_fill u, rocblas_operation t, int n, int k, const float* al, const float* A, int lda, const float* be, float* C, int ldc) { return rocblas_ssyrk(h, u, t, n, k, al, A, lda, be, C, ldc); }
inline rocblas_status xgemm(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, const double* al, const double* A, int lda, const double* B, int ldb, const double* be, double* C, int ldc) { return rocblas_dgemm(h, ta, tb, m, n, k, al, A, lda, B, ldb, be, C, ldc); }
inline rocblas_status xgemm(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, const float* al, const float* A, int lda, const float* B, int ldb, const float* be, float* C, int ldc) { return rocblas_sgemm(h, ta, tb, m, n, k, al, A, lda, B, ldb, be, C, ldc); }
inline rocblas_status xgemm_sb(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, const double* al, const double* A, int lda, rocblas_stride sa, const double* B, int ldb, rocblas_stride sb, const double* be, double* C, int ldc, rocblas_stride sc, int batch) { return rocblas_dgemm_strided_batched(h, ta, tb, m, n, k, al, A, lda, sa, B, ldb, sb, be, C, ldc, sc, batch); }
inline rocblas_status xgemm_sb(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, const float* al, const float* A, int lda, rocblas_stride sa, const float* B, int ldb, rocblas_stride sb, const float* be, float* C, int ldc, rocblas_stride sc, int batch) { return rocblas_sgemm_strided_batched(h, ta, tb, m, n, k, al, A, lda, sa, B, ldb, sb, be, C, ldc, sc, batch); }
inline rocblas_status xtrsv(rocblas_handle h, rocblas_fill u, rocblas_operation t, rocblas_diagonal d, int n, const double* A, int lda, double* x, int inc) { return rocblas_dtrsv(h, u, t, d, n, A, lda, x, inc); }
inline rocblas_status xtrsv(rocblas_handle h, rocblas_fill u, rocblas_operation t, rocblas_diagonal d, int n, const float* A, int lda, float* x, int inc) { return rocblas_strsv(h, u, t, d, n, A, lda, x, inc); }

// ---- small helper kernels -------------------------------------------------------------------------------
// out = a * I + b * X + c * Y   (M x M, column-major; Y may be null)
template <typename T>
__global__ __launch_bounds__(256) void mat_combine_kernel(T* __restrict__ out, int M, T a, T b, const T* __restrict__ X, T cc, const T* __restrict__ Y) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)M * M) return;
    const int i = (int)(idx % M), j = (int)(idx / M);
    T v = b * X[idx];
    if (Y) v = tfma<T>(cc, Y[idx], v);
    if (i == j) v += a;
    out[idx] = v;
}

// AAt = sum_b slab[b] (fixed order), lower triangle summed and mirrored so the result is exactly symmetric
template <typename T>
__global__ __launch_bounds__(256) void slab_reduce_sym_kernel(const T* __restrict__ slabs, int nslab, int M, T* __restrict__ out, int accumulate) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t mm = (int64_t)M * M;
    if (idx >= mm) return;
    const int i = (int)(idx % M), j = (int)(idx / M);
    if (i < j) return;
    T s = accumulate ? out[idx] : T(0);  // later groups of slabs continue the running sum of the earlier ones
    for (int b = 0; b < nslab; ++b) s += slabs[(int64_t)b * mm + idx];
    out[idx] = s;
    out[(int64_t)i * M + j] = s;
}

template <typename T>
__global__ __launch_bounds__(256) void trace_kernel(const T* __restrict__ Mc, int M, double* __restrict__ slot) {
    __shared__ double smem[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < M; i += blockDim.x) s += (double)Mc[(int64_t)i * M + i];
    s = block_sum(s, smem);
    if (threadIdx.x == 0) slot[0] = s;
}

// y = a * x (length n);  y2 = b * x optional
template <typename T>
__global__ __launch_bounds__(256) void scale2_kernel(const T* __restrict__ x, int64_t n, T a, T* __restrict__ y, T b, T* __restrict__ y2) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T xv = x[i];
        y[i] = a * xv;
        if (y2) y2[i] = b * xv;
    }
}

// out = a*x + b*y
template <typename T>
__global__ __launch_bounds__(256) void axpby_kernel(T* __restrict__ out, T a, const T* __restrict__ x, T b, const T* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = a * x[i] + b * y[i];
}

// scalar-only gradient terms (added once, by the shard that owns row 0)
__global__ void grad_scalar_terms_kernel(double* __restrict__ out, int D, const double* __restrict__ sc, const double* __restrict__ trBinv,
                                         double N, double M, double f, double s, double tau, double T) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    out[D] += sc[5] / f - N / (2.0 * tau * s);
    out[D + 1] += sc[2] + 0.5 * sc[3] + (M - trBinv[0]) / (2.0 * s) - N / (2.0 * s) + N * f / (2.0 * tau * s * s) - T / (2.0 * tau * s);
    out[D + 2] += sc[4];
}

// prediction epilogue: per new point n
//   mean[n] = cg_mean[n] + sum_m tmp2[m][n] c[m] + mu ; var[n] = f + sum tmp2^2 - sum tmp1^2    (models.py:347-351)
template <typename T>
__global__ __launch_bounds__(256) void predict_finish_kernel(const T* __restrict__ tmp1, const T* __restrict__ tmp2, int64_t ld, int M,
                                                             const T* __restrict__ cvec, int64_t n_new, T mu, T f, T* __restrict__ mean_io,
                                                             T* __restrict__ var_out) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_new) return;
    T s1 = 0, s2 = 0, sm = 0;
    for (int m = 0; m < M; ++m) {
        const T a = tmp1[(int64_t)m * ld + n], b = tmp2[(int64_t)m * ld + n];
        s1 = tfma<T>(a, a, s1);
        s2 = tfma<T>(b, b, s2);
        sm = tfma<T>(b, cvec[m], sm);
    }
    mean_io[n] = mean_io[n] + sm + mu;
    var_out[n] = f + s2 - s1;
}

// Kus panel for prediction: out[m*ld + n] = var*kappa(z_m, xnew_n)
template <typename T, int KIND, int DP>
__global__ __launch_bounds__(256) void kus_kernel(const T* __restrict__ Zs, const T* __restrict__ XsNew, int64_t n_new, int64_t ld, int M, T var,
                                                  T* __restrict__ out) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_new) return;
    T x[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) x[d] = XsNew[n * DP + d];
    const int m0 = blockIdx.y * 32, m1 = min(M, m0 + 32);
    for (int m = m0; m < m1; ++m) {
        T d2 = 0;
#pragma unroll
        for (int d = 0; d < DP; ++d) {
            const T df = Zs[(int64_t)m * DP + d] - x[d];
            d2 = tfma<T>(df, df, d2);
        }
        out[(int64_t)m * ld + n] = var * kappa_from_d2<T, KIND>(d2);
    }
}

inline int grid1d_full(int64_t n) { return (int)((n + 255) / 256); }  // one thread per element, no stride loop

inline int grid1d(int64_t n) {
    int64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

// per-call device temporaries: freed on every exit path of the function that owns the holder
struct DevTemps {
    std::vector<void*> ptrs;
    ~DevTemps() { for (void* p : ptrs) if (p) (void)hipFree(p); }
    int alloc(cglb_ctx* c, void** out, size_t bytes) {
        *out = nullptr;
        hipError_t e = hipMalloc(out, bytes ? bytes : 16);
        if (e != hipSuccess) return cglb_fail(c, CGLB_ERR_HIP, std::string("hipMalloc of a temporary: ") + hipGetErrorString(e));
        ptrs.push_back(*out);
        return CGLB_OK;
    }
};

int dalloc(cglb_ctx* c, void** p, size_t bytes) {
    if (bytes == 0) bytes = 16;
    HIP_CHECK(c, hipMalloc(p, bytes));
    return CGLB_OK;
}

int read_scalars(cglb_ctx* c, const double* dev, double* host, int n) {
    HIP_CHECK(c, hipMemcpyAsync(host, dev, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return CGLB_OK;
}

inline double tau_of(const cglb_ctx* c) { return 1.0 + c->var / c->noise - c->trace_AAt / (double)c->N; }

// ---- common terms -------------------------------------------------------------------------------------------
template <typename T>
int setup_local_impl(cglb_ctx* c) {
    if (!c->have_data || !c->have_hypers) return cglb_fail(c, CGLB_ERR_STATE, "set_data and set_hypers must precede setup");
    const int M = c->M;
    // K_uu + jitter I -> L (models.py:200-202)
    c->have_Linv = false;
    CGLB_TRY(launch_kuu(c));
    if (c->chol_mode == 1) CGLB_TRY(launch_cholesky_lower(c, c->Lc, (int*)c->info_dev));  // blocked LDS Cholesky (kernels_chol.hip)
    else BLAS_CHECK(c, xpotrf(c->blas, rocblas_fill_lower, M, (T*)c->Lc, M, c->info_dev));
    rocblas_int info = 0;
    double ldiag[2] = {1.0, 1.0};
    CGLB_TRY(launch_diag_minmax(c, c->Lc, c->scal + S_LDIAG));   // rides on the read-back of the factorisation status
    HIP_CHECK(c, hipMemcpyAsync(&info, c->info_dev, sizeof(info), hipMemcpyDeviceToHost, c->stream));
    CGLB_TRY(read_scalars(c, c->scal + S_LDIAG, ldiag, 2));
    if (info != 0) return cglb_fail(c, CGLB_ERR_NOT_PD, "cholesky(K_uu + jitter I) failed: leading minor " + std::to_string(info) + " not positive definite");
    c->L_diag_ratio = ldiag[0] > 0.0 ? ldiag[1] / ldiag[0] : 1.0e300;
    CGLB_TRY(launch_tri_clean(c, c->Lc, 1));
    if (c->precond_mode == 1) {  // explicit L^-1 in both orientations for the implicit preconditioner
        HIP_CHECK(c, hipMemcpyAsync(c->Linv, c->Lc, (size_t)M * M * c->esz, hipMemcpyDeviceToDevice, c->stream));
        BLAS_CHECK(c, xtrtri(c->blas, rocblas_fill_lower, rocblas_diagonal_non_unit, M, (T*)c->Linv, M, c->info_dev + 2));
        CGLB_TRY(launch_transpose(c, c->Linv, c->LinvT));
        rocblas_int info_inv = 0;
        HIP_CHECK(c, hipMemcpyAsync(&info_inv, c->info_dev + 2, sizeof(info_inv), hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(c, hipStreamSynchronize(c->stream));
        if (info_inv != 0) return cglb_fail(c, CGLB_ERR_NOT_PD, "inverse of L failed: zero pivot " + std::to_string(info_inv));
        c->have_Linv = true;
    }
    // K_uf shard -> A = L^-1 K_uf / sigma  (models.py:196-197, :206).  Column-major view: At (nloc x M) L^T = Kuf^T / sigma.
    if (c->nloc > 0) {
        CGLB_TRY(launch_kuf(c));
        const T alpha = (T)(1.0 / std::sqrt(c->noise));
        BLAS_CHECK(c, xtrsm(c->blas, rocblas_side_right, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit,
                            (int)c->nloc, M, &alpha, (const T*)c->Lc, M, (T*)c->At, (int)c->lda));
        // partial A A^T (models.py:207): C = At^T At.  The contraction runs over nloc (10^5) with a 1024^2 output, a shape
        // rocBLAS syrk/gemm serialise badly (88 ms measured); split K into chunks with a strided-batched TN GEMM into
        // slabs and sum the slabs in fixed order (reproducible).
        const T one = 1, zero = 0;
        const int64_t kc = 2048;
        const int nfull = (int)(c->nloc / kc);
        const int64_t rem = c->nloc - (int64_t)nfull * kc;
        const int nslab_total = nfull + (rem > 0 ? 1 : 0);
        // The slabs are processed in groups of at most `group` (64 at M = 1024 = 512 MB; fewer for larger M so that a group stays
        // within ~1 GB): scratch stays bounded for any nloc, and the groups are added to A A^T in order (fixed order: reproducible).
        int group = (int)(((size_t)1 << 30) / ((size_t)M * M * c->esz));
        if (group < 4) group = 4;
        if (group > 64) group = 64;
        if (group > nslab_total) group = nslab_total;
        const size_t need = (size_t)group * M * M * c->esz;
        if (need > c->slab_cap) {
            if (c->slabs) HIP_CHECK(c, hipFree(c->slabs));
            c->slabs = nullptr;
            HIP_CHECK(c, hipMalloc(&c->slabs, need));
            c->slab_cap = need;
        }
        T* slabs = (T*)c->slabs;
        const T* At = (const T*)c->At;
        // Only the lower block triangle is computed (the slab sum below reads i >= j and mirrors): with `bs`-wide blocks that is
        // nb (nb + 1) / 2 of nb^2 block products - 75 % of the flops at M = 1024 with the default bs = 512 (256-wide blocks lose more in GEMM efficiency than they save: 9.6 vs 9.4 ms of setup).
        const int bs = (c->aat_block > 0 && M % c->aat_block == 0 && M >= 2 * c->aat_block) ? c->aat_block : M;
        for (int s0 = 0; s0 < nslab_total; s0 += group) {
            const int ns = std::min(group, nslab_total - s0);            // slabs of this group
            const int nf = std::min(ns, std::max(nfull - s0, 0));         // ... of which full 2048-column chunks
            const bool tail = (s0 + ns == nslab_total) && rem > 0;        // the short last chunk belongs to this group
            for (int bj = 0; bj < M; bj += bs)
                for (int bi = bj; bi < M; bi += bs) {
                    const T* Ai = At + (int64_t)bi * c->lda + (int64_t)s0 * kc;  // columns bi.. of the column-major (nloc x M) view
                    const T* Aj = At + (int64_t)bj * c->lda + (int64_t)s0 * kc;
                    T* Cij = slabs + bi + (int64_t)bj * M;
                    if (nf > 0)
                        BLAS_CHECK(c, xgemm_sb(c->blas, rocblas_operation_transpose, rocblas_operation_none, bs, bs, (int)kc, &one, Ai, (int)c->lda, kc,
                                               Aj, (int)c->lda, kc, &zero, Cij, M, (rocblas_stride)M * M, nf));
                    if (tail)
                        BLAS_CHECK(c, xgemm(c->blas, rocblas_operation_transpose, rocblas_operation_none, bs, bs, (int)rem, &one,
                                            Ai + (int64_t)nf * kc, (int)c->lda, Aj + (int64_t)nf * kc, (int)c->lda, &zero,
                                            Cij + (int64_t)nf * M * M, M));
                }
            hipLaunchKernelGGL((slab_reduce_sym_kernel<T>), dim3(grid1d_full((int64_t)M * M)), dim3(256), 0, c->stream, (const T*)slabs, ns, M,
                               (T*)c->AAt, s0 > 0 ? 1 : 0);
            CGLB_LAUNCH_CHECK(c);
        }
    } else {
        HIP_CHECK(c, hipMemsetAsync(c->AAt, 0, (size_t)M * M * c->esz, c->stream));
    }
    c->have_local = true;
    c->have_terms = false;
    return CGLB_OK;
}

template <typename T>
int setup_finish_impl(cglb_ctx* c) {
    if (!c->have_local) return cglb_fail(c, CGLB_ERR_STATE, "setup_local must precede setup_finish");
    const int M = c->M;
    const size_t mm = (size_t)M * M * c->esz;
    // B = AA^T + I, LB = chol(B), tr(AA^T)  (models.py:208-211)
    HIP_CHECK(c, hipMemcpyAsync(c->LBc, c->AAt, mm, hipMemcpyDeviceToDevice, c->stream));
    CGLB_TRY(launch_add_identity_trace(c, c->LBc, c->scal + S_TRACE));
    if (c->chol_mode == 1) CGLB_TRY(launch_cholesky_lower(c, c->LBc, (int*)c->info_dev));
    else BLAS_CHECK(c, xpotrf(c->blas, rocblas_fill_lower, M, (T*)c->LBc, M, c->info_dev));
    CGLB_TRY(launch_tri_clean(c, c->LBc, 1));
    CGLB_TRY(launch_sum_log_diag(c, c->LBc, c->scal + S_SUMLOG));
    // explicit triangular inverse of LB in both orientations (contiguous rows for the two products of
    // LB^-T LB^-1 u, conjugate_gradient.py:106-107)
    HIP_CHECK(c, hipMemcpyAsync(c->LBinv, c->LBc, mm, hipMemcpyDeviceToDevice, c->stream));
    BLAS_CHECK(c, xtrtri(c->blas, rocblas_fill_lower, rocblas_diagonal_non_unit, M, (T*)c->LBinv, M, c->info_dev + 1));
    CGLB_TRY(launch_transpose(c, c->LBinv, c->LBinvT));
    rocblas_int info[2] = {0, 0};
    double sc[2];
    HIP_CHECK(c, hipMemcpyAsync(info, c->info_dev, sizeof(info), hipMemcpyDeviceToHost, c->stream));
    CGLB_TRY(read_scalars(c, c->scal + S_TRACE, sc, 2));
    if (info[0] != 0) return cglb_fail(c, CGLB_ERR_NOT_PD, "cholesky(A A^T + I) failed: leading minor " + std::to_string(info[0]));
    if (info[1] != 0) return cglb_fail(c, CGLB_ERR_NOT_PD, "inverse of LB failed: zero pivot " + std::to_string(info[1]));
    c->trace_AAt = sc[0];
    c->sum_log_diag_LB = sc[1];
    c->have_terms = true;
    return CGLB_OK;
}

// u = A r for the local column shard: stored panel (reference form, conjugate_gradient.py:105) or implicitly as
// sigma^-1 L^-1 (K_uf r) with the tiled pair kernel (no pass over the 819 MB panel)
int precond_u_any(cglb_ctx* c, const void* r_local, void* u_out) {
    if (c->precond_mode == 0) return launch_gemv_u(c, r_local, u_out);
    const char* pcol = (const char*)r_local - (size_t)c->r0 * c->esz;  // the pair kernel indexes columns absolutely
    CGLB_TRY(launch_pairs_rect(c, c->Zh, c->zah, c->M, c->Xh, c->xah, pcol, c->r0, c->r1, c->w_q));
    CGLB_TRY(launch_tri_rowdot(c, c->LinvT, c->w_q, 1, u_out));
    return launch_scale(c, u_out, 1.0 / std::sqrt(c->noise), c->M);
}
// z = (r - A^T t)/noise, rz = r^T z (conjugate_gradient.py:110-113); implicit form: A^T t = K_fu (L^-T t / sigma)
int precond_z_any(cglb_ctx* c, const void* r_local, const void* t, void* z_local, double* rz_slot) {
    if (c->precond_mode == 0) return launch_precond_z(c, r_local, t, z_local, rz_slot);
    CGLB_TRY(launch_tri_rowdot(c, c->Linv, t, 0, c->w_q));
    CGLB_TRY(launch_scale(c, c->w_q, 1.0 / std::sqrt(c->noise), c->M));
    const char* xrow = (const char*)c->Xh + (size_t)c->r0 * c->Dp * c->esz;
    const char* arow = (const char*)c->xah + (size_t)c->r0 * c->esz;
    CGLB_TRY(launch_pairs_rect(c, xrow, arow, c->nloc, c->Zh, c->zah, c->w_q, 0, c->M, c->tpart));
    return launch_precond_z_from(c, r_local, c->tpart, z_local, rz_slot);
}

int precond_single(cglb_ctx* c, const void* r, void* z, double* rz_slot) {
    CGLB_TRY(precond_u_any(c, r, c->w_u));
    CGLB_TRY(launch_tri_apply(c, c->w_u, c->w_t));
    CGLB_TRY(precond_z_any(c, r, c->w_t, z, rz_slot));
    return CGLB_OK;
}

int require_terms(cglb_ctx* c) {
    if (!c->have_terms) return cglb_fail(c, CGLB_ERR_STATE, "common terms not computed (call cglb_setup)");
    return CGLB_OK;
}
int require_single(cglb_ctx* c) {
    if (c->r0 != 0 || c->r1 != c->N) return cglb_fail(c, CGLB_ERR_STATE, "fused call needs a single shard covering all rows");
    return CGLB_OK;
}

// look-ahead threshold: speculate on the next mat-vec while 1/2 r^T P r of the PREVIOUS iteration exceeds factor x max_error
// (option "pcg_lookahead": 0 off, 1 the default factor, k >= 2 that factor)
inline double lookahead_factor(const cglb_ctx* c) { return c->pcg_lookahead >= 2 ? (double)c->pcg_lookahead : CGLB_LOOKAHEAD_FACTOR; }

// ---- PCG (conjugate_gradient.py:41-86) ----------------------------------------------------------------------
int pcg_impl(cglb_ctx* c, const void* b, void* v, double max_error, int max_iter, int restart_iter, int* steps, double* half_rz) {
    double* S = c->scal;
    // A weighted copy p o w left behind by the LAST update of an earlier solve (loop left without a look-ahead mat-vec) must not be
    // taken for the operand of this solve's first mat-vec: the direction vector is rewritten below.
    c->pwh_src = nullptr;
    // :57-61  Av = A v ; r = b - Av ; z, rz = P(r) ; p = z
    // A cold start (v == 0, models.py:59-68) gives Av == 0 and r == b exactly, so the mat-vec is skipped in that case;
    // the result is bit-identical to computing it.
    double vnorm = 0.0;
    CGLB_TRY(launch_dot(c, v, v, c->nloc, S + S_TMP));
    CGLB_TRY(read_scalars(c, S + S_TMP, &vnorm, 1));
    if (vnorm == 0.0) {
        HIP_CHECK(c, hipMemcpyAsync(c->w_r, b, (size_t)c->nloc * c->esz, hipMemcpyDeviceToDevice, c->stream));
    } else {
        CGLB_TRY(launch_kff_matvec(c, v, c->w_Kv, nullptr));
        CGLB_TRY(launch_residual(c, c->w_r, b, c->w_Kv));
    }
    double *s_rz = S + S_RZ, *s_nrz = S + S_NRZ;  // the two slots swap roles every iteration (:76) instead of being copied
    CGLB_TRY(precond_single(c, c->w_r, c->w_z, s_rz));
    CGLB_TRY(launch_update_p(c, c->w_p, c->w_z, s_rz, s_rz, 1, -1, true));  // :61 p = z (+ the weighted copy for the first mat-vec)
    double rz = 0;
    CGLB_TRY(read_scalars(c, s_rz, &rz, 1));
    // The stop predicate (:65) is evaluated on the host, like the reference's (:80-81).  Look-ahead: while the residual is
    // still far above the tolerance (more than CGLB_LOOKAHEAD_FACTOR = 32x after the PREVIOUS iteration), the mat-vec of the next iteration is enqueued
    // before the host waits for this iteration's scalar, so the GPU does not idle over the read-back.  If the predicate then
    // says stop, that mat-vec was wasted (it only writes Ap and the p.Ap slot): results are identical either way.
    int i = 0;
    bool ahead = false;
    while (0.5 * rz > max_error && i < max_iter) {  // :65
        if (!ahead) CGLB_TRY(launch_kff_matvec(c, c->w_p, c->w_Ap, S + S_PAP));                    // :66 and (p*Ap).sum()
        const int restart = (restart_iter > 0) && (i % restart_iter == restart_iter - 1);          // :70
        CGLB_TRY(launch_update_v_r(c, v, c->w_r, c->w_p, c->w_Ap, s_rz, S + S_PAP, !restart));  // :67-68, :72
        if (restart) {
            CGLB_TRY(launch_kff_matvec(c, v, c->w_Kv, nullptr));
            CGLB_TRY(launch_residual(c, c->w_r, b, c->w_Kv));
        }
        CGLB_TRY(precond_single(c, c->w_r, c->w_z, s_nrz));                                         // :73
        CGLB_TRY(launch_update_p(c, c->w_p, c->w_z, s_nrz, s_rz, restart, -1, true));               // :75 (+ weighted copy for :66)
        std::swap(s_rz, s_nrz);                                                                     // :76
        HIP_CHECK(c, hipMemcpyAsync(c->host_scal, s_rz, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(c, hipEventRecord(c->scal_event, c->stream));
        ahead = c->pcg_lookahead && (i + 1 < max_iter) && (0.5 * rz > lookahead_factor(c) * max_error);  // rz: still the value of the previous iteration
        if (ahead) CGLB_TRY(launch_kff_matvec(c, c->w_p, c->w_Ap, S + S_PAP));
        HIP_CHECK(c, hipEventSynchronize(c->scal_event));                                            // host test of :65 (and the sync of :80-81)
        rz = c->host_scal[0];
        ++i;
    }
    if (steps) *steps = i;
    if (half_rz) *half_rz = 0.5 * rz;  // :83
    return CGLB_OK;
}

// ---- objective phases -----------------------------------------------------------------------------------------
int obj_phase1(cglb_ctx* c, const void* v_full, void* u_partial) {
    const char* y_loc = (const char*)c->y + (size_t)c->r0 * c->esz;
    CGLB_TRY(launch_sub_scalar(c, c->w_e, y_loc, c->mean, c->nloc));       // models.py:253-254
    CGLB_TRY(launch_kff_matvec(c, v_full, c->w_Kv, nullptr));              // :280
    CGLB_TRY(launch_residual(c, c->w_r, c->w_e, c->w_Kv));                 // :281
    CGLB_TRY(precond_u_any(c, c->w_r, u_partial));                         // first half of precon(r), :282
    return CGLB_OK;
}

// Phase 1 without the mat-vec of models.py:280 (option "final_matvec" = 0): straight after a solve, w_r still holds the residual the
// PCG recurrence carries, r = e - K v up to the rounding of its updates (exact at the start of the solve and after every restart step,
// conjugate_gradient.py:58,72), so K v = e - r costs one vector kernel instead of N^2 pair evaluations.  The reference recomputes
// `cov @ v`; the two differ at the level of the mat-vec's own rounding (measured: DESIGN.md section 5).
int obj_phase1_reuse(cglb_ctx* c, void* u_partial) {
    CGLB_TRY(launch_residual(c, c->w_Kv, c->w_e, c->w_r));                 // K v = e - r   (w_e = y - mean was the solve's right-hand side)
    CGLB_TRY(precond_u_any(c, c->w_r, u_partial));
    return CGLB_OK;
}

int obj_phase2(cglb_ctx* c, const void* v_full, const void* u, double* sc_partial, void* aw_partial) {
    const char* v_loc = (const char*)v_full + (size_t)c->r0 * c->esz;
    CGLB_TRY(launch_tri_apply(c, u, c->w_t));
    CGLB_TRY(precond_z_any(c, c->w_r, c->w_t, c->w_z, c->scal + S_TMP));  // w = P r
    CGLB_TRY(launch_obj_scalars(c, v_loc, c->w_r, c->w_Kv, c->w_z, sc_partial));
    CGLB_TRY(launch_gemv_u(c, c->w_z, aw_partial));                           // A w  (for c = Kuu^-1 Kuf w)
    return CGLB_OK;
}

// L^-1 (lower, column-major) for the gradient algebra: one rocBLAS trtri per evaluation, after which the three M x M triangular
// solves and the trsv of the adjoints are plain GEMMs / a row-dot kernel (rocBLAS trsm with an M x M right-hand side inverts the
// 128-wide diagonal blocks itself and then issues ~20 small GEMMs: 0.27 ms each at M = 1024, trsv 0.17 ms).  L comes from a
// successful Cholesky factorisation (positive diagonal), so the inversion cannot meet a zero pivot.
template <typename T>
int ensure_Linv(cglb_ctx* c) {
    if (c->have_Linv) return CGLB_OK;
    const int M = c->M;
    HIP_CHECK(c, hipMemcpyAsync(c->Linv, c->Lc, (size_t)M * M * c->esz, hipMemcpyDeviceToDevice, c->stream));
    BLAS_CHECK(c, xtrtri(c->blas, rocblas_fill_lower, rocblas_diagonal_non_unit, M, (T*)c->Linv, M, c->info_dev + 2));
    c->have_Linv = true;
    c->Linv_unchecked = true;  // info_dev[2] is read with the next host read-back of the evaluation (obj_finish)
    return CGLB_OK;
}

template <typename T>
int obj_phase3_impl(cglb_ctx* c, const void* v_full, const double* sc, const void* aw, double* out, const void* u_full_cyclic = nullptr) {
    const int M = c->M, D = c->D;
    const size_t glen = (size_t)CGLB_GRAD_LEN(D, M);
    const double s = c->noise, f = c->var, sigma = std::sqrt(s), tau = tau_of(c);
    const T one = 1, zero = 0;
    HIP_CHECK(c, hipMemsetAsync(out, 0, glen * sizeof(double), c->stream));
    if (!c->Guf) CGLB_TRY(dalloc(c, &c->Guf, (size_t)M * c->lda * c->esz));
    // B^-1 = LB^-T LB^-1 and its trace
    BLAS_CHECK(c, xgemm(c->blas, rocblas_operation_transpose, rocblas_operation_none, M, M, M, &one, (const T*)c->LBinv, M, (const T*)c->LBinv, M,
                        &zero, (T*)c->Mtmp, M));
    hipLaunchKernelGGL((trace_kernel<T>), dim3(1), dim3(256), 0, c->stream, (const T*)c->Mtmp, M, c->scal + S_TRBINV);
    // c = Kuu^-1 Kuf w = sigma L^-T (A w); mhalf = -c/2
    // grad_trsm: 0 products with the explicit L^-1; 1 rocBLAS trsm / trsv; 2 (default) the products followed by ONE step of iterative
    // refinement against L itself (x += L^-1 (b - L x): two more M^3 GEMMs per solve, ~50 us each at M = 1024, against 270 us for a
    // rocBLAS trsm with an M x M right-hand side) - the forward error of a product with an explicit inverse is ~ eps cond(L) |L^-1||b|,
    // one refinement step brings it to that of a backward-stable solve, ~ eps cond(L) |x| (tools/zgrad_owner.py, tools/grad_trsm_ab.py)
    const bool use_inv = c->grad_trsm != 1;
    const bool refine = c->grad_trsm == 2;
    const T minus_one = -1;
    if (use_inv) {
        CGLB_TRY(ensure_Linv<T>(c));
        if (!c->Mtmp3) CGLB_TRY(dalloc(c, &c->Mtmp3, (size_t)M * M * c->esz));
        if (refine && !c->Mtmp4) CGLB_TRY(dalloc(c, &c->Mtmp4, (size_t)M * M * c->esz));
        CGLB_TRY(launch_tri_rowdot(c, c->Linv, aw, 0, c->w_t2));  // (L^-T x)_i = column i of L^-1 (contiguous) . x
        if (refine) {
            CGLB_TRY(launch_tri_rowdot(c, c->Lc, c->w_t2, 0, c->w_q));       // L^T x: column i of L (contiguous) . x
            CGLB_TRY(launch_residual(c, c->w_t, aw, c->w_q, M));              // aw - L^T x
            CGLB_TRY(launch_tri_rowdot(c, c->Linv, c->w_t, 0, c->w_q));
            CGLB_TRY(launch_axpy(c, c->w_t2, 1.0, c->w_q, M));
        }
    } else {
        HIP_CHECK(c, hipMemcpyAsync(c->w_t2, aw, (size_t)M * c->esz, hipMemcpyDeviceToDevice, c->stream));
        BLAS_CHECK(c, xtrsv(c->blas, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit, M, (const T*)c->Lc, M, (T*)c->w_t2, 1));
    }
    hipLaunchKernelGGL((scale2_kernel<T>), dim3(grid1d(M)), dim3(256), 0, c->stream, (const T*)c->w_t2, (int64_t)M, (T)sigma, (T*)c->w_t2,
                       (T)(-0.5 * sigma), (T*)c->w_t);
    // Guf = (1/sigma) L^-T (I/tau - B^-1) A   (+ c w^T applied on the fly)
    hipLaunchKernelGGL((mat_combine_kernel<T>), dim3(grid1d_full((int64_t)M * M)), dim3(256), 0, c->stream, (T*)c->Mtmp2, M, (T)(1.0 / tau), (T)-1,
                       (const T*)c->Mtmp, (T)0, (const T*)nullptr);
    const T inv_sigma = (T)(1.0 / sigma);
    const T* Tuf = (const T*)c->Mtmp2;  // (1/sigma) L^-T (I/tau - B^-1)
    if (use_inv) {
        BLAS_CHECK(c, xgemm(c->blas, rocblas_operation_transpose, rocblas_operation_none, M, M, M, &inv_sigma, (const T*)c->Linv, M,
                            (const T*)c->Mtmp2, M, &zero, (T*)c->Mtmp3, M));
        if (refine) {  // R = S - sigma L^T X ; X += (1/sigma) L^-T R
            const T msigma = (T)(-sigma);
            HIP_CHECK(c, hipMemcpyAsync(c->Mtmp4, c->Mtmp2, (size_t)M * M * c->esz, hipMemcpyDeviceToDevice, c->stream));
            BLAS_CHECK(c, xgemm(c->blas, rocblas_operation_transpose, rocblas_operation_none, M, M, M, &msigma, (const T*)c->Lc, M,
                                (const T*)c->Mtmp3, M, &one, (T*)c->Mtmp4, M));
            BLAS_CHECK(c, xgemm(c->blas, rocblas_operation_transpose, rocblas_operation_none, M, M, M, &inv_sigma, (const T*)c->Linv, M,
                                (const T*)c->Mtmp4, M, &one, (T*)c->Mtmp3, M));
        }
        Tuf = (const T*)c->Mtmp3;
    } else {
        BLAS_CHECK(c, xtrsm(c->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit, M, M, &inv_sigma,
                            (const T*)c->Lc, M, (T*)c->Mtmp2, M));
    }
    if (c->nloc > 0) {
        BLAS_CHECK(c, xgemm(c->blas, rocblas_operation_none, rocblas_operation_transpose, (int)c->nloc, M, M, &one, (const T*)c->At, (int)c->lda,
                            Tuf, M, &zero, (T*)c->Guf, (int)c->lda));
        CGLB_TRY(launch_grad_kuf(c, c->w_t2, c->w_z, out));
        if (!u_full_cyclic) {
            // u = w + v/2 ; N^2 bilinear pass over the local rows
            const T* v_loc = (const T*)v_full + c->r0;
            hipLaunchKernelGGL((axpby_kernel<T>), dim3(grid1d(c->nloc)), dim3(256), 0, c->stream, (T*)c->w_Ap, (T)1, (const T*)c->w_z, (T)0.5, v_loc,
                               c->nloc);
            CGLB_TRY(launch_grad_kff(c, v_full, c->w_Ap, c->scal + S_TMP2));
            hipLaunchKernelGGL((axpby_kernel<double>), dim3(1), dim3(64), 0, c->stream, out, 1.0, (const double*)out, 1.0,
                               (const double*)(c->scal + S_TMP2), (int64_t)D);
        }
    }
    if (u_full_cyclic) {  // this rank's cyclic share of the global symmetric N^2 form (u gathered by the caller)
        CGLB_TRY(launch_grad_kff_cyclic(c, v_full, u_full_cyclic, c->scal + S_TMP2));
        hipLaunchKernelGGL((axpby_kernel<double>), dim3(1), dim3(64), 0, c->stream, out, 1.0, (const double*)out, 1.0,
                           (const double*)(c->scal + S_TMP2), (int64_t)D);
    }
    if (c->r0 == 0) {
        // Guu = L^-T [ (I - B^-1)/2 - (AA^T)/(2 tau) ] L^-1  - c c^T/2
        hipLaunchKernelGGL((mat_combine_kernel<T>), dim3(grid1d_full((int64_t)M * M)), dim3(256), 0, c->stream, (T*)c->Mtmp, M, (T)0.5, (T)-0.5,
                           (const T*)c->Mtmp, (T)(-0.5 / tau), (const T*)c->AAt);
        if (use_inv) {  // L^-T S L^-1 as two GEMMs (each followed by its refinement step)
            const size_t mm = (size_t)M * M * c->esz;
            BLAS_CHECK(c, xgemm(c->blas, rocblas_operation_transpose, rocblas_operation_none, M, M, M, &one, (const T*)c->Linv, M,
                                (const T*)c->Mtmp, M, &zero, (T*)c->Mtmp3, M));
            if (refine) {  // X = L^-T S:  R = S - L^T X ; X += L^-T R
                HIP_CHECK(c, hipMemcpyAsync(c->Mtmp4, c->Mtmp, mm, hipMemcpyDeviceToDevice, c->stream));
                BLAS_CHECK(c, xgemm(c->blas, rocblas_operation_transpose, rocblas_operation_none, M, M, M, &minus_one, (const T*)c->Lc, M,
                                    (const T*)c->Mtmp3, M, &one, (T*)c->Mtmp4, M));
                BLAS_CHECK(c, xgemm(c->blas, rocblas_operation_transpose, rocblas_operation_none, M, M, M, &one, (const T*)c->Linv, M,
                                    (const T*)c->Mtmp4, M, &one, (T*)c->Mtmp3, M));
            }
            BLAS_CHECK(c, xgemm(c->blas, rocblas_operation_none, rocblas_operation_none, M, M, M, &one, (const T*)c->Mtmp3, M,
                                (const T*)c->Linv, M, &zero, (T*)c->Mtmp, M));
            if (refine) {  // Y = X L^-1:  R = X - Y L ; Y += R L^-1
                HIP_CHECK(c, hipMemcpyAsync(c->Mtmp4, c->Mtmp3, mm, hipMemcpyDeviceToDevice, c->stream));
                BLAS_CHECK(c, xgemm(c->blas, rocblas_operation_none, rocblas_operation_none, M, M, M, &minus_one, (const T*)c->Mtmp, M,
                                    (const T*)c->Lc, M, &one, (T*)c->Mtmp4, M));
                BLAS_CHECK(c, xgemm(c->blas, rocblas_operation_none, rocblas_operation_none, M, M, M, &one, (const T*)c->Mtmp4, M,
                                    (const T*)c->Linv, M, &one, (T*)c->Mtmp, M));
            }
        } else {
            BLAS_CHECK(c, xtrsm(c->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit, M, M, &one,
                                (const T*)c->Lc, M, (T*)c->Mtmp, M));
            BLAS_CHECK(c, xtrsm(c->blas, rocblas_side_right, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, M, M, &one,
                                (const T*)c->Lc, M, (T*)c->Mtmp, M));
        }
        CGLB_TRY(launch_grad_kuu(c, c->Mtmp, c->w_t2, c->w_t, out));
        hipLaunchKernelGGL(grad_scalar_terms_kernel, dim3(1), dim3(64), 0, c->stream, out, D, sc, (const double*)(c->scal + S_TRBINV), (double)c->N,
                           (double)M, f, s, tau, c->trace_AAt);
    }
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

double logdet_value(const cglb_ctx* c) {
    const double N = (double)c->N;
    return -c->sum_log_diag_LB - 0.5 * N * std::log(c->noise) - 0.5 * N * std::log(tau_of(c));  // models.py:236-243
}

int obj_finish(cglb_ctx* c, const double* sc_dev, double* out4) {
    double sc[8];
    rocblas_int info_inv = 0;
    if (c->Linv_unchecked) HIP_CHECK(c, hipMemcpyAsync(&info_inv, c->info_dev + 2, sizeof(info_inv), hipMemcpyDeviceToHost, c->stream));
    CGLB_TRY(read_scalars(c, sc_dev, sc, 8));
    if (c->Linv_unchecked) {
        c->Linv_unchecked = false;
        if (info_inv != 0) { c->have_Linv = false; return cglb_fail(c, CGLB_ERR_NOT_PD, "inverse of L (gradient algebra) failed: zero pivot " + std::to_string(info_inv)); }
    }
    const double N = (double)c->N;
    const double lower = sc[0], upper = sc[0] + 0.5 * sc[1];           // models.py:283-284
    const double logdet = logdet_value(c);
    const double cst = -0.5 * N * std::log(2.0 * M_PI);                  // models.py:162-163
    out4[0] = -upper + logdet + cst;                                     // models.py:286, :169
    out4[1] = lower;
    out4[2] = upper;
    out4[3] = logdet;
    return CGLB_OK;
}


// ================================ N ranks inside the library (include/cglb_hip.h, "N ranks inside the library") ================================
// Same scheme as cglb_amd/distributed.py: SymShardedCGLB (the host-driven twin that the gloo tests exercise with CPU local ops): the global
// upper triangle of K_ff dealt to the ranks by cyclic row blocks, replicated full-length p, Ap, v, r, b, the Nystrom panel column-sharded over
// contiguous rows, three collectives per PCG iteration - issued here on the context stream.
// ---- phase timing of an evaluation (option "eval_profile") -------------------------------------------------------------------
int eval_mark(cglb_ctx* c) {
    if (!c->eval_profile) return CGLB_OK;
    if (c->eval_events_used >= c->eval_events.size()) {
        hipEvent_t ev;
        HIP_CHECK(c, hipEventCreate(&ev));
        c->eval_events.push_back(ev);
    }
    HIP_CHECK(c, hipEventRecord(c->eval_events[c->eval_events_used++], c->stream));
    return CGLB_OK;
}
int eval_collect(cglb_ctx* c) {
    for (size_t q = 0; q + 5 <= c->eval_events_used; q += 5) {
        HIP_CHECK(c, hipEventSynchronize(c->eval_events[q + 4]));
        for (int k = 0; k < 4; ++k) {
            float ms = 0.f;
            HIP_CHECK(c, hipEventElapsedTime(&ms, c->eval_events[q + k], c->eval_events[q + k + 1]));
            c->eval_ms[k] += ms;
        }
        c->eval_count += 1;
    }
    c->eval_events_used = 0;
    return CGLB_OK;
}

inline ncclDataType_t nccl_type(const cglb_ctx* c, bool as_double) { return (as_double || c->dtype == CGLB_F64) ? ncclDouble : ncclFloat; }

int require_comm(cglb_ctx* c) {
    if (!c->comm) return cglb_fail(c, CGLB_ERR_STATE, "no communicator: call cglb_comm_init_rccl or cglb_comm_init_callbacks first");
    if (c->precond_mode != 0) return cglb_fail(c, CGLB_ERR_STATE, "the N-rank path needs the stored-panel preconditioner (precond_mode 0)");
    return CGLB_OK;
}

int comm_fail(cglb_ctx* c, const char* what, int code) { return cglb_fail(c, CGLB_ERR_COMM, std::string(what) + " failed with code " + std::to_string(code)); }

// in-place sum over ranks of `count` elements (vector element type, or double when as_double)
int comm_allreduce(cglb_ctx* c, void* buf, int64_t count, bool as_double = false) {
    cglb_comm_state* m = c->comm;
    m->n_allreduce++;
    if (m->kind == 1) {
        const ncclResult_t r = ncclAllReduce(buf, buf, (size_t)count, nccl_type(c, as_double), ncclSum, (ncclComm_t)m->nccl, c->stream);
        if (r != ncclSuccess) return cglb_fail(c, CGLB_ERR_COMM, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
        return CGLB_OK;
    }
    const int rc = m->ar(m->user, buf, count, as_double ? CGLB_F64 : c->dtype, (void*)c->stream);
    return rc == 0 ? CGLB_OK : comm_fail(c, "all-reduce callback", rc);
}

// in-place all-gather: rank g's `count` elements sit at element g * count of buf
int comm_allgather(cglb_ctx* c, void* buf, int64_t count) {
    cglb_comm_state* m = c->comm;
    m->n_allgather++;
    if (m->kind == 1) {
        const char* mine = (const char*)buf + (size_t)m->rank * (size_t)count * c->esz;
        const ncclResult_t r = ncclAllGather(mine, buf, (size_t)count, nccl_type(c, false), (ncclComm_t)m->nccl, c->stream);
        if (r != ncclSuccess) return cglb_fail(c, CGLB_ERR_COMM, std::string("ncclAllGather: ") + ncclGetErrorString(r));
        return CGLB_OK;
    }
    const int rc = m->ag(m->user, buf, count, c->dtype, (void*)c->stream);
    return rc == 0 ? CGLB_OK : comm_fail(c, "all-gather callback", rc);
}

void comm_free(cglb_ctx* c) {
    cglb_comm_state* m = c->comm;
    if (!m) return;
    if (c->stream) (void)hipStreamSynchronize(c->stream); else (void)hipDeviceSynchronize();
    if (m->kind == 1 && m->nccl) (void)ncclCommDestroy((ncclComm_t)m->nccl);
    void* ptrs[] = {m->p, m->r, m->Ap, m->Kv, m->b, m->zseg, m->ubuf, m->u, m->aw, m->sc, m->grad, m->gat};
    for (void* q : ptrs) if (q) (void)hipFree(q);
    delete m;
    c->comm = nullptr;
}

int comm_alloc(cglb_ctx* c, int world, int rank) {
    if (world < 1 || rank < 0 || rank >= world) return cglb_fail(c, CGLB_ERR_BAD_ARG, "bad world/rank");
    const int64_t per = (c->N + world - 1) / world;
    const int64_t r0 = std::min<int64_t>((int64_t)rank * per, c->N), r1 = std::min<int64_t>((int64_t)(rank + 1) * per, c->N);
    if (c->r0 != r0 || c->r1 != r1)
        return cglb_fail(c, CGLB_ERR_BAD_ARG, "the context owns rows [" + std::to_string(c->r0) + "," + std::to_string(c->r1) + ") but rank " + std::to_string(rank) + " of " +
                                                  std::to_string(world) + " must own [" + std::to_string(r0) + "," + std::to_string(r1) + ")");
    comm_free(c);
    cglb_comm_state* m = new (std::nothrow) cglb_comm_state();
    if (!m) return cglb_fail(c, CGLB_ERR_BAD_ARG, "out of host memory");
    c->comm = m;
    m->world = world; m->rank = rank; m->per = per;
    const size_t e = c->esz, N = (size_t)c->N, M = (size_t)c->M;
    void** vecs[] = {&m->p, &m->r, &m->Ap, &m->Kv, &m->b};
    for (void** q : vecs) CGLB_TRY(dalloc(c, q, N * e));
    CGLB_TRY(dalloc(c, &m->zseg, (size_t)world * (size_t)(per + 1) * e));
    CGLB_TRY(dalloc(c, &m->ubuf, (size_t)world * (size_t)per * e));
    CGLB_TRY(dalloc(c, &m->u, M * e));
    CGLB_TRY(dalloc(c, &m->aw, M * e));
    CGLB_TRY(dalloc(c, (void**)&m->sc, 8 * sizeof(double)));
    CGLB_TRY(dalloc(c, (void**)&m->grad, (size_t)CGLB_GRAD_LEN(c->D, c->M) * sizeof(double)));
    HIP_CHECK(c, hipMemsetAsync(m->zseg, 0, (size_t)world * (size_t)(per + 1) * e, c->stream));
    HIP_CHECK(c, hipMemsetAsync(m->ubuf, 0, (size_t)world * (size_t)per * e, c->stream));
    c->par_world = world;  // cglb_set_parallel: the cyclic deal of the symmetric K_ff work
    c->par_rank = rank;
    return CGLB_OK;
}

int dist_setup(cglb_ctx* c) {
    CGLB_DISPATCH_T(c->dtype, CGLB_TRY(setup_local_impl<T>(c)));
    CGLB_TRY(comm_allreduce(c, c->AAt, (int64_t)c->M * c->M));
    CGLB_DISPATCH_T(c->dtype, CGLB_TRY(setup_finish_impl<T>(c)));
    return CGLB_OK;
}

// out = (K_ff + noise I) x on every rank (rank 0's partial carries the noise term)
int dist_matvec(cglb_ctx* c, const void* x_full, void* out_full) {
    CGLB_TRY(launch_kff_sym_cyclic(c, x_full, out_full));
    return comm_allreduce(c, out_full, c->N);
}

// z = P r (conjugate_gradient.py:73) gathered in segments of per + 1 elements: slice g = rank g's rows and its partial of r^T z
int dist_precond_gather(cglb_ctx* c, const void* r_full) {
    cglb_comm_state* m = c->comm;
    const char* r_loc = (const char*)r_full + (size_t)c->r0 * c->esz;
    char* slot = (char*)m->zseg + (size_t)m->rank * (size_t)(m->per + 1) * c->esz;
    if (c->nloc > 0) CGLB_TRY(launch_gemv_u(c, r_loc, m->u));
    else HIP_CHECK(c, hipMemsetAsync(m->u, 0, (size_t)c->M * c->esz, c->stream));
    CGLB_TRY(comm_allreduce(c, m->u, c->M));
    if (c->nloc > 0) {
        CGLB_TRY(launch_tri_apply(c, m->u, c->w_t));
        CGLB_TRY(launch_precond_z(c, r_loc, c->w_t, slot, nullptr, slot + (size_t)m->per * c->esz));
    } else {
        HIP_CHECK(c, hipMemsetAsync(slot + (size_t)m->per * c->esz, 0, c->esz, c->stream));
    }
    return comm_allgather(c, m->zseg, m->per + 1);
}

// ... then rz_new = r^T z from the gathered partials (rank order) and p = z + p rz_new / rz_old, or p = z (:75)
int dist_precond_direction(cglb_ctx* c, double* rz_new, const double* rz_old, int restart) {
    cglb_comm_state* m = c->comm;
    CGLB_TRY(dist_precond_gather(c, m->r));
    return launch_update_p_seg(c, m->p, m->zseg, c->N, m->per, m->world, rz_new, rz_old, restart);
}

// conjugate_gradient.py:41-86 on replicated full vectors - the N-rank twin of pcg_impl (same look-ahead, same stop rule)
int dist_pcg_impl(cglb_ctx* c, const void* b, void* v, double max_error, int max_iter, int restart_iter, int* steps, double* half_rz) {
    cglb_comm_state* m = c->comm;
    double* S = c->scal;
    const int64_t N = c->N;
    c->pwh_src = nullptr;
    double vnorm = 0.0;
    CGLB_TRY(launch_dot(c, v, v, N, S + S_TMP));
    CGLB_TRY(read_scalars(c, S + S_TMP, &vnorm, 1));
    if (vnorm == 0.0) {  // cold start: A v == 0 and r == b exactly (v is replicated: every rank takes the same branch)
        HIP_CHECK(c, hipMemcpyAsync(m->r, b, (size_t)N * c->esz, hipMemcpyDeviceToDevice, c->stream));
    } else {
        CGLB_TRY(dist_matvec(c, v, m->Kv));                                                        // :57
        CGLB_TRY(launch_residual(c, m->r, b, m->Kv, N));                                           // :58
    }
    double *s_rz = S + S_RZ, *s_nrz = S + S_NRZ;
    CGLB_TRY(dist_precond_direction(c, s_rz, s_rz, 1));                                            // :59, :61
    double rz = 0;
    CGLB_TRY(read_scalars(c, s_rz, &rz, 1));
    if (!std::isfinite(rz)) return cglb_fail(c, CGLB_ERR_COMM, "r^T P r is not finite at the start of the solve");  // gathered numbers: same on every rank
    int i = 0;
    bool ahead = false;
    while (0.5 * rz > max_error && i < max_iter) {                                                 // :65
        if (!ahead) CGLB_TRY(dist_matvec(c, m->p, m->Ap));                                         // :66
        CGLB_TRY(launch_dot(c, m->p, m->Ap, N, S + S_PAP));                                        // :67
        const int restart = (restart_iter > 0) && (i % restart_iter == restart_iter - 1);          // :70
        CGLB_TRY(launch_update_v_r(c, v, m->r, m->p, m->Ap, s_rz, S + S_PAP, !restart, N));        // :68, :72
        if (restart) {
            CGLB_TRY(dist_matvec(c, v, m->Kv));
            CGLB_TRY(launch_residual(c, m->r, b, m->Kv, N));
        }
        CGLB_TRY(dist_precond_direction(c, s_nrz, s_rz, restart));                                 // :73, :75
        std::swap(s_rz, s_nrz);                                                                    // :76
        HIP_CHECK(c, hipMemcpyAsync(c->host_scal, s_rz, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(c, hipEventRecord(c->scal_event, c->stream));
        ahead = c->pcg_lookahead && (i + 1 < max_iter) && (0.5 * rz > lookahead_factor(c) * max_error);
        if (ahead) CGLB_TRY(dist_matvec(c, m->p, m->Ap));                                          // kernel + all-reduce enqueued before the host waits
        HIP_CHECK(c, hipEventSynchronize(c->scal_event));
        rz = c->host_scal[0];  // a function of all-gathered numbers only: every rank reads the same value and leaves the loop together
        if (!std::isfinite(rz)) return cglb_fail(c, CGLB_ERR_COMM, "r^T P r is not finite after iteration " + std::to_string(i));
        ++i;
    }
    if (steps) *steps = i;
    if (half_rz) *half_rz = 0.5 * rz;
    return CGLB_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void unpack_pairs_kernel(const T* __restrict__ gat, int64_t pern, int64_t n, T* __restrict__ a, T* __restrict__ b) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t g = i / pern, k = i - g * pern;
    a[i] = gat[g * 2 * pern + k];
    b[i] = gat[g * 2 * pern + pern + k];
}

}  // namespace

// =================================================== C ABI ====================================================
extern "C" {

int cglb_version(void) { return 100; }

const char* cglb_last_error(const cglb_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int cglb_ctx_create(cglb_ctx** out, int64_t n_total, int64_t row_begin, int64_t row_end, int d, int m, int dtype, int kernel_kind, int device,
                    void* stream) {
    g_create_error.clear();
    if (!out) { g_create_error = "out is NULL"; return CGLB_ERR_BAD_ARG; }
    *out = nullptr;
    if (d > CGLB_MAX_D) {
        g_create_error = "input dimension D = " + std::to_string(d) + " exceeds the supported maximum of " + std::to_string(CGLB_MAX_D);
        return CGLB_ERR_BAD_ARG;
    }
    if (n_total <= 0 || row_begin < 0 || row_end < row_begin || row_end > n_total || d <= 0 || m <= 0 ||
        (dtype != CGLB_F64 && dtype != CGLB_F32) || (kernel_kind != CGLB_RBF && kernel_kind != CGLB_MATERN32)) {
        g_create_error = "bad argument to cglb_ctx_create";
        return CGLB_ERR_BAD_ARG;
    }
    if (n_total > 2000000000LL || (int64_t)m > 65536) { g_create_error = "problem too large for 32-bit BLAS dimensions"; return CGLB_ERR_BAD_ARG; }
    cglb_ctx* c = new (std::nothrow) cglb_ctx();
    if (!c) { g_create_error = "out of host memory"; return CGLB_ERR_BAD_ARG; }
    c->N = n_total; c->r0 = row_begin; c->r1 = row_end; c->nloc = row_end - row_begin;
    c->lda = (c->nloc + 7) & ~(int64_t)7; if (c->lda == 0) c->lda = 8;
    c->D = d; c->Dp = pad_dim(d); c->M = m; c->dtype = dtype; c->kind = kernel_kind; c->device = device;
    c->Dh = mid_dim(d, dtype);
    c->esz = dtype == CGLB_F64 ? 8 : 4; c->stream = (hipStream_t)stream;
    auto fail = [&](int rc) { g_create_error = c->err; cglb_ctx_destroy(c); return rc; };
#define CR(expr) do { int _rc = (expr); if (_rc != CGLB_OK) return fail(_rc); } while (0)
    { hipError_t e = hipSetDevice(device); if (e != hipSuccess) { c->err = std::string("hipSetDevice: ") + hipGetErrorString(e); return fail(CGLB_ERR_HIP); } }
    { rocblas_status s = rocblas_create_handle(&c->blas); if (s != rocblas_status_success) { c->err = "rocblas_create_handle failed"; c->blas = nullptr; return fail(CGLB_ERR_BLAS); } }
    { rocblas_status s = rocblas_set_stream(c->blas, c->stream); if (s != rocblas_status_success) { c->err = "rocblas_set_stream failed"; return fail(CGLB_ERR_BLAS); } }
    const size_t e = c->esz, N = (size_t)c->N, nl = (size_t)c->nloc, M = (size_t)m, Dp = (size_t)c->Dp;
    CR(dalloc(c, &c->X, N * d * e)); CR(dalloc(c, &c->y, N * e)); CR(dalloc(c, &c->Z, M * d * e));
    CR(dalloc(c, &c->Xs, N * Dp * e)); CR(dalloc(c, &c->xa, N * e)); CR(dalloc(c, &c->Zs, M * Dp * e)); CR(dalloc(c, &c->za, M * e));
    CR(dalloc(c, &c->Zh, M * Dp * e)); CR(dalloc(c, &c->zah, M * e)); CR(dalloc(c, &c->Linv, M * M * e)); CR(dalloc(c, &c->LinvT, M * M * e));
    CR(dalloc(c, &c->w_q, M * e));
    const size_t hotN = c->Dp > CGLB_MAX_D_NARROW ? 0 : N;  // the hot operand set exists for the register-resident pair kernels only
    CR(dalloc(c, &c->Xh, c->Dh > 0 ? N * (size_t)c->Dh * e : hotN * Dp * e)); CR(dalloc(c, &c->Xhsq, hotN * Dp * e)); CR(dalloc(c, &c->xah, N * e)); CR(dalloc(c, &c->wh, N * e)); CR(dalloc(c, &c->pwh, N * e)); CR(dalloc(c, (void**)&c->exp_tab, CGLB_TAB_SIZE * sizeof(double)));
    {   // 2^x table of the pair kernels: 2^((k + 1/2)/T) for the floor/fract range reduction (devmath.h)
        std::vector<double> tab(CGLB_TAB_SIZE);
        for (int k = 0; k < CGLB_TAB_SIZE; ++k) {
            const double v = std::exp2(((double)k + 0.5) / (double)CGLB_TAB_SIZE);  // glibc exp2: < 1 ulp
            uint64_t bits;
            std::memcpy(&bits, &v, 8);
            bits -= (uint64_t)k << (52 - CGLB_TAB_BITS);  // pre-compensated for the one-add scaling (devmath.h exp2_tab_scale)
            std::memcpy(&tab[k], &bits, 8);
        }
        hipError_t e3 = hipMemcpy(c->exp_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e3 != hipSuccess) { c->err = "exp table upload failed"; return fail(CGLB_ERR_HIP); }
    }
    CR(dalloc(c, &c->At, M * (size_t)c->lda * e));
    CR(dalloc(c, &c->Lc, M * M * e)); CR(dalloc(c, &c->LBc, M * M * e)); CR(dalloc(c, &c->LBinv, M * M * e)); CR(dalloc(c, &c->LBinvT, M * M * e));
    CR(dalloc(c, &c->AAt, M * M * e)); CR(dalloc(c, &c->Mtmp, M * M * e)); CR(dalloc(c, &c->Mtmp2, M * M * e));
    CR(dalloc(c, (void**)&c->info_dev, 4 * sizeof(rocblas_int)));
    CR(dalloc(c, &c->w_r, nl * e)); CR(dalloc(c, &c->w_z, nl * e)); CR(dalloc(c, &c->w_p, nl * e)); CR(dalloc(c, &c->w_Ap, nl * e));
    CR(dalloc(c, &c->w_Kv, nl * e)); CR(dalloc(c, &c->w_e, nl * e)); CR(dalloc(c, &c->w_pfull, N * e));
    CR(dalloc(c, &c->w_u, M * e)); CR(dalloc(c, &c->w_t, M * e)); CR(dalloc(c, &c->w_t2, M * e));
    CR(dalloc(c, &c->tpart, ((M + 63) / 64) * nl * e));
    CR(dalloc(c, (void**)&c->dotpart, DOTPART_CAP * sizeof(double)));
    CR(dalloc(c, (void**)&c->scal, 64 * sizeof(double)));
    CR(dalloc(c, (void**)&c->gradbuf, (size_t)CGLB_GRAD_LEN(d, m) * sizeof(double)));
    { hipError_t e2 = hipMemsetAsync(c->scal, 0, 64 * sizeof(double), c->stream); if (e2 != hipSuccess) { c->err = "memset failed"; return fail(CGLB_ERR_HIP); } }
    {
        hipError_t e4 = hipHostMalloc((void**)&c->host_scal, 8 * sizeof(double), hipHostMallocDefault);
        if (e4 == hipSuccess) e4 = hipEventCreateWithFlags(&c->scal_event, hipEventDisableTiming);
        if (e4 != hipSuccess) { c->err = std::string("pinned scalar buffer: ") + hipGetErrorString(e4); return fail(CGLB_ERR_HIP); }
    }
#undef CR
    *out = c;
    return CGLB_OK;
}

int cglb_ctx_destroy(cglb_ctx* c) {
    if (!c) return CGLB_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream); else (void)hipDeviceSynchronize();
    comm_free(c);
    wide_free(c);
    void* ptrs[] = {c->X, c->y, c->Z, c->Xs, c->xa, c->Zs, c->za, c->Xh, c->Xhsq, c->xah, c->wh, c->pwh, c->exp_tab, c->At, c->Lc, c->LBc, c->LBinv, c->LBinvT, c->AAt, c->Mtmp, c->Mtmp2, c->Mtmp3, c->Guf,
                    c->info_dev, c->w_r, c->w_z, c->w_p, c->w_Ap, c->w_Kv, c->w_e, c->w_pfull, c->w_u, c->w_t, c->w_t2, c->kpart, c->tpart,
                    c->dotpart, c->scal, c->gpart, c->gradbuf, c->slabs, c->fragA, c->fragB, c->sym_items, c->Zh, c->zah, c->Linv, c->LinvT, c->w_q, c->ppart, c->chol_blk, c->uwh, c->Mtmp4};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (hipEvent_t ev : c->k1_events) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : c->eval_events) (void)hipEventDestroy(ev);
    if (c->host_scal) (void)hipHostFree(c->host_scal);
    if (c->scal_event) (void)hipEventDestroy(c->scal_event);
    if (c->blas) (void)rocblas_destroy_handle(c->blas);
    delete c;
    return CGLB_OK;
}

int cglb_set_option(cglb_ctx* c, const char* name, int64_t value) {
    if (!c || !name) return CGLB_ERR_BAD_ARG;
    if (!strcmp(name, "kff_variant")) c->kff_variant = (int)value;
    else if (!strcmp(name, "kff_jsplit")) c->kff_jsplit = (int)value;
    else if (!strcmp(name, "kff_rows")) c->kff_rows = (int)value;
    else if (!strcmp(name, "sym_chunk")) c->sym_chunk_opt = value;
    else if (!strcmp(name, "pcg_lookahead")) c->pcg_lookahead = (int)value;
    else if (!strcmp(name, "sym_order")) c->sym_order = (int)value;
    else if (!strcmp(name, "aat_block")) c->aat_block = (int)value;
    else if (!strcmp(name, "grad_gram")) c->grad_gram = (int)value;
    else if (!strcmp(name, "k1_profile")) {  // 1: start timing every launch of the symmetric pair kernel (counters reset), 0: stop
        if (value) { c->k1_events_used = 0; c->k1_ms_total = 0.0; c->k1_launches = 0; }
        else CGLB_TRY(k1_profile_collect(c));
        c->k1_profile = value != 0;
    }
    else if (!strcmp(name, "eval_profile")) {  // 1: time the phases of every cglb_objective_and_grad from now on (counters reset), 0: stop
        if (value) { c->eval_events_used = 0; c->eval_count = 0; for (double& t : c->eval_ms) t = 0.0; }
        else CGLB_TRY(eval_collect(c));
        c->eval_profile = value != 0;
    }
    else if (!strcmp(name, "precision")) {
        if (value != CGLB_PREC_EXACT && value != CGLB_PREC_FAST && value != CGLB_PREC_LOW)
            return cglb_fail(c, CGLB_ERR_BAD_ARG, "precision must be 0 (exact), 1 (fast, default) or 2 (low)");
        c->precision = (int)value;
    }
    else if (!strcmp(name, "drop_weighted_operand")) c->pwh_src = nullptr;  // the vector last written by cglb_vec_update_p_seg is about to change
    else if (!strcmp(name, "final_matvec")) c->final_matvec = (int)value;
    else if (!strcmp(name, "wide_grad_sym")) c->wide_grad_sym = (int)value;
    else if (!strcmp(name, "wide_reg")) { c->wide_reg = (int)value; c->pwh_src = nullptr; }
    else if (!strcmp(name, "chol_mode")) c->chol_mode = (int)value;
    else if (!strcmp(name, "grad_trsm")) c->grad_trsm = (int)value;
    else if (!strcmp(name, "precond_mode")) { c->precond_mode = (int)value; c->have_local = c->have_terms = false; }
    else return cglb_fail(c, CGLB_ERR_BAD_ARG, std::string("unknown option ") + name);
    return CGLB_OK;
}

int cglb_set_data(cglb_ctx* c, const void* X, const void* y) {
    if (c) c->obj_valid = false;
    if (!c || !X || !y) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    HIP_CHECK(c, hipSetDevice(c->device));
    const size_t nx = (size_t)c->N * c->D;
    HIP_CHECK(c, hipMemcpyAsync(c->X, X, nx * c->esz, hipMemcpyDefault, c->stream));
    HIP_CHECK(c, hipMemcpyAsync(c->y, y, (size_t)c->N * c->esz, hipMemcpyDefault, c->stream));
    // column means (centre of the Gram form; any centre is exact in exact arithmetic)
    std::vector<char> host(nx * c->esz);
    HIP_CHECK(c, hipMemcpyAsync(host.data(), c->X, nx * c->esz, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(c, hipStreamSynchronize(c->stream));
    for (int d = 0; d < c->D; ++d) c->xmean[d] = 0.0;
    for (int64_t i = 0; i < c->N; ++i)
        for (int d = 0; d < c->D; ++d)
            c->xmean[d] += c->dtype == CGLB_F64 ? ((const double*)host.data())[i * c->D + d] : (double)((const float*)host.data())[i * c->D + d];
    for (int d = 0; d < c->D; ++d) c->xmean[d] /= (double)c->N;
    for (int d = 0; d < c->D; ++d) c->xrange[d] = 0.0;
    c->xradius2 = 0.0;
    for (int64_t i = 0; i < c->N; ++i) {
        double r2 = 0.0;
        for (int d = 0; d < c->D; ++d) {
            const double x = c->dtype == CGLB_F64 ? ((const double*)host.data())[i * c->D + d] : (double)((const float*)host.data())[i * c->D + d];
            c->xrange[d] = std::fmax(c->xrange[d], std::fabs(x - c->xmean[d]));
            r2 += (x - c->xmean[d]) * (x - c->xmean[d]);
        }
        c->xradius2 = std::fmax(c->xradius2, r2);
    }
    c->have_data = true;
    c->have_local = c->have_terms = false;
    return CGLB_OK;
}

int cglb_set_hypers(cglb_ctx* c, const double* lengthscales, double variance, double noise, double mean, const void* Z, double jitter) {
    if (c) c->obj_valid = false;
    if (!c || !lengthscales || !Z) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    if (!c->have_data) return cglb_fail(c, CGLB_ERR_STATE, "set_data must precede set_hypers");
    if (!(variance > 0) || !(noise > 0) || !(jitter >= 0) || !std::isfinite(mean)) return cglb_fail(c, CGLB_ERR_BAD_ARG, "variance/noise must be positive, jitter >= 0");
    for (int d = 0; d < c->D; ++d)
        if (!(lengthscales[d] > 0) || !std::isfinite(lengthscales[d])) return cglb_fail(c, CGLB_ERR_BAD_ARG, "lengthscales must be positive");
    HIP_CHECK(c, hipSetDevice(c->device));
    for (int d = 0; d < c->D; ++d) c->ls[d] = lengthscales[d];
    c->var = variance; c->noise = noise; c->mean = mean; c->jitter = jitter;
    HIP_CHECK(c, hipMemcpyAsync(c->Z, Z, (size_t)c->M * c->D * c->esz, hipMemcpyDefault, c->stream));
    c->have_hypers = true;
    c->pwh_src = nullptr;  // the column weights change with the hypers
    {   // |a_i + a_j + xs_i.xs_j| <= 2 max|xs|^2 (scaled units: octaves for RBF, octaves^2 for Matern).  The unclamped 2^x of
        // the hot loops needs the exponent inside [-1000, 0] octaves (exp2_tab_scale) and its hot-unit integer below 2^30.
        const double ks = (c->kind == CGLB_RBF) ? std::sqrt(CGLB_LOG2E) : CGLB_SQRT3 * CGLB_LOG2E;
        double s2 = 0.0;
        for (int d = 0; d < c->D; ++d) { const double v = c->xrange[d] * ks / c->ls[d]; s2 += v * v; }
        const double oct = (c->kind == CGLB_RBF) ? 2.0 * s2 : 2.0 * std::sqrt(s2);
        const bool int_ok = 2.0 * s2 * CGLB_HOT_UNITS * (c->kind == CGLB_RBF ? 1.0 : CGLB_HOT_UNITS) < 1.0e9;
        // (RBF: the symmetric kernel's unweighted factor 2^(a_i + x_i.x_j) spans [-1.5 s2, 0.5 s2] octaves; fp32 exp2 range is +-126)
        const double oct_max = (c->dtype == CGLB_F32 && c->kind == CGLB_RBF) ? 200.0 : 0.95 * CGLB_EXP_FLOOR_OCT;
        c->exp_clamp = !(int_ok && oct < oct_max);
        // Matern-3/2: the unclamped fast-level kernels keep d2 = a_i + a_j - 2 x_i.x_j positive by a bias in the row seeds instead of a
        // clamp per pair (devmath.h).  amax = max_i a_i in hot units^2 <= hot_scale^2 s2; beyond the admissible bias the clamped variant runs.
        c->m32_bias = 0.0;
        if (c->kind == CGLB_MATERN32) {
            const double hs = cglb_hot_scale(c);
            double lmin = c->ls[0];
            for (int d = 1; d < c->D; ++d) lmin = std::fmin(lmin, c->ls[d]);
            const double amax = std::fmin(s2, ks * ks * c->xradius2 / (lmin * lmin)) * hs * hs;  // max_i a_i: box bound or ball bound, whichever is tighter
            c->m32_bias = std::fmax(CGLB_M32_BIAS_FACTOR * (5.0 * c->D + 3.0) * amax, 1.0e-200);  // > 0 even when every point sits on the mean
            if (c->m32_bias > CGLB_M32_BIAS_MAX) c->exp_clamp = true;
        }
    }
    if (is_wide(c)) {  // D > 32: one scaled operand set (+ its squares), Gram products through rocBLAS (kernels_wide.hip)
        CGLB_TRY(wide_after_hypers(c));
        if (c->Dh > 0) {   // mid width: the symmetric mat-vec stays register-resident (kernels_kff_sym.hip) and needs its hot operands
            CGLB_TRY(wide_prep_hot(c));
            CGLB_TRY(launch_hot_weights(c));
        }
    } else {
        CGLB_TRY(launch_prep_scaled(c, c->X, c->N, c->Xs, c->xa));
        CGLB_TRY(launch_prep_scaled(c, c->X, c->N, c->Xh, c->xah, true));
        CGLB_TRY(launch_hot_weights(c));
        CGLB_TRY(launch_hot_squares(c));
        CGLB_TRY(launch_prep_scaled(c, c->Z, c->M, c->Zs, c->za));
        CGLB_TRY(launch_prep_scaled(c, c->Z, c->M, c->Zh, c->zah, true));
    }
    c->frag_valid = false;  // the pre-permuted operands of the experimental matrix-pipe variant are rebuilt on its first use
    c->have_local = c->have_terms = false;
    return CGLB_OK;
}

int cglb_shard_setup_local(cglb_ctx* c) {
    if (!c) return CGLB_ERR_BAD_ARG;
    HIP_CHECK(c, hipSetDevice(c->device));
    CGLB_DISPATCH_T(c->dtype, return setup_local_impl<T>(c));
}
void* cglb_aat_buffer(cglb_ctx* c) { return c ? c->AAt : nullptr; }
int cglb_shard_setup_finish(cglb_ctx* c) {
    if (!c) return CGLB_ERR_BAD_ARG;
    HIP_CHECK(c, hipSetDevice(c->device));
    CGLB_DISPATCH_T(c->dtype, return setup_finish_impl<T>(c));
}
int cglb_setup(cglb_ctx* c) {
    if (!c) return CGLB_ERR_BAD_ARG;
    CGLB_TRY(require_single(c));
    CGLB_TRY(cglb_shard_setup_local(c));
    return cglb_shard_setup_finish(c);
}

int cglb_logdet(cglb_ctx* c, double* logdet) {
    if (!c || !logdet) return CGLB_ERR_BAD_ARG;
    CGLB_TRY(require_terms(c));
    *logdet = logdet_value(c);
    return CGLB_OK;
}

int cglb_matvec(cglb_ctx* c, const void* p_full, void* out_local) {
    if (!c || !p_full || !out_local) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    if (!c->have_hypers) return cglb_fail(c, CGLB_ERR_STATE, "set_hypers must precede matvec");
    HIP_CHECK(c, hipSetDevice(c->device));
    return launch_kff_matvec(c, p_full, out_local, nullptr);
}

int cglb_matvec_dot(cglb_ctx* c, const void* p_full, void* out_local, void* pdot) {
    if (!c || !p_full || !out_local || !pdot) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    if (!c->have_hypers) return cglb_fail(c, CGLB_ERR_STATE, "set_hypers must precede matvec");
    HIP_CHECK(c, hipSetDevice(c->device));
    return launch_kff_matvec(c, p_full, out_local, (double*)pdot);
}

int cglb_shard_rhs(cglb_ctx* c, void* out_local) {
    if (!c || !out_local) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    if (!c->have_hypers) return cglb_fail(c, CGLB_ERR_STATE, "set_hypers must precede rhs");
    HIP_CHECK(c, hipSetDevice(c->device));
    return launch_sub_scalar(c, out_local, (const char*)c->y + (size_t)c->r0 * c->esz, c->mean, c->nloc);
}

int cglb_rhs_full(cglb_ctx* c, void* out_full) {
    if (!c || !out_full) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    if (!c->have_hypers) return cglb_fail(c, CGLB_ERR_STATE, "set_hypers must precede rhs");
    HIP_CHECK(c, hipSetDevice(c->device));
    return launch_sub_scalar(c, out_full, c->y, c->mean, c->N);
}

int cglb_cross_matvec(cglb_ctx* c, const void* xnew, int64_t n_new, const void* v_full, void* out) {
    if (!c || !xnew || !v_full || !out || n_new < 0) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "bad argument") : CGLB_ERR_BAD_ARG;
    if (!c->have_hypers) return cglb_fail(c, CGLB_ERR_STATE, "set_hypers must precede cross_matvec");
    if (n_new == 0) return CGLB_OK;
    HIP_CHECK(c, hipSetDevice(c->device));
    void *xr = nullptr, *xs = nullptr, *xa = nullptr;
    DevTemps tmp;  // freed on every path below
    CGLB_TRY(tmp.alloc(c, &xr, (size_t)n_new * c->D * c->esz));
    CGLB_TRY(tmp.alloc(c, &xs, (size_t)n_new * c->Dp * c->esz));
    CGLB_TRY(tmp.alloc(c, &xa, (size_t)n_new * c->esz));
    int rc = CGLB_OK;
    hipError_t e = hipMemcpyAsync(xr, xnew, (size_t)n_new * c->D * c->esz, hipMemcpyDefault, c->stream);
    if (e != hipSuccess) rc = cglb_fail(c, CGLB_ERR_HIP, "copy of xnew failed");
    if (rc == CGLB_OK) rc = launch_prep_scaled(c, xr, n_new, xs, xa, true);  // rows of the pair kernel: hot units
    if (rc == CGLB_OK) rc = launch_cross_matvec(c, xs, xa, n_new, v_full, out);
    (void)hipStreamSynchronize(c->stream);  // the temporaries are released when `tmp` goes out of scope
    return rc;
}

int cglb_precond_apply(cglb_ctx* c, const void* r, void* z, double* rz) {
    if (c) c->obj_valid = false;
    if (!c || !r || !z) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    CGLB_TRY(require_single(c));
    CGLB_TRY(require_terms(c));
    HIP_CHECK(c, hipSetDevice(c->device));
    CGLB_TRY(precond_single(c, r, z, c->scal + S_TMP));
    if (rz) CGLB_TRY(read_scalars(c, c->scal + S_TMP, rz, 1));
    return CGLB_OK;
}

int cglb_shard_precond_u(cglb_ctx* c, const void* r_local, void* u_partial) {
    // a rank whose row shard is empty (nloc == 0: more ranks than row blocks) passes an empty vector, whose pointer may be NULL:
    // its partial u is zero
    if (!c || (!r_local && c->nloc > 0) || !u_partial) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    CGLB_TRY(require_terms(c));
    HIP_CHECK(c, hipSetDevice(c->device));
    if (c->nloc == 0) { HIP_CHECK(c, hipMemsetAsync(u_partial, 0, (size_t)c->M * c->esz, c->stream)); return CGLB_OK; }
    return precond_u_any(c, r_local, u_partial);
}
int cglb_shard_precond_z(cglb_ctx* c, const void* r_local, const void* u, void* z_local, void* rz_partial) {
    if (c) c->obj_valid = false;
    if (!c || !r_local || !u || !z_local) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;  // rz_partial may be NULL
    CGLB_TRY(require_terms(c));
    HIP_CHECK(c, hipSetDevice(c->device));
    CGLB_TRY(launch_tri_apply(c, u, c->w_t));
    return precond_z_any(c, r_local, c->w_t, z_local, (double*)rz_partial);
}

int cglb_shard_precond_z_seg(cglb_ctx* c, const void* r_local, const void* u, void* z_slot, int64_t per) {
    if (c) c->obj_valid = false;
    if (!c || (!r_local && c->nloc > 0) || !u || !z_slot || per < 0) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "bad argument") : CGLB_ERR_BAD_ARG;
    if (per < c->nloc) return cglb_fail(c, CGLB_ERR_BAD_ARG, "precond_z_seg: slice shorter than the local rows");
    if (c->precond_mode != 0) return cglb_fail(c, CGLB_ERR_STATE, "precond_z_seg needs the stored-panel preconditioner (precond_mode 0)");
    CGLB_TRY(require_terms(c));
    HIP_CHECK(c, hipSetDevice(c->device));
    if (c->nloc == 0) {  // empty shard: nothing to precondition, the partial of r^T z is zero
        HIP_CHECK(c, hipMemsetAsync((char*)z_slot + (size_t)per * c->esz, 0, c->esz, c->stream));
        return CGLB_OK;
    }
    CGLB_TRY(launch_tri_apply(c, u, c->w_t));
    return launch_precond_z(c, r_local, c->w_t, z_slot, nullptr, (char*)z_slot + (size_t)per * c->esz);
}

int cglb_shard_dot(cglb_ctx* c, const void* a_local, const void* b_local, void* out) {
    if (!c || !a_local || !b_local || !out) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    HIP_CHECK(c, hipSetDevice(c->device));
    return launch_dot(c, a_local, b_local, c->nloc, (double*)out);
}
int cglb_shard_update_v_r(cglb_ctx* c, void* v_local, void* r_local, const void* p_local, const void* Ap_local, const void* rz, const void* pAp,
                          int update_r) {
    if (!c || !v_local || !r_local || !p_local || !Ap_local || !rz || !pAp) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    HIP_CHECK(c, hipSetDevice(c->device));
    return launch_update_v_r(c, v_local, r_local, p_local, Ap_local, (const double*)rz, (const double*)pAp, update_r);
}
int cglb_shard_residual(cglb_ctx* c, void* r_local, const void* b_local, const void* Kv_local) {
    if (!c || !r_local || !b_local || !Kv_local) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    HIP_CHECK(c, hipSetDevice(c->device));
    return launch_residual(c, r_local, b_local, Kv_local);
}
int cglb_shard_update_p(cglb_ctx* c, void* p_local, const void* z_local, const void* new_rz, const void* rz, int restart) {
    if (!c || !p_local || !z_local || !new_rz || !rz) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    HIP_CHECK(c, hipSetDevice(c->device));
    return launch_update_p(c, p_local, z_local, (const double*)new_rz, (const double*)rz, restart);
}

// ---- cyclic-symmetric multi-GPU path (include/cglb_hip.h) ---------------------------------------------------------
int cglb_set_parallel(cglb_ctx* c, int world, int rank) {
    if (!c || world < 1 || rank < 0 || rank >= world) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "bad world/rank") : CGLB_ERR_BAD_ARG;
    c->par_world = world;
    c->par_rank = rank;
    return CGLB_OK;
}
int cglb_matvec_cyclic(cglb_ctx* c, const void* p_full, void* out_full_partial) {
    if (!c || !p_full || !out_full_partial) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    if (!c->have_hypers) return cglb_fail(c, CGLB_ERR_STATE, "set_hypers must precede matvec");
    HIP_CHECK(c, hipSetDevice(c->device));
    return launch_kff_sym_cyclic(c, p_full, out_full_partial);
}
int cglb_vec_dot(cglb_ctx* c, int64_t n, const void* a, const void* b, void* out) {
    if (!c || !a || !b || !out || n < 0) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "bad argument") : CGLB_ERR_BAD_ARG;
    HIP_CHECK(c, hipSetDevice(c->device));
    return launch_dot(c, a, b, n, (double*)out);
}
int cglb_vec_update_v_r(cglb_ctx* c, int64_t n, void* v, void* r, const void* p, const void* Ap, const void* rz, const void* pAp, int update_r) {
    if (!c || !v || !r || !p || !Ap || !rz || !pAp || n < 0) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "bad argument") : CGLB_ERR_BAD_ARG;
    HIP_CHECK(c, hipSetDevice(c->device));
    return launch_update_v_r(c, v, r, p, Ap, (const double*)rz, (const double*)pAp, update_r, n);
}
int cglb_vec_residual(cglb_ctx* c, int64_t n, void* r, const void* b, const void* Kv) {
    if (!c || !r || !b || !Kv || n < 0) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "bad argument") : CGLB_ERR_BAD_ARG;
    HIP_CHECK(c, hipSetDevice(c->device));
    return launch_residual(c, r, b, Kv, n);
}
int cglb_vec_update_p(cglb_ctx* c, int64_t n, void* p, const void* z, const void* new_rz, const void* rz, int restart) {
    if (!c || !p || !z || !new_rz || !rz || n < 0) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "bad argument") : CGLB_ERR_BAD_ARG;
    HIP_CHECK(c, hipSetDevice(c->device));
    return launch_update_p(c, p, z, (const double*)new_rz, (const double*)rz, restart, n);
}
int cglb_vec_update_p_seg(cglb_ctx* c, int64_t n, int64_t per, int world, void* p, const void* zseg, void* new_rz, const void* rz, int restart) {
    if (!c || !p || !zseg || !new_rz || !rz || n <= 0 || per <= 0 || world <= 0 || (int64_t)world * per < n)
        return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "bad argument") : CGLB_ERR_BAD_ARG;
    HIP_CHECK(c, hipSetDevice(c->device));
    return launch_update_p_seg(c, p, zseg, n, per, world, (double*)new_rz, (const double*)rz, restart);
}
int cglb_vec_axpy(cglb_ctx* c, int64_t n, double alpha, const void* x, void* y) {
    if (!c || !x || !y || n < 0) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "bad argument") : CGLB_ERR_BAD_ARG;
    HIP_CHECK(c, hipSetDevice(c->device));
    return launch_axpy(c, y, alpha, x, n);
}
int cglb_shard_obj_phase1_kv(cglb_ctx* c, const void* Kv_local, void* u_partial) {
    if (c) c->obj_valid = false;
    if (!c || !Kv_local || !u_partial) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    CGLB_TRY(require_terms(c));
    HIP_CHECK(c, hipSetDevice(c->device));
    const char* y_loc = (const char*)c->y + (size_t)c->r0 * c->esz;
    CGLB_TRY(launch_sub_scalar(c, c->w_e, y_loc, c->mean, c->nloc));
    HIP_CHECK(c, hipMemcpyAsync(c->w_Kv, Kv_local, (size_t)c->nloc * c->esz, hipMemcpyDeviceToDevice, c->stream));
    CGLB_TRY(launch_residual(c, c->w_r, c->w_e, c->w_Kv));
    return precond_u_any(c, c->w_r, u_partial);
}
int cglb_shard_obj_w(cglb_ctx* c, void* w_local_out) {
    if (!c || !w_local_out) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    HIP_CHECK(c, hipSetDevice(c->device));
    HIP_CHECK(c, hipMemcpyAsync(w_local_out, c->w_z, (size_t)c->nloc * c->esz, hipMemcpyDeviceToDevice, c->stream));
    return CGLB_OK;
}
int cglb_shard_obj_phase3_cyclic(cglb_ctx* c, const void* v_full, const void* u_full, const void* sc, const void* aw, void* grad_partial) {
    if (!c || !v_full || !u_full || !sc || !aw || !grad_partial) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    CGLB_TRY(require_terms(c));
    HIP_CHECK(c, hipSetDevice(c->device));
    CGLB_DISPATCH_T(c->dtype, return obj_phase3_impl<T>(c, v_full, (const double*)sc, aw, (double*)grad_partial, u_full));
}

int cglb_pcg_solve(cglb_ctx* c, const void* b, void* v_inout, double max_error, int max_cg_iter, int restart_cg_iter, int* steps, double* half_rz) {
    if (c) c->obj_valid = false;
    if (!c || !b || !v_inout) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    CGLB_TRY(require_single(c));
    CGLB_TRY(require_terms(c));
    HIP_CHECK(c, hipSetDevice(c->device));
    return pcg_impl(c, b, v_inout, max_error, max_cg_iter, restart_cg_iter, steps, half_rz);
}

int cglb_shard_obj_phase1(cglb_ctx* c, const void* v_full, void* u_partial) {
    if (c) c->obj_valid = false;
    if (!c || !v_full || !u_partial) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    CGLB_TRY(require_terms(c));
    HIP_CHECK(c, hipSetDevice(c->device));
    return obj_phase1(c, v_full, u_partial);
}
int cglb_shard_obj_phase2(cglb_ctx* c, const void* v_full, const void* u, void* sc_partial, void* aw_partial) {
    if (!c || !v_full || !u || !sc_partial || !aw_partial) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    CGLB_TRY(require_terms(c));
    HIP_CHECK(c, hipSetDevice(c->device));
    return obj_phase2(c, v_full, u, (double*)sc_partial, aw_partial);
}
int cglb_shard_obj_phase3(cglb_ctx* c, const void* v_full, const void* sc, const void* aw, void* grad_partial) {
    if (!c || !v_full || !sc || !aw || !grad_partial) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    CGLB_TRY(require_terms(c));
    HIP_CHECK(c, hipSetDevice(c->device));
    CGLB_DISPATCH_T(c->dtype, return obj_phase3_impl<T>(c, v_full, (const double*)sc, aw, (double*)grad_partial));
}
int cglb_shard_obj_finish(cglb_ctx* c, const void* sc, double* out4) {
    if (!c || !sc || !out4) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    CGLB_TRY(require_terms(c));
    HIP_CHECK(c, hipSetDevice(c->device));
    return obj_finish(c, (const double*)sc, out4);
}

int cglb_objective_and_grad(cglb_ctx* c, void* v_inout, int run_cg, double max_error, int max_cg_iter, int restart_cg_iter, double* out4,
                            double* grad, int* steps, double* half_rz) {
    if (!c || !v_inout || !out4) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    c->obj_valid = false;
    CGLB_TRY(require_single(c));
    if (c->eval_profile && c->eval_events_used + 5 > 5 * 512) CGLB_TRY(eval_collect(c));  // bounded pool
    CGLB_TRY(eval_mark(c));
    CGLB_TRY(cglb_setup(c));                                                        // models.py:155
    CGLB_TRY(eval_mark(c));
    if (steps) *steps = 0;
    if (half_rz) *half_rz = std::nan("");
    if (run_cg) {                                                                   // models.py:262-278
        CGLB_TRY(launch_sub_scalar(c, c->w_e, c->y, c->mean, c->nloc));
        // w_e is reused by phase1, which recomputes it; pcg reads it as b
        CGLB_TRY(pcg_impl(c, c->w_e, v_inout, max_error, max_cg_iter, restart_cg_iter, steps, half_rz));
    }
    CGLB_TRY(eval_mark(c));
    double* sc = c->scal + S_SC;
    if (run_cg && !c->final_matvec) CGLB_TRY(obj_phase1_reuse(c, c->w_u));
    else CGLB_TRY(obj_phase1(c, v_inout, c->w_u));
    CGLB_TRY(obj_phase2(c, v_inout, c->w_u, sc, c->w_u));  // aw overwrites u after u has been consumed (stream order)
    CGLB_TRY(eval_mark(c));
    if (grad) {
        CGLB_DISPATCH_T(c->dtype, CGLB_TRY(obj_phase3_impl<T>(c, v_inout, sc, c->w_u, c->gradbuf)));
        HIP_CHECK(c, hipMemcpyAsync(grad, c->gradbuf, (size_t)CGLB_GRAD_LEN(c->D, c->M) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    CGLB_TRY(eval_mark(c));
    CGLB_TRY(obj_finish(c, sc, out4));
    c->obj_valid = true;  // r = e - K v and w = P r of this evaluation stay in the work vectors (cglb_objective_grad_v)
    return CGLB_OK;
}

int cglb_objective_grad_v(cglb_ctx* c, void* gv_out) {
    if (!c || !gv_out) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    CGLB_TRY(require_single(c));
    if (!c->obj_valid) return cglb_fail(c, CGLB_ERR_STATE, "cglb_objective_grad_v must directly follow cglb_objective_and_grad");
    HIP_CHECK(c, hipSetDevice(c->device));
    // bound = -upper + ..., upper = v^T e - v^T K v / 2 + r^T P r / 2, r = e - K v (K incl. the noise term)  =>
    // d bound / d v = -(e - K v) + K P r = K w - r
    CGLB_TRY(launch_kff_matvec(c, c->w_z, c->w_Ap, nullptr));
    return launch_residual(c, gv_out, c->w_Ap, c->w_r);
}

// ---- inducing-point initialisation (config.py:55-65; kernels_select.hip) -------------------------------------------
int cglb_select_inducing(cglb_ctx* c, const double* lengthscales, double variance, double jitter, int64_t* indices_out, void* Z_out,
                         double* trace_out) {
    if (!c || !lengthscales || !indices_out) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    if (!c->have_data) return cglb_fail(c, CGLB_ERR_STATE, "set_data must precede select_inducing");
    if (!(variance > 0) || !(jitter >= 0)) return cglb_fail(c, CGLB_ERR_BAD_ARG, "variance must be positive, jitter >= 0");
    for (int d = 0; d < c->D; ++d)
        if (!(lengthscales[d] > 0) || !std::isfinite(lengthscales[d])) return cglb_fail(c, CGLB_ERR_BAD_ARG, "lengthscales must be positive");
    HIP_CHECK(c, hipSetDevice(c->device));
    // the scaled operand buffers of the context serve as scratch: whatever set_hypers had put there is invalidated
    double saved[CGLB_MAX_D];
    for (int d = 0; d < c->D; ++d) { saved[d] = c->ls[d]; c->ls[d] = lengthscales[d]; }
    const int rc_prep = launch_prep_scaled(c, c->X, c->N, c->Xs, c->xa);
    for (int d = 0; d < c->D; ++d) c->ls[d] = saved[d];
    c->have_hypers = c->have_local = c->have_terms = false;
    if (rc_prep != CGLB_OK) return rc_prep;
    const int M = (int)(c->M < c->N ? c->M : c->N);
    long long* chosen_dev = nullptr;
    double* trace_dev = nullptr;
    DevTemps tmp;
    CGLB_TRY(tmp.alloc(c, (void**)&chosen_dev, (size_t)M * sizeof(long long)));
    CGLB_TRY(tmp.alloc(c, (void**)&trace_dev, sizeof(double)));
    int rc = launch_select_inducing(c, variance, jitter, chosen_dev, Z_out, trace_dev);
    if (rc == CGLB_OK) {
        std::vector<long long> host(M);
        hipError_t e = hipMemcpy(host.data(), chosen_dev, (size_t)M * sizeof(long long), hipMemcpyDeviceToHost);
        double tr = 0.0;
        if (e == hipSuccess) e = hipMemcpy(&tr, trace_dev, sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = cglb_fail(c, CGLB_ERR_HIP, hipGetErrorString(e));
        for (int m = 0; m < M; ++m) indices_out[m] = (int64_t)host[m];
        if (trace_out) *trace_out = tr;
    }
    return rc;
}

// res = (y - mean) - Kv over the local rows, u_partial = A_loc res  (models.py:318, :335, :340)
static int predict_u_local(cglb_ctx* c, const void* Kv_local, void* u_partial) {
    const char* y_loc = (const char*)c->y + (size_t)c->r0 * c->esz;
    CGLB_TRY(launch_sub_scalar(c, c->w_e, y_loc, c->mean, c->nloc));
    CGLB_TRY(launch_residual(c, c->w_r, c->w_e, Kv_local));
    if (c->nloc == 0) { HIP_CHECK(c, hipMemsetAsync(u_partial, 0, (size_t)c->M * c->esz, c->stream)); return CGLB_OK; }
    return launch_gemv_u(c, c->w_r, u_partial);
}

// c = LB^-1 u / sigma (:343) into w_u, then for the n_new points of xnew: cg_mean (:334), K_us panel, tmp1, tmp2 (:344-345), mean / var (:347-351)
static int predict_rows(cglb_ctx* c, const void* v_full, const void* u, const void* xnew, int64_t n_new, void* f_mean, void* f_var) {
    const int M = c->M;
    const int64_t ld = (n_new + 7) & ~(int64_t)7;
    void *xr = nullptr, *xs = nullptr, *xa = nullptr, *t1 = nullptr, *t2 = nullptr;
    DevTemps tmp;  // freed on every path below
    CGLB_TRY(tmp.alloc(c, &xr, (size_t)n_new * c->D * c->esz));
    CGLB_TRY(tmp.alloc(c, &xs, (size_t)n_new * c->Dp * c->esz));
    CGLB_TRY(tmp.alloc(c, &xa, (size_t)n_new * c->esz));
    CGLB_TRY(tmp.alloc(c, &t1, (size_t)M * ld * c->esz));
    CGLB_TRY(tmp.alloc(c, &t2, (size_t)M * ld * c->esz));
    auto body = [&]() -> int {
        HIP_CHECK(c, hipMemcpyAsync(xr, xnew, (size_t)n_new * c->D * c->esz, hipMemcpyDefault, c->stream));
        CGLB_TRY(launch_prep_scaled(c, xr, n_new, xs, xa, true));               // hot units for the pair kernel
        CGLB_TRY(launch_cross_matvec(c, xs, xa, n_new, v_full, f_mean));       // cg_mean = ksf @ v   (models.py:334)
        CGLB_TRY(launch_prep_scaled(c, xr, n_new, xs, xa, false));             // plain scaled units for the K_us panel
        if (u != c->w_u) HIP_CHECK(c, hipMemcpyAsync(c->w_u, u, (size_t)M * c->esz, hipMemcpyDeviceToDevice, c->stream));
        CGLB_DISPATCH_T(c->dtype, {
            const T one = 1;
            const T inv_sigma = (T)(1.0 / std::sqrt(c->noise));
            // c = LB^-1 a_res / sigma (:343)
            BLAS_CHECK(c, xtrsv(c->blas, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, M, (const T*)c->LBc, M, (T*)c->w_u, 1));
            hipLaunchKernelGGL((scale2_kernel<T>), dim3(grid1d(M)), dim3(256), 0, c->stream, (const T*)c->w_u, (int64_t)M, inv_sigma, (T*)c->w_u, (T)0, (T*)nullptr);
            // tmp1 = L^-1 Kus (:344), tmp2 = LB^-1 tmp1 (:345); panels stored [M][ld] row-major == (ld x M) column-major
            if (is_wide(c)) {
                CGLB_TRY(wide_kus(c, xs, xa, n_new, ld, t1));
            } else {
                dim3 grid((unsigned)((n_new + 255) / 256), (unsigned)((M + 31) / 32));
                CGLB_DISPATCH_KIND(c->kind, CGLB_DISPATCH_DP(c->Dp, hipLaunchKernelGGL((kus_kernel<T, KIND, DP>), grid, dim3(256), 0, c->stream, (const T*)c->Zs,
                                                                                        (const T*)xs, n_new, ld, M, (T)c->var, (T*)t1)));
            }
            BLAS_CHECK(c, xtrsm(c->blas, rocblas_side_right, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit, (int)n_new, M, &one,
                                (const T*)c->Lc, M, (T*)t1, (int)ld));
            HIP_CHECK(c, hipMemcpyAsync(t2, t1, (size_t)M * ld * c->esz, hipMemcpyDeviceToDevice, c->stream));
            BLAS_CHECK(c, xtrsm(c->blas, rocblas_side_right, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit, (int)n_new, M, &one,
                                (const T*)c->LBc, M, (T*)t2, (int)ld));
            hipLaunchKernelGGL((predict_finish_kernel<T>), dim3((unsigned)((n_new + 255) / 256)), dim3(256), 0, c->stream, (const T*)t1, (const T*)t2, ld, M,
                               (const T*)c->w_u, n_new, (T)c->mean, (T)c->var, (T*)f_mean, (T*)f_var);
        });
        CGLB_LAUNCH_CHECK(c);
        return CGLB_OK;
    };
    const int rc = body();
    (void)hipStreamSynchronize(c->stream);  // before `tmp` releases the buffers the kernels use
    return rc;
}

int cglb_predict(cglb_ctx* c, const void* v_full, const void* xnew, int64_t n_new, void* f_mean, void* f_var) {
    if (c) c->obj_valid = false;
    if (!c || !v_full || !xnew || !f_mean || !f_var || n_new < 0) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "bad argument") : CGLB_ERR_BAD_ARG;
    CGLB_TRY(require_single(c));
    CGLB_TRY(require_terms(c));
    if (n_new == 0) return CGLB_OK;
    HIP_CHECK(c, hipSetDevice(c->device));
    CGLB_TRY(launch_kff_matvec(c, v_full, c->w_Kv, nullptr));               // cov @ v            (:335)
    CGLB_TRY(predict_u_local(c, c->w_Kv, c->w_u));                          // a_res = A @ res    (:340)
    return predict_rows(c, v_full, c->w_u, xnew, n_new, f_mean, f_var);
}

int cglb_shard_predict_u(cglb_ctx* c, const void* Kv_local, void* u_partial) {
    if (c) c->obj_valid = false;
    if (!c || (!Kv_local && c->nloc > 0) || !u_partial) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    CGLB_TRY(require_terms(c));
    HIP_CHECK(c, hipSetDevice(c->device));
    return predict_u_local(c, Kv_local, u_partial);
}

int cglb_shard_predict_rows(cglb_ctx* c, const void* v_full, const void* u, const void* xnew, int64_t n_new, void* f_mean, void* f_var) {
    if (c) c->obj_valid = false;
    if (!c || !v_full || !u || n_new < 0 || (n_new > 0 && (!xnew || !f_mean || !f_var))) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "bad argument") : CGLB_ERR_BAD_ARG;
    CGLB_TRY(require_terms(c));
    if (n_new == 0) return CGLB_OK;
    HIP_CHECK(c, hipSetDevice(c->device));
    return predict_rows(c, v_full, u, xnew, n_new, f_mean, f_var);
}


// ---- N ranks inside the library --------------------------------------------------------------------------------------
int cglb_comm_get_unique_id(void* id_out) {
    if (!id_out) return CGLB_ERR_BAD_ARG;
    static_assert(sizeof(ncclUniqueId) == CGLB_COMM_ID_BYTES, "RCCL unique id size");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return CGLB_ERR_COMM;
    std::memcpy(id_out, &id, sizeof(id));
    return CGLB_OK;
}

int cglb_comm_init_rccl(cglb_ctx* c, const void* unique_id, int world, int rank) {
    if (!c || !unique_id) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    HIP_CHECK(c, hipSetDevice(c->device));
    CGLB_TRY(comm_alloc(c, world, rank));
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = ncclCommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) {
        comm_free(c);
        return cglb_fail(c, CGLB_ERR_COMM, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
    }
    c->comm->kind = 1;
    c->comm->nccl = (void*)comm;
    return CGLB_OK;
}

int cglb_comm_init_callbacks(cglb_ctx* c, int world, int rank, cglb_allreduce_fn allreduce, cglb_allgather_fn allgather, void* user) {
    if (!c || !allreduce || !allgather) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    HIP_CHECK(c, hipSetDevice(c->device));
    CGLB_TRY(comm_alloc(c, world, rank));
    c->comm->kind = 2;
    c->comm->ar = allreduce; c->comm->ag = allgather; c->comm->user = user;
    return CGLB_OK;
}

int cglb_comm_destroy(cglb_ctx* c) {
    if (!c) return CGLB_ERR_BAD_ARG;
    (void)hipSetDevice(c->device);
    comm_free(c);
    c->par_world = 1; c->par_rank = 0;
    return CGLB_OK;
}

int cglb_dist_setup(cglb_ctx* c) {
    if (!c) return CGLB_ERR_BAD_ARG;
    CGLB_TRY(require_comm(c));
    HIP_CHECK(c, hipSetDevice(c->device));
    return dist_setup(c);
}

int cglb_dist_matvec(cglb_ctx* c, const void* x_full, void* out_full) {
    if (!c || !x_full || !out_full) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    CGLB_TRY(require_comm(c));
    if (!c->have_hypers) return cglb_fail(c, CGLB_ERR_STATE, "set_hypers must precede matvec");
    HIP_CHECK(c, hipSetDevice(c->device));
    return dist_matvec(c, x_full, out_full);
}

int cglb_dist_precond_apply(cglb_ctx* c, const void* r_full, void* z_full, double* rz) {
    if (c) c->obj_valid = false;
    if (!c || !r_full || !z_full) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    CGLB_TRY(require_comm(c));
    CGLB_TRY(require_terms(c));
    HIP_CHECK(c, hipSetDevice(c->device));
    CGLB_TRY(dist_precond_gather(c, r_full));
    // p = z through the segmented update (restart form): un-segments the gathered z and sums the partials of r^T z in rank order
    CGLB_TRY(launch_update_p_seg(c, z_full, c->comm->zseg, c->N, c->comm->per, c->comm->world, c->scal + S_TMP, c->scal + S_TMP, 1));
    c->pwh_src = nullptr;  // z_full is the caller's vector, not a direction the next mat-vec will consume
    if (rz) CGLB_TRY(read_scalars(c, c->scal + S_TMP, rz, 1));
    return CGLB_OK;
}

int cglb_dist_pcg_solve(cglb_ctx* c, const void* b_full, void* v_full_inout, double max_error, int max_cg_iter, int restart_cg_iter, int* steps,
                        double* half_rz) {
    if (c) c->obj_valid = false;
    if (!c || !b_full || !v_full_inout) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    CGLB_TRY(require_comm(c));
    CGLB_TRY(require_terms(c));
    HIP_CHECK(c, hipSetDevice(c->device));
    return dist_pcg_impl(c, b_full, v_full_inout, max_error, max_cg_iter, restart_cg_iter, steps, half_rz);
}

int cglb_dist_objective_and_grad(cglb_ctx* c, void* v, int run_cg, double max_error, int max_cg_iter, int restart_cg_iter, double* out4,
                                 double* grad, int* steps, double* half_rz) {
    if (!c || !v || !out4) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "NULL argument") : CGLB_ERR_BAD_ARG;
    c->obj_valid = false;
    CGLB_TRY(require_comm(c));
    HIP_CHECK(c, hipSetDevice(c->device));
    cglb_comm_state* m = c->comm;
    CGLB_TRY(dist_setup(c));                                                        // models.py:155
    if (steps) *steps = 0;
    if (half_rz) *half_rz = std::nan("");
    if (run_cg) {                                                                   // models.py:262-278
        CGLB_TRY(launch_sub_scalar(c, m->b, c->y, c->mean, c->N));
        CGLB_TRY(dist_pcg_impl(c, m->b, v, max_error, max_cg_iter, restart_cg_iter, steps, half_rz));
    }
    if (run_cg && !c->final_matvec) CGLB_TRY(launch_residual(c, m->Kv, m->b, m->r, c->N));  // K v = e - r of the recurrence (option "final_matvec")
    else CGLB_TRY(dist_matvec(c, v, m->Kv));                                        // models.py:280
    // phase 1 with K v given: r = e - K v on the local rows, u_partial = A_loc r
    const char* y_loc = (const char*)c->y + (size_t)c->r0 * c->esz;
    CGLB_TRY(launch_sub_scalar(c, c->w_e, y_loc, c->mean, c->nloc));
    HIP_CHECK(c, hipMemcpyAsync(c->w_Kv, (const char*)m->Kv + (size_t)c->r0 * c->esz, (size_t)c->nloc * c->esz, hipMemcpyDeviceToDevice, c->stream));
    CGLB_TRY(launch_residual(c, c->w_r, c->w_e, c->w_Kv));
    if (c->nloc > 0) CGLB_TRY(launch_gemv_u(c, c->w_r, m->u));
    else HIP_CHECK(c, hipMemsetAsync(m->u, 0, (size_t)c->M * c->esz, c->stream));
    CGLB_TRY(comm_allreduce(c, m->u, c->M));
    CGLB_TRY(obj_phase2(c, v, m->u, m->sc, m->aw));
    CGLB_TRY(comm_allreduce(c, m->sc, 8, true));
    if (grad) {
        CGLB_TRY(comm_allreduce(c, m->aw, c->M));
        // u = w + v/2 on the local rows, gathered: the cyclic share of the N^2 gradient form needs it in full
        char* u_loc = (char*)m->ubuf + (size_t)m->rank * (size_t)m->per * c->esz;
        HIP_CHECK(c, hipMemcpyAsync(u_loc, c->w_z, (size_t)c->nloc * c->esz, hipMemcpyDeviceToDevice, c->stream));
        CGLB_TRY(launch_axpy(c, u_loc, 0.5, (const char*)v + (size_t)c->r0 * c->esz, c->nloc));
        CGLB_TRY(comm_allgather(c, m->ubuf, m->per));
        CGLB_DISPATCH_T(c->dtype, CGLB_TRY(obj_phase3_impl<T>(c, v, m->sc, m->aw, m->grad, m->ubuf)));
        CGLB_TRY(comm_allreduce(c, m->grad, (int64_t)CGLB_GRAD_LEN(c->D, c->M), true));
        HIP_CHECK(c, hipMemcpyAsync(grad, m->grad, (size_t)CGLB_GRAD_LEN(c->D, c->M) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    return obj_finish(c, m->sc, out4);
}

int cglb_dist_predict(cglb_ctx* c, const void* v_full, const void* xnew, int64_t n_new, void* f_mean, void* f_var) {
    if (c) c->obj_valid = false;
    if (!c || !v_full || !xnew || !f_mean || !f_var || n_new < 0) return c ? cglb_fail(c, CGLB_ERR_BAD_ARG, "bad argument") : CGLB_ERR_BAD_ARG;
    CGLB_TRY(require_comm(c));
    CGLB_TRY(require_terms(c));
    if (n_new == 0) return CGLB_OK;
    HIP_CHECK(c, hipSetDevice(c->device));
    cglb_comm_state* m = c->comm;
    CGLB_TRY(dist_matvec(c, v_full, m->Kv));                                                            // cov @ v  (models.py:335)
    CGLB_TRY(predict_u_local(c, (const char*)m->Kv + (size_t)c->r0 * c->esz, m->u));                    // a_res partial (:340)
    CGLB_TRY(comm_allreduce(c, m->u, c->M));
    // the new points are dealt to the ranks in contiguous slices of pern rows; slice g lands at [g * 2 pern, ...): mean then variance
    const int64_t pern = (n_new + m->world - 1) / m->world;
    const int64_t a = std::min<int64_t>((int64_t)m->rank * pern, n_new), b = std::min<int64_t>((int64_t)(m->rank + 1) * pern, n_new);
    const size_t need = (size_t)m->world * 2 * (size_t)pern * c->esz;
    if (need > m->gat_cap) {
        if (m->gat) HIP_CHECK(c, hipFree(m->gat));
        m->gat = nullptr; m->gat_cap = 0;
        HIP_CHECK(c, hipMalloc(&m->gat, need));
        m->gat_cap = need;
    }
    char* mine = (char*)m->gat + (size_t)m->rank * 2 * (size_t)pern * c->esz;
    if (b > a) {
        // xnew may be host or device memory: element offsets are the same either way
        CGLB_TRY(predict_rows(c, v_full, m->u, (const char*)xnew + (size_t)a * c->D * c->esz, b - a, mine, mine + (size_t)pern * c->esz));
    }
    CGLB_TRY(comm_allgather(c, m->gat, 2 * pern));
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((unpack_pairs_kernel<T>), dim3((unsigned)((n_new + 255) / 256)), dim3(256), 0, c->stream,
                                                 (const T*)m->gat, pern, n_new, (T*)f_mean, (T*)f_var));
    CGLB_LAUNCH_CHECK(c);
    HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return CGLB_OK;
}

int cglb_get_matrix(cglb_ctx* c, int which, void* dst) {
    if (!c || !dst) return CGLB_ERR_BAD_ARG;
    CGLB_TRY(require_terms(c));
    HIP_CHECK(c, hipSetDevice(c->device));
    const size_t e = c->esz;
    if (which == 0) {  // A [M][nloc] (strip the lda padding)
        HIP_CHECK(c, hipMemcpy2DAsync(dst, (size_t)c->nloc * e, c->At, (size_t)c->lda * e, (size_t)c->nloc * e, (size_t)c->M, hipMemcpyDefault, c->stream));
    } else if (which == 1 || which == 2) {  // column-major lower -> row-major lower via transpose into Mtmp2
        CGLB_TRY(launch_transpose(c, which == 1 ? c->Lc : c->LBc, c->Mtmp2));
        HIP_CHECK(c, hipMemcpyAsync(dst, c->Mtmp2, (size_t)c->M * c->M * e, hipMemcpyDefault, c->stream));
    } else {
        return cglb_fail(c, CGLB_ERR_BAD_ARG, "unknown matrix id");
    }
    HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return CGLB_OK;
}

int cglb_get_stat(cglb_ctx* c, const char* name, double* value) {
    if (!c || !name || !value) return CGLB_ERR_BAD_ARG;
    if (!strcmp(name, "k1_ms_total") || !strcmp(name, "k1_launches")) {
        CGLB_TRY(k1_profile_collect(c));
        *value = !strcmp(name, "k1_ms_total") ? c->k1_ms_total : (double)c->k1_launches;
        return CGLB_OK;
    }
    {
        static const char* names[] = {"eval_setup_ms", "eval_pcg_ms", "eval_final_ms", "eval_grad_ms"};
        for (int k = 0; k < 4; ++k)
            if (!strcmp(name, names[k])) { CGLB_TRY(eval_collect(c)); *value = c->eval_ms[k]; return CGLB_OK; }
        if (!strcmp(name, "eval_count")) { CGLB_TRY(eval_collect(c)); *value = (double)c->eval_count; return CGLB_OK; }
    }
    if (!strcmp(name, "k1_pairs_per_launch")) { *value = c->sym_pairs; return CGLB_OK; }  // of the most recent symmetric launch geometry
    if (!strcmp(name, "kpart_bytes")) { *value = (double)c->kpart_cap; return CGLB_OK; }     // partial-sum slabs of the mat-vec
    if (!strcmp(name, "comm_allreduce_calls")) { *value = c->comm ? (double)c->comm->n_allreduce : 0.0; return CGLB_OK; }
    if (!strcmp(name, "comm_allgather_calls")) { *value = c->comm ? (double)c->comm->n_allgather : 0.0; return CGLB_OK; }
    if (!strcmp(name, "L_diag_ratio")) { *value = c->L_diag_ratio; return CGLB_OK; }         // of the last cglb_setup (chooses the gradient algebra)
    return cglb_fail(c, CGLB_ERR_BAD_ARG, std::string("unknown statistic ") + name);
}

int cglb_time_kernel(cglb_ctx* c, int which, int reps, double* ms_avg) {
    if (c) c->obj_valid = false;
    if (!c || !ms_avg || reps <= 0) return CGLB_ERR_BAD_ARG;
    if (!c->have_hypers) return cglb_fail(c, CGLB_ERR_STATE, "set_hypers must precede timing");
    if (which != 0 && which != 3 && which != 4 && which != 5) CGLB_TRY(require_terms(c));
    HIP_CHECK(c, hipSetDevice(c->device));
    // operands: y as a generic vector (values do not change the instruction stream)
    hipEvent_t e0, e1;
    HIP_CHECK(c, hipEventCreate(&e0));
    HIP_CHECK(c, hipEventCreate(&e1));
    int rc = CGLB_OK;
    auto once = [&]() -> int {
        if (which == 0) return launch_kff_matvec(c, c->y, c->w_Ap, nullptr);
        if (which == 1) return precond_single(c, (const char*)c->y + (size_t)c->r0 * c->esz, c->w_z, c->scal + S_TMP);
        if (which == 2) return launch_grad_kff(c, c->y, (const char*)c->y + (size_t)c->r0 * c->esz, c->scal + S_TMP2);
        if (which == 4) {  // this rank's cyclic share of the global symmetric mat-vec (pair kernel alone), multi-GPU path
            c->kff_skip_combine = true;
            const int r = launch_kff_sym_cyclic(c, c->y, c->w_pfull);
            c->kff_skip_combine = false;
            return r;
        }
        if (which == 5) {  // K_uu + jitter I and its blocked Cholesky (the first factorisation of the common terms)
            c->have_local = c->have_terms = false;  // L is overwritten without the clean-up of the strict upper triangle
            CGLB_TRY(launch_kuu(c));
            return launch_cholesky_lower(c, c->Lc, (int*)c->info_dev);
        }
        if (which == 3) {  // the pair kernel of the mat-vec alone (what rocprofv3 lists as kff_matvec_kernel)
            c->kff_skip_combine = true;
            const int r = launch_kff_matvec(c, c->y, c->w_Ap, nullptr);
            c->kff_skip_combine = false;
            return r;
        }
        return cglb_fail(c, CGLB_ERR_BAD_ARG, "unknown kernel id");
    };
    rc = once();  // warm-up (also sizes the work buffers)
    if (rc == CGLB_OK) {
        (void)hipEventRecord(e0, c->stream);
        for (int i = 0; i < reps && rc == CGLB_OK; ++i) rc = once();
        (void)hipEventRecord(e1, c->stream);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *ms_avg = (double)ms / reps;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

}  // extern "C"
