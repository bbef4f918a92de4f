// Wide inputs (D > 32).  The reference takes any of the Wilson UCI sets (cglb_experiments/datasets.py:47-76: up to D = 385); its own
// experiments stop at D = 27 (xpert-main.toml:28), which is what the register-resident pair kernels (kernels_kff*.hip, kernels_grad.hip)
// are built for: a lane keeps its row operands in VGPRs and the column operands arrive as scalars.  Beyond 32 dimensions that no longer
// fits - and it no longer is the right shape either: the Gram part of the pair value,
//     RBF       kappa_ij = 2^(a_i + a_j + x_i.x_j),                    a = -|x|^2/2   (x scaled by sqrt(log2 e)/l)
//     Matern32  kappa_ij = (1 + r ln2) 2^(-r), r^2 = a_i + a_j - 2 x_i.x_j,  a = |x|^2      (x scaled by sqrt3 log2 e/l)
// is a contraction with k = D >= 33: a true GEMM, which belongs on the matrix cores.  So this path forms tiles G = X_I X_J^T with rocBLAS
// (fp64 MFMA) and runs the kernel profile as a streaming pass over the tile:
//   * K_ff mat-vec (models.py:251-252, conjugate_gradient.py:57,66,72): row sums sum_j kappa_ij p_j straight off the Gram tile (the
//     kernel values are never stored), column slices of 64 per workgroup, fixed-order partial sums;
//   * K_uf / K_uu / K_us panels (models.py:196-201, :337): the tile IS the result (in place);
//   * the N^2 form of the lengthscale gradient (row G of SURVEY 8a): with the derivative factors H of a tile materialised,
//       sum_ij u_i h_ij v_j (x_id - x_jd)^2 = sum_i x_id^2 R_i - 2 sum_i x_id u_i T_id + sum_j x_jd^2 C_j,
//       R = u o (H v), C = v o (H^T u), T = H (v o X)   -> two GEMVs and one GEMM per tile;
//   * the panel gradients (K_uf, K_uu adjoints): W = (G + c w^T) o H in one pass, then T = W X (GEMM), row and column sums (GEMV).
// Everything is deterministic (no atomics).  Accuracy: the Gram form loses ~eps (|x_i|^2 + |x_j|^2) absolute on the exponent (the
// direct differences of the narrow panel kernels do not); with centred inputs that is 1e-13 relative on kernel values at |x|^2 ~ 10^3.
#include <algorithm>

#include "devmath.h"
#include "dispatch.h"

namespace {

constexpr int64_t TILE = 4096;   // rows x columns of a Gram tile (134 MB in fp64: stays in the 256-MB Infinity Cache between GEMM and pass)
constexpr int CSLICE = 64;       // columns per workgroup of the row-sum pass

inline rocblas_status wgemm(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, double al, const double* A, int lda,
                            const double* B, int ldb, double be, double* C, int ldc) { return rocblas_dgemm(h, ta, tb, m, n, k, &al, A, lda, B, ldb, &be, C, ldc); }
inline rocblas_status wgemm(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, double al, const float* A, int lda,
                            const float* B, int ldb, double be, float* C, int ldc) {
    const float a = (float)al, b = (float)be;
    return rocblas_sgemm(h, ta, tb, m, n, k, &a, A, lda, B, ldb, &b, C, ldc);
}
inline rocblas_status wgemm_sb(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, const double* A, int lda, rocblas_stride sa,
                               const double* B, int ldb, rocblas_stride sb, double* C, int ldc, rocblas_stride sc, int batch) {
    const double one = 1.0, zero = 0.0;
    return rocblas_dgemm_strided_batched(h, ta, tb, m, n, k, &one, A, lda, sa, B, ldb, sb, &zero, C, ldc, sc, batch);
}
inline rocblas_status wgemm_sb(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, const float* A, int lda, rocblas_stride sa,
                               const float* B, int ldb, rocblas_stride sb, float* C, int ldc, rocblas_stride sc, int batch) {
    const float one = 1.0f, zero = 0.0f;
    return rocblas_sgemm_strided_batched(h, ta, tb, m, n, k, &one, A, lda, sa, B, ldb, sb, &zero, C, ldc, sc, batch);
}
inline rocblas_status wgemv(rocblas_handle h, rocblas_operation t, int m, int n, double al, const double* A, int lda, const double* x, double be, double* y) {
    return rocblas_dgemv(h, t, m, n, &al, A, lda, x, 1, &be, y, 1);
}
inline rocblas_status wgemv(rocblas_handle h, rocblas_operation t, int m, int n, double al, const float* A, int lda, const float* x, double be, float* y) {
    const float a = (float)al, b = (float)be;
    return rocblas_sgemv(h, t, m, n, &a, A, lda, x, 1, &b, y, 1);
}

int wensure(cglb_ctx* c, void** p, size_t* cap, size_t need) {
    if (need <= *cap && *p) return CGLB_OK;
    if (*p) HIP_CHECK(c, hipFree(*p));
    *p = nullptr;
    *cap = 0;
    HIP_CHECK(c, hipMalloc(p, need ? need : 16));
    *cap = need;
    return CGLB_OK;
}

// ---- operand preparation: xs = (x - centre) * scale, a = norm term, xsq = xs o xs ---------------------------------------------------
template <typename T, int KIND>
__global__ __launch_bounds__(256) void wide_prep_kernel(const T* __restrict__ X, int64_t n, int D, const double* __restrict__ center,
                                                        const double* __restrict__ scale, T* __restrict__ Xs, T* __restrict__ xa, T* __restrict__ Xsq) {
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // one wave per row: coalesced along d
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    double s2 = 0.0;
    for (int d = lane; d < D; d += 64) {
        const T v = (T)(((double)X[i * D + d] - center[d]) * scale[d]);
        Xs[i * D + d] = v;
        if (Xsq) Xsq[i * D + d] = v * v;
        s2 += (double)v * (double)v;
    }
    s2 = wave_sum(s2);
    if (lane == 0) xa[i] = (KIND == CGLB_RBF) ? (T)(-0.5 * s2) : (T)s2;
}

// Hot operand set of a mid-width context (32 < D <= 96, fp64): xh = (x - centre) * scale * hot, zero padded to Dh, a = |xh|^2 term -
// the layout kernels_kff_sym.hip streams (one wave per row)
template <typename T, int KIND>
__global__ __launch_bounds__(256) void wide_prep_hot_kernel(const T* __restrict__ X, int64_t n, int D, int Dh, const double* __restrict__ center,
                                                            const double* __restrict__ scale, double hot, T* __restrict__ Xh, T* __restrict__ xah) {
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    double s2 = 0.0;
    for (int d = lane; d < Dh; d += 64) {
        T v = T(0);
        if (d < D) v = (T)(((double)X[i * D + d] - center[d]) * (scale[d] * hot));
        Xh[i * Dh + d] = v;
        s2 += (double)v * (double)v;
    }
    s2 = wave_sum(s2);
    if (lane == 0) xah[i] = (KIND == CGLB_RBF) ? (T)(-0.5 * s2) : (T)s2;
}

// kernel value / derivative factor from a Gram entry.  MODE 0: kappa; 1: h with dk/dl_d = var h delta_d^2 / l_d (RBF: kappa; Matern-3/2: 3 2^(-r))
template <typename T, int KIND, int MODE> __device__ __forceinline__ T wide_profile(T g, T ai, T aj) {
    if (KIND == CGLB_RBF) return exp2_neg(tmin<T>(ai + aj + g, T(0)));   // the exact exponent is <= 0; round-off may leave it a few ulp above
    const T d2 = tmax<T>(tfma<T>(T(-2), g, ai + aj), T(0));
    const T r = sqrt_pos(d2);
    const T e = exp2_neg(-r);
    return MODE == 0 ? tfma<T>(r, T(CGLB_LN2), T(1)) * e : T(3) * e;
}

// in place: G[i + j ld] <- scale * profile(G, a_row[i], a_col[j]) (+ diag on i == j); one thread per element, coalesced along i
template <typename T, int KIND, int MODE>
__global__ __launch_bounds__(256) void wide_profile_kernel(T* __restrict__ G, int64_t ld, int64_t nr, int64_t nc, const T* __restrict__ arow,
                                                           const T* __restrict__ acol, T scale, T diag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j = blockIdx.y;
    if (i >= nr || j >= nc) return;
    T k = scale * wide_profile<T, KIND, MODE>(G[i + j * ld], arow[i], acol[j]);
    if (diag != T(0) && i == j) k += diag;
    G[i + j * ld] = k;
}

// row sums of a Gram tile against p: part[s][i] = sum_{j in slice s} kappa(G_ij) p_j.  Block = 256 rows x CSLICE columns.
template <typename T, int KIND>
__global__ __launch_bounds__(256) void wide_rowsum_kernel(const T* __restrict__ G, int64_t ld, int64_t nr, int64_t nc, const T* __restrict__ arow,
                                                          const T* __restrict__ acol, const T* __restrict__ p, T* __restrict__ part) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j0 = (int64_t)blockIdx.y * CSLICE;
    const int64_t j1 = j0 + CSLICE < nc ? j0 + CSLICE : nc;
    if (i >= nr) return;
    const T ai = arow[i];
    T acc = 0;
    for (int64_t j = j0; j < j1; ++j) acc = tfma<T>(wide_profile<T, KIND, 0>(G[i + j * ld], ai, acol[j]), p[j], acc);
    part[(int64_t)blockIdx.y * nr + i] = acc;
}

// out[i] (+)= var * sum_s part[s][i] (+ noise * pdiag[i] once)
template <typename T>
__global__ __launch_bounds__(256) void wide_rowsum_reduce_kernel(const T* __restrict__ part, int nslice, int64_t nr, T var, T* __restrict__ out, int accumulate,
                                                                 T noise, const T* __restrict__ pdiag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nr) return;
    T s = 0;
    for (int q = 0; q < nslice; ++q) s += part[(int64_t)q * nr + i];
    T o = var * s;
    if (accumulate) o += out[i];
    if (pdiag) o = tfma<T>(noise, pdiag[i], o);
    out[i] = o;
}

// Symmetric use of a Gram tile that lies right of the diagonal: kappa_ij serves out_i += kappa p_j (row sums, as above) AND
// out_j += kappa p_i (column sums: a wave reduction per column, 12 shuffle/add per 64 pairs).  pcpart[(bx * 4 + wave)][j].
template <typename T, int KIND>
__global__ __launch_bounds__(256) void wide_rowcol_kernel(const T* __restrict__ G, int64_t ld, int64_t nr, int64_t nc, const T* __restrict__ arow,
                                                          const T* __restrict__ acol, const T* __restrict__ prow, const T* __restrict__ pcol,
                                                          T* __restrict__ part, T* __restrict__ pcpart) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j0 = (int64_t)blockIdx.y * CSLICE;
    const int64_t j1 = j0 + CSLICE < nc ? j0 + CSLICE : nc;
    const bool live = i < nr;
    const int64_t ii = live ? i : nr - 1;
    const T ai = arow[ii];
    const T pi = live ? prow[ii] : T(0);
    const int lane = threadIdx.x & 63;
    const int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    T acc = 0;
    for (int64_t j = j0; j < j1; ++j) {
        const T k = live ? wide_profile<T, KIND, 0>(G[ii + j * ld], ai, acol[j]) : T(0);
        acc = tfma<T>(k, pcol[j], acc);
        const T cs = wave_sum(k * pi);
        if (lane == 0) pcpart[chunk * nc + j] = cs;
    }
    if (live) part[(int64_t)blockIdx.y * nr + i] = acc;
}

// Gradient pass, one tile: G <- H = derivative factor of the Gram entries (in place, for the moment GEMMs that follow) and, in the same
// pass over the tile, the row sums against v_J (and u_J) and the column sums against u_I (and v_I) that four GEMVs over the stored tile
// used to take.  Block = 256 rows x CSLICE columns; row partials part_*[slice][i], column partials pc_*[(bx * 4 + wave)][j].
// BOTH: the tile also stands for its mirror image (symmetric use), which needs the second sum of each kind.
template <typename T, int KIND, bool BOTH>
__global__ __launch_bounds__(256) void wide_grad_profile_kernel(T* __restrict__ G, int64_t ld, int64_t nr, int64_t nc, const T* __restrict__ arow,
                                                                const T* __restrict__ acol, const T* __restrict__ urow, const T* __restrict__ vrow,
                                                                const T* __restrict__ ucol, const T* __restrict__ vcol, T* __restrict__ part_v,
                                                                T* __restrict__ part_u, T* __restrict__ pc_u, T* __restrict__ pc_v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j0 = (int64_t)blockIdx.y * CSLICE;
    const int64_t j1 = j0 + CSLICE < nc ? j0 + CSLICE : nc;
    const bool live = i < nr;
    const int64_t ii = live ? i : nr - 1;
    const T ai = arow[ii];
    const T ui = live ? urow[ii] : T(0);
    const T vi = (BOTH && live) ? vrow[ii] : T(0);
    const int lane = threadIdx.x & 63;
    const int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    T av = 0, au = 0;
    for (int64_t j = j0; j < j1; ++j) {
        T h = T(0);
        if (live) {
            h = wide_profile<T, KIND, 1>(G[ii + j * ld], ai, acol[j]);
            G[ii + j * ld] = h;
        }
        av = tfma<T>(h, vcol[j], av);
        const T cu = wave_sum(h * ui);
        if (lane == 0) pc_u[chunk * nc + j] = cu;
        if constexpr (BOTH) {
            au = tfma<T>(h, ucol[j], au);
            const T cv = wave_sum(h * vi);
            if (lane == 0) pc_v[chunk * nc + j] = cv;
        }
    }
    if (live) {
        part_v[(int64_t)blockIdx.y * nr + i] = av;
        if constexpr (BOTH) part_u[(int64_t)blockIdx.y * nr + i] = au;
    }
}
// out[i] += a[i] * sum_q part[q][i]   (fixed order)
template <typename T>
__global__ __launch_bounds__(256) void wide_partsum_mulacc_kernel(const T* __restrict__ part, int nq, int64_t n, const T* __restrict__ a, T* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T s = 0;
    for (int q = 0; q < nq; ++q) s += part[(int64_t)q * n + i];
    out[i] = tfma<T>(a[i], s, out[i]);
}

template <typename T> __global__ __launch_bounds__(256) void wide_fill_kernel(T* __restrict__ x, int64_t n, T v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = v;
}
// y[i] (+)= a[i] * b[i]
template <typename T> __global__ __launch_bounds__(256) void wide_mulacc_kernel(T* __restrict__ y, const T* __restrict__ a, const T* __restrict__ b, int64_t n, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = accumulate ? tfma<T>(a[i], b[i], y[i]) : a[i] * b[i];
}
// VX[i][d] = v[i] * Xs[i][d]
template <typename T> __global__ __launch_bounds__(256) void wide_rowscale_kernel(const T* __restrict__ Xs, const T* __restrict__ v, int64_t n, int D, T* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n * D) out[idx] = v[idx / D] * Xs[idx];
}
// out[d] = sum_i w[i] A[i][d] B[i][d]  (one block per d; B may be null = 1)
template <typename T>
__global__ __launch_bounds__(256) void wide_coldot_kernel(const T* __restrict__ A, const T* __restrict__ B, const T* __restrict__ w, int64_t n, int D,
                                                          double* __restrict__ out) {
    __shared__ double smem[16];
    const int d = blockIdx.x;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += (double)w[i] * (double)A[i * D + d] * (B ? (double)B[i * D + d] : 1.0);
    s = block_sum(s, smem);
    if (threadIdx.x == 0) out[d] = s;
}

// panel pass: S[n + m ld] = (G[m ldg + n] + c_m w_n) * h(gram), block partial of sum (G + c w^T) kappa.  gram arrives in S.
template <typename T, int KIND>
__global__ __launch_bounds__(256) void wide_panel_weight_kernel(T* __restrict__ S, int64_t ld, const T* __restrict__ G, int64_t ldg, const T* __restrict__ cvec,
                                                                const T* __restrict__ wvec, const T* __restrict__ acol, const T* __restrict__ az, int64_t ncols,
                                                                double* __restrict__ fpart) {
    __shared__ double smem[16];
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y;
    double f = 0.0;
    if (n < ncols) {
        T g = G[(int64_t)m * ldg + n];
        if (cvec) g = tfma<T>(cvec[m], wvec[n], g);
        const T gram = S[n + (int64_t)m * ld];
        const T kap = wide_profile<T, KIND, 0>(gram, acol[n], az[m]);
        const T h = (KIND == CGLB_RBF) ? kap : wide_profile<T, KIND, 1>(gram, acol[n], az[m]);
        S[n + (int64_t)m * ld] = g * h;
        f = (double)g * (double)kap;
    }
    f = block_sum(f, smem);
    if (threadIdx.x == 0) fpart[(int64_t)m * gridDim.x + blockIdx.x] = f;
}

// packed-gradient assembly of a panel: dl[d] += var sl_d (sum_m z_md^2 R_m - 2 z_md T_md + xc_d); dvar += sum fpart; dZ[m][d] += -zf var sz_d (z_md R_m - T_md)
template <typename T>
__global__ __launch_bounds__(256) void wide_panel_finish_kernel(const T* __restrict__ Zs, const T* __restrict__ Tm, const T* __restrict__ R, const double* __restrict__ xc,
                                                                const double* __restrict__ fpart, int64_t nf, int M, int D, const double* __restrict__ sl,
                                                                const double* __restrict__ sz, double var, double zfactor, double* __restrict__ out) {
    __shared__ double smem[16];
    const int b = blockIdx.x;
    if (b < D) {
        double s = 0.0;
        for (int m = threadIdx.x; m < M; m += blockDim.x) {
            const double z = (double)Zs[(int64_t)m * D + b];
            s += z * (z * (double)R[m] - 2.0 * (double)Tm[(int64_t)m * D + b]);
        }
        s = block_sum(s, smem);
        if (threadIdx.x == 0) out[b] += var * sl[b] * (s + xc[b]);
    } else if (b == D) {
        double s = 0.0;
        for (int64_t k = threadIdx.x; k < nf; k += blockDim.x) s += fpart[k];
        s = block_sum(s, smem);
        if (threadIdx.x == 0) out[D] += s;
    } else {
        const int64_t idx = (int64_t)(b - D - 1) * blockDim.x + threadIdx.x;
        if (idx < (int64_t)M * D) {
            const int m = (int)(idx / D), d = (int)(idx % D);
            out[D + 3 + idx] += -zfactor * var * sz[d] * ((double)Zs[idx] * (double)R[m] - (double)Tm[idx]);
        }
    }
}

// dl[d] = var sl_d (a_d - 2 b_d + c_d)
__global__ void wide_dl_finish_kernel(const double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ cc, const double* __restrict__ sl, int D,
                                      double var, double* __restrict__ out) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d < D) out[d] = var * sl[d] * (a[d] - 2.0 * b[d] + cc[d]);
}

inline double kscale_of(const cglb_ctx* c) { return (c->kind == CGLB_RBF) ? sqrt(CGLB_LOG2E) : CGLB_SQRT3 * CGLB_LOG2E; }

// device copies of centre / scale and the per-dimension gradient factors: wsmall = [sl (D): 1/(l ks^2) | sz (D): 1/(l ks) | 3 x D scratch]
int upload_scales(cglb_ctx* c) {
    const int D = c->D;
    if (!c->wcenter) {
        HIP_CHECK(c, hipMalloc((void**)&c->wcenter, (size_t)D * sizeof(double)));
        HIP_CHECK(c, hipMalloc((void**)&c->wscale, (size_t)D * sizeof(double)));
        HIP_CHECK(c, hipMalloc((void**)&c->wsmall, (size_t)5 * D * sizeof(double)));
    }
    std::vector<double> h(4 * (size_t)D);
    const double ks = kscale_of(c);
    for (int d = 0; d < D; ++d) {
        h[d] = c->xmean[d];
        h[D + d] = ks / c->ls[d];
        h[2 * D + d] = 1.0 / (c->ls[d] * ks * ks);
        h[3 * D + d] = 1.0 / (c->ls[d] * ks);
    }
    HIP_CHECK(c, hipMemcpyAsync(c->wcenter, h.data(), (size_t)D * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(c, hipMemcpyAsync(c->wscale, h.data() + D, (size_t)D * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(c, hipMemcpyAsync(c->wsmall, h.data() + 2 * D, (size_t)2 * D * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(c, hipStreamSynchronize(c->stream));  // `h` is pageable host memory
    return CGLB_OK;
}

int ensure_ones(cglb_ctx* c) {
    const int64_t n = std::max<int64_t>(c->N, c->M);
    if (c->wones) return CGLB_OK;
    HIP_CHECK(c, hipMalloc(&c->wones, (size_t)n * c->esz));
    HIP_CHECK(c, hipMalloc(&c->wR, (size_t)n * c->esz));
    HIP_CHECK(c, hipMalloc(&c->wC, (size_t)n * c->esz));
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((wide_fill_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (T*)c->wones, n, (T)1));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// Gram tile of rows [i0, i0+nr) of A against rows [j0, j0+nc) of B (both row-major [.][D]) into c->wtile, column-major nr x nc, ld = nr
template <typename T>
int gram_tile(cglb_ctx* c, const T* A, int64_t nr, const T* B, int64_t nc, T* out, int64_t ld) {
    BLAS_CHECK(c, wgemm(c->blas, rocblas_operation_transpose, rocblas_operation_none, (int)nr, (int)nc, c->D, 1.0, A, c->D, B, c->D, 0.0, out, (int)ld));
    return CGLB_OK;
}

template <typename T, int KIND>
int matvec_impl(cglb_ctx* c, const T* XsRow, const T* xaRow, int64_t row0_global, int64_t nrows, const T* p_full, T* out, bool diag_noise,
                double* pdot_slot, int tile_stride, int tile_offset) {
    const int64_t N = c->N;
    CGLB_TRY(wensure(c, &c->wtile, &c->wtile_cap, (size_t)TILE * TILE * sizeof(T)));
    // [row-sum partials: (TILE / CSLICE) x TILE | column-sum partials: (TILE / 64) x TILE]
    CGLB_TRY(wensure(c, &c->wpart, &c->wpart_cap, (size_t)2 * (TILE / CSLICE) * TILE * sizeof(T)));
    T* G = (T*)c->wtile;
    T* part = (T*)c->wpart;
    T* pcpart = part + (size_t)(TILE / CSLICE) * TILE;
    // The square of the training inputs (all rows against all columns: the single-GPU mat-vec and the cyclic multi-GPU form) is evaluated
    // on its upper block triangle only, every tile right of the diagonal serving its transposed contribution as well: half the Gram GEMMs
    // and half the kernel evaluations.  Rectangular products (new points, a row shard) take every tile.
    const bool sym = (XsRow == (const T*)c->Xs) && row0_global == 0 && nrows == N;
    if (sym) HIP_CHECK(c, hipMemsetAsync(out, 0, (size_t)N * sizeof(T), c->stream));
    int64_t t = 0;
    for (int64_t i0 = 0; i0 < nrows; i0 += TILE, ++t) {
        const int64_t nr = std::min(TILE, nrows - i0);
        const bool mine = !(tile_stride > 1 && (t % tile_stride) != tile_offset);
        if (!mine) {  // another rank's row tile: nothing of it is added to this partial
            if (!sym) HIP_CHECK(c, hipMemsetAsync(out + i0, 0, (size_t)nr * sizeof(T), c->stream));
            continue;
        }
        for (int64_t j0 = sym ? i0 : 0; j0 < N; j0 += TILE) {
            const int64_t nc = std::min(TILE, N - j0);
            CGLB_TRY(gram_tile<T>(c, XsRow + i0 * c->D, nr, (const T*)c->Xs + j0 * c->D, nc, G, nr));
            const int nslice = (int)((nc + CSLICE - 1) / CSLICE);
            const unsigned gx = (unsigned)((nr + 255) / 256);
            const bool both = sym && j0 > i0;
            if (both)
                hipLaunchKernelGGL((wide_rowcol_kernel<T, KIND>), dim3(gx, (unsigned)nslice), dim3(256), 0, c->stream, (const T*)G, nr, nr, nc, xaRow + i0,
                                   (const T*)c->xa + j0, p_full + i0, p_full + j0, part, pcpart);
            else
                hipLaunchKernelGGL((wide_rowsum_kernel<T, KIND>), dim3(gx, (unsigned)nslice), dim3(256), 0, c->stream, (const T*)G, nr, nr, nc, xaRow + i0,
                                   (const T*)c->xa + j0, p_full + j0, part);
            const bool last = j0 + TILE >= N;
            // sym: `out` was zeroed and every tile accumulates; otherwise the first column tile overwrites
            hipLaunchKernelGGL((wide_rowsum_reduce_kernel<T>), dim3(gx), dim3(256), 0, c->stream, (const T*)part, nslice, nr, (T)c->var, out + i0,
                               (sym || j0 > 0) ? 1 : 0, (T)c->noise, (diag_noise && last) ? p_full + row0_global + i0 : (const T*)nullptr);
            if (both)
                hipLaunchKernelGGL((wide_rowsum_reduce_kernel<T>), dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, c->stream, (const T*)pcpart, (int)(gx * 4), nc,
                                   (T)c->var, out + j0, 1, (T)0, (const T*)nullptr);
            CGLB_LAUNCH_CHECK(c);
        }
    }
    if (pdot_slot) CGLB_TRY(launch_dot(c, p_full + row0_global, out, nrows, pdot_slot));
    return CGLB_OK;
}

}  // namespace

// =========================================================== launchers ===========================================================
void wide_free(cglb_ctx* c) {
    void* ptrs[] = {c->wlong, c->Xsq, c->Zsq, c->wtile, c->wpart, c->wS1, c->wVX, c->wR, c->wC, c->wones, c->wpanel, c->wcenter, c->wscale, c->wsmall};
    for (void* p : ptrs) if (p) (void)hipFree(p);
}

// xs, a (and xs o xs when asked) of n raw rows; centre and scale come from the device copies refreshed by wide_after_hypers
int wide_prep_scaled(cglb_ctx* c, const void* Xraw, int64_t n, void* Xs_out, void* xa_out, void* Xsq_out) {
    if (n == 0) return CGLB_OK;
    CGLB_TRY(upload_scales(c));   // the lengthscales may have changed since the last call (set_hypers, select_inducing): D doubles
    CGLB_DISPATCH_T(c->dtype, CGLB_DISPATCH_KIND(c->kind, hipLaunchKernelGGL((wide_prep_kernel<T, KIND>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, c->stream,
                                                                             (const T*)Xraw, n, c->D, (const double*)c->wcenter, (const double*)c->wscale,
                                                                             (T*)Xs_out, (T*)xa_out, (T*)Xsq_out)));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// after the lengthscales changed: device copies of the scales, scaled operands of X and Z with their squares
int wide_after_hypers(cglb_ctx* c) {
    if (!c->Xsq) HIP_CHECK(c, hipMalloc(&c->Xsq, (size_t)c->N * c->D * c->esz));
    if (!c->Zsq) HIP_CHECK(c, hipMalloc(&c->Zsq, (size_t)c->M * c->D * c->esz));
    CGLB_TRY(ensure_ones(c));
    CGLB_TRY(wide_prep_scaled(c, c->X, c->N, c->Xs, c->xa, c->Xsq));
    CGLB_TRY(wide_prep_scaled(c, c->Z, c->M, c->Zs, c->za, c->Zsq));
    return CGLB_OK;
}

int wide_prep_hot(cglb_ctx* c) {   // after wide_after_hypers: the device copies of centre and scale are current
    if (c->Dh == 0 || c->N == 0) return CGLB_OK;
    CGLB_DISPATCH_T(c->dtype, CGLB_DISPATCH_KIND(c->kind, hipLaunchKernelGGL((wide_prep_hot_kernel<T, KIND>), dim3((unsigned)((c->N + 3) / 4)), dim3(256), 0, c->stream,
                                                                             (const T*)c->X, c->N, c->D, c->Dh, (const double*)c->wcenter, (const double*)c->wscale,
                                                                             cglb_hot_scale(c), (T*)c->Xh, (T*)c->xah)));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// At (nloc x M column-major, ld = lda) <- var * K(x_n, z_m) for the local rows
int wide_kuf(cglb_ctx* c) {
    if (c->nloc == 0) return CGLB_OK;
    CGLB_DISPATCH_T(c->dtype, CGLB_DISPATCH_KIND(c->kind, {
        CGLB_TRY(gram_tile<T>(c, (const T*)c->Xs + c->r0 * c->D, c->nloc, (const T*)c->Zs, c->M, (T*)c->At, c->lda));
        hipLaunchKernelGGL((wide_profile_kernel<T, KIND, 0>), dim3((unsigned)((c->nloc + 255) / 256), (unsigned)c->M), dim3(256), 0, c->stream, (T*)c->At, c->lda,
                           c->nloc, (int64_t)c->M, (const T*)c->xa + c->r0, (const T*)c->za, (T)c->var, (T)0);
    }));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

int wide_kuu(cglb_ctx* c) {
    const int M = c->M;
    CGLB_DISPATCH_T(c->dtype, CGLB_DISPATCH_KIND(c->kind, {
        CGLB_TRY(gram_tile<T>(c, (const T*)c->Zs, M, (const T*)c->Zs, M, (T*)c->Lc, M));
        hipLaunchKernelGGL((wide_profile_kernel<T, KIND, 0>), dim3((unsigned)((M + 255) / 256), (unsigned)M), dim3(256), 0, c->stream, (T*)c->Lc, (int64_t)M, (int64_t)M,
                           (int64_t)M, (const T*)c->za, (const T*)c->za, (T)c->var, (T)c->jitter);
    }));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// out[m * ld + n] = var * K(z_m, xnew_n)   (models.py:337)
int wide_kus(cglb_ctx* c, const void* XsNew, const void* xaNew, int64_t n_new, int64_t ld, void* out) {
    CGLB_DISPATCH_T(c->dtype, CGLB_DISPATCH_KIND(c->kind, {
        CGLB_TRY(gram_tile<T>(c, (const T*)XsNew, n_new, (const T*)c->Zs, c->M, (T*)out, ld));
        hipLaunchKernelGGL((wide_profile_kernel<T, KIND, 0>), dim3((unsigned)((n_new + 255) / 256), (unsigned)c->M), dim3(256), 0, c->stream, (T*)out, ld, n_new,
                           (int64_t)c->M, (const T*)xaNew, (const T*)c->za, (T)c->var, (T)0);
    }));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// out[i] = var sum_j kappa(row_i, x_j) p_j over all N columns (+ noise p_{row0_global + i} when diag_noise).  Row tiles can be dealt
// round-robin (tile_stride, tile_offset: the cyclic multi-GPU form - rows of other ranks' tiles are zeroed in this partial).
int wide_matvec(cglb_ctx* c, const void* XsRow, const void* xaRow, int64_t row0_global, int64_t nrows, const void* p_full, void* out, bool diag_noise,
                double* pdot_slot, int tile_stride, int tile_offset) {
    if (nrows == 0) return CGLB_OK;
    CGLB_DISPATCH_T(c->dtype, CGLB_DISPATCH_KIND(c->kind, return (matvec_impl<T, KIND>(c, (const T*)XsRow, (const T*)xaRow, row0_global, nrows, (const T*)p_full,
                                                                                       (T*)out, diag_noise, pdot_slot, tile_stride, tile_offset))));
}

// out_dl[d] (device double[D], overwritten) = sum_{i in rows, j} u_i dK_ij/dl_d v_j  for the rows [row0, row0 + nrows) of X (u_rows indexed from row0)
int wide_grad_kff(cglb_ctx* c, const void* v_full_, const void* u_rows_, int64_t row0, int64_t nrows, int tile_stride, int tile_offset, double* out_dl) {
    const int D = c->D;
    const int64_t N = c->N;
    if (nrows == 0) { HIP_CHECK(c, hipMemsetAsync(out_dl, 0, sizeof(double) * D, c->stream)); return CGLB_OK; }
    CGLB_TRY(ensure_ones(c));
    CGLB_DISPATCH_T(c->dtype, CGLB_DISPATCH_KIND(c->kind, {
        const T *v = (const T*)v_full_, *u = (const T*)u_rows_;
        CGLB_TRY(wensure(c, &c->wtile, &c->wtile_cap, (size_t)TILE * TILE * sizeof(T)));
        CGLB_TRY(wensure(c, &c->wS1, &c->wS1_cap, (size_t)2 * N * D * sizeof(T)));   // [T' (nrows x D) | VX (N x D)]
        T* G = (T*)c->wtile;
        T* Tp = (T*)c->wS1;
        T* VX = Tp + (size_t)N * D;
        T *R = (T*)c->wR, *C = (T*)c->wC;   // R: rows of this call, C: all columns
        CGLB_TRY(wensure(c, &c->wpart, &c->wpart_cap, (size_t)4 * (TILE / CSLICE) * TILE * sizeof(T)));   // two row-partial and two column-partial slabs
        T* tmp = (T*)c->wpart;
        HIP_CHECK(c, hipMemsetAsync(Tp, 0, (size_t)nrows * D * sizeof(T), c->stream));
        HIP_CHECK(c, hipMemsetAsync(R, 0, (size_t)nrows * sizeof(T), c->stream));
        HIP_CHECK(c, hipMemsetAsync(C, 0, (size_t)N * sizeof(T), c->stream));
        hipLaunchKernelGGL((wide_rowscale_kernel<T>), dim3((unsigned)((N * D + 255) / 256)), dim3(256), 0, c->stream, (const T*)c->Xs, v, N, D, VX);
        const bool sym = row0 == 0 && nrows == N && c->wide_grad_sym;
        int64_t t = 0;
        for (int64_t i0 = 0; i0 < nrows; i0 += TILE, ++t) {
            if (tile_stride > 1 && (t % tile_stride) != tile_offset) continue;
            const int64_t nr = std::min(TILE, nrows - i0);
            // The rows are all of X (one GPU, or the cyclic deal of the row tiles): H is symmetric, so a tile right of the diagonal also stands
            // for its mirror image - its derivative factors are formed once and the three sums of the mirrored tile (R_J, C_I, T'_J) are
            // taken from the same H with the operands swapped: half the Gram GEMMs and profile passes.
            for (int64_t j0 = sym ? i0 : 0; j0 < N; j0 += TILE) {
                const int64_t nc = std::min(TILE, N - j0);
                CGLB_TRY(gram_tile<T>(c, (const T*)c->Xs + (row0 + i0) * D, nr, (const T*)c->Xs + j0 * D, nc, G, nr));
                const bool both = sym && j0 > i0;
                {   // derivative factors in place + the row / column sums of the tile in the same pass
                    const unsigned gx = (unsigned)((nr + 255) / 256);
                    const int nslice = (int)((nc + CSLICE - 1) / CSLICE), nchunk = (int)gx * 4;
                    T* part_v = tmp;                                   // [nslice][nr]
                    T* part_u = part_v + (int64_t)nslice * nr;         // [nslice][nr]
                    T* pc_u = part_u + (int64_t)nslice * nr;           // [nchunk][nc]
                    T* pc_v = pc_u + (int64_t)nchunk * nc;             // [nchunk][nc]
                    const T* ar = (const T*)c->xa + row0 + i0;
                    const T* ac = (const T*)c->xa + j0;
                    if (both)
                        hipLaunchKernelGGL((wide_grad_profile_kernel<T, KIND, true>), dim3(gx, (unsigned)nslice), dim3(256), 0, c->stream, G, nr, nr, nc, ar, ac, u + i0,
                                           v + i0, u + j0, v + j0, part_v, part_u, pc_u, pc_v);
                    else
                        hipLaunchKernelGGL((wide_grad_profile_kernel<T, KIND, false>), dim3(gx, (unsigned)nslice), dim3(256), 0, c->stream, G, nr, nr, nc, ar, ac, u + i0,
                                           (const T*)nullptr, (const T*)nullptr, v + j0, part_v, part_u, pc_u, pc_v);
                    // R_I += u_I o (H v_J) ; C_J += v_J o (H^T u_I)
                    hipLaunchKernelGGL((wide_partsum_mulacc_kernel<T>), dim3(gx), dim3(256), 0, c->stream, (const T*)part_v, nslice, nr, u + i0, R + i0);
                    hipLaunchKernelGGL((wide_partsum_mulacc_kernel<T>), dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, c->stream, (const T*)pc_u, nchunk, nc, v + j0, C + j0);
                    if (both) {   // the mirror image: C_I += v_I o (H u_J) ; R_J += u_J o (H^T v_I) ; T'_J += H^T (v o X)_I
                        hipLaunchKernelGGL((wide_partsum_mulacc_kernel<T>), dim3(gx), dim3(256), 0, c->stream, (const T*)part_u, nslice, nr, v + i0, C + i0);
                        hipLaunchKernelGGL((wide_partsum_mulacc_kernel<T>), dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, c->stream, (const T*)pc_v, nchunk, nc, u + j0, R + j0);
                        BLAS_CHECK(c, wgemm(c->blas, rocblas_operation_none, rocblas_operation_none, D, (int)nc, (int)nr, 1.0, (const T*)VX + i0 * D, D, (const T*)G, (int)nr,
                                            1.0, Tp + j0 * D, D));
                    }
                }
                // T'_I += H (v o X)_J      (row-major [i][D] == column-major D x nr)
                BLAS_CHECK(c, wgemm(c->blas, rocblas_operation_none, rocblas_operation_transpose, D, (int)nr, (int)nc, 1.0, (const T*)VX + j0 * D, D, (const T*)G, (int)nr,
                                    1.0, Tp + i0 * D, D));
            }
        }
        double* sm = c->wsmall + 2 * (size_t)D;   // [a | b | cc]
        // a_d = sum_i x_id^2 R_i,  b_d = sum_i u_i x_id T'_id,  cc_d = sum_j x_jd^2 C_j
        hipLaunchKernelGGL((wide_coldot_kernel<T>), dim3(D), dim3(256), 0, c->stream, (const T*)c->Xsq + row0 * D, (const T*)nullptr, (const T*)R, nrows, D, sm);
        hipLaunchKernelGGL((wide_coldot_kernel<T>), dim3(D), dim3(256), 0, c->stream, (const T*)c->Xs + row0 * D, (const T*)Tp, u, nrows, D, sm + D);
        hipLaunchKernelGGL((wide_coldot_kernel<T>), dim3(D), dim3(256), 0, c->stream, (const T*)c->Xsq, (const T*)nullptr, (const T*)C, N, D, sm + 2 * D);
        hipLaunchKernelGGL(wide_dl_finish_kernel, dim3((D + 255) / 256), dim3(256), 0, c->stream, (const double*)sm, (const double*)(sm + D), (const double*)(sm + 2 * D),
                           (const double*)c->wsmall, D, c->var, out_dl);
    }));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// out[i] = sum_b slabs[b * count + i] in fixed order
template <typename T>
__global__ __launch_bounds__(256) void wide_slab_sum_kernel(const T* __restrict__ slabs, int nb, int64_t count, T* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    T s = 0;
    for (int b = 0; b < nb; ++b) s += slabs[(int64_t)b * count + i];
    out[i] = s;
}

// C (m x n, ld m) = A (m x k, lda) B (k x n, ldb) for a contraction far longer than the output is wide (k = N against m n = D M): one
// rocBLAS GEMM of that shape keeps a handful of workgroups busy (measured 5.4 ms for 7.9 GFLOP at N = 50 000, D = 77, M = 1024).  The
// contraction is cut into chunks of 1024, one strided-batched GEMM writes a slab per chunk and the slabs are added in fixed order.
template <typename T>
int gemm_long_k(cglb_ctx* c, int m, int n, int64_t k, const T* A, int lda, const T* B, int ldb, T* C) {
    const int64_t kc = 1024;
    const int nfull = (int)(k / kc);
    const int64_t rem = k - (int64_t)nfull * kc;
    const int nb = nfull + (rem > 0 ? 1 : 0);
    if (nb <= 1) {
        BLAS_CHECK(c, wgemm(c->blas, rocblas_operation_none, rocblas_operation_none, m, n, (int)k, 1.0, A, lda, B, ldb, 0.0, C, m));
        return CGLB_OK;
    }
    const int64_t count = (int64_t)m * n;
    CGLB_TRY(wensure(c, &c->wlong, &c->wlong_cap, (size_t)nb * count * sizeof(T)));
    T* slabs = (T*)c->wlong;
    if (nfull > 0)
        BLAS_CHECK(c, wgemm_sb(c->blas, rocblas_operation_none, rocblas_operation_none, m, n, (int)kc, A, lda, (rocblas_stride)(kc * lda), B, ldb, (rocblas_stride)kc,
                               slabs, m, (rocblas_stride)count, nfull));
    if (rem > 0)
        BLAS_CHECK(c, wgemm(c->blas, rocblas_operation_none, rocblas_operation_none, m, n, (int)rem, 1.0, A + (int64_t)nfull * kc * lda, lda, B + (int64_t)nfull * kc, ldb,
                            0.0, slabs + (int64_t)nfull * count, m));
    hipLaunchKernelGGL((wide_slab_sum_kernel<T>), dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, (const T*)slabs, nb, count, C);
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// Packed-gradient contributions of an adjoint panel G (row m contiguous, ld = ldg; + optional rank-1 term cvec[m] wvec[n]) of
// K(z_m, col_n): same contract as grad_panel (kernels_grad.hip).  XsCol / xaCol / XsqCol: scaled operands of the ncols columns.
int wide_grad_panel(cglb_ctx* c, const void* G, int64_t ldg, const void* cvec, const void* wvec, const void* XsCol, const void* xaCol, const void* XsqCol,
                    int64_t ncols, double zfactor, double* out) {
    if (ncols == 0) return CGLB_OK;
    const int D = c->D, M = c->M;
    CGLB_TRY(ensure_ones(c));
    const int64_t ld = (ncols + 7) & ~(int64_t)7;
    CGLB_DISPATCH_T(c->dtype, CGLB_DISPATCH_KIND(c->kind, {
        CGLB_TRY(wensure(c, &c->wpanel, &c->wpanel_cap, (size_t)M * ld * sizeof(T)));
        T* S = (T*)c->wpanel;
        const int nbx = (int)((ncols + 255) / 256);
        CGLB_TRY(wensure(c, &c->wS1, &c->wS1_cap, std::max((size_t)2 * c->N * D * sizeof(T), (size_t)M * D * sizeof(T) + (size_t)M * nbx * sizeof(double) + 64)));
        T* Tm = (T*)c->wS1;                                                     // [M][D]
        double* fpart = (double*)((char*)c->wS1 + (((size_t)M * D * sizeof(T) + 63) & ~(size_t)63));
        T *R = (T*)c->wR, *C = (T*)c->wC;
        CGLB_TRY(gram_tile<T>(c, (const T*)XsCol, ncols, (const T*)c->Zs, M, S, ld));   // S[n + m ld] = xs_n . zs_m
        hipLaunchKernelGGL((wide_panel_weight_kernel<T, KIND>), dim3((unsigned)nbx, (unsigned)M), dim3(256), 0, c->stream, S, ld, (const T*)G, ldg, (const T*)cvec,
                           (const T*)wvec, (const T*)xaCol, (const T*)c->za, ncols, fpart);
        // R_m = sum_n W_mn, C_n = sum_m W_mn, T_md = sum_n W_mn xs_nd
        BLAS_CHECK(c, wgemv(c->blas, rocblas_operation_transpose, (int)ncols, M, 1.0, (const T*)S, (int)ld, (const T*)c->wones, 0.0, R));
        BLAS_CHECK(c, wgemv(c->blas, rocblas_operation_none, (int)ncols, M, 1.0, (const T*)S, (int)ld, (const T*)c->wones, 0.0, C));
        CGLB_TRY((gemm_long_k<T>(c, D, M, ncols, (const T*)XsCol, D, (const T*)S, (int)ld, Tm)));
        double* xc = c->wsmall + 2 * (size_t)D;   // xc_d = sum_n xs_nd^2 C_n
        hipLaunchKernelGGL((wide_coldot_kernel<T>), dim3(D), dim3(256), 0, c->stream, (const T*)XsqCol, (const T*)nullptr, (const T*)C, ncols, D, xc);
        const int zblocks = (int)(((int64_t)M * D + 255) / 256);
        hipLaunchKernelGGL((wide_panel_finish_kernel<T>), dim3(D + 1 + zblocks), dim3(256), 0, c->stream, (const T*)c->Zs, (const T*)Tm, (const T*)R, (const double*)xc,
                           (const double*)fpart, (int64_t)M * nbx, M, D, (const double*)c->wsmall, (const double*)(c->wsmall + D), c->var, zfactor, out);
    }));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}
