// K4 — vector primitives of the PCG loop (conjugate_gradient.py:67-75) and
// K2/K3 — the Nystrom preconditioner pieces (conjugate_gradient.py:95-113) with the stored panel A.
// All HBM-bound streaming kernels; reductions use block partials + a fixed-order finalize (reproducible).
#include "devmath.h"
#include "dispatch.h"

__global__ __launch_bounds__(256) void finalize_sum_kernel2(const double* __restrict__ partials, int n, double* __restrict__ out,
                                                            double scale) {
    __shared__ double smem[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += partials[i];
    s = block_sum(s, smem);
    if (threadIdx.x == 0) out[0] = s * scale;
}

static inline int vec_grid(int64_t n, int per_block = 1024) {
    int64_t g = (n + per_block - 1) / per_block;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

// ---- dot ----------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dot_kernel(const T* __restrict__ a, const T* __restrict__ b, int64_t n,
                                                  double* __restrict__ dotpart) {
    __shared__ double smem[16];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        s += (double)a[i] * (double)b[i];
    s = block_sum(s, smem);
    if (threadIdx.x == 0) dotpart[blockIdx.x] = s;
}

int launch_dot(cglb_ctx* c, const void* a, const void* b, int64_t n, double* out_slot) {
    const int grid = vec_grid(n);
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((dot_kernel<T>), dim3(grid), dim3(256), 0, c->stream, (const T*)a,
                                                 (const T*)b, n, c->dotpart));
    CGLB_LAUNCH_CHECK(c);
    hipLaunchKernelGGL(finalize_sum_kernel2, dim3(1), dim3(256), 0, c->stream, (const double*)c->dotpart, grid, out_slot, 1.0);
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// ---- v += gamma p ; r -= gamma Ap  (gamma = rz / pAp from device scalars) ------------------------------
template <typename T>
__global__ __launch_bounds__(256) void update_v_r_kernel(T* __restrict__ v, T* __restrict__ r, const T* __restrict__ p,
                                                         const T* __restrict__ Ap, int64_t n, const double* __restrict__ rz,
                                                         const double* __restrict__ pAp, int update_r) {
    const T gamma = (T)(rz[0] / pAp[0]);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        v[i] = tfma<T>(gamma, p[i], v[i]);
        if (update_r) r[i] = tfma<T>(-gamma, Ap[i], r[i]);
    }
}

int launch_update_v_r(cglb_ctx* c, void* v, void* r, const void* p, const void* Ap, const double* rz, const double* pAp,
                      int update_r, int64_t n) {
    if (n < 0) n = c->nloc;
    if (n == 0) return CGLB_OK;
    const int grid = vec_grid(n, 256);
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((update_v_r_kernel<T>), dim3(grid), dim3(256), 0, c->stream, (T*)v, (T*)r,
                                                 (const T*)p, (const T*)Ap, n, rz, pAp, update_r));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// ---- r = b - Kv ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void residual_kernel(T* __restrict__ r, const T* __restrict__ b, const T* __restrict__ Kv, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        r[i] = b[i] - Kv[i];
}

int launch_residual(cglb_ctx* c, void* r, const void* b, const void* Kv, int64_t n) {
    if (n < 0) n = c->nloc;
    if (n == 0) return CGLB_OK;
    const int grid = vec_grid(n, 256);
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((residual_kernel<T>), dim3(grid), dim3(256), 0, c->stream, (T*)r, (const T*)b,
                                                 (const T*)Kv, n));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// ---- p = z + p * new_rz / rz  (or p = z on restart) --------------------------------------------------------
// wh != null (RBF, unclamped range): the pre-weighted operand pw = p * wh of the symmetric pair kernel (folded column norm,
// kernels_kff_sym.hip) is written in the same pass, so the next mat-vec needs no separate weighting launch.
template <typename T>
__global__ __launch_bounds__(256) void update_p_kernel(T* __restrict__ p, const T* __restrict__ z, int64_t n,
                                                       const double* __restrict__ new_rz, const double* __restrict__ rz, int restart,
                                                       const T* __restrict__ wh, T* __restrict__ pw) {
    const T beta = restart ? T(0) : (T)(new_rz[0] / rz[0]);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T pi = restart ? z[i] : tfma<T>(beta, p[i], z[i]);
        p[i] = pi;
        if (wh) pw[i] = pi * wh[i];
    }
}

// wh/pw of the fused weighting: only for a full-length vector (all N rows) of a context whose symmetric kernel folds the column norm
static inline bool fuse_weights(const cglb_ctx* c, int64_t n) {
    return c->kind == CGLB_RBF && !c->exp_clamp && c->have_hypers && n == c->N && (!is_wide(c) || mid_reg(c));
}

// fuse: also write the weighted copy for the NEXT symmetric mat-vec of p.  Only for callers that own the loop (the fused PCG and the
// segmented multi-GPU update): the copy is valid only while p is not modified before that mat-vec, which a caller of the generic
// cglb_vec_update_p / cglb_shard_update_p entry points has not promised.
int launch_update_p(cglb_ctx* c, void* p, const void* z, const double* new_rz, const double* rz, int restart, int64_t n, bool fuse) {
    if (n < 0) n = c->nloc;
    if (n == 0) return CGLB_OK;
    const int grid = vec_grid(n, 256);
    const bool fw = fuse && fuse_weights(c, n);
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((update_p_kernel<T>), dim3(grid), dim3(256), 0, c->stream, (T*)p, (const T*)z,
                                                 n, new_rz, rz, restart, fw ? (const T*)c->wh : (const T*)nullptr, (T*)c->pwh));
    CGLB_LAUNCH_CHECK(c);
    c->pwh_src = fw ? p : nullptr;  // consumed (and cleared) by the next symmetric mat-vec of exactly this vector
    return CGLB_OK;
}

// Segmented form for the cyclic multi-GPU driver: z arrives all-gathered in `world` slices of per + 1 elements, slice g = rank g's
// rows [g*per, (g+1)*per) followed by ONE extra element: that rank's partial of r^T z over its own rows.  new_rz = the sum of the
// `world` partials in rank order - every rank adds the same numbers in the same order, so the stop test of the host loop
// (conjugate_gradient.py:65) is taken on a value that is identical on all ranks BY CONSTRUCTION (it no longer depends on the
// replicated vector arithmetic staying bit-identical).  Also p = z + p new_rz / rz (:75) and the fused operand weighting.
template <typename T>
__global__ __launch_bounds__(256) void update_p_seg_kernel(T* __restrict__ p, const T* __restrict__ zseg, int64_t n, int64_t per, int world,
                                                           double* __restrict__ new_rz_out, const double* __restrict__ rz, int restart,
                                                           const T* __restrict__ wh, T* __restrict__ pw) {
    double nrz = 0.0;
    for (int g = 0; g < world; ++g) nrz += (double)zseg[(int64_t)g * (per + 1) + per];
    const T beta = restart ? T(0) : (T)(nrz / rz[0]);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T zi = zseg[i + i / per];  // slice g = i / per starts at g * (per + 1)
        const T pi = restart ? zi : tfma<T>(beta, p[i], zi);
        p[i] = pi;
        if (wh) pw[i] = pi * wh[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) new_rz_out[0] = nrz;
}

int launch_update_p_seg(cglb_ctx* c, void* p, const void* zseg, int64_t n, int64_t per, int world, double* new_rz_out, const double* rz,
                        int restart) {
    if (n <= 0 || per <= 0 || world <= 0) return cglb_fail(c, CGLB_ERR_BAD_ARG, "update_p_seg: bad geometry");
    const int grid = vec_grid(n, 256);
    const bool fw = fuse_weights(c, n);
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((update_p_seg_kernel<T>), dim3(grid), dim3(256), 0, c->stream, (T*)p, (const T*)zseg, n, per,
                                                 world, new_rz_out, rz, restart, fw ? (const T*)c->wh : (const T*)nullptr, (T*)c->pwh));
    CGLB_LAUNCH_CHECK(c);
    c->pwh_src = fw ? p : nullptr;
    return CGLB_OK;
}

// ---- y += alpha x  (adds the noise * p diagonal term to an all-reduced K_ff p) ----------------------------------
template <typename T>
__global__ __launch_bounds__(256) void axpy_kernel(T* __restrict__ y, T alpha, const T* __restrict__ x, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = tfma<T>(alpha, x[i], y[i]);
}
int launch_axpy(cglb_ctx* c, void* y, double alpha, const void* x, int64_t n) {
    if (n == 0) return CGLB_OK;
    const int grid = vec_grid(n, 256);
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((axpy_kernel<T>), dim3(grid), dim3(256), 0, c->stream, (T*)y, (T)alpha, (const T*)x, n));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// ---- e = y - mean ----------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sub_scalar_kernel(T* __restrict__ out, const T* __restrict__ y, T mean, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = y[i] - mean;
}

int launch_sub_scalar(cglb_ctx* c, void* out, const void* y_local, double mean, int64_t n) {
    if (n == 0) return CGLB_OK;
    const int grid = vec_grid(n, 256);
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((sub_scalar_kernel<T>), dim3(grid), dim3(256), 0, c->stream, (T*)out,
                                                 (const T*)y_local, (T)mean, n));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// ---- u = A_loc r : GU_ROWS rows of the [M][nloc] panel per block (HBM-bound, 16-B loads) ------------------------------
// grid = (ceil(M / GU_ROWS), nsplit): block (b, s) reduces columns [s*chunk, (s+1)*chunk) of rows GU_ROWS*b ...; partials [M][nsplit].
// One load of r serves GU_ROWS panel rows (with one row per block r was re-read from L2 once per panel row: as many L2 bytes as
// HBM bytes) and GU_ROWS independent 16-B panel loads are in flight per thread.
#define GU_ROWS 4
typedef double gu_d2 __attribute__((ext_vector_type(2)));  // native vectors: accepted by the non-temporal load builtin
typedef float gu_f4 __attribute__((ext_vector_type(4)));
template <typename T>
__global__ __launch_bounds__(256) void gemv_u_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ r, int64_t nloc, int M,
                                                     int64_t chunk, double* __restrict__ upart, T* __restrict__ u_direct) {
    __shared__ double smem[GU_ROWS][4];
    const int m0 = blockIdx.x * GU_ROWS;
    const int64_t n0 = (int64_t)blockIdx.y * chunk;
    const int64_t n1 = (n0 + chunk < nloc) ? n0 + chunk : nloc;
    const T* __restrict__ row[GU_ROWS];
#pragma unroll
    for (int q = 0; q < GU_ROWS; ++q) row[q] = A + (int64_t)(m0 + q < M ? m0 + q : M - 1) * lda;  // clamped rows are computed and dropped
    constexpr int V = 16 / sizeof(T);
    double s[GU_ROWS];
#pragma unroll
    for (int q = 0; q < GU_ROWS; ++q) s[q] = 0.0;
    int64_t i = n0 + (int64_t)threadIdx.x * V;
    // vector body when the row starts are 16-B aligned (lda is a multiple of 8 elements)
    const bool aligned = ((((uintptr_t)(row[0] + n0)) | ((uintptr_t)(r + n0)) | (uintptr_t)(lda * sizeof(T))) & 15) == 0;
    if (aligned) {
        using VT = typename std::conditional<sizeof(T) == 8, gu_d2, gu_f4>::type;
        for (; i + V <= n1; i += (int64_t)blockDim.x * V) {
            const VT b = *reinterpret_cast<const VT*>(r + i);
            VT a[GU_ROWS];
#pragma unroll
            for (int q = 0; q < GU_ROWS; ++q) a[q] = CGLB_STREAM_LOAD(reinterpret_cast<const VT*>(row[q] + i));
#pragma unroll
            for (int q = 0; q < GU_ROWS; ++q) {
                if constexpr (sizeof(T) == 8) {
                    s[q] += (double)a[q].x * (double)b.x + (double)a[q].y * (double)b.y;
                } else {
                    s[q] += (double)a[q].x * (double)b.x + (double)a[q].y * (double)b.y + (double)a[q].z * (double)b.z + (double)a[q].w * (double)b.w;
                }
            }
        }
        // tail (at most V-1 elements, handled by the thread whose i landed there)
        for (int64_t k = i; k < n1 && k < i + V; ++k) {
#pragma unroll
            for (int q = 0; q < GU_ROWS; ++q) s[q] += (double)row[q][k] * (double)r[k];
        }
    } else {
        for (int64_t k = n0 + threadIdx.x; k < n1; k += blockDim.x) {
#pragma unroll
            for (int q = 0; q < GU_ROWS; ++q) s[q] += (double)row[q][k] * (double)r[k];
        }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < GU_ROWS; ++q) {
        const double w = wave_sum(s[q]);
        if (lane == 0) smem[q][wid] = w;
    }
    __syncthreads();
    if (threadIdx.x < GU_ROWS && m0 + (int)threadIdx.x < M) {
        const int q = threadIdx.x;
        const double tot = (smem[q][0] + smem[q][1]) + (smem[q][2] + smem[q][3]);
        if (u_direct) u_direct[m0 + q] = (T)tot;  // one column split: this IS u[m], no finalize launch
        else upart[(int64_t)(m0 + q) * gridDim.y + blockIdx.y] = tot;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gemv_u_finalize_kernel(const double* __restrict__ upart, int M, int nsplit, T* __restrict__ u) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    double s = 0.0;
    for (int k = 0; k < nsplit; ++k) s += upart[(int64_t)m * nsplit + k];
    u[m] = (T)s;
}

int launch_gemv_u(cglb_ctx* c, const void* r_local, void* u_out) {
    int nsplit = 1;
    // enough blocks to fill the chip: M * nsplit >= ~2048
    while ((int64_t)c->M / GU_ROWS * nsplit < 2048 && nsplit < 64 && c->nloc / (nsplit * 2) >= 4096) nsplit *= 2;
    int64_t chunk = (c->nloc + nsplit - 1) / nsplit;
    chunk = (chunk + 7) & ~(int64_t)7;  // keeps 16-B alignment of chunk starts when nloc*esz is 16-B aligned
    if (chunk == 0) chunk = 8;
    nsplit = (int)((c->nloc + chunk - 1) / chunk);
    if (nsplit < 1) nsplit = 1;
    const size_t need = (size_t)c->M * nsplit * sizeof(double);
    if (need > c->gpart_cap) {
        if (c->gpart) HIP_CHECK(c, hipFree(c->gpart));
        c->gpart = nullptr;
        HIP_CHECK(c, hipMalloc((void**)&c->gpart, need));
        c->gpart_cap = need;
    }
    dim3 grid((unsigned)((c->M + GU_ROWS - 1) / GU_ROWS), (unsigned)nsplit);
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((gemv_u_kernel<T>), grid, dim3(256), 0, c->stream, (const T*)c->At, c->lda,
                                                 (const T*)r_local, c->nloc, c->M, chunk, c->gpart, nsplit == 1 ? (T*)u_out : (T*)nullptr));
    CGLB_LAUNCH_CHECK(c);
    if (nsplit > 1) {
        CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((gemv_u_finalize_kernel<T>), dim3((c->M + 255) / 256), dim3(256), 0, c->stream,
                                                     (const double*)c->gpart, c->M, nsplit, (T*)u_out));
        CGLB_LAUNCH_CHECK(c);
    }
    return CGLB_OK;
}

// ---- t = LB^-T (LB^-1 u): two triangular products with the explicit triangular inverse -----------------
// Wrows: matrix whose row i is contiguous (ld = M); out[i] = sum_{j in [lo_i, hi_i)} Wrows[i][j] x[j]
// lower = 1: j in [0, i];  lower = 0: j in [i, M).   One wave per output element.
template <typename T>
__global__ __launch_bounds__(256) void tri_rowdot_kernel(const T* __restrict__ Wrows, const T* __restrict__ x, int M, int lower,
                                                         T* __restrict__ out) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= M) return;
    const int lo = lower ? 0 : wave;
    const int hi = lower ? wave + 1 : M;
    const T* __restrict__ row = Wrows + (int64_t)wave * M;
    double s = 0.0;
    for (int j = lo + lane; j < hi; j += 64) s += (double)row[j] * (double)x[j];
    s = wave_sum(s);
    if (lane == 0) out[wave] = (T)s;
}

int launch_tri_rowdot(cglb_ctx* c, const void* Wrows, const void* x, int lower, void* out) {
    const int grid = (c->M * 64 + 255) / 256;
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((tri_rowdot_kernel<T>), dim3(grid), dim3(256), 0, c->stream, (const T*)Wrows, (const T*)x, c->M,
                                                 lower, (T*)out));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void scale_kernel(T* __restrict__ x, T a, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] *= a;
}
int launch_scale(cglb_ctx* c, void* x, double a, int64_t n) {
    if (n == 0) return CGLB_OK;
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((scale_kernel<T>), dim3(vec_grid(n, 256)), dim3(256), 0, c->stream, (T*)x, (T)a, n));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

int launch_tri_apply(cglb_ctx* c, const void* u, void* t_out) {
    const int grid = (c->M * 64 + 255) / 256;
    // y = LB^-1 u : rows of LB^-1 are contiguous in LBinvT's column-major storage (== row-major LB^-1)
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((tri_rowdot_kernel<T>), dim3(grid), dim3(256), 0, c->stream, (const T*)c->LBinvT,
                                                 (const T*)u, c->M, 1, (T*)c->w_t2));
    CGLB_LAUNCH_CHECK(c);
    // t = LB^-T y : rows of LB^-T are the columns of LB^-1, contiguous in LBinv's column-major storage
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((tri_rowdot_kernel<T>), dim3(grid), dim3(256), 0, c->stream, (const T*)c->LBinv,
                                                 (const T*)c->w_t2, c->M, 0, (T*)t_out));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// ---- rp = r - A_loc^T t ; z = rp / noise ; rz partial = sum rp*r / noise ---------------------------------
// stage 1: column sums over an m-chunk: tpart[s][n] = sum_{m in chunk s} A[m][n] t[m]   (coalesced along n)
template <typename T>
__global__ __launch_bounds__(256) void gemv_t_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ t, int64_t nloc, int M,
                                                     int mchunk, T* __restrict__ tpart) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nloc) return;
    const int m0 = blockIdx.y * mchunk;
    const int m1 = min(M, m0 + mchunk);
    T s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int m = m0;
    for (; m + 4 <= m1; m += 4) {
        s0 = tfma<T>(CGLB_STREAM_LOAD(A + (int64_t)m * lda + n), t[m], s0);
        s1 = tfma<T>(CGLB_STREAM_LOAD(A + (int64_t)(m + 1) * lda + n), t[m + 1], s1);
        s2 = tfma<T>(CGLB_STREAM_LOAD(A + (int64_t)(m + 2) * lda + n), t[m + 2], s2);
        s3 = tfma<T>(CGLB_STREAM_LOAD(A + (int64_t)(m + 3) * lda + n), t[m + 3], s3);
    }
    for (; m < m1; ++m) s0 = tfma<T>(A[(int64_t)m * lda + n], t[m], s0);
    tpart[(int64_t)blockIdx.y * nloc + n] = (s0 + s1) + (s2 + s3);
}

template <typename T>
__global__ __launch_bounds__(256) void precond_z_kernel(const T* __restrict__ r, const T* __restrict__ tpart, int msplit, int64_t nloc,
                                                        T inv_noise, T* __restrict__ z, double* __restrict__ dotpart) {
    __shared__ double smem[16];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nloc; i += (int64_t)gridDim.x * blockDim.x) {
        T p = 0;
        for (int s = 0; s < msplit; ++s) p += tpart[(int64_t)s * nloc + i];
        const T rp = r[i] - p;
        z[i] = rp * inv_noise;
        acc += (double)rp * (double)r[i];
    }
    acc = block_sum(acc, smem);
    if (threadIdx.x == 0) dotpart[blockIdx.x] = acc;
}

// z = (r - Ks)/noise, rz = sum (r - Ks) r / noise, with Ks = A^T t already formed (implicit preconditioner)
int launch_precond_z_from(cglb_ctx* c, const void* r_local, const void* Ks_local, void* z_local, double* rz_slot) {
    const int g2 = vec_grid(c->nloc, 256);
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((precond_z_kernel<T>), dim3(g2), dim3(256), 0, c->stream, (const T*)r_local,
                                                 (const T*)Ks_local, 1, c->nloc, (T)(1.0 / c->noise), (T*)z_local, c->dotpart));
    CGLB_LAUNCH_CHECK(c);
    if (rz_slot) {  // callers that form r^T z themselves (cyclic multi-GPU driver) pass no slot
        hipLaunchKernelGGL(finalize_sum_kernel2, dim3(1), dim3(256), 0, c->stream, (const double*)c->dotpart, g2, rz_slot, 1.0 / c->noise);
        CGLB_LAUNCH_CHECK(c);
    }
    return CGLB_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void finalize_sum_T_kernel(const double* __restrict__ partials, int n, T* __restrict__ out, double scale) {
    __shared__ double smem[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += partials[i];
    s = block_sum(s, smem);
    if (threadIdx.x == 0) out[0] = (T)(s * scale);
}

int launch_precond_z(cglb_ctx* c, const void* r_local, const void* t, void* z_local, double* rz_slot, void* rz_slot_T) {
    const int mchunk = 64;
    const int msplit = (c->M + mchunk - 1) / mchunk;
    if (c->nloc > 0) {
        dim3 grid((unsigned)((c->nloc + 255) / 256), (unsigned)msplit);
        CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((gemv_t_kernel<T>), grid, dim3(256), 0, c->stream, (const T*)c->At, c->lda,
                                                     (const T*)t, c->nloc, c->M, mchunk, (T*)c->tpart));
        CGLB_LAUNCH_CHECK(c);
    }
    const int g2 = vec_grid(c->nloc, 256);
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((precond_z_kernel<T>), dim3(g2), dim3(256), 0, c->stream, (const T*)r_local,
                                                 (const T*)c->tpart, msplit, c->nloc, (T)(1.0 / c->noise), (T*)z_local, c->dotpart));
    CGLB_LAUNCH_CHECK(c);
    if (rz_slot) {  // callers that form r^T z themselves pass no slot
        hipLaunchKernelGGL(finalize_sum_kernel2, dim3(1), dim3(256), 0, c->stream, (const double*)c->dotpart, g2, rz_slot, 1.0 / c->noise);
        CGLB_LAUNCH_CHECK(c);
    }
    if (rz_slot_T) {  // partial r^T z in the vector's element type: the extra element of this rank's all-gather slice
        CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((finalize_sum_T_kernel<T>), dim3(1), dim3(256), 0, c->stream, (const double*)c->dotpart, g2,
                                                     (T*)rz_slot_T, 1.0 / c->noise));
        CGLB_LAUNCH_CHECK(c);
    }
    return CGLB_OK;
}

// ---- small M x M helpers (column-major) -------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void tri_clean_kernel(T* __restrict__ Mc, int M, int keep_lower) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)M * M) return;
    const int i = (int)(idx % M), j = (int)(idx / M);  // column-major: element (i, j)
    if (keep_lower ? (i < j) : (i > j)) Mc[idx] = 0;
}
int launch_tri_clean(cglb_ctx* c, void* Mc, int keep_lower) {
    const int64_t tot = (int64_t)c->M * c->M;
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((tri_clean_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream,
                                                 (T*)Mc, c->M, keep_lower));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ src, T* __restrict__ dst, int M) {
    __shared__ T tile[16][17];
    const int bx = blockIdx.x * 16, by = blockIdx.y * 16;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    if (bx + tx < M && by + ty < M) tile[ty][tx] = src[(int64_t)(by + ty) * M + bx + tx];
    __syncthreads();
    if (by + tx < M && bx + ty < M) dst[(int64_t)(bx + ty) * M + by + tx] = tile[tx][ty];
}
int launch_transpose(cglb_ctx* c, const void* src, void* dst) {
    dim3 grid((c->M + 15) / 16, (c->M + 15) / 16);
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((transpose_kernel<T>), grid, dim3(256), 0, c->stream, (const T*)src, (T*)dst, c->M));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void add_identity_trace_kernel(T* __restrict__ Mc, int M, double* __restrict__ trace_slot) {
    __shared__ double smem[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        const T d = Mc[(int64_t)i * M + i];
        s += (double)d;
        Mc[(int64_t)i * M + i] = d + T(1);
    }
    s = block_sum(s, smem);
    if (threadIdx.x == 0) trace_slot[0] = s;
}
int launch_add_identity_trace(cglb_ctx* c, void* Mc, double* trace_slot) {
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((add_identity_trace_kernel<T>), dim3(1), dim3(256), 0, c->stream, (T*)Mc, c->M, trace_slot));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void sum_log_diag_kernel(const T* __restrict__ Mc, int M, double* __restrict__ slot) {
    __shared__ double smem[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < M; i += blockDim.x) s += log((double)Mc[(int64_t)i * M + i]);
    s = block_sum(s, smem);
    if (threadIdx.x == 0) slot[0] = s;
}
int launch_sum_log_diag(cglb_ctx* c, const void* Mc, double* slot) {
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((sum_log_diag_kernel<T>), dim3(1), dim3(256), 0, c->stream, (const T*)Mc, c->M, slot));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// slot[0] = min, slot[1] = max of the diagonal of a column-major M x M matrix (the spread of diag(L) is a cheap proxy of cond(L))
template <typename T>
__global__ __launch_bounds__(256) void diag_minmax_kernel(const T* __restrict__ Mc, int M, double* __restrict__ slot) {
    __shared__ double smin[256], smax[256];
    double lo = 1.0e300, hi = -1.0e300;
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        const double d = (double)Mc[(int64_t)i * M + i];
        lo = fmin(lo, d);
        hi = fmax(hi, d);
    }
    smin[threadIdx.x] = lo;
    smax[threadIdx.x] = hi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            smin[threadIdx.x] = fmin(smin[threadIdx.x], smin[threadIdx.x + s]);
            smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { slot[0] = smin[0]; slot[1] = smax[0]; }
}
int launch_diag_minmax(cglb_ctx* c, const void* Mc, double* slot) {
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((diag_minmax_kernel<T>), dim3(1), dim3(256), 0, c->stream, (const T*)Mc, c->M, slot));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void symmetrize_lower_kernel(T* __restrict__ Mc, int M) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)M * M) return;
    const int i = (int)(idx % M), j = (int)(idx / M);
    if (i < j) Mc[idx] = Mc[(int64_t)i * M + j];  // (i,j) upper <- (j,i) lower
}
int launch_symmetrize_lower(cglb_ctx* c, void* Mc) {
    const int64_t tot = (int64_t)c->M * c->M;
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((symmetrize_lower_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                                                 c->stream, (T*)Mc, c->M));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// ---- scalars of the bound assembly (models.py:280-284) and of the gradient -------------------------------
// sc[0] = sum v (r + Kv/2)      lower bound          (models.py:283)
// sc[1] = sum w r               error_bound r^T P r   (models.py:282), w = P r
// sc[2] = sum (w + v/2) v
// sc[3] = sum w^2
// sc[4] = sum (v + w)
// sc[5] = sum (w + v/2) (Kv - noise v)   = u^T K_ff v
template <typename T>
__global__ __launch_bounds__(256) void obj_scalars_kernel(const T* __restrict__ v, const T* __restrict__ r, const T* __restrict__ Kv,
                                                          const T* __restrict__ w, int64_t n, double noise, double* __restrict__ part) {
    __shared__ double smem[16];
    double a[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double vi = v[i], ri = r[i], kvi = Kv[i], wi = w[i];
        const double ui = wi + 0.5 * vi;
        a[0] += vi * (ri + 0.5 * kvi);
        a[1] += wi * ri;
        a[2] += ui * vi;
        a[3] += wi * wi;
        a[4] += vi + wi;
        a[5] += ui * (kvi - noise * vi);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double s = block_sum(a[k], smem);
        if (threadIdx.x == 0) part[(int64_t)k * gridDim.x + blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(256) void obj_scalars_finalize_kernel(const double* __restrict__ part, int nblk, double* __restrict__ sc8) {
    __shared__ double smem[16];
    for (int k = 0; k < 6; ++k) {
        double s = 0.0;
        for (int i = threadIdx.x; i < nblk; i += blockDim.x) s += part[(int64_t)k * nblk + i];
        s = block_sum(s, smem);
        if (threadIdx.x == 0) sc8[k] = s;
        __syncthreads();
    }
    if (threadIdx.x == 0) { sc8[6] = 0.0; sc8[7] = 0.0; }
}

int launch_obj_scalars(cglb_ctx* c, const void* v_local, const void* r, const void* Kv, const void* w, double* sc8) {
    const int grid = vec_grid(c->nloc, 256) > 1024 ? 1024 : vec_grid(c->nloc, 256);
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((obj_scalars_kernel<T>), dim3(grid), dim3(256), 0, c->stream, (const T*)v_local,
                                                 (const T*)r, (const T*)Kv, (const T*)w, c->nloc, c->noise, c->dotpart));
    CGLB_LAUNCH_CHECK(c);
    hipLaunchKernelGGL(obj_scalars_finalize_kernel, dim3(1), dim3(256), 0, c->stream, (const double*)c->dotpart, grid, sc8);
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}
