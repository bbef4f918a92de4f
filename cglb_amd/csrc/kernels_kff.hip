// K1 — implicit kernel mat-vec  out_i = var * sum_j kappa(x_i, x_j) p_j (+ noise p_i)
// Reference seam: `A @ p` with A = kernel(x).add_diag(sigma^2)  (models.py:251-252;
// conjugate_gradient.py:57,66,72).  K_ff is never materialised.
//
// Roofline: vector-fp64 ALU.  Algorithmic HBM traffic is N(D+2)w bytes per mat-vec against N^2 pair
// evaluations, so the kernel is built around VALU issue, not bandwidth:
//   * each lane owns R rows x_i (R*DP values + norm term in VGPRs for the whole launch);
//   * the column operand (xs_j, a_j, p_j) is wave-uniform, so it is fetched with scalar loads into
//     SGPRs and fed straight into v_fma_f64 as the scalar source: no VGPRs, no LDS traffic, no bank
//     conflicts for the streamed side;
//   * pair value by the Gram form (DP fma + 1 add), table-driven 2^x (devmath.h);
//   * columns are split over blockIdx.y; partial row sums go to a [jsplit][nrows] slab and are
//     combined in fixed order (bitwise reproducible, no atomics).
#include "devmath.h"
#include "dispatch.h"

template <typename T, int KIND, int DP, int R, bool CLAMP, int PREC>
__global__ __launch_bounds__(256) void kff_matvec_kernel(const T* __restrict__ XsRow, const T* __restrict__ xaRow,
                                                         int64_t nrows, const T* __restrict__ Xs,
                                                         const T* __restrict__ xa, const T* __restrict__ p, int64_t col0,
                                                         int64_t col1, int64_t jchunk, T* __restrict__ part,
                                                         const double* __restrict__ exp_tab) {
    __shared__ double tab[CGLB_TAB_SIZE];
    load_exp_table(tab, exp_tab);
    const int64_t rbase = (int64_t)blockIdx.x * (256 * R) + threadIdx.x;
    T xi[R][DP], ai[R], acc[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        int64_t row = rbase + (int64_t)k * 256;
        row = row < nrows ? row : nrows - 1;
#pragma unroll
        for (int d = 0; d < DP; ++d) xi[k][d] = XsRow[row * DP + d];
        ai[k] = (KIND == CGLB_RBF) ? xaRow[row] : T(-0.5) * xaRow[row];  // seed of the Gram chain
        acc[k] = 0;
    }
    const int64_t j0 = col0 + (int64_t)blockIdx.y * jchunk;
    const int64_t j1 = (j0 + jchunk < col1) ? j0 + jchunk : col1;
    for (int64_t j = j0; j < j1; ++j) {
        const T aj = xa[j];
        const T pj = p[j];
        T xj[DP];
#pragma unroll
        for (int d = 0; d < DP; ++d) xj[d] = Xs[j * DP + d];
        T gram[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            T g = ai[k];
#pragma unroll
            for (int d = 0; d < DP; ++d) g = tfma<T>(xi[k][d], xj[d], g);
            gram[k] = g;
        }
        KappaPend<T> kp[R];
        kappa_hot_begin_batch<T, KIND, CLAMP, false, PREC, R>(gram, aj, tab, kp);  // the R range reductions share one rounding-mode window
#pragma unroll
        for (int k = 0; k < R; ++k) {
            kappa_hot_poly<T, KIND, PREC>(kp[k]);
            acc[k] = tfma<T>(kappa_hot_end<T, KIND>(kp[k]), pj, acc[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int64_t row = rbase + (int64_t)k * 256;
        if (row < nrows) part[(int64_t)blockIdx.y * nrows + row] = acc[k];
    }
}

// out[i] = var * sum_s part[s][i] + noise * pdiag[i];  optional block partials of pdiag[i]*out[i].
template <typename T>
__global__ __launch_bounds__(256) void kff_combine_kernel(const T* __restrict__ part, int jsplit, int64_t nrows, T var,
                                                          T noise, const T* __restrict__ pdiag, T* __restrict__ out,
                                                          double* __restrict__ dotpart) {
    __shared__ double smem[16];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double contrib = 0.0;
    if (i < nrows) {
        T s = 0;
        for (int js = 0; js < jsplit; ++js) s += part[(int64_t)js * nrows + i];
        T o = var * s;
        if (pdiag) {
            o = tfma<T>(noise, pdiag[i], o);
            contrib = (double)pdiag[i] * (double)o;
        }
        out[i] = o;
    }
    if (dotpart) {
        const double bs = block_sum(contrib, smem);
        if (threadIdx.x == 0) dotpart[blockIdx.x] = bs;
    }
}

__global__ __launch_bounds__(256) void finalize_sum_kernel(const double* __restrict__ partials, int n, double* __restrict__ out) {
    __shared__ double smem[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += partials[i];
    s = block_sum(s, smem);
    if (threadIdx.x == 0) out[0] = s;
}

// fixed-order combine of the partial slab c->kpart [jsplit][nrows] (+ noise p, + optional p.out partials)
// Rectangular pair kernel with the software pipelining of the symmetric kernel (kernels_kff_sym.hip): one (row block, column chunk)
// work item per wave, the column operand fetched with scalar loads one column ahead of its use, the R table reads of a column issued
// together.  Serves the implicit Nystrom preconditioner (K_uf r: 1024 rows x N columns; K_fu s: N rows x 1024 columns), whose
// launches are small: the chunk is chosen so that a launch has thousands of items.  Always the range-clamped 2^x (inducing points may
// lie outside the training range).  part[k][row] = sum over the columns of chunk k; kff_combine_kernel adds the chunks in order.
template <typename T, int KIND, int DP, int R, int PREC>
__global__ __launch_bounds__(256, (sizeof(T) == 4 ? 4 : 1)) void kff_rect_kernel(const T* __restrict__ XsRow, const T* __restrict__ xaRow, int64_t nrows,
                                                       const T* __restrict__ XsCol, const T* __restrict__ xaCol, const T* __restrict__ pcol,
                                                       int64_t col0, int64_t col1, int64_t chunk, int nchunk, int64_t nitems,
                                                       T* __restrict__ part, const double* __restrict__ exp_tab) {
    __shared__ double tab[CGLB_TAB_SIZE];
    load_exp_table(tab, exp_tab);
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform
    if (item >= nitems) return;
    const int64_t rb = item / nchunk, k = item - rb * nchunk;
    const int64_t rbase = rb * (64 * R);
    T xi[R][DP], ai[R], acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t row = rbase + r * 64 + lane;
        row = row < nrows ? row : nrows - 1;
#pragma unroll
        for (int d = 0; d < DP; ++d) xi[r][d] = XsRow[row * DP + d];
        ai[r] = (KIND == CGLB_RBF) ? xaRow[row] : T(-0.5) * xaRow[row];  // seed of the Gram chain
        acc[r] = 0;
    }
    const int64_t j0 = col0 + k * chunk;
    const int64_t j1 = (j0 + chunk < col1) ? j0 + chunk : col1;
    T xj[DP], aj = xaCol[j0], pj = pcol[j0];
#pragma unroll
    for (int d = 0; d < DP; ++d) xj[d] = XsCol[j0 * DP + d];
    for (int64_t j = j0; j < j1; ++j) {
        const int64_t jn = (j + 1 < j1) ? j + 1 : j;  // next column (the last one re-reads itself)
        T xn[DP];
        const T an = xaCol[jn], pn = pcol[jn];
#pragma unroll
        for (int d = 0; d < DP; ++d) xn[d] = XsCol[jn * DP + d];
        __builtin_amdgcn_sched_barrier(0);  // issue the prefetch first; it is consumed a whole column later
        T gram[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            T g = ai[r];
#pragma unroll
            for (int d = 0; d < DP; ++d) g = tfma<T>(xi[r][d], xj[d], g);
            gram[r] = g;
        }
        KappaPend<T> kp[R];
        kappa_hot_begin_batch<T, KIND, true, false, PREC, R>(gram, aj, tab, kp);
        __builtin_amdgcn_sched_barrier(0);  // all R table reads are in flight here ...
#pragma unroll
        for (int r = 0; r < R; ++r) kappa_hot_poly<T, KIND, PREC>(kp[r]);
        __builtin_amdgcn_sched_barrier(0);  // ... and are first needed here
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = tfma<T>(kappa_hot_end<T, KIND>(kp[r]), pj, acc[r]);
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("" : "+v"(acc[r]));  // keep the accumulation of a column in its column
        __builtin_amdgcn_sched_barrier(0);
        aj = an;
        pj = pn;
#pragma unroll
        for (int d = 0; d < DP; ++d) xj[d] = xn[d];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = rbase + r * 64 + lane;
        if (row < nrows) part[k * nrows + row] = acc[r];
    }
}

// Same sum for many slabs and few rows (K_uf r of the implicit preconditioner: ~900 slabs x 1024 rows): a block takes 64 rows, its
// 16 waves add the slabs g, g+16, ... (4 loads in flight per lane) and the 16 partial sums are added in fixed order through LDS.
template <typename T>
__global__ __launch_bounds__(1024) void kff_combine_wide_kernel(const T* __restrict__ part, int jsplit, int64_t nrows, T var, T* __restrict__ out) {
    __shared__ T gsum[16][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    T a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (i < nrows) {
        int js = g;
        for (; js + 48 < jsplit; js += 64) {
            a0 += part[(int64_t)js * nrows + i];
            a1 += part[(int64_t)(js + 16) * nrows + i];
            a2 += part[(int64_t)(js + 32) * nrows + i];
            a3 += part[(int64_t)(js + 48) * nrows + i];
        }
        for (; js < jsplit; js += 16) a0 += part[(int64_t)js * nrows + i];
    }
    gsum[g][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (g == 0 && i < nrows) {
        T s = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += gsum[q][lane];
        out[i] = var * s;
    }
}

template <typename T>
static int kff_combine(cglb_ctx* c, int64_t jsplit, int64_t nrows, T* out, const T* pdiag, T noise, double* pdot_slot) {
    if (c->kff_skip_combine) return CGLB_OK;
    const int cgrid = (int)((nrows + 255) / 256);
    if (pdot_slot && cgrid > DOTPART_CAP) return cglb_fail(c, CGLB_ERR_BAD_ARG, "row shard too large for dot partials");
    hipLaunchKernelGGL((kff_combine_kernel<T>), dim3(cgrid), dim3(256), 0, c->stream, (const T*)c->kpart, (int)jsplit, nrows,
                       (T)c->var, noise, pdiag, out, pdot_slot ? c->dotpart : nullptr);
    CGLB_LAUNCH_CHECK(c);
    if (pdot_slot) {
        hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(256), 0, c->stream, (const double*)c->dotpart, cgrid, pdot_slot);
        CGLB_LAUNCH_CHECK(c);
    }
    return CGLB_OK;
}

static inline int rows_per_thread(const cglb_ctx* c) {
    int r = c->kff_rows;
    if (c->Dp > 16) r = 1;
    else if (c->Dp > 8 && r > 2) r = 2;
    if (r != 1 && r != 2 && r != 4) r = 4;
    return r;
}

// Launches the plain pair kernel for rows (XsRow, xaRow, nrows) against columns [col0, col1) into `part` ([slots][nrows]).
template <typename T, int KIND, int DP>
static int kff_pairs_range(cglb_ctx* c, const T* XsRow, const T* xaRow, int64_t nrows, const T* p_full, int64_t col0, int64_t col1, T* part,
                           int64_t max_slots, int64_t* nslots, const T* XsCol = nullptr, const T* xaCol = nullptr) {
    if (!XsCol) { XsCol = (const T*)c->Xh; xaCol = (const T*)c->xah; }  // default column operand: the training inputs
    const int R = rows_per_thread(c);
    const int64_t ncols = col1 - col0;
    const int64_t bx = (nrows + 256 * R - 1) / (256 * R);
    int64_t jsplit = c->kff_jsplit > 0 ? c->kff_jsplit : (8192 + bx - 1) / bx;
    if (jsplit > max_slots) jsplit = max_slots;
    if (jsplit > (ncols + 63) / 64) jsplit = (ncols + 63) / 64;
    if (jsplit < 1) jsplit = 1;
    int64_t jchunk = (ncols + jsplit - 1) / jsplit;
    jchunk = (jchunk + 1) & ~(int64_t)1;
    jsplit = (ncols + jchunk - 1) / jchunk;
    *nslots = jsplit;
    dim3 grid((unsigned)bx, (unsigned)jsplit);
#define KFF_LAUNCH(RR)                                                                                               \
    CGLB_DISPATCH_PREC(c, {                                                                                          \
        if (c->exp_clamp)                                                                                            \
            hipLaunchKernelGGL((kff_matvec_kernel<T, KIND, DP, RR, true, PREC>), grid, dim3(256), 0, c->stream, XsRow, \
                               xaRow, nrows, XsCol, xaCol, p_full, col0, col1, jchunk, part,                          \
                               (const double*)c->exp_tab);                                                          \
        else                                                                                                         \
            hipLaunchKernelGGL((kff_matvec_kernel<T, KIND, DP, RR, false, PREC>), grid, dim3(256), 0, c->stream, XsRow, \
                               xaRow, nrows, XsCol, xaCol, p_full, col0, col1, jchunk, part,                          \
                               (const double*)c->exp_tab);                                                          \
    })
    if (R == 4) {
        if constexpr (DP <= 8) { KFF_LAUNCH(4); } else if constexpr (DP <= 16) { KFF_LAUNCH(2); } else { KFF_LAUNCH(1); }
    } else if (R == 2) {
        if constexpr (DP <= 16) { KFF_LAUNCH(2); } else { KFF_LAUNCH(1); }
    } else {
        KFF_LAUNCH(1);
    }
#undef KFF_LAUNCH
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

template <typename T, int KIND, int DP>
static int kff_generic(cglb_ctx* c, const T* XsRow, const T* xaRow, int64_t nrows, const T* p_full, T* out, const T* pdiag,
                       T noise, double* pdot_slot) {
    if (nrows == 0) return CGLB_OK;
    const size_t need = (size_t)512 * nrows * sizeof(T);
    if (need > c->kpart_cap) {
        if (c->kpart) HIP_CHECK(c, hipFree(c->kpart));
        c->kpart = nullptr;
        HIP_CHECK(c, hipMalloc(&c->kpart, need));
        c->kpart_cap = need;
    }
    int64_t jsplit = 1;
    CGLB_TRY((kff_pairs_range<T, KIND, DP>(c, XsRow, xaRow, nrows, p_full, 0, c->N, (T*)c->kpart, 512, &jsplit)));
    return kff_combine<T>(c, jsplit, nrows, out, pdiag, noise, pdot_slot);
}

// Rectangular kernel product with explicit row and column operands (hot-scaled):
//   out[i] = var * sum_{j in [col0,col1)} kappa(row_i, col_j) pcol[j]      (pcol is indexed by the absolute column index)
// Used by the implicit Nystrom preconditioner (K_uf r and K_fu s).
template <typename T, int KIND, int DP>
static int kff_rect_generic(cglb_ctx* c, const T* XsRow, const T* xaRow, int64_t nrows, const T* XsCol, const T* xaCol, const T* pcol, int64_t col0,
                            int64_t col1, T* out) {
    constexpr int R = sizeof(T) == 8 ? ((DP <= 4) ? 8 : (DP <= 8) ? 4 : (DP <= 16 ? 2 : 1)) : ((DP <= 4) ? 8 : (DP <= 16) ? 4 : 2);
    const int64_t ncols = col1 - col0;
    const int64_t nrb = (nrows + 64 * R - 1) / (64 * R);
    // enough (row block, chunk) items to fill the chip (~4k waves), chunks of at least 64 columns, multiples of 16
    int64_t nchunk = (4096 + nrb - 1) / nrb;
    if (nchunk > (ncols + 63) / 64) nchunk = (ncols + 63) / 64;
    if (nchunk > 2048) nchunk = 2048;
    if (nchunk < 1) nchunk = 1;
    int64_t chunk = (ncols + nchunk - 1) / nchunk;
    chunk = (chunk + 15) / 16 * 16;
    nchunk = (ncols + chunk - 1) / chunk;
    const size_t need = (size_t)nchunk * nrows * sizeof(T);
    if (need > c->ppart_cap) {
        if (c->ppart) HIP_CHECK(c, hipFree(c->ppart));
        c->ppart = nullptr;
        HIP_CHECK(c, hipMalloc(&c->ppart, need));
        c->ppart_cap = need;
    }
    const int64_t nitems = nrb * nchunk;
    const unsigned grid = (unsigned)((nitems + 3) / 4);
    CGLB_DISPATCH_PREC(c, hipLaunchKernelGGL((kff_rect_kernel<T, KIND, DP, R, PREC>), dim3(grid), dim3(256), 0, c->stream, XsRow, xaRow, nrows, XsCol, xaCol,
                                             pcol, col0, col1, chunk, (int)nchunk, nitems, (T*)c->ppart, (const double*)c->exp_tab));
    CGLB_LAUNCH_CHECK(c);
    if (nchunk > 32) {  // many slabs: the one-thread-per-row sum would be a serial chain of nchunk loads on nrows threads
        hipLaunchKernelGGL((kff_combine_wide_kernel<T>), dim3((unsigned)((nrows + 63) / 64)), dim3(1024), 0, c->stream, (const T*)c->ppart, (int)nchunk, nrows,
                           (T)c->var, out);
    } else {
        const int cgrid = (int)((nrows + 255) / 256);
        hipLaunchKernelGGL((kff_combine_kernel<T>), dim3(cgrid), dim3(256), 0, c->stream, (const T*)c->ppart, (int)nchunk, nrows, (T)c->var, (T)0,
                           (const T*)nullptr, out, (double*)nullptr);
    }
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// Rectangular kernel product with explicit row and column operands (hot-scaled):
//   out[i] = var * sum_{j in [col0,col1)} kappa(row_i, col_j) pcol[j]      (pcol is indexed by the absolute column index)
// Used by the implicit Nystrom preconditioner (K_uf r and K_fu s).
int launch_pairs_rect(cglb_ctx* c, const void* XsRow, const void* xaRow, int64_t nrows, const void* XsCol, const void* xaCol, const void* pcol,
                      int64_t col0, int64_t col1, void* out) {
    if (nrows == 0) return CGLB_OK;
    if (is_wide(c)) return cglb_fail(c, CGLB_ERR_STATE, "the implicit preconditioner (precond_mode 1) is not available for inputs wider than 32 dimensions");
    if (col1 <= col0) {
        HIP_CHECK(c, hipMemsetAsync(out, 0, (size_t)nrows * c->esz, c->stream));
        return CGLB_OK;
    }
    CGLB_DISPATCH_ALL(c, return (kff_rect_generic<T, KIND, DP>(c, (const T*)XsRow, (const T*)xaRow, nrows, (const T*)XsCol, (const T*)xaCol, (const T*)pcol,
                                                               col0, col1, (T*)out)));
    return CGLB_OK;
}

// plain pair kernel of the local row shard against columns [col0, col1) (used by the symmetric path for the
// off-diagonal column ranges of a shard); at most 512 slots are written to `part`.
int launch_kff_plain_range(cglb_ctx* c, const void* p_full, int64_t col0, int64_t col1, void* part, int64_t* nslots) {
    CGLB_DISPATCH_ALL(c, return (kff_pairs_range<T, KIND, DP>(c, (const T*)c->Xh + c->r0 * DP, (const T*)c->xah + c->r0, c->nloc, (const T*)p_full,
                                                              col0, col1, (T*)part, 512, nslots)));
    return CGLB_OK;
}

// out_local = K_ff[rows,:] p + noise p[rows]; if pdot_slot != null also sum_i p_i out_i over local rows.
int launch_kff_matvec(cglb_ctx* c, const void* p_full, void* out_local, double* pdot_slot) {
    if (c->nloc == 0) return CGLB_OK;
    if (is_wide(c) && mid_reg(c) && c->nloc == c->N) return launch_kff_sym_mid(c, p_full, out_local, pdot_slot, false);
    if (is_wide(c))
        return wide_matvec(c, (const char*)c->Xs + (size_t)c->r0 * c->Dp * c->esz, (const char*)c->xa + (size_t)c->r0 * c->esz, c->r0, c->nloc, p_full, out_local, true,
                           pdot_slot, 1, 0);
    if (c->kff_variant == 2) return launch_kff_sym(c, p_full, out_local, pdot_slot);  // symmetric form (kernels_kff_sym.hip)
    if (c->kff_variant == 1 && c->dtype == CGLB_F64 && (c->r0 & 15) == 0) {  // matrix-pipe Gram path (kernels_kff_mfma.hip)
        int64_t jsplit = 1;
        CGLB_TRY(launch_kff_mfma_pairs(c, (const double*)p_full, &jsplit));
        return kff_combine<double>(c, jsplit, c->nloc, (double*)out_local, (const double*)p_full + c->r0, c->noise, pdot_slot);
    }
    CGLB_DISPATCH_ALL(c, return (kff_generic<T, KIND, DP>(c, (const T*)c->Xh + c->r0 * DP, (const T*)c->xah + c->r0, c->nloc,
                                                          (const T*)p_full, (T*)out_local, (const T*)p_full + c->r0,
                                                          (T)c->noise, pdot_slot)));
    return CGLB_OK;
}

// out[i] = var * sum_j kappa(xnew_i, x_j) v_j over all N columns (no diagonal term).
int launch_cross_matvec(cglb_ctx* c, const void* Xs_new, const void* xa_new, int64_t n_new, const void* v_full, void* out) {
    if (is_wide(c)) return wide_matvec(c, Xs_new, xa_new, 0, n_new, v_full, out, false, nullptr, 1, 0);
    // new points may lie far outside the training range: always take the range-clamped 2^x here
    const bool saved = c->exp_clamp;
    c->exp_clamp = true;
    int rc = CGLB_OK;
    CGLB_DISPATCH_ALL(c, rc = (kff_generic<T, KIND, DP>(c, (const T*)Xs_new, (const T*)xa_new, n_new, (const T*)v_full,
                                                        (T*)out, (const T*)nullptr, (T)0, nullptr)));
    c->exp_clamp = saved;
    return rc;
}
